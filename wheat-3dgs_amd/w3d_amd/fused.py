"""Bindings of the two fused "next-row" kernels: photometric loss (N1) and Adam (N2)."""
import ctypes

import torch

from ._lib import check, lib, ptr, stream_ptr

_vp, _i32, _u64, _f = ctypes.c_void_p, ctypes.c_int32, ctypes.c_uint64, ctypes.c_float
lib.w3d_l1_ssim_sizes.argtypes = [_i32, _i32, _i32, ctypes.POINTER(_u64)]
lib.w3d_l1_ssim_sizes.restype = ctypes.c_int
lib.w3d_l1_ssim_fwd_bwd.argtypes = [_i32, _i32, _i32, _vp, _vp, _f, _vp, _vp, _vp, _vp]
lib.w3d_l1_ssim_fwd_bwd.restype = ctypes.c_int
lib.w3d_adam_step.argtypes = [_u64, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _f, _f, _i32, _vp]
lib.w3d_adam_step.restype = ctypes.c_int


def l1_ssim_fwd_bwd(image, gt, lambda_dssim=0.2):
    """(loss scalar tensor, dL/dimage) for image, gt of shape (C,H,W) on the GPU."""
    if not image.is_cuda:
        raise RuntimeError("fused loss needs GPU tensors")
    dev = image.device
    img = image.detach().float().contiguous()
    g = gt.detach().float().contiguous()
    if img.shape != g.shape or img.dim() != 3:
        raise RuntimeError("image and gt must both be (C,H,W)")
    C, H, W = img.shape
    sb = _u64()
    check(lib.w3d_l1_ssim_sizes(C, H, W, ctypes.byref(sb)))
    with torch.cuda.device(dev):
        scratch = torch.empty(sb.value, dtype=torch.uint8, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        grad = torch.empty_like(img)
        check(lib.w3d_l1_ssim_fwd_bwd(C, H, W, ptr(img), ptr(g), float(lambda_dssim), ptr(loss), ptr(grad),
                                      ptr(scratch), stream_ptr(dev)))
    return loss, grad


def adam_step(p, g, m, v, lr, beta1, beta2, eps, bc1, bc2, zero_grad=False):
    """In-place Adam on four 1-D fp32 views that share their alignment."""
    n = p.numel()
    if n == 0:
        return
    with torch.cuda.device(p.device):
        check(lib.w3d_adam_step(n, ptr(p), ptr(g), ptr(m), ptr(v), float(lr), float(beta1), float(beta2), float(eps),
                                float(bc1), float(bc2), int(bool(zero_grad)), stream_ptr(p.device)))
