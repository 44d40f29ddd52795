"""Bindings of the two fused "next-row" kernels: photometric loss (N1) and Adam (N2)."""
import ctypes

import torch

from ._lib import check, lib, ptr, stream_ptr

_vp, _i32, _u64, _f = ctypes.c_void_p, ctypes.c_int32, ctypes.c_uint64, ctypes.c_float
lib.w3d_l1_ssim_sizes.argtypes = [_i32, _i32, _i32, ctypes.POINTER(_u64)]
lib.w3d_l1_ssim_sizes.restype = ctypes.c_int
lib.w3d_l1_ssim_fwd_bwd.argtypes = [_i32, _i32, _i32, _vp, _vp, _f, _vp, _vp, _vp, _vp]
lib.w3d_l1_ssim_fwd_bwd.restype = ctypes.c_int
lib.w3d_l1_ssim_values.argtypes = [_i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]
lib.w3d_l1_ssim_values.restype = ctypes.c_int
lib.w3d_l1_ssim_grad.argtypes = [_i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
lib.w3d_l1_ssim_grad.restype = ctypes.c_int
lib.w3d_add_densification_stats.argtypes = [_i32, _vp, _vp, _vp, _vp, _vp]
lib.w3d_add_densification_stats.restype = ctypes.c_int
lib.w3d_adam_step.argtypes = [_u64, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _f, _f, _i32, _vp]
lib.w3d_adam_step.restype = ctypes.c_int
lib.w3d_densify_compact.argtypes = [_i32, ctypes.POINTER(_i32), _i32, _i32, _u64, _u64, _u64, _u64] + [_vp] * 10
lib.w3d_densify_compact.restype = ctypes.c_int


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


def l1_ssim_values(image, gt):
    """(mean |image - gt|, ssim(image, gt), scratch) — pass A of the fused loss alone; `scratch` holds the maps
    l1_ssim_grad needs (w3d_l1_ssim_values / w3d_l1_ssim_grad, include/w3d.h)."""
    if not image.is_cuda:
        raise RuntimeError("fused loss needs GPU tensors")
    dev = image.device
    img, g = image.detach(), gt.detach()
    if img.shape != g.shape or img.dim() != 3 or img.dtype != torch.float32 or g.dtype != torch.float32 or \
            not img.is_contiguous() or not g.is_contiguous():
        raise RuntimeError("image and gt must both be contiguous fp32 (C,H,W)")
    C, H, W = img.shape
    sb = _u64()
    check(lib.w3d_l1_ssim_sizes(C, H, W, ctypes.byref(sb)))
    with torch.cuda.device(dev):
        scratch = torch.empty(sb.value, dtype=torch.uint8, device=dev)
        l1 = torch.empty((), dtype=torch.float32, device=dev)
        s = torch.empty((), dtype=torch.float32, device=dev)
        check(lib.w3d_l1_ssim_values(C, H, W, ptr(img), ptr(g), ptr(l1), ptr(s), ptr(scratch), stream_ptr(dev)))
    return l1, s, scratch


def l1_ssim_grad(image, gt, w_l1, w_ssim, scratch):
    """dL/dimage = w_l1 * dL1/dimage + w_ssim * dSSIM/dimage with the two upstream gradients as DEVICE scalars (None = 0),
    from the maps l1_ssim_values left in `scratch`."""
    dev = image.device
    C, H, W = image.shape
    f = lambda t: None if t is None else t.detach().to(torch.float32).contiguous()  # noqa: E731
    w1, w2 = f(w_l1), f(w_ssim)
    with torch.cuda.device(dev):
        grad = torch.empty_like(image)
        check(lib.w3d_l1_ssim_grad(C, H, W, ptr(image.detach()), ptr(gt.detach()), ptr(w1), ptr(w2), ptr(grad), ptr(scratch),
                                   stream_ptr(dev)))
    return grad


def add_densification_stats(grad2d, update_filter, accum, denom):
    """In place: accum[f] += ||grad2d[f, :2]||, denom[f] += 1 for a boolean filter, in one kernel."""
    P = grad2d.shape[0]
    with torch.cuda.device(grad2d.device):
        check(lib.w3d_add_densification_stats(P, ptr(grad2d), ptr(update_filter), ptr(accum), ptr(denom),
                                              stream_ptr(grad2d.device)))


def adam_step(p, g, m, v, lr, beta1, beta2, eps, bc1, bc2, zero_grad=False):
    """In-place Adam on four 1-D fp32 views that share their alignment."""
    n = p.numel()
    if n == 0:
        return
    with torch.cuda.device(p.device):
        check(lib.w3d_adam_step(n, ptr(p), ptr(g), ptr(m), ptr(v), float(lr), float(beta1), float(beta2), float(eps),
                                float(bc1), float(bc2), int(bool(zero_grad)), stream_ptr(p.device)))


def densify_compact(block_dims, xyz_block, scaling_block, P_old, src_rows, n_keep, n_child0, param_old, m_old, v_old,
                    param_new, m_new, v_new, child_xyz=None, child_scaling=None):
    """One-pass row compaction of the flat parameter buffer and the Adam moments (csrc/w3d_densify.hip).
    src_rows: int32 (P_new,) source row of every output row; rows >= n_keep get zero moments; rows >= n_child0
    take their xyz / scaling from child_xyz / child_scaling."""
    if not param_old.is_cuda:
        raise RuntimeError("densify_compact needs GPU tensors")
    P_new = int(src_rows.numel())
    dims = (_i32 * len(block_dims))(*[int(d) for d in block_dims])
    src = src_rows.to(torch.int32).contiguous()
    cx = None if child_xyz is None else child_xyz.float().contiguous()
    cs = None if child_scaling is None else child_scaling.float().contiguous()
    with torch.cuda.device(param_old.device):
        check(lib.w3d_densify_compact(len(block_dims), dims, int(xyz_block), int(scaling_block), int(P_old), P_new,
                                      int(n_keep), int(n_child0), ptr(src), ptr(param_old), ptr(m_old), ptr(v_old),
                                      ptr(param_new), ptr(m_new), ptr(v_new), ptr(cx), ptr(cs), stream_ptr(param_old.device)))
