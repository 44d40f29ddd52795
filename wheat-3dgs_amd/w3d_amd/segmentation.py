"""Host logic around the FlashSplat contribution render: turning the
additive per-(object, Gaussian) counts into labels, as reference run_3d_seg.py:54-72 does
(pinned by tests/golden/multi_instance_opt.npz) — vectorised over objects instead of a Python
loop with one tqdm step per object — plus the per-view mask work of that loop on the device (csrc/w3d_mask.hip):
binarising the decoded mask image, and find_match's alpha > 0.5 -> bounding box -> IoU scoring without copying the
alpha image to the host."""
import ctypes

import numpy as np
import torch


def multi_instance_opt(all_contrib: torch.Tensor, gamma: float = 0.0) -> torch.Tensor:
    """all_contrib (K, P) additive counts -> bool (K, P): S[i, j] = Gaussian j belongs to object i.
    For every object the pair (rest, own) = (sum - own, own) is L2-normalised over the pair,
    `gamma` is added to the rest score, and the larger one wins (ties go to "rest")."""
    total = all_contrib.sum(dim=0, keepdim=True)
    rest = total - all_contrib
    norm = torch.sqrt(rest * rest + all_contrib * all_contrib).clamp_min(1e-12)
    return (all_contrib / norm) > (rest / norm + gamma)


def accumulate_counts(render_fn, cameras, masks, obj_num=1):
    """Sum of `used_count` over views (reference run_3d_seg.py:91-97).  render_fn(cam, mask) must
    return the dict of flashsplat_render."""
    total = None
    for cam, mask in zip(cameras, masks):
        with torch.no_grad():
            uc = render_fn(cam, mask)["used_count"]
        total = uc.clone() if total is None else total + uc
    return total


def accumulate_counts_raw(model, cameras, masks, bg_color, obj_num=1):
    """The same sum for the flat GaussianModel on the GPU, accumulated IN the kernel: every view's scatter adds into one
    (obj_num+1, P) buffer, so no per-view count tensor is allocated, zeroed or added (2.4 GB each at 300 labels x 2 M)."""
    from .fused_step import render_raw
    P = model.num_points
    total = torch.zeros(obj_num + 1, P, dtype=torch.float32, device=model.flat.device)
    with torch.no_grad():
        for cam, mask in zip(cameras, masks):
            render_raw(cam, model, bg_color, flash=dict(gt_mask=mask, num_obj=obj_num, accumulate_into=total))
    return total


def _mask_lib():
    from ._lib import lib
    if not getattr(lib, "_w3d_mask_bound", False):
        vp, i32 = ctypes.c_void_p, ctypes.c_int32
        lib.w3d_mask_binarize.argtypes = [i32, i32, i32, vp, vp, vp]
        lib.w3d_mask_binarize.restype = ctypes.c_int
        lib.w3d_mask_iou.argtypes = [i32, i32, i32, vp, ctypes.c_float, vp, vp, vp]
        lib.w3d_mask_iou.restype = ctypes.c_int
        lib._w3d_mask_bound = True
    return lib


def binarize_mask_device(image, device="cuda"):
    """binarize_mask(PILtoTorch(image, resolution)) of reference run_3d_seg.py:88-89 / utils/wheatgs_utils.py:14-37,
    computed on the device: `image` is the decoded mask — a PIL image, or an (H,W) / (H,W,C) uint8 array — already at
    the camera's resolution; its 8-bit pixels are uploaded as they are and a (H,W) float32 image of 0 / 1 comes back
    (1 where any channel is non-zero; the reference's normalisation to [0,1] does not change which pixels are > 0)."""
    from ._lib import check, ptr, stream_ptr
    a = np.asarray(image)
    if a.dtype == np.bool_:
        a = a.astype(np.uint8)
    if a.dtype != np.uint8 or a.ndim not in (2, 3):
        raise ValueError("mask image must be 8-bit with shape (H,W) or (H,W,C)")
    H, W = int(a.shape[0]), int(a.shape[1])
    C = 1 if a.ndim == 2 else int(a.shape[2])
    if C not in (1, 3):
        # (an RGBA PNG would become all ones through its opaque alpha channel; reference binarize_mask,
        #  utils/wheatgs_utils.py:26-37, raises for anything but 1 or 3 channels too)
        raise ValueError("Mask tensor should have 1 or 3 channels")
    dev = torch.device(device)
    pix = torch.from_numpy(np.ascontiguousarray(a)).to(dev, non_blocking=True)
    out = torch.empty(H, W, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(_mask_lib().w3d_mask_binarize(H, W, C, ptr(pix), ptr(out), stream_ptr(dev)))
    return out


def mask_iou_device(alpha, masks=None, thresh=0.5):
    """find_match's scoring on the device (reference run_3d_seg.py:127-163): pred = alpha > thresh.
    alpha (H,W) or (1,H,W) float32 on the GPU; masks None or (K,H,W) uint8 / bool / float (non-zero = inside).
    Returns (iou (K,) float64 tensor on the HOST, bbox (x_min, y_min, x_max, y_max) or None, pred pixel count) —
    one small device-to-host copy instead of the alpha image and K numpy passes.  IoU as utils/wheatgs_utils.py:94-103
    (0 when the union is empty), bbox as get_bbox_from_mask :45-53."""
    from ._lib import check, ptr, stream_ptr
    if not alpha.is_cuda:
        raise RuntimeError("mask_iou_device needs the alpha image on the GPU")
    dev = alpha.device
    a = alpha.detach().reshape(alpha.shape[-2], alpha.shape[-1]).to(torch.float32).contiguous()
    H, W = int(a.shape[0]), int(a.shape[1])
    K = 0
    m = None
    if masks is not None:
        m = masks.to(dev)
        m = (m != 0).to(torch.uint8).contiguous() if m.dtype != torch.uint8 else m.contiguous()
        if m.dim() == 2:
            m = m[None]
        if tuple(m.shape[-2:]) != (H, W):
            raise RuntimeError("masks must have the alpha image's height and width")
        K = int(m.shape[0])
    out = torch.empty(2 * K + 5, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(_mask_lib().w3d_mask_iou(H, W, K, ptr(a), float(thresh), ptr(m), ptr(out), stream_ptr(dev)))
    h = out.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    inter, union = h[0:2 * K:2], h[1:2 * K:2]
    iou = torch.from_numpy(np.where(union > 0, inter / np.maximum(union, 1), 0.0))
    n_pred = int(h[2 * K + 4])
    bbox = None if n_pred == 0 else (int(h[2 * K]), int(h[2 * K + 1]), int(h[2 * K + 2]), int(h[2 * K + 3]))
    return iou, bbox, n_pred
