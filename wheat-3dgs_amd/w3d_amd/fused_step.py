"""The fused training-step path: the rasterizer called on the
PRE-ACTIVATION parameters of GaussianModel, gradients written straight into the flat gradient
bucket, densification statistics updated in the same backward kernel.  No autograd graph, no
activation / cat / split / accumulate kernels: PyTorch only owns the memory and the stream.

Same arithmetic as render() + loss.backward() of the drop-in path (reference
gaussian_renderer/__init__.py:22-106 followed by autograd through scene/gaussian_model.py:101-121);
tests/test_gpu_fused.py checks the two paths against each other.
"""
import ctypes
import math
import os

import torch

from ._lib import W3DView, check, lib, ptr, stream_ptr
from .rasterizer import GaussianRasterizationSettings, _View

_vp, _i32 = ctypes.c_void_p, ctypes.c_int32


class W3DRawParams(ctypes.Structure):
    _fields_ = [(n, _vp) for n in ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")]


class W3DRawGrads(ctypes.Structure):
    _fields_ = [(n, _vp) for n in ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")]


class W3DDensifyStats(ctypes.Structure):
    _fields_ = [(n, _vp) for n in ("dL_dmeans2D", "grad2d_norm", "radii", "xyz_gradient_accum", "denom", "max_radii2D")]


lib.w3d_forward_stage1_raw.argtypes = [ctypes.POINTER(W3DView), _i32, ctypes.POINTER(W3DRawParams), _vp, _vp, _vp, _vp, _vp]
lib.w3d_forward_stage1_raw.restype = ctypes.c_int
lib.w3d_backward_raw.argtypes = [ctypes.POINTER(W3DView), _i32, ctypes.POINTER(W3DRawParams), _vp, _vp, _vp, _vp, _vp,
                                 ctypes.POINTER(W3DRawGrads), ctypes.POINTER(W3DDensifyStats), _vp, _vp]
lib.w3d_backward_raw.restype = ctypes.c_int


class W3DAdamFused(ctypes.Structure):
    _fields_ = [("exp_avg", W3DRawGrads), ("exp_avg_sq", W3DRawGrads), ("lr", ctypes.c_float * 6), ("skip", _i32 * 6),
                ("beta1", ctypes.c_float), ("beta2", ctypes.c_float), ("eps", ctypes.c_float),
                ("bias_correction1", ctypes.c_float), ("bias_correction2", ctypes.c_float)]


lib.w3d_backward_raw_adam.argtypes = [ctypes.POINTER(W3DView), _i32, ctypes.POINTER(W3DRawGrads), _vp, _vp, _vp, _vp, _vp,
                                      ctypes.POINTER(W3DAdamFused), ctypes.POINTER(W3DDensifyStats), _vp, _vp]
lib.w3d_backward_raw_adam.restype = ctypes.c_int
_BLOCK_ORDER = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
lib.w3d_flash_reblend.argtypes = [ctypes.POINTER(W3DView), _i32, _vp, _vp, ctypes.c_uint64, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp]
lib.w3d_flash_reblend.restype = ctypes.c_int
lib.w3d_backward_raw_lowrank.argtypes = [ctypes.POINTER(W3DView), _i32, ctypes.POINTER(W3DRawParams), _vp, _vp, _vp, _vp, _vp,
                                         ctypes.POINTER(W3DRawGrads), _vp, ctypes.POINTER(W3DDensifyStats), _vp, _vp]
lib.w3d_backward_raw_lowrank.restype = ctypes.c_int
lib.w3d_backward_blend_dcolor.argtypes = [ctypes.POINTER(W3DView), _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
lib.w3d_backward_blend_dcolor.restype = ctypes.c_int
lib.w3d_sh_adam_lowrank.argtypes = [_i32, _i32, _i32] + [_vp] * 9 + [ctypes.c_float, ctypes.c_float, _i32, _i32] + \
    [ctypes.c_float] * 5 + [_vp]
lib.w3d_sh_adam_lowrank.restype = ctypes.c_int
GEO_BLOCKS = ("xyz", "opacity", "scaling", "rotation")      # their gradients are all-reduced as they are (11 floats)
SH_BLOCKS = ("f_dc", "f_rest")                               # rebuilt on every rank from the exchanged dL/dRGB


def _raw_params(model):
    p = W3DRawParams()
    p.xyz, p.f_dc, p.f_rest = model._xyz.data_ptr(), model._features_dc.data_ptr(), model._features_rest.data_ptr()
    p.opacity, p.scaling, p.rotation = model._opacity.data_ptr(), model._scaling.data_ptr(), model._rotation.data_ptr()
    return p


class ListCapacity:
    """Speculative sizing of the per-tile list buffer so that the forward needs NO host sync.

    The list length R (= num_rendered) is only known on the device after stage 1.  The synchronous path
    copies it to the host and waits (one pipeline bubble per view).  Here the list is allocated from the
    largest R seen so far (x `slack`); stage 2 is enqueued immediately; R travels to pinned host memory
    with an async copy + event, and `finish()` — called after the backward has been enqueued, so the GPU
    always has work queued — reports whether the guess held.  The fill kernel never writes past the
    capacity it was given; on overflow the caller grows the capacity and repeats the view."""

    def __init__(self, slack=1.25):
        self.slack = slack
        self.known = 0          # largest R observed

    def guess(self):
        return int(self.known * self.slack) + 1024 if self.known else 0

    def observe(self, R):
        self.known = max(self.known, int(R))


_capacity = ListCapacity()

# Depth-layered binning (w3d_view.depth_layers = 2) in the asynchronous forward: bin and blend the front
# ~28 % of the depth-ordered Gaussians, then only the tiles that are still open.  Identical outputs.
# OFF by default: it only pays when the global depth order follows the per-tile order (fronto-parallel
# views).  On the benchmark's tilted overhead cameras the depth gradient across the image is as large as
# the slab is thick, the front layer closes only the near side of the image, and the second pass costs
# more than it saves (measured 269 vs 319 iters/s).
DEPTH_LAYERS = os.environ.get("W3D_DEPTH_LAYERS", "0") == "2"


def finish(handle):
    """Wait for the counters of an asynchronous forward; True if the list capacity sufficed."""
    pend = handle.get("pending")
    if pend is None:
        return True
    pinned, ev = pend
    ev.synchronize()
    handle["num_visible"], handle["num_rendered"] = int(pinned[0]), int(pinned[1])
    handle["suspect_tiles"] = int(pinned[8])      # tiles whose depth-cut list ended before they saturated
    handle["pending"] = None
    if handle["suspect_tiles"] == 0 or handle["depth_cut"] is None:
        _capacity.observe(handle["num_rendered"])
    return handle["num_rendered"] <= handle["capacity"] and handle["suspect_tiles"] == 0


def render_raw(cam, model, bg_color, scaling_modifier=1.0, sync=True, depth_cut=None, want_cut=False, flash=None):
    """Forward on the raw parameters.  Returns the dict of render() (minus viewspace_points) plus a
    `handle` for backward_raw().  sync=False: no host synchronisation (see ListCapacity); the caller
    must call finish(handle) before trusting the outputs.

    depth_cut (T,) float tensor from an earlier visit of this camera: per-tile depth beyond which
    nothing is binned (w3d_view.tile_depth_cut) — finish() is False when a tile turned out to need more
    (repeat the view with depth_cut=None).  want_cut: handle["depth_cut_out"] receives the cuts for the
    next visit."""
    dev = model.flat.device
    if not model.flat.is_cuda:
        raise RuntimeError("the fused step needs the model on the GPU; there is no CPU path")
    P = model.num_points
    H, W = int(cam.image_height), int(cam.image_width)
    s = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), bg_color,
                                      scaling_modifier, cam.world_view_transform, cam.full_proj_transform,
                                      model.active_sh_degree, cam.camera_center, False, False)
    view = _View(s, (model.max_sh_degree + 1) ** 2, dev)
    T = ((W + 15) // 16) * ((H + 15) // 16)
    cut_out = None
    if depth_cut is not None:
        if sync:
            raise ValueError("depth cuts need the asynchronous forward (sync=False)")
        assert depth_cut.numel() == T and depth_cut.dtype == torch.float32 and depth_cut.is_cuda
        view.c.tile_depth_cut = depth_cut.data_ptr()
    if want_cut:
        cut_out = torch.empty(T, dtype=torch.float32, device=dev)
        view.c.tile_depth_cut_out = cut_out.data_ptr()
    prm = _raw_params(model)
    with torch.cuda.device(dev):
        stream = stream_ptr(dev)
        sb, tb = ctypes.c_uint64(), ctypes.c_uint64()
        check(lib.w3d_forward_sizes(P, H, W, ctypes.byref(sb), ctypes.byref(tb)))
        state = torch.empty(sb.value, dtype=torch.uint8, device=dev)
        scratch = torch.empty(tb.value, dtype=torch.uint8, device=dev)
        radii = torch.empty(P, dtype=torch.int32, device=dev)
        guess = 0 if sync else _capacity.guess()
        pending = None
        if guess == 0:
            counts = (ctypes.c_uint32 * 2)()
            check(lib.w3d_forward_stage1_raw(ctypes.byref(view.c), P, ctypes.byref(prm), ptr(radii), ptr(state),
                                             ptr(scratch), ctypes.cast(counts, _vp), stream))
            R, V = int(counts[1]), int(counts[0])
            _capacity.observe(R)
        else:
            view.c.depth_layers = 2 if DEPTH_LAYERS else 0
            check(lib.w3d_forward_stage1_raw(ctypes.byref(view.c), P, ctypes.byref(prm), ptr(radii), ptr(state),
                                             ptr(scratch), None, stream))
            R, V = guess, -1
        plist = torch.empty(max(R, 1), dtype=torch.int32, device=dev)
        color = torch.empty(3, H, W, dtype=torch.float32, device=dev)
        depth = torch.empty(1, H, W, dtype=torch.float32, device=dev)
        alpha = torch.empty(1, H, W, dtype=torch.float32, device=dev)
        gt_mask = used_count = contrib_num = proj_xy = gs_depth = None
        num_obj = 0
        if flash is not None:
            # FlashSplat outputs (reference gaussian_renderer/__init__.py:194-204): forward only
            num_obj = int(flash["num_obj"])
            if num_obj < 1:
                raise RuntimeError("num_obj must be >= 1")
            gt_mask = flash.get("gt_mask")
            if gt_mask is not None:
                gt_mask = gt_mask.detach().to(device=dev, dtype=torch.float32).contiguous()
                if tuple(gt_mask.shape[-2:]) != (H, W) or gt_mask.numel() != H * W:
                    raise RuntimeError("gt_mask must have dimensions (image_height, image_width)")
            used_count = torch.zeros(num_obj + 1, P, dtype=torch.float32, device=dev)
            contrib_num = torch.empty(H, W, dtype=torch.int32, device=dev)
            proj_xy = torch.empty(P, 2, dtype=torch.float32, device=dev)
            gs_depth = torch.empty(P, dtype=torch.float32, device=dev)
        check(lib.w3d_forward_stage2(ctypes.byref(view.c), P, ptr(state), ptr(scratch), ptr(plist), ctypes.c_uint64(R),
                                     ptr(color), ptr(depth), ptr(alpha), ptr(gt_mask), num_obj, ptr(used_count),
                                     ptr(contrib_num), ptr(proj_xy), ptr(gs_depth), stream))
        if guess != 0:
            # counters (offset 0 of the state) travel to pinned memory AFTER the blend so that the depth-cut
            # verdict is included; finish() waits on the event
            pinned = torch.empty(16, dtype=torch.int32, pin_memory=True)
            pinned.copy_(state[:64].view(torch.int32), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            pending = (pinned, ev)
    handle = dict(view=view, P=P, state=state, point_list=plist, radii=radii, num_rendered=R, num_visible=V,
                  capacity=R, pending=pending, depth_cut=depth_cut, depth_cut_out=cut_out, suspect_tiles=0)
    out = {"render": color, "radii": radii, "depth": depth, "alpha": alpha, "handle": handle}
    if flash is not None:
        out.update(contrib_num=contrib_num, used_count=used_count, proj_xy=proj_xy, gs_depth=gs_depth)
    return out


def backward_raw(model, handle, dL_dimage, dL_ddepth=None, dL_dalpha=None, update_stats=False, want_norm=False,
                 want_means2D=False):
    """Backward into model.flat_grad (OVERWRITTEN).  update_stats: add_densification_stats and the
    max_radii2D update happen inside the kernel (single-GPU step).  want_norm: also return the
    per-Gaussian ||dL/dmean2D|| (the view-parallel exchange needs it before reduction)."""
    dev = model.flat.device
    P, view = handle["P"], handle["view"]
    if P != model.num_points:
        raise RuntimeError("model was resized between forward and backward")
    prm = _raw_params(model)
    g = W3DRawGrads()
    for n, attr in (("xyz", "xyz"), ("f_dc", "f_dc"), ("f_rest", "f_rest"), ("opacity", "opacity"),
                    ("scaling", "scaling"), ("rotation", "rotation")):
        setattr(g, n, model._p[attr].grad.data_ptr())
    st = W3DDensifyStats()
    gnorm = torch.empty(P, dtype=torch.float32, device=dev) if want_norm else None
    m2d = torch.empty(P, 3, dtype=torch.float32, device=dev) if want_means2D else None
    st.dL_dmeans2D = None if m2d is None else m2d.data_ptr()
    st.grad2d_norm = None if gnorm is None else gnorm.data_ptr()
    st.radii = handle["radii"].data_ptr()
    if update_stats:
        st.xyz_gradient_accum, st.denom = model.xyz_gradient_accum.data_ptr(), model.denom.data_ptr()
        st.max_radii2D = model.max_radii2D.data_ptr()
    with torch.cuda.device(dev):
        sb = ctypes.c_uint64()
        check(lib.w3d_backward_sizes(P, ctypes.byref(sb)))
        scratch = torch.empty(sb.value, dtype=torch.uint8, device=dev)
        check(lib.w3d_backward_raw(ctypes.byref(view.c), P, ctypes.byref(prm), ptr(handle["state"]),
                                   ptr(handle["point_list"]), ptr(dL_dimage.contiguous()), ptr(dL_ddepth), ptr(dL_dalpha),
                                   ctypes.byref(g), ctypes.byref(st), ptr(scratch), stream_ptr(dev)))
    return gnorm, m2d


def backward_raw_adam(model, handle, dL_dimage, skip=(), want_norm=True, update_stats=False):
    """Backward with the optimizer fused in (single GPU): the kernel that finishes each Gaussian's gradient applies
    FlatAdam's update to the parameter blocks and both moments in place, so the 59*P gradient bucket is neither written
    nor read back (model.flat_grad is left untouched).  If the forward overflowed its speculative list buffer the kernel
    updates NOTHING; the caller checks finish(handle) and only then calls model.optimizer.note_fused_step().
    Returns the per-Gaussian ||dL/dmean2D|| (0 for culled) when want_norm."""
    import math
    dev = model.flat.device
    P, view = handle["P"], handle["view"]
    opt = model.optimizer
    if P != model.num_points:
        raise RuntimeError("model was resized between forward and backward")
    prm, ad = W3DRawGrads(), W3DAdamFused()
    sl = model.block_slices()
    b1, b2 = opt.betas
    t = opt.step_count + 1
    for i, n in enumerate(_BLOCK_ORDER):
        a, _ = sl[n]
        setattr(prm, n, model._p[n].data_ptr())
        setattr(ad.exp_avg, n, opt.exp_avg.data_ptr() + 4 * a)
        setattr(ad.exp_avg_sq, n, opt.exp_avg_sq.data_ptr() + 4 * a)
        ad.lr[i] = float(opt.lrs[n])
        ad.skip[i] = int(n in skip)
    ad.beta1, ad.beta2, ad.eps = float(b1), float(b2), float(opt.eps)
    ad.bias_correction1, ad.bias_correction2 = 1.0 - math.pow(b1, t), 1.0 - math.pow(b2, t)
    st = W3DDensifyStats()
    gnorm = torch.empty(P, dtype=torch.float32, device=dev) if want_norm else None
    st.grad2d_norm = None if gnorm is None else gnorm.data_ptr()
    st.radii = handle["radii"].data_ptr()
    if update_stats:     # add_densification_stats + max_radii2D inside the kernel, applied only if the view is final
        st.xyz_gradient_accum, st.denom = model.xyz_gradient_accum.data_ptr(), model.denom.data_ptr()
        st.max_radii2D = model.max_radii2D.data_ptr()
    with torch.cuda.device(dev):
        sb = ctypes.c_uint64()
        check(lib.w3d_backward_sizes(P, ctypes.byref(sb)))
        scratch = torch.empty(sb.value, dtype=torch.uint8, device=dev)
        check(lib.w3d_backward_raw_adam(ctypes.byref(view.c), P, ctypes.byref(prm), ptr(handle["state"]),
                                        ptr(handle["point_list"]), ptr(dL_dimage.contiguous()), None, None,
                                        ctypes.byref(ad), ctypes.byref(st), ptr(scratch), stream_ptr(dev)))
    return gnorm


def backward_blend_dcolor(model, handle, dL_dimage):
    """First half of backward_raw_lowrank: the blend backward and the (P,3) clamp-masked dL/dRGB it implies — everything the
    other ranks need for the SH gradient — so that its all-gather can be issued before the per-Gaussian backward is even
    enqueued.  Call backward_raw_lowrank(model, handle, None) afterwards."""
    dev = model.flat.device
    P, view = handle["P"], handle["view"]
    if P != model.num_points:
        raise RuntimeError("model was resized between forward and backward")
    dcol = torch.empty(P, 3, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        sb = ctypes.c_uint64()
        check(lib.w3d_backward_sizes(P, ctypes.byref(sb)))
        scratch = torch.empty(sb.value, dtype=torch.uint8, device=dev)
        check(lib.w3d_backward_blend_dcolor(ctypes.byref(view.c), P, ptr(handle["state"]), ptr(handle["point_list"]),
                                            ptr(dL_dimage.contiguous()), None, None, ptr(dcol), ptr(scratch), stream_ptr(dev)))
    handle["bwd_scratch"] = scratch
    return dcol


def backward_raw_lowrank(model, handle, dL_dimage, want_norm=True):
    """Backward of the view-parallel step: gradients of the geometry blocks (xyz, opacity, scaling, rotation) into
    model.flat_grad, and instead of the 48-float SH gradient rows the (P,3) clamp-masked dL/dRGB per Gaussian — the SH
    gradient of one view is basis(view direction) x dL/dRGB, so that is all the other ranks need (sh_adam_lowrank).
    Returns (||dL/dmean2D|| or None, dcolor (P,3))."""
    dev = model.flat.device
    P, view = handle["P"], handle["view"]
    if P != model.num_points:
        raise RuntimeError("model was resized between forward and backward")
    prm = _raw_params(model)
    g = W3DRawGrads()
    for n in GEO_BLOCKS:
        setattr(g, n, model._p[n].grad.data_ptr())
    st = W3DDensifyStats()
    gnorm = torch.empty(P, dtype=torch.float32, device=dev) if want_norm else None
    st.grad2d_norm = None if gnorm is None else gnorm.data_ptr()
    st.radii = handle["radii"].data_ptr()
    with torch.cuda.device(dev):
        if dL_dimage is None:
            # second half: backward_blend_dcolor already ran the blend backward into the handle's scratch
            scratch = handle.pop("bwd_scratch")
            check(lib.w3d_backward_raw_lowrank(ctypes.byref(view.c), P, ctypes.byref(prm), ptr(handle["state"]),
                                               ptr(handle["point_list"]), None, None, None, ctypes.byref(g), None,
                                               ctypes.byref(st), ptr(scratch), stream_ptr(dev)))
            return gnorm, None
        dcol = torch.empty(P, 3, dtype=torch.float32, device=dev)
        sb = ctypes.c_uint64()
        check(lib.w3d_backward_sizes(P, ctypes.byref(sb)))
        scratch = torch.empty(sb.value, dtype=torch.uint8, device=dev)
        check(lib.w3d_backward_raw_lowrank(ctypes.byref(view.c), P, ctypes.byref(prm), ptr(handle["state"]),
                                           ptr(handle["point_list"]), ptr(dL_dimage.contiguous()), None, None,
                                           ctypes.byref(g), ptr(dcol), ctypes.byref(st), ptr(scratch), stream_ptr(dev)))
    return gnorm, dcol


def sh_adam_lowrank(model, dcolor_all, campos_all, skip=(), rows=None):
    """Adam step of f_dc / f_rest from the colour gradients of ALL views of this iteration (dcolor_all (V,P,3), campos_all
    (V,3)): dL/dSH[k] = sum_v basis_k(normalize(xyz - campos_v)) * dcolor_v, summed in view order.  The optimizer's step
    counter must already be advanced for this iteration.  GPU: csrc sh_adam_lowrank_kernel, in place; CPU (host-logic
    tests): the same formula with torch ops through FlatAdam's CPU path."""
    opt = model.optimizer
    P, V = model.num_points, int(dcolor_all.shape[0])
    r0, r1 = (0, P) if rows is None else rows          # rows=(r0, r1): dcolor_all holds only these Gaussians
    n = r1 - r0
    assert dcolor_all.shape[1] == n
    deg = int(model.active_sh_degree)
    if model.max_sh_degree != 3:
        raise RuntimeError("the low-rank exchange is written for 16 SH coefficients")
    if not model.flat.is_cuda:
        from .sh import sh_basis
        if rows is not None:
            raise RuntimeError("row chunks are a GPU-path feature")
        xyz = model._p["xyz"].detach()
        grad = torch.zeros(P, 16, 3, dtype=torch.float32)
        for v in range(V):                                   # view order, as in the kernel
            dirs = xyz - campos_all[v][None]
            dirs = dirs / dirs.norm(dim=1, keepdim=True)
            basis = sh_basis(deg, dirs)                      # (P, (deg+1)^2)
            grad[:, :basis.shape[1]] += basis[:, :, None] * dcolor_all[v][:, None, :]
        model._p["f_dc"].grad.copy_(grad[:, :1])
        model._p["f_rest"].grad.copy_(grad[:, 1:])
        opt.step(only=SH_BLOCKS, skip=skip, advance=False)
        return
    sl = model.block_slices()
    bc1, bc2 = opt.bias_corrections()
    b1, b2 = opt.betas
    m, v = opt.exp_avg, opt.exp_avg_sq
    (a_dc, _), (a_rest, _) = sl["f_dc"], sl["f_rest"]
    dev = model.flat.device
    d_all = dcolor_all.contiguous()
    cp = campos_all.to(device=dev, dtype=torch.float32).contiguous()
    with torch.cuda.device(dev):
        check(lib.w3d_sh_adam_lowrank(n, V, deg, ptr(cp), model._p["xyz"].data_ptr() + 12 * r0, ptr(d_all),
                                      model._p["f_dc"].data_ptr() + 12 * r0, model._p["f_rest"].data_ptr() + 180 * r0,
                                      m.data_ptr() + 4 * (a_dc + 3 * r0), v.data_ptr() + 4 * (a_dc + 3 * r0),
                                      m.data_ptr() + 4 * (a_rest + 45 * r0), v.data_ptr() + 4 * (a_rest + 45 * r0),
                                      float(opt.lrs["f_dc"]),
                                      float(opt.lrs["f_rest"]), int("f_dc" in skip), int("f_rest" in skip), float(b1),
                                      float(b2), float(opt.eps), float(bc1), float(bc2), stream_ptr(dev)))


def flash_reblend(pkg, gt_mask, num_obj):
    """FlashSplat contribution counts of ANOTHER label image for a view already rendered by render_raw(..., flash=...):
    only the blend runs again, on the state and per-tile lists of that forward (run_3d_seg.py renders every view once per
    object mask — preprocessing, depth sort and binning do not depend on the mask).  Returns (used_count (num_obj+1, P),
    contrib_num (H, W)); pkg["render"] / ["depth"] / ["alpha"] are rewritten with the same values."""
    h = pkg["handle"]
    view, P = h["view"], h["P"]
    color = pkg["render"]
    dev = color.device
    H, W = int(color.shape[-2]), int(color.shape[-1])
    num_obj = int(num_obj)
    if num_obj < 1:
        raise RuntimeError("num_obj must be >= 1")
    if h.get("pending") is not None:
        raise RuntimeError("finish() the forward before re-blending it")
    gt = gt_mask.detach().to(device=dev, dtype=torch.float32).contiguous()
    if tuple(gt.shape[-2:]) != (H, W) or gt.numel() != H * W:
        raise RuntimeError("gt_mask must have dimensions (image_height, image_width)")
    used_count = torch.zeros(num_obj + 1, P, dtype=torch.float32, device=dev)
    contrib_num = torch.empty(H, W, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(lib.w3d_flash_reblend(ctypes.byref(view.c), P, ptr(h["state"]), ptr(h["point_list"]),
                                    ctypes.c_uint64(h["capacity"]), ptr(color), ptr(pkg["depth"]), ptr(pkg["alpha"]), ptr(gt),
                                    num_obj, ptr(used_count), ptr(contrib_num), stream_ptr(dev)))
    return used_count, contrib_num
