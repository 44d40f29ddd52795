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

import torch

from ._lib import W3DView, check, lib, ptr, stream_ptr
from .rasterizer import (GaussianRasterizationSettings, adapt_list_share, list_share_of, _View, backward_scratch,
                         list_capacity, scratch_done)

_vp, _i32 = ctypes.c_void_p, ctypes.c_int32


class W3DRawParams(ctypes.Structure):
    _fields_ = [(n, _vp) for n in ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")]


class W3DRawGrads(ctypes.Structure):
    _fields_ = [(n, _vp) for n in ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")]


class W3DDensifyStats(ctypes.Structure):
    _fields_ = [(n, _vp) for n in ("dL_dmeans2D", "grad2d_norm", "radii", "xyz_gradient_accum", "denom", "max_radii2D")]


lib.w3d_forward_stage1_raw.argtypes = [ctypes.POINTER(W3DView), _i32, ctypes.POINTER(W3DRawParams), _vp, _vp, _vp, _vp, _vp]
lib.w3d_forward_stage1_raw.restype = ctypes.c_int
lib.w3d_forward_stage1_raw_subset.argtypes = [ctypes.POINTER(W3DView), _i32, ctypes.POINTER(W3DRawParams), _vp, _vp, _vp, _vp, _vp, _vp]
lib.w3d_forward_stage1_raw_subset.restype = ctypes.c_int
lib.w3d_backward_raw.argtypes = [ctypes.POINTER(W3DView), _i32, ctypes.POINTER(W3DRawParams), _vp, _vp, _vp, _vp, _vp,
                                 ctypes.POINTER(W3DRawGrads), ctypes.POINTER(W3DDensifyStats), _vp, _vp]
lib.w3d_backward_raw.restype = ctypes.c_int


class W3DAdamFused(ctypes.Structure):
    _fields_ = [("exp_avg", W3DRawGrads), ("exp_avg_sq", W3DRawGrads), ("lr", ctypes.c_float * 6), ("skip", _i32 * 6),
                ("beta1", ctypes.c_float), ("beta2", ctypes.c_float), ("eps", ctypes.c_float),
                ("bias_correction1", ctypes.c_float * 6), ("bias_correction2", ctypes.c_float * 6)]


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
lib.w3d_pack_gradient_rows.argtypes = [_i32, _vp, ctypes.POINTER(W3DRawGrads), _vp, ctypes.c_float, _vp, ctypes.c_uint32, _vp, _vp]
lib.w3d_pack_gradient_rows.restype = ctypes.c_int
lib.w3d_backward_raw_rows.argtypes = [ctypes.POINTER(W3DView), _i32, ctypes.POINTER(W3DRawParams), _vp, _vp, _vp, _vp, _vp, ctypes.c_float,
                                      _vp, ctypes.c_uint32, _vp, _vp, _vp]
lib.w3d_backward_raw_rows.restype = ctypes.c_int
lib.w3d_apply_gradient_rows.argtypes = [_i32, _vp, _vp, ctypes.c_uint32, _vp, ctypes.POINTER(W3DRawGrads), _vp, _vp]
lib.w3d_apply_gradient_rows.restype = ctypes.c_int
lib.w3d_index_gradient_rows.argtypes = [_i32, _i32, _vp, _vp, ctypes.c_uint32, _vp, _vp, _vp]
lib.w3d_index_gradient_rows.restype = ctypes.c_int
lib.w3d_rows_norm_sum.argtypes = [_i32, _i32, _vp, ctypes.c_uint32, _vp, _vp, _vp, _vp]
lib.w3d_rows_norm_sum.restype = ctypes.c_int
lib.w3d_rows_norm_accumulate.argtypes = [_i32, _i32, _vp, ctypes.c_uint32, _vp, _vp, _vp, _vp]
lib.w3d_rows_norm_accumulate.restype = ctypes.c_int
lib.w3d_track_visibility.argtypes = [_i32, _vp, _vp, _vp, _vp]
lib.w3d_track_visibility.restype = ctypes.c_int
lib.w3d_rows_adam.argtypes = [_i32, _i32, _i32, _vp, _vp, ctypes.c_uint32, _vp, _vp, ctypes.POINTER(W3DRawGrads),
                              ctypes.POINTER(W3DAdamFused), _vp]
lib.w3d_rows_adam.restype = ctypes.c_int
ROW_FLOATS = 16          # {index bits, ||dL/dmean2D||, dL/dRGB[3], d xyz[3], d opacity, d scaling[3], d rotation[4]}
GEO_BLOCKS = ("xyz", "opacity", "scaling", "rotation")      # their gradients are all-reduced as they are (11 floats)
SH_BLOCKS = ("f_dc", "f_rest")                               # rebuilt on every rank from the exchanged dL/dRGB


def _raw_params(model):
    p = W3DRawParams()
    p.xyz, p.f_dc, p.f_rest = model._xyz.data_ptr(), model._features_dc.data_ptr(), model._features_rest.data_ptr()
    p.opacity, p.scaling, p.rotation = model._opacity.data_ptr(), model._scaling.data_ptr(), model._rotation.data_ptr()
    return p


def finish(handle):
    """Wait for the counters of an asynchronous forward; True if the list capacity sufficed."""
    pend = handle.get("pending")
    if pend is None:
        return True
    pinned, ev = pend
    ev.synchronize()
    handle["num_visible"], handle["num_rendered"] = int(pinned[0]) & 0xFFFFFFFF, int(pinned[1]) & 0xFFFFFFFF
    handle["pending"] = None
    handle["cap"].observe(handle["num_rendered"])
    return handle["num_rendered"] <= handle["capacity"]


def render_raw(cam, model, bg_color, scaling_modifier=1.0, sync=True, flash=None, used_mask=None, color_only=False):
    """Forward on the raw parameters.  Returns the dict of render() (minus viewspace_points) plus a
    `handle` for backward_raw().  sync=False: no host synchronisation (rasterizer.ListCapacity); the caller
    must call finish(handle) before trusting the outputs, and repeat the view when it returns False.
    used_mask: (P,) bool — only these Gaussians are rendered (flashsplat_render(used_mask=...), reference
    gaussian_renderer/__init__.py:151-156,168-170,186-187); the others are culled inside the preprocess kernel, so no
    subset of the parameter blocks is gathered.  Per-Gaussian outputs keep P rows (zeros on the rows left out).
    The model's `tile_cull` / `deterministic` attributes select the two optional behaviours (rasterizer.py header).
    color_only: the depth and alpha images are not wanted (Trainer.step_fused: reference train_vanilla_3dgs.py:74-80 feeds only
    `render` to the loss) — `depth` / `alpha` of the result are None and the blend skips the two channels."""
    dev = model.flat.device
    if not model.flat.is_cuda:
        raise RuntimeError("the fused step needs the model on the GPU; there is no CPU path")
    P = model.num_points
    H, W = int(cam.image_height), int(cam.image_width)
    s = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), bg_color,
                                      scaling_modifier, cam.world_view_transform, cam.full_proj_transform,
                                      model.active_sh_degree, cam.camera_center, False, False,
                                      bool(getattr(model, "tile_cull", True)), bool(getattr(model, "deterministic", False)),
                                      list_share_of(model))
    view = _View(s, (model.max_sh_degree + 1) ** 2, dev)
    prm = _raw_params(model)
    um = None
    if used_mask is not None:
        if used_mask.dtype != torch.bool or used_mask.dim() != 1 or used_mask.shape[0] != P or used_mask.device != dev:
            raise RuntimeError("used_mask must be a (num_points,) bool tensor on the model's device")
        um = used_mask.contiguous()
    with torch.cuda.device(dev):
        stream = stream_ptr(dev)
        sb, tb = ctypes.c_uint64(), ctypes.c_uint64()
        check(lib.w3d_forward_sizes(P, H, W, ctypes.byref(sb), ctypes.byref(tb)))
        state = torch.empty(sb.value, dtype=torch.uint8, device=dev)
        scratch = torch.empty(tb.value, dtype=torch.uint8, device=dev)
        radii = torch.empty(P, dtype=torch.int32, device=dev)
        # (a subset render says nothing about the list length of a full one, and vice versa: separate hints)
        cap = list_capacity(model if um is None else used_mask, H, W)
        guess = 0 if sync else cap.guess()
        pending = None
        if guess == 0:
            counts = (ctypes.c_uint32 * 2)()
            check(lib.w3d_forward_stage1_raw_subset(ctypes.byref(view.c), P, ctypes.byref(prm), ptr(um), ptr(radii),
                                                    ptr(state), ptr(scratch), ctypes.cast(counts, _vp), stream))
            R, V = int(counts[1]), int(counts[0])
            cap.observe(R)
        else:
            check(lib.w3d_forward_stage1_raw_subset(ctypes.byref(view.c), P, ctypes.byref(prm), ptr(um), ptr(radii),
                                                    ptr(state), ptr(scratch), None, stream))
            R, V = guess, -1
            # the counters start their way to pinned host memory right after stage 1, BEFORE stage 2 is enqueued:
            # whoever waits for them (finish()) is released while the GPU is still busy with stage 2
            pinned = torch.empty(2, dtype=torch.int32, pin_memory=True)
            pinned.copy_(state[:8].view(torch.int32), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            pending = (pinned, ev)
        plist = torch.empty(max(R, 1), dtype=torch.int32, device=dev)
        color = torch.empty(3, H, W, dtype=torch.float32, device=dev)
        depth = alpha = None
        if not (color_only and flash is None):
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
            # the kernel ADDS into used_count: a caller that sums over views (run_3d_seg.py:95-97 all_counts += used_count)
            # hands its running total in and saves a (num_obj+1, P) allocation, memset and add per view — 2.4 GB each at
            # 300 labels x 2 M Gaussians (eval_wheatgs.py:99-105)
            used_count = flash.get("accumulate_into")
            if used_count is None:
                used_count = torch.zeros(num_obj + 1, P, dtype=torch.float32, device=dev)
            elif tuple(used_count.shape) != (num_obj + 1, P) or used_count.dtype != torch.float32 or not used_count.is_contiguous() \
                    or used_count.device != dev:
                raise RuntimeError("accumulate_into must be a contiguous float32 (num_obj+1, P) tensor on the model's device")
            contrib_num = torch.empty(H, W, dtype=torch.int32, device=dev)
            proj_xy = torch.empty(P, 2, dtype=torch.float32, device=dev)
            gs_depth = torch.empty(P, dtype=torch.float32, device=dev)
        check(lib.w3d_forward_stage2(ctypes.byref(view.c), P, ptr(state), ptr(scratch), ptr(plist), ctypes.c_uint64(R),
                                     ptr(color), ptr(depth), ptr(alpha), ptr(gt_mask), num_obj, ptr(used_count),
                                     ptr(contrib_num), ptr(proj_xy), ptr(gs_depth), stream))
    handle = dict(view=view, P=P, state=state, point_list=plist, radii=radii, num_rendered=R, num_visible=V,
                  capacity=R, pending=pending, cap=cap)
    if getattr(model, "debug_keep_scratch", False):       # (rasterizer.debug_depth_buckets reads the sort's grid from it)
        handle["scratch"] = scratch
    out = {"render": color, "radii": radii, "depth": depth, "alpha": alpha, "handle": handle}
    if flash is not None:
        out.update(contrib_num=contrib_num, used_count=used_count, proj_xy=proj_xy, gs_depth=gs_depth)
    return out


def backward_raw(model, handle, dL_dimage, dL_ddepth=None, dL_dalpha=None, update_stats=False, want_norm=False,
                 want_means2D=False, into=None):
    """Backward into model.flat_grad (OVERWRITTEN) — or into `into`, a flat buffer of the same layout.  update_stats:
    add_densification_stats and the max_radii2D update happen inside the kernel (single-GPU step).  want_norm: also
    return the per-Gaussian ||dL/dmean2D|| (the view-parallel exchange needs it before reduction)."""
    dev = model.flat.device
    P, view = handle["P"], handle["view"]
    if P != model.num_points:
        raise RuntimeError("model was resized between forward and backward")
    prm = _raw_params(model)
    g = W3DRawGrads()
    sl = model.block_slices()
    for n in _BLOCK_ORDER:
        setattr(g, n, (model.flat_grad if into is None else into).data_ptr() + 4 * sl[n][0])
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
        scratch = backward_scratch(view, P, handle["point_list"], dev, owner=model)
        check(lib.w3d_backward_raw(ctypes.byref(view.c), P, ctypes.byref(prm), ptr(handle["state"]),
                                   ptr(handle["point_list"]), ptr(dL_dimage.contiguous()), ptr(dL_ddepth), ptr(dL_dalpha),
                                   ctypes.byref(g), ctypes.byref(st), ptr(scratch), stream_ptr(dev)))
        scratch_done(view, model)
    return gnorm, m2d


class _RasterizeRawFn(torch.autograd.Function):
    """render() under autograd on the flat GaussianModel — what an unmodified train_vanilla_3dgs.py:73-80 runs through
    (render(), then loss.backward()).  The graph has ONE node between the six parameter tensors and the image: the
    kernels apply exp / sigmoid / normalize / the dc-rest split themselves and chain their derivatives, so none of the
    activation, cat and split kernels of reference scene/gaussian_model.py:101-121 (nor their autograd backward) is
    launched.  The backward writes the parameter gradients straight into the model's flat gradient bucket and hands
    autograd VIEWS of it: with .grad None (optimizer.zero_grad(set_to_none=True), train_vanilla_3dgs.py:115) the engine
    adopts them without a copy or an accumulation pass."""

    @staticmethod
    def forward(ctx, means2D, xyz, f_dc, f_rest, opacity, scaling, rotation, model, cam, bg, scaling_modifier):
        pkg = render_raw(cam, model, bg, scaling_modifier, sync=False)
        if not finish(pkg["handle"]):                 # list buffer too small: repeat with the exact size
            pkg = render_raw(cam, model, bg, scaling_modifier, sync=True)
        adapt_list_share(model, pkg["handle"])        # (speed only: which list grid the next renders of this model use)
        ctx.model, ctx.handle = model, pkg["handle"]
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(pkg["radii"])
        return pkg["render"], pkg["radii"], pkg["depth"], pkg["alpha"]

    @staticmethod
    def backward(ctx, g_color, g_radii, g_depth, g_alpha):
        model, handle = ctx.model, ctx.handle
        dev = model.flat.device
        H, W = handle["view"].c.image_height, handle["view"].c.image_width
        if g_color is None:
            g_color = torch.zeros(3, H, W, dtype=torch.float32, device=dev)
        f32 = lambda t: None if t is None else t.to(torch.float32).contiguous()  # noqa: E731
        # The kernel OVERWRITES its gradient buffers.  Straight into the bucket only when no gradient is being
        # accumulated there (every .grad None); otherwise into a temporary that autograd adds to the existing .grad.
        # And only ONCE per backward pass: two render() calls on one model feeding one loss.backward() (a multi-view loss)
        # both see .grad None — their nodes run before any AccumulateGrad — so the first claims the bucket and every
        # later node writes a temporary that the engine adds (zero_grad / step release the claim).
        direct = not getattr(model, "_bucket_claimed", False) and all(p.grad is None for p in model._p.values())
        if direct:
            model._bucket_claimed = True
        into = None if direct else torch.empty_like(model.flat_grad)
        _, m2d = backward_raw(model, handle, f32(g_color), f32(g_depth), f32(g_alpha), want_means2D=True, into=into)
        ctx.handle = None
        if direct:
            grads = {n: model.grad_view(n) for n in _BLOCK_ORDER}
        else:
            sl = model.block_slices()
            grads = {n: into[sl[n][0]:sl[n][1]].view(model._p[n].shape) for n in _BLOCK_ORDER}
        return (m2d, grads["xyz"], grads["f_dc"], grads["f_rest"], grads["opacity"], grads["scaling"], grads["rotation"],
                None, None, None, None)


def render_raw_autograd(cam, model, bg_color, scaling_modifier=1.0):
    """The dict of render() (reference gaussian_renderer/__init__.py:99-106) through _RasterizeRawFn."""
    xyz = model._p["xyz"]
    # (reference gaussian_renderer/__init__.py:33-37 builds a zeros tensor per call only to collect its .grad: the zeros
    #  are shared — nothing writes to them — and every call gets its own leaf over them)
    z = getattr(model, "_screenspace_zeros", None)
    if z is None or z.shape != xyz.shape or z.device != xyz.device:
        z = model._screenspace_zeros = torch.zeros_like(xyz)
    screenspace_points = z.detach().requires_grad_(True)
    p = model._p
    color, radii, depth, alpha = _RasterizeRawFn.apply(screenspace_points, p["xyz"], p["f_dc"], p["f_rest"], p["opacity"],
                                                       p["scaling"], p["rotation"], model, cam, bg_color, scaling_modifier)
    return {"render": color, "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii,
            "depth": depth, "alpha": alpha}


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
    for i, n in enumerate(_BLOCK_ORDER):
        a, _ = sl[n]
        setattr(prm, n, model._p[n].data_ptr())
        setattr(ad.exp_avg, n, opt.exp_avg.data_ptr() + 4 * a)
        setattr(ad.exp_avg_sq, n, opt.exp_avg_sq.data_ptr() + 4 * a)
        ad.lr[i] = float(opt.lrs[n])
        ad.skip[i] = int(n in skip)
        # the update this kernel applies is step t+1 of the block (note_fused_step advances the counters afterwards)
        ad.bias_correction1[i], ad.bias_correction2[i] = opt.bias_corrections(n, ahead=1)
    ad.beta1, ad.beta2, ad.eps = float(b1), float(b2), float(opt.eps)
    st = W3DDensifyStats()
    gnorm = torch.empty(P, dtype=torch.float32, device=dev) if want_norm else None
    st.grad2d_norm = None if gnorm is None else gnorm.data_ptr()
    st.radii = handle["radii"].data_ptr()
    if update_stats:     # add_densification_stats + max_radii2D inside the kernel, applied only if the view is final
        st.xyz_gradient_accum, st.denom = model.xyz_gradient_accum.data_ptr(), model.denom.data_ptr()
        st.max_radii2D = model.max_radii2D.data_ptr()
    with torch.cuda.device(dev):
        scratch = backward_scratch(view, P, handle["point_list"], dev, owner=model)
        check(lib.w3d_backward_raw_adam(ctypes.byref(view.c), P, ctypes.byref(prm), ptr(handle["state"]),
                                        ptr(handle["point_list"]), ptr(dL_dimage.contiguous()), None, None,
                                        ctypes.byref(ad), ctypes.byref(st), ptr(scratch), stream_ptr(dev)))
        scratch_done(view, model)
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
        scratch = backward_scratch(view, P, handle["point_list"], dev, owner=model)      # (clean again after the second half)
        check(lib.w3d_backward_blend_dcolor(ctypes.byref(view.c), P, ptr(handle["state"]), ptr(handle["point_list"]),
                                            ptr(dL_dimage.contiguous()), None, None, ptr(dcol), ptr(scratch), stream_ptr(dev)))
    handle["bwd_scratch"] = scratch
    kept = getattr(model, "_w3d_bwd_scratch", None) if view.c.records_kept_clean else None
    handle["bwd_scratch_token"] = None if kept is None else (kept, kept.generation)
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
        setattr(g, n, model.grad_view(n).data_ptr())
    st = W3DDensifyStats()
    gnorm = torch.empty(P, dtype=torch.float32, device=dev) if want_norm else None
    st.grad2d_norm = None if gnorm is None else gnorm.data_ptr()
    st.radii = handle["radii"].data_ptr()
    with torch.cuda.device(dev):
        if dL_dimage is None:
            # second half: backward_blend_dcolor already ran the blend backward into the handle's scratch
            scratch = handle.pop("bwd_scratch")
            token = handle.pop("bwd_scratch_token", None)
            if token is not None and token[0].generation != token[1]:
                raise RuntimeError("backward_raw_lowrank: another backward of this model ran between backward_blend_dcolor and "
                                   "its second half and reused the kept gradient records (rasterizer.KeptScratch): the blend "
                                   "records of this view are gone — run the two halves back to back")
            check(lib.w3d_backward_raw_lowrank(ctypes.byref(view.c), P, ctypes.byref(prm), ptr(handle["state"]),
                                               ptr(handle["point_list"]), None, None, None, ctypes.byref(g), None,
                                               ctypes.byref(st), ptr(scratch), stream_ptr(dev)))
            scratch_done(view, model)
            return gnorm, None
        dcol = torch.empty(P, 3, dtype=torch.float32, device=dev)
        scratch = backward_scratch(view, P, handle["point_list"], dev, owner=model)
        check(lib.w3d_backward_raw_lowrank(ctypes.byref(view.c), P, ctypes.byref(prm), ptr(handle["state"]),
                                           ptr(handle["point_list"]), ptr(dL_dimage.contiguous()), None, None,
                                           ctypes.byref(g), ptr(dcol), ctypes.byref(st), ptr(scratch), stream_ptr(dev)))
        scratch_done(view, model)
    return gnorm, dcol


def backward_raw_rows(model, handle, dL_dimage, norm_scale=1.0):
    """Backward of the view-parallel step in its sparse form (include/w3d.h w3d_backward_raw_rows): blend backward, then the
    per-Gaussian backward appends the non-zero 64-B gradient rows {index, ||dL/dmean2D|| * norm_scale, dL/dRGB, 11 geometry
    gradients} itself — backward_raw_lowrank + pack_gradient_rows without the dense arrays in between.  model.flat_grad is not
    touched.  Returns (rows (P, 16) float32 of which the first `count` are filled, count (1,) int32 on the device)."""
    dev = model.flat.device
    P, view = handle["P"], handle["view"]
    if P != model.num_points:
        raise RuntimeError("model was resized between forward and backward")
    prm = _raw_params(model)
    rows = torch.empty(max(P, 1), ROW_FLOATS, dtype=torch.float32, device=dev)
    count = torch.empty(1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        scratch = backward_scratch(view, P, handle["point_list"], dev, owner=model)
        check(lib.w3d_backward_raw_rows(ctypes.byref(view.c), P, ctypes.byref(prm), ptr(handle["state"]), ptr(handle["point_list"]),
                                        ptr(dL_dimage.contiguous()), None, None, float(norm_scale), ptr(rows), rows.shape[0],
                                        ptr(count), ptr(scratch), stream_ptr(dev)))
        scratch_done(view, model)
    return rows, count


def sh_adam_lowrank(model, dcolor_all, campos_all, skip=(), rows=None):
    """Adam step of f_dc / f_rest from the colour gradients of ALL views of this iteration (dcolor_all (V,P,3), campos_all
    (V,3)): dL/dSH[k] = sum_v basis_k(normalize(xyz - campos_v)) * dcolor_v, summed in view order.  The optimizer's step
    counter must already be advanced for this iteration.  csrc sh_adam_lowrank_kernel, in place (a model on the CPU is
    refused: w3d_amd/_host_twins.py)."""
    opt = model.optimizer
    P, V = model.num_points, int(dcolor_all.shape[0])
    r0, r1 = (0, P) if rows is None else rows          # rows=(r0, r1): dcolor_all holds only these Gaussians
    n = r1 - r0
    assert dcolor_all.shape[1] == n
    deg = int(model.active_sh_degree)
    if model.max_sh_degree != 3:
        raise RuntimeError("the low-rank exchange is written for 16 SH coefficients")
    if not model.flat.is_cuda:
        from ._host_twins import twin
        return twin("sh_adam_lowrank", "sh_adam_lowrank")(model, dcolor_all, campos_all, skip, rows)
    sl = model.block_slices()
    assert opt.steps["f_dc"] == opt.steps["f_rest"] or "f_dc" in skip or "f_rest" in skip
    bc1, bc2 = opt.bias_corrections("f_rest" if "f_dc" in skip else "f_dc")
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


def _geo_grads(model):
    g = W3DRawGrads()
    for n in GEO_BLOCKS:
        setattr(g, n, model.grad_view(n).data_ptr())
    return g


def pack_gradient_rows(model, dcolor, grad2d_norm=None, norm_scale=1.0):
    """The non-zero rows of this view's gradient (include/w3d.h w3d_pack_gradient_rows): dcolor (P,3) and the geometry blocks
    of model.flat_grad as backward_raw_lowrank left them.  Returns (rows (P, 16) float32 of which the first `count` are
    filled, count (1,) int32 on the device).  One kernel (a model on the CPU is refused: w3d_amd/_host_twins.py)."""
    P = model.num_points
    dev = model.flat.device
    rows = torch.empty(max(P, 1), ROW_FLOATS, dtype=torch.float32, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    if not model.flat.is_cuda:
        from ._host_twins import twin
        return twin("pack_gradient_rows", "pack_gradient_rows")(model, dcolor, grad2d_norm, norm_scale, rows, count)
    g = _geo_grads(model)
    with torch.cuda.device(dev):
        check(lib.w3d_pack_gradient_rows(P, ptr(dcolor.contiguous()), ctypes.byref(g), ptr(grad2d_norm), float(norm_scale),
                                         ptr(rows), rows.shape[0], ptr(count), stream_ptr(dev)))
    return rows, count


def apply_gradient_rows(model, rows, count, max_rows, dcolor_view, norm_sum=None):
    """Apply ONE view's packed rows (include/w3d.h w3d_apply_gradient_rows): dcolor_view (P,3) receives the colour rows, the
    geometry blocks of model.flat_grad and norm_sum (P,) are ADDED to.  `count` is a (1,) int32 device tensor: the kernel
    reads it, the host does not.  Call once per view in view order on zeroed buffers."""
    P = model.num_points
    if not model.flat.is_cuda:
        from ._host_twins import twin
        return twin("apply_gradient_rows", "apply_gradient_rows")(model, rows, count, max_rows, dcolor_view, norm_sum)
    dev = model.flat.device
    assert rows.is_contiguous() and dcolor_view.is_contiguous() and count.dtype == torch.int32
    g = _geo_grads(model)
    with torch.cuda.device(dev):
        check(lib.w3d_apply_gradient_rows(P, ptr(rows), ptr(count), int(max_rows), ptr(dcolor_view), ctypes.byref(g),
                                          ptr(norm_sum), stream_ptr(dev)))


class GatheredRows:
    """The all-gathered gradient rows of one iteration, indexed per Gaussian (GPU path of the sparse exchange): rows_all
    (V, cap, 16), counts (V,) int32 on the device, viewmask (P,) / slots (V, P) int32 built by w3d_index_gradient_rows."""

    def __init__(self, model, rows_all, counts, index_bufs=None):
        """index_bufs: optional (viewmask (>= P,), slots (>= V*P,)) int32 buffers the caller keeps from step to step (the
        Trainer does: 4*(V+1)*P bytes — 72 MB at 2 M Gaussians and 8 ranks — that need not be re-allocated every step)."""
        self.rows_all, self.counts = rows_all, counts
        self.V, self.cap = int(rows_all.shape[0]), int(rows_all.shape[1])
        P, dev = model.num_points, model.flat.device
        assert rows_all.is_contiguous() and rows_all.shape[2] == ROW_FLOATS and counts.dtype == torch.int32
        self.P = P
        n = max(P, 1)
        if index_bufs is not None and index_bufs[0].numel() >= n and index_bufs[1].numel() >= self.V * n and \
                index_bufs[0].device == dev:
            self.viewmask, self.slots = index_bufs[0][:n], index_bufs[1][:self.V * n].view(self.V, n)
        else:
            self.viewmask = torch.empty(n, dtype=torch.int32, device=dev)
            self.slots = torch.empty(self.V, n, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            check(lib.w3d_index_gradient_rows(P, self.V, ptr(rows_all), ptr(counts), self.cap, ptr(self.viewmask), ptr(self.slots),
                                              stream_ptr(dev)))

    def norm_sum(self):
        """(P,) sum over the views, in view order, of the rows' ||dL/dmean2D||."""
        dev = self.rows_all.device
        out = torch.empty(max(self.P, 1), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            check(lib.w3d_rows_norm_sum(self.P, self.V, ptr(self.rows_all), self.cap, ptr(self.viewmask), ptr(self.slots), ptr(out),
                                        stream_ptr(dev)))
        return out[:self.P]

    def norm_accumulate(self, accum):
        """accum (P,) or (P,1) float32 += the views' sum of the rows' ||dL/dmean2D|| (one addition per Gaussian, view order)."""
        dev = self.rows_all.device
        assert accum.is_contiguous() and accum.numel() == self.P and accum.dtype == torch.float32 and accum.device == dev
        with torch.cuda.device(dev):
            check(lib.w3d_rows_norm_accumulate(self.P, self.V, ptr(self.rows_all), self.cap, ptr(self.viewmask), ptr(self.slots),
                                               ptr(accum), stream_ptr(dev)))


def rows_adam(model, gathered, campos_all, skip=()):
    """Replicated optimizer step of ALL six blocks from the gathered rows (include/w3d.h w3d_rows_adam): SH gradient rebuilt
    per view direction, geometry gradients summed, both in view order, torch.optim.Adam's update in place — one kernel.
    The optimizer's step counters must already be advanced for this iteration (FlatAdam.advance)."""
    opt = model.optimizer
    if model.max_sh_degree != 3:
        raise RuntimeError("the row exchange is written for 16 SH coefficients")
    if model.num_points != gathered.P:
        raise RuntimeError("model was resized between the exchange and the optimizer step")
    dev = model.flat.device
    prm, ad = W3DRawGrads(), W3DAdamFused()
    sl = model.block_slices()
    b1, b2 = opt.betas
    for i, n in enumerate(_BLOCK_ORDER):
        a, _ = sl[n]
        setattr(prm, n, model._p[n].data_ptr())
        setattr(ad.exp_avg, n, opt.exp_avg.data_ptr() + 4 * a)
        setattr(ad.exp_avg_sq, n, opt.exp_avg_sq.data_ptr() + 4 * a)
        ad.lr[i] = float(opt.lrs[n])
        ad.skip[i] = int(n in skip)
        ad.bias_correction1[i], ad.bias_correction2[i] = (1.0, 1.0) if n in skip else opt.bias_corrections(n)
    ad.beta1, ad.beta2, ad.eps = float(b1), float(b2), float(opt.eps)
    cp = campos_all.to(device=dev, dtype=torch.float32).contiguous()
    with torch.cuda.device(dev):
        check(lib.w3d_rows_adam(model.num_points, gathered.V, int(model.active_sh_degree), ptr(cp), ptr(gathered.rows_all),
                                gathered.cap, ptr(gathered.viewmask), ptr(gathered.slots), ctypes.byref(prm), ctypes.byref(ad),
                                stream_ptr(dev)))


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
