"""Host side of the rasterizer: torch.autograd wrappers over the C-ABI (include/w3d.h).

Mirrors the Python surface of the two rasterizer packages the reference imports
(gaussian_renderer/__init__.py:14,18-19): ``GaussianRasterizationSettings`` +
``GaussianRasterizer`` with the call signature used at gaussian_renderer/__init__.py:89-97
(4 outputs) and :194-204 (FlashSplat, 8 outputs).  PyTorch is plumbing only: it owns device
memory (outputs, state, scratch, lists all come from its caching allocator, so the host code's
``torch.cuda.empty_cache()`` calls — run_3d_seg.py:99 — keep working) and the stream.
"""
import ctypes
from typing import NamedTuple, Optional

import torch

from . import _lib
from ._lib import W3DView, check, lib, ptr, stream_ptr


# This module keeps NO mutable state (two rasterizers are constructed per process).  The two
# behaviour switches travel in the settings tuple, behind the reference's own fields and with the reference's behaviour as
# default, so a call site that builds the tuple by keyword (gaussian_renderer/__init__.py:40-53) is unaffected:
#   tile_cull      exact footprint culling of tile instances (w3d_view.tile_cull): outputs unchanged, per-tile lists ~40 %
#                  shorter.  False gives the published bounding-square lists, comparable entry by entry with another
#                  implementation (tests/test_gpu_parity.py relies on that).
#   deterministic  w3d_view.deterministic: the blend backward stores every (tile, Gaussian) contribution in the slot of its
#                  list entry and adds them per Gaussian in tile order instead of float atomics — bit-identical gradients
#                  from run to run, about 2x the backward time (debugging; tests that compare k-step parameters bit for bit).
#   list_share     w3d_view.list_share (speed only; needs tile_cull and the atomic backward): 0 = one list per 16x16 tile,
#                  1 = one per 32x16 pair of tiles, 2 = one per 32x32 block.  The blend still runs one wave per 16x16 tile.
# The raw-parameter path (fused_step.py) reads the same switches from attributes of the GaussianModel it is given.
LIST_SHARE_DEFAULT = 1


def list_share_of(obj):
    """obj.list_share; when that is missing or None: what a Trainer last chose for this object from the measured walk fraction
    (obj._list_share_chosen, train.Trainer.adapt_list_share), else the default."""
    x = getattr(obj, "list_share", None)
    if x is None:
        x = getattr(obj, "_list_share_chosen", None)
    return LIST_SHARE_DEFAULT if x is None else int(x)


# list_share from what the scene does (results never depend on it): rho = (sum of the tiles' walk lengths) /
# (sum of the lengths of the lists they read), measured on the view just rendered — one reduction of the camera's walk array —
# every SHARE_PROBE_EVERY renders of an owner (a GaussianModel); below SHARE_RHO[0]: 32x32 cells, below [1]: 32x16, else one list
# per tile, +- SHARE_HYST around a bound.  Same-box A/B on three scenes: profiles/r05/ab_list_share.txt.
SHARE_PROBE_EVERY = 64
SHARE_RHO = (0.30, 0.60)
SHARE_HYST = 0.04


def adapt_list_share(owner, handle, every=SHARE_PROBE_EVERY, combine=None):
    """Choose owner._list_share_chosen for the following renders from the walked fraction of this one (handle: a finished forward of
    fused_step.render_raw).  A caller's explicit owner.list_share is left alone.  Returns the running rho (or None).
    combine: rho -> rho, called on every probe — a view-parallel job passes the mean over its ranks (Trainer.adapt_list_share), so that
    all ranks run the same mode: every rank probes its OWN camera, and per-rank choices made their timings, list lengths and buffer
    capacities diverge."""
    if getattr(owner, "list_share", None) is not None:
        return None
    st = getattr(owner, "_list_share_state", None)
    if st is None:
        st = owner._list_share_state = {"calls": 0, "rho": None}
    st["calls"] += 1
    if st["calls"] > 3 and st["calls"] % every:
        return st["rho"]
    view = handle["view"]
    mode = int(view.c.list_share) if (view.c.tile_cull and not view.c.deterministic) else 0
    lists_read = float(handle["num_rendered"]) * (1, 2, 4)[mode]
    if lists_read <= 0 and combine is None:
        return st["rho"]
    # (an upper estimate of the lists' total length as the tiles read them: clipped border cells and odd grid sizes read less)
    rho = float(view.tile_walk_hint.sum()) / lists_read if lists_read > 0 else 1.0         # (host-synchronous: 7 500 integers)
    if combine is not None:
        rho = float(combine(rho))                        # (a collective: every rank reaches this line in the same call)
    r = st["rho"] = rho if st["rho"] is None else 0.5 * (st["rho"] + rho)
    (lo, hi), h = SHARE_RHO, SHARE_HYST
    want = 2 if r < lo else 1 if r < hi else 0
    if want != mode:        # hysteresis: a bound only counts once it is crossed by SHARE_HYST in the direction of the change
        edge = lo if {want, mode} == {1, 2} else hi if {want, mode} == {0, 1} else None
        if edge is not None and abs(r - edge) < h:
            want = mode
    if want != getattr(owner, "_list_share_chosen", None):
        # (the list-capacity hints were learnt under the other grid: num_rendered differs up to 3x between the modes)
        hints = getattr(owner, "_w3d_list_hints", None)
        if hints:
            hints.clear()
    owner._list_share_chosen = want
    return r


class GaussianRasterizationSettings(NamedTuple):
    """Fields and order of the reference call site gaussian_renderer/__init__.py:40-53."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool
    tile_cull: bool = True
    deterministic: bool = False
    list_share: int = LIST_SHARE_DEFAULT


class FlashSplatRasterizationSettings(NamedTuple):
    """The 14-field variant of gaussian_renderer/__init__.py:132-147."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool
    mask_grad: bool = False
    num_obj: int = 2
    tile_cull: bool = True
    deterministic: bool = False
    list_share: int = LIST_SHARE_DEFAULT


class ListCapacity:
    """Speculative sizing of the per-tile list buffer, so that the forward's only host wait overlaps GPU work.

    The list length R (= num_rendered) is only known on the device after stage 1.  Waiting for it before stage 2 can be
    launched leaves the GPU idle for the host's wake-up + allocation + launch.  Instead the list is allocated from the
    largest R seen so far for this image size (x `slack`), stage 2 is enqueued right behind stage 1, and only then does
    the host wait for the counters — which the GPU produced BEFORE it started stage 2, so the wait ends while stage 2
    is still running.  The fill kernel never writes past the capacity it was given; if R turns out larger, stage 2 is
    simply run again with the exact size (outputs are overwritten).  A hint only: results never depend on it."""

    def __init__(self, slack=1.25):
        self.slack = slack
        self.known = 0          # largest R observed

    def guess(self):
        return int(self.known * self.slack) + 1024 if self.known else 0

    def observe(self, R):
        self.known = max(self.known, int(R))


def list_capacity(owner, H, W) -> ListCapacity:
    """The hint for image size (H, W), kept WITH THE CALLER'S DATA — an attribute of `owner`, the object that survives from
    one call to the next on the caller's side (the GaussianModel of the raw-parameter path; the means3D tensor the drop-in
    module is handed, which is the reference's `_xyz` nn.Parameter until a densification replaces it) — not in this module:
    nothing is shared between two models or two rasterizers of one process, and the hint dies with the data it describes.
    An owner that cannot carry attributes simply gets a fresh (empty) hint: the first-view, synchronous path."""
    hints = getattr(owner, "_w3d_list_hints", None)
    if hints is None:
        hints = {}
        try:
            owner._w3d_list_hints = hints
        except (AttributeError, TypeError):
            pass
    key = (int(H), int(W))
    cap = hints.get(key)
    if cap is None:
        cap = hints[key] = ListCapacity()
    return cap


def _f32c(t: Optional[torch.Tensor], device):
    if t is None:
        return None
    if t.device != device:
        raise ValueError(f"tensor on {t.device}, expected {device}")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _require_gpu(t: torch.Tensor):
    if not t.is_cuda:
        raise RuntimeError("the MI355X rasterizer needs its inputs on the GPU (device='cuda'); there is no CPU path")


class _View:
    """Keeps the device tensors of a w3d_view alive next to the ctypes struct."""

    def __init__(self, s, sh_coeffs, device):
        self.bg = _f32c(s.bg, device)
        self.vm = _f32c(s.viewmatrix, device)
        self.pm = _f32c(s.projmatrix, device)
        self.cp = _f32c(s.campos, device)
        if self.bg.numel() != 3 or self.vm.numel() != 16 or self.pm.numel() != 16 or self.cp.numel() != 3:
            raise ValueError("bg/viewmatrix/projmatrix/campos must have 3/16/16/3 elements")
        v = W3DView()
        v.image_height, v.image_width = int(s.image_height), int(s.image_width)
        v.tanfovx, v.tanfovy = float(s.tanfovx), float(s.tanfovy)
        v.scale_modifier = float(s.scale_modifier)
        v.sh_degree, v.sh_coeffs = int(s.sh_degree), int(sh_coeffs)
        v.prefiltered, v.debug = int(bool(s.prefiltered)), int(bool(s.debug))
        v.bg, v.viewmatrix = self.bg.data_ptr(), self.vm.data_ptr()
        v.projmatrix, v.campos = self.pm.data_ptr(), self.cp.data_ptr()
        v.tile_cull = int(bool(getattr(s, "tile_cull", True)))
        self.deterministic = bool(getattr(s, "deterministic", False))
        # (the forward must already know that the backward will be the deterministic one: shared lists are an atomic-mode layout)
        v.deterministic = int(self.deterministic)
        v.list_share = list_share_of(s)
        # per-camera walk-length hint of the blend forward (w3d_view.tile_walk_hint; speed only): kept on the camera's own
        # view-matrix tensor — the object a training loop hands in again every time it renders that camera
        tiles = ((v.image_width + 15) // 16) * ((v.image_height + 15) // 16)
        owner = s.viewmatrix
        hint = getattr(owner, "_w3d_tile_walk", None)
        if hint is None or hint.numel() != tiles or hint.device != device:
            hint = torch.zeros(tiles, dtype=torch.int32, device=device)
            try:
                owner._w3d_tile_walk = hint
            except (AttributeError, TypeError):
                pass
        self.tile_walk_hint = hint
        v.tile_walk_hint = hint.data_ptr()
        self.c = v


def _check_variants(shs, colors_precomp, scales, rotations, cov3D_precomp):
    # same messages/behaviour as the rasterizer modules the reference binds
    if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
        raise Exception('Please provide excatly one of either SHs or precomputed colors!')
    if ((scales is None or rotations is None) and cov3D_precomp is None) or \
            ((scales is not None or rotations is not None) and cov3D_precomp is not None):
        raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')


def _forward_impl(settings, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                  flash=None):
    """Runs stage 1 + stage 2.  Returns (color, radii, depth, alpha, saved, extras)."""
    _require_gpu(means3D)
    dev = means3D.device
    hint_owner = means3D            # (the caller's tensor object, before any conversion below)
    if means3D.dim() != 2 or means3D.shape[1] != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    P = int(means3D.shape[0])
    H, W = int(settings.image_height), int(settings.image_width)
    means3D = _f32c(means3D, dev)
    shs, colors_precomp = _f32c(shs, dev), _f32c(colors_precomp, dev)
    opacities = _f32c(opacities, dev)
    scales, rotations, cov3D_precomp = _f32c(scales, dev), _f32c(rotations, dev), _f32c(cov3D_precomp, dev)
    sh_coeffs = int(shs.shape[1]) if shs is not None and shs.dim() == 3 else 0
    if shs is not None and (shs.dim() != 3 or shs.shape[0] != P or shs.shape[2] != 3):
        raise RuntimeError("shs must have dimensions (num_points, coeffs, 3)")
    view = _View(settings, sh_coeffs, dev)
    with torch.cuda.device(dev):
        stream = stream_ptr(dev)
        state_b, scratch_b = ctypes.c_uint64(), ctypes.c_uint64()
        check(lib.w3d_forward_sizes(P, H, W, ctypes.byref(state_b), ctypes.byref(scratch_b)))
        state = torch.empty(state_b.value, dtype=torch.uint8, device=dev)
        scratch = torch.empty(scratch_b.value, dtype=torch.uint8, device=dev)
        radii = torch.empty(P, dtype=torch.int32, device=dev)
        cap = list_capacity(hint_owner, H, W)
        guess = cap.guess()
        pending = None
        if guess == 0:
            # first view of this size: nothing to go by, wait for R before the list is allocated
            counts = (ctypes.c_uint32 * 2)()
            check(lib.w3d_forward_stage1(ctypes.byref(view.c), P, ptr(means3D), ptr(shs), ptr(colors_precomp),
                                         ptr(opacities), ptr(scales), ptr(rotations), ptr(cov3D_precomp), ptr(radii),
                                         ptr(state), ptr(scratch), ctypes.cast(counts, ctypes.c_void_p), stream))
            num_visible, num_rendered = int(counts[0]), int(counts[1])
            cap.observe(num_rendered)
            list_len = num_rendered
        else:
            # speculative list size (ListCapacity): the counters start their way to the host right after stage 1
            check(lib.w3d_forward_stage1(ctypes.byref(view.c), P, ptr(means3D), ptr(shs), ptr(colors_precomp),
                                         ptr(opacities), ptr(scales), ptr(rotations), ptr(cov3D_precomp), ptr(radii),
                                         ptr(state), ptr(scratch), None, stream))
            pinned = torch.empty(2, dtype=torch.int32, pin_memory=True)
            pinned.copy_(state[:8].view(torch.int32), non_blocking=True)
            pending = torch.cuda.Event()
            pending.record()
            list_len = guess
        point_list = torch.empty(max(list_len, 1), dtype=torch.int32, device=dev)
        color = torch.empty(3, H, W, dtype=torch.float32, device=dev)
        depth = torch.empty(1, H, W, dtype=torch.float32, device=dev)
        alpha = torch.empty(1, H, W, dtype=torch.float32, device=dev)
        extras = None
        gt_mask = used_count = contrib_num = proj_xy = gs_depth = None
        num_obj = 0
        if flash is not None:
            num_obj = int(flash["num_obj"])
            gt_mask = flash.get("gt_mask")
            if gt_mask is not None:
                gt_mask = _f32c(gt_mask, dev)
                if tuple(gt_mask.shape[-2:]) != (H, W) or gt_mask.numel() != H * W:
                    raise RuntimeError("gt_mask must have dimensions (image_height, image_width)")
            used_count = torch.zeros(num_obj + 1, P, dtype=torch.float32, device=dev)
            contrib_num = torch.empty(H, W, dtype=torch.int32, device=dev)
            proj_xy = torch.empty(P, 2, dtype=torch.float32, device=dev)
            gs_depth = torch.empty(P, dtype=torch.float32, device=dev)
            extras = (contrib_num, used_count, proj_xy, gs_depth)

        def stage2():
            check(lib.w3d_forward_stage2(ctypes.byref(view.c), P, ptr(state), ptr(scratch), ptr(point_list),
                                         ctypes.c_uint64(list_len), ptr(color), ptr(depth), ptr(alpha),
                                         ptr(gt_mask), num_obj, ptr(used_count), ptr(contrib_num), ptr(proj_xy),
                                         ptr(gs_depth), stream))
        stage2()
        if pending is not None:
            pending.synchronize()          # the GPU is inside stage 2 by now: this wait costs it nothing
            num_visible, num_rendered = int(pinned[0]) & 0xFFFFFFFF, int(pinned[1]) & 0xFFFFFFFF
            cap.observe(num_rendered)
            if num_rendered > list_len:    # the guess was too small: same stage again with the exact size
                list_len = num_rendered
                point_list = torch.empty(list_len, dtype=torch.int32, device=dev)
                if used_count is not None:
                    used_count.zero_()     # (stage 2 accumulates into it)
                stage2()
    saved = dict(view=view, P=P, means3D=means3D, shs=shs, colors_precomp=colors_precomp, opacities=opacities,
                 scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp, state=state,
                 point_list=point_list, num_rendered=num_rendered, num_visible=num_visible)
    return color, radii, depth, alpha, saved, extras


class KeptScratch:
    """A backward scratch buffer an owner (a GaussianModel) keeps from call to call, so that the per-Gaussian gradient records
    at its start need no zeroing pass (include/w3d.h, w3d_view.records_kept_clean): zero-filled once, then every backward
    hands it back clean.  `clean` is False while a backward that dirtied it has not been followed by the call that
    consumes (and clears) the records — begin() then zero-fills again.  One backward at a time per owner: successive
    backwards of a model are ordered by the stream they are enqueued on (autograd and the trainer both use the current
    stream); backwards of ONE model enqueued on different streams at once would share the records."""

    def __init__(self):
        self.buf, self.clean = None, False
        self.generation = 0           # counts begin() calls: the token of the backward that currently owns the records

    def begin(self, nbytes, dev):
        self.generation += 1
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != dev:
            self.buf = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        elif not self.clean:
            self.buf.zero_()
        self.clean = False            # dirty from now on; done() after the consuming call
        return self.buf

    def done(self):
        self.clean = True


def backward_scratch(view, P, point_list, dev, owner=None):
    """The scratch buffer of a backward call; selects the deterministic mode in the C struct when `view` asks for it.
    owner: an object that keeps the buffer between calls (KeptScratch on owner._w3d_bwd_scratch) — the backward then skips
    its zeroing pass; the caller must call owner._w3d_bwd_scratch.done() once the per-Gaussian backward has been enqueued."""
    sb = ctypes.c_uint64()
    view.c.records_kept_clean = 0
    if view.deterministic:
        cap = int(point_list.numel())
        view.c.deterministic, view.c.det_list_capacity = 1, cap
        check(lib.w3d_backward_det_sizes(P, cap, ctypes.byref(sb)))
    else:
        view.c.deterministic, view.c.det_list_capacity = 0, 0
        check(lib.w3d_backward_sizes(P, ctypes.byref(sb)))
        if owner is not None:
            kept = getattr(owner, "_w3d_bwd_scratch", None)
            if kept is None:
                kept = owner._w3d_bwd_scratch = KeptScratch()
            view.c.records_kept_clean = 1
            return kept.begin(sb.value, dev)
    return torch.empty(sb.value, dtype=torch.uint8, device=dev)


def scratch_done(view, owner):
    """The per-Gaussian backward that consumes the records has been enqueued: the kept buffer is clean again."""
    if owner is not None and view.c.records_kept_clean:
        owner._w3d_bwd_scratch.done()


def _backward_impl(saved, grad_color, grad_depth, grad_alpha):
    view, P = saved["view"], saved["P"]
    dev = saved["means3D"].device
    shs, colors_precomp = saved["shs"], saved["colors_precomp"]
    scales, cov3D_precomp = saved["scales"], saved["cov3D_precomp"]
    H, W = view.c.image_height, view.c.image_width
    grad_color = _f32c(grad_color, dev) if grad_color is not None else torch.zeros(3, H, W, device=dev)
    grad_depth, grad_alpha = _f32c(grad_depth, dev), _f32c(grad_alpha, dev)
    e = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)  # noqa: E731
    g_means3D, g_means2D, g_opac = e(P, 3), e(P, 3), e(P, 1)
    g_shs = e(*shs.shape) if shs is not None else None
    g_colors = e(P, 3) if colors_precomp is not None else None
    g_scales = e(P, 3) if scales is not None else None
    g_rots = e(P, 4) if scales is not None else None
    g_cov = e(P, 6) if cov3D_precomp is not None else None
    if P == 0:
        return g_means3D, g_means2D, g_shs, g_colors, g_opac, g_scales, g_rots, g_cov
    with torch.cuda.device(dev):
        scratch = backward_scratch(view, P, saved["point_list"], dev)
        check(lib.w3d_backward(ctypes.byref(view.c), P, ptr(saved["means3D"]), ptr(shs), ptr(colors_precomp),
                               ptr(saved["opacities"]), ptr(scales), ptr(saved["rotations"]), ptr(cov3D_precomp),
                               ptr(saved["state"]), ptr(saved["point_list"]), ptr(grad_color), ptr(grad_depth),
                               ptr(grad_alpha), ptr(g_means3D), ptr(g_means2D), ptr(g_colors), ptr(g_shs),
                               ptr(g_opac), ptr(g_scales), ptr(g_rots), ptr(g_cov), ptr(scratch), stream_ptr(dev)))
    return g_means3D, g_means2D, g_shs, g_colors, g_opac, g_scales, g_rots, g_cov


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, settings):
        color, radii, depth, alpha, saved, _ = _forward_impl(settings, means3D, sh, colors_precomp, opacities,
                                                             scales, rotations, cov3Ds_precomp)
        ctx.saved = saved
        ctx.set_materialize_grads(False)   # unused depth/alpha outputs arrive as None, not zero images
        ctx.means2D_shape = None if means2D is None else tuple(means2D.shape)
        ctx.opac_shape = tuple(opacities.shape)
        ctx.mark_non_differentiable(radii)
        return color, radii, depth, alpha

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_alpha):
        saved = ctx.saved
        g_means3D, g_means2D, g_shs, g_colors, g_opac, g_scales, g_rots, g_cov = _backward_impl(
            saved, grad_color, grad_depth, grad_alpha)
        if ctx.means2D_shape is not None and tuple(g_means2D.shape) != ctx.means2D_shape:
            g_means2D = None   # means2D is only a gradient carrier; mismatched proxies get nothing
        ctx.saved = None
        return (g_means3D, g_means2D, g_shs, g_colors, g_opac.reshape(ctx.opac_shape), g_scales, g_rots, g_cov, None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, settings)


class GaussianRasterizer(torch.nn.Module):
    """Drop-in for diff_gaussian_rasterization.GaussianRasterizer (depth/alpha fork): returns
    (color (3,H,W), radii (P,) int32, depth (1,H,W), alpha (1,H,W))."""

    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """Boolean near-plane visibility (never called by Wheat-3DGS; kept for API completeness)."""
        with torch.no_grad():
            vm = self.raster_settings.viewmatrix
            z = positions @ vm[:3, 2] + vm[3, 2]
            return z > 0.2

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        _check_variants(shs, colors_precomp, scales, rotations, cov3D_precomp)
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, self.raster_settings)


class FlashSplatRasterizer(torch.nn.Module):
    """Drop-in for flashsplat_rasterization.GaussianRasterizer as called at
    gaussian_renderer/__init__.py:194-204: forward-only (every call site is under no_grad and
    mask_grad=False), returns (color, radii, depth, alpha, contrib_num, used_count, proj_xy, gs_depth)
    with used_count of shape (num_obj+1, P)."""

    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, means3D, means2D, opacities, gt_mask=None, unique_label=None, shs=None, colors_precomp=None,
                scales=None, rotations=None, cov3D_precomp=None):
        _check_variants(shs, colors_precomp, scales, rotations, cov3D_precomp)
        s = self.raster_settings
        if getattr(s, "mask_grad", False):
            # rejected, not deferred (include/w3d.h, w3d_forward_stage2): the reference hard-codes mask_grad=False
            # (gaussian_renderer/__init__.py:145) and calls the FlashSplat rasterizer under no_grad only
            raise NotImplementedError("mask_grad=True: the FlashSplat outputs of this rasterizer are forward-only — Wheat-3DGS constructs "
                                      "its settings with mask_grad=False (gaussian_renderer/__init__.py:145) and never differentiates them")
        with torch.no_grad():
            color, radii, depth, alpha, _, extras = _forward_impl(
                s, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                flash=dict(gt_mask=gt_mask, num_obj=getattr(s, "num_obj", 2)))
        contrib_num, used_count, proj_xy, gs_depth = extras
        return color, radii, depth, alpha, contrib_num, used_count, proj_xy, gs_depth


KNN_GRID_FROM = 4096      # below this the brute-force kernel is at least as fast


def dist2_knn3(points: torch.Tensor, method: str = "auto") -> torch.Tensor:
    """distCUDA2 (reference scene/gaussian_model.py:148): (N,3) fp32 cuda -> (N,) mean squared
    distance to the 3 nearest other points."""
    _require_gpu(points)
    pts = _f32c(points, points.device)
    if pts.dim() != 2 or pts.shape[1] != 3:
        raise RuntimeError("points must have dimensions (num_points, 3)")
    N = int(pts.shape[0])
    out = torch.empty(N, dtype=torch.float32, device=pts.device)
    with torch.cuda.device(pts.device):
        if N < KNN_GRID_FROM or method == "brute":
            check(lib.w3d_knn_dist2(N, ptr(pts), ptr(out), stream_ptr(pts.device)))
        else:       # same values bit for bit, O(N) through a uniform grid built on the device
            sb = ctypes.c_uint64()
            check(lib.w3d_knn_sizes(N, ctypes.byref(sb)))
            scratch = torch.empty(sb.value, dtype=torch.uint8, device=pts.device)
            check(lib.w3d_knn_dist2_grid(N, ptr(pts), ptr(out), ptr(scratch), stream_ptr(pts.device)))
    return out


def debug_tile_ranges(saved):
    v = saved["view"].c
    T = ((v.image_width + 15) // 16) * ((v.image_height + 15) // 16)
    dev = saved["state"].device
    out = torch.empty(T, 2, dtype=torch.int32, device=dev)
    check(lib.w3d_debug_tile_ranges(v.image_height, v.image_width, saved["P"], ptr(saved["state"]), ptr(out), stream_ptr(dev)))
    return out


def debug_tile_rects(saved):
    """(P,4) int32 {rect lo, rect hi, mask lo, mask hi} of a forward with tile_cull (include/w3d.h w3d_debug_tile_rects)."""
    v = saved["view"].c
    dev = saved["state"].device
    out = torch.empty(saved["P"], 4, dtype=torch.int32, device=dev)
    lib.w3d_debug_tile_rects.argtypes = [ctypes.c_int32] * 3 + [ctypes.c_void_p] * 3
    lib.w3d_debug_tile_rects.restype = ctypes.c_int
    check(lib.w3d_debug_tile_rects(v.image_height, v.image_width, saved["P"], ptr(saved["state"]), ptr(out), stream_ptr(dev)))
    return out


def debug_tile_schedule(saved):
    """(8, cap) int64 numpy array of the block -> (tile, part) entries the last blend kernel on this state ran
    (include/w3d.h w3d_debug_tile_schedule)."""
    import numpy as np
    v = saved["view"].c
    lib.w3d_debug_tile_schedule.argtypes = [ctypes.c_int32] * 3 + [ctypes.c_void_p] * 3
    lib.w3d_debug_tile_schedule.restype = ctypes.c_int
    cap = ctypes.c_uint32()
    check(lib.w3d_debug_tile_schedule(v.image_height, v.image_width, saved["P"], None, None, ctypes.byref(cap)))
    out = np.empty(8 * cap.value, np.uint32)
    torch.cuda.synchronize(saved["state"].device)
    check(lib.w3d_debug_tile_schedule(v.image_height, v.image_width, saved["P"], ptr(saved["state"]), out.ctypes.data, ctypes.byref(cap)))
    return out.astype(np.int64).reshape(8, cap.value)


def debug_depth_buckets(saved):
    """(bstart[1025], brange[1024, 2]) int64 numpy arrays: the depth sort's bucket grid of this forward (include/w3d.h
    w3d_debug_depth_buckets).  The forward must have kept its scratch buffer (model.debug_keep_scratch = True for render_raw)."""
    import numpy as np
    if saved.get("scratch") is None:
        raise RuntimeError("this forward did not keep its scratch buffer (set model.debug_keep_scratch = True)")
    v = saved["view"].c
    lib.w3d_debug_depth_buckets.argtypes = [ctypes.c_int32] * 3 + [ctypes.c_void_p] * 3
    lib.w3d_debug_depth_buckets.restype = ctypes.c_int
    bstart, brange = np.empty(1025, np.uint32), np.empty((1024, 2), np.uint32)
    torch.cuda.synchronize(saved["scratch"].device)
    check(lib.w3d_debug_depth_buckets(v.image_height, v.image_width, saved["P"], ptr(saved["scratch"]), bstart.ctypes.data, brange.ctypes.data))
    return bstart.astype(np.int64), brange.astype(np.int64)


def debug_gaussian_records(saved):
    """(P,16) float32: the 64-B per-Gaussian records of a forward (include/w3d.h w3d_debug_gaussian_records); rows of culled
    Gaussians (radii == 0) are not written by the forward."""
    v = saved["view"].c
    dev = saved["state"].device
    out = torch.empty(saved["P"], 16, dtype=torch.float32, device=dev)
    lib.w3d_debug_gaussian_records.argtypes = [ctypes.c_int32] * 3 + [ctypes.c_void_p] * 3
    lib.w3d_debug_gaussian_records.restype = ctypes.c_int
    check(lib.w3d_debug_gaussian_records(v.image_height, v.image_width, saved["P"], ptr(saved["state"]), ptr(out), stream_ptr(dev)))
    return out


def debug_pixel_state(saved):
    v = saved["view"].c
    dev = saved["state"].device
    ft = torch.empty(v.image_height, v.image_width, dtype=torch.float32, device=dev)
    nc = torch.empty(v.image_height, v.image_width, dtype=torch.int32, device=dev)
    check(lib.w3d_debug_pixel_state(v.image_height, v.image_width, saved["P"], ptr(saved["state"]), ptr(ft), ptr(nc), stream_ptr(dev)))
    return ft, nc
