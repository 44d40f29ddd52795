"""render() / flashsplat_render(): same names, arguments and returned dict keys as reference
gaussian_renderer/__init__.py:22-106 and :109-218 (pinned by tests/golden/render_marshalling.json),
so the reference's train / seg / render scripts can call this module unchanged."""
import math

import torch

from .rasterizer import (list_share_of, FlashSplatRasterizationSettings, FlashSplatRasterizer,
                         GaussianRasterizationSettings, GaussianRasterizer)


def _sh_python(pc, viewpoint_camera):
    from .sh import eval_sh
    shs_view = pc.get_features.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
    dir_pp = pc.get_xyz - viewpoint_camera.camera_center.repeat(pc.get_features.shape[0], 1)
    dir_pp = dir_pp / dir_pp.norm(dim=1, keepdim=True)
    return torch.clamp_min(eval_sh(pc.active_sh_degree, shs_view, dir_pp) + 0.5, 0.0)


def _settings(cls, cam, pc, bg_color, scaling_modifier, debug, **extra):
    return cls(image_height=int(cam.image_height), image_width=int(cam.image_width),
               tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), bg=bg_color,
               scale_modifier=scaling_modifier, viewmatrix=cam.world_view_transform,
               projmatrix=cam.full_proj_transform, sh_degree=pc.active_sh_degree, campos=cam.camera_center,
               prefiltered=False, debug=debug, tile_cull=bool(getattr(pc, "tile_cull", True)),
               deterministic=bool(getattr(pc, "deterministic", False)),
               list_share=list_share_of(pc), **extra)


# render() on this package's flat GaussianModel hands the PRE-ACTIVATION parameter blocks to the kernels (one autograd
# node, no activation / cat kernels — fused_step._RasterizeRawFn).  False: always marshal activated tensors through
# GaussianRasterizer exactly as the reference's render() does (what any other GaussianModel gets anyway).
RAW_AUTOGRAD = True


def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None):
    xyz = pc.get_xyz
    if (override_color is None and hasattr(pc, "flat") and pc.flat.is_cuda and
            not pipe.compute_cov3D_python and not pipe.convert_SHs_python and pc.max_sh_degree == 3):
        if not torch.is_grad_enabled():
            # evaluation renders (reference render.py:24-35, eval loops: all under no_grad) on the flat model: the
            # raw-parameter forward — no exp / sigmoid / normalize launches, no cat of the (P,16,3) features
            from .fused_step import render_raw
            r = render_raw(viewpoint_camera, pc, bg_color, scaling_modifier)
            return {"render": r["render"], "viewspace_points": torch.zeros_like(xyz), "visibility_filter": r["radii"] > 0,
                    "radii": r["radii"], "depth": r["depth"], "alpha": r["alpha"]}
        if RAW_AUTOGRAD:
            # training renders (train_vanilla_3dgs.py:73): same kernels behind one autograd node
            from .fused_step import render_raw_autograd
            return render_raw_autograd(viewpoint_camera, pc, bg_color, scaling_modifier)
    screenspace_points = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True, device=xyz.device) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass
    rasterizer = GaussianRasterizer(raster_settings=_settings(GaussianRasterizationSettings, viewpoint_camera, pc,
                                                              bg_color, scaling_modifier, False))
    scales = rotations = cov3D_precomp = shs = colors_precomp = None
    if pipe.compute_cov3D_python:
        cov3D_precomp = pc.get_covariance(scaling_modifier)
    else:
        scales, rotations = pc.get_scaling, pc.get_rotation
    if override_color is None:
        if pipe.convert_SHs_python:
            colors_precomp = _sh_python(pc, viewpoint_camera)
        else:
            shs = pc.get_features
    else:
        colors_precomp = override_color
    rendered_image, radii, rendered_depth, rendered_alpha = rasterizer(
        means3D=xyz, means2D=screenspace_points, shs=shs, colors_precomp=colors_precomp, opacities=pc.get_opacity,
        scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)
    return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
            "radii": radii, "depth": rendered_depth, "alpha": rendered_alpha}


def _subset_rows(used_mask):
    """Row indices a boolean mask selects (what `t[used_mask]` gathers), computed ONCE per mask: run_3d_seg.py hands the
    same obj_used_mask tensor to ~30 views in a row (:130-134, :362), and every boolean-mask index would pay a nonzero +
    host sync of its own.  Kept on the mask tensor itself together with its version counter, so an in-place change of the
    mask is noticed and nothing outlives the mask."""
    c = getattr(used_mask, "_w3d_rows", None)
    if c is None or c[0] != used_mask._version:
        c = (used_mask._version, used_mask.nonzero(as_tuple=True)[0])
        used_mask._w3d_rows = c
    return c[1]


def flashsplat_render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None, gt_mask=None,
                      used_mask=None, unique_label=None, setpdb=False, obj_num=2):
    xyz = pc.get_xyz
    subset = used_mask is not None
    if (not torch.is_grad_enabled() and override_color is None and hasattr(pc, "flat") and pc.flat.is_cuda and
            (not subset or (isinstance(used_mask, torch.Tensor) and used_mask.dtype == torch.bool and used_mask.dim() == 1
                            and used_mask.shape[0] == xyz.shape[0] and used_mask.device == xyz.device)) and
            not pipe.compute_cov3D_python and not pipe.convert_SHs_python and pc.max_sh_degree == 3):
        # every call site of the reference runs under no_grad (run_3d_seg.py:91,130,362): take the raw-parameter forward —
        # no exp / sigmoid / normalize launches and no cat of the (P,16,3) features (0.77 GB of traffic at 2 M) per view.
        # used_mask (find_match / the refine rounds, ~30 views per object mask): the mask goes INTO the preprocess kernel as a
        # cull, instead of four boolean-index gathers of the activated parameter blocks; the per-Gaussian outputs are then
        # gathered to the reference's subset indexing (rows = used_mask.nonzero()) with the cached row list.
        from .fused_step import finish, render_raw
        flash = dict(gt_mask=gt_mask, num_obj=obj_num)
        r = render_raw(viewpoint_camera, pc, bg_color, scaling_modifier, sync=not subset, flash=flash, used_mask=used_mask)
        if subset and not finish(r["handle"]):          # speculative list size was too small: once more, exact
            r = render_raw(viewpoint_camera, pc, bg_color, scaling_modifier, sync=True, flash=flash, used_mask=used_mask)
        radii, used_count, proj_xy, gs_depth = r["radii"], r["used_count"], r["proj_xy"], r["gs_depth"]
        if subset:
            rows = _subset_rows(used_mask)
            radii, used_count = radii.index_select(0, rows), used_count.index_select(1, rows)
            proj_xy, gs_depth = proj_xy.index_select(0, rows), gs_depth.index_select(0, rows)
        return {"render": r["render"], "viewspace_points": torch.zeros_like(xyz), "visibility_filter": radii > 0,
                "radii": radii, "alpha": r["alpha"], "depth": r["depth"], "contrib_num": r["contrib_num"],
                "used_count": used_count, "proj_xy": proj_xy, "gs_depth": gs_depth}
    screenspace_points = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True, device=xyz.device) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass
    rasterizer = FlashSplatRasterizer(raster_settings=_settings(
        FlashSplatRasterizationSettings, viewpoint_camera, pc, bg_color, scaling_modifier, pipe.debug,
        mask_grad=False, num_obj=obj_num))
    sub = (lambda t: t[used_mask]) if used_mask is not None else (lambda t: t)
    means3D, opacity = sub(xyz), sub(pc.get_opacity)
    scales = rotations = cov3D_precomp = shs = colors_precomp = None
    if pipe.compute_cov3D_python:
        cov3D_precomp = pc.get_covariance(scaling_modifier)
    else:
        scales, rotations = sub(pc.get_scaling), sub(pc.get_rotation)
    if override_color is None:
        if pipe.convert_SHs_python:
            colors_precomp = _sh_python(pc, viewpoint_camera)
        else:
            shs = sub(pc.get_features)
    else:
        colors_precomp = override_color
    rendered_image, radii, depth, alpha, contrib_num, used_count, proj_xy, gs_depth = rasterizer(
        gt_mask=gt_mask, unique_label=unique_label, means3D=means3D, means2D=screenspace_points, shs=shs,
        colors_precomp=colors_precomp, opacities=opacity, scales=scales, rotations=rotations,
        cov3D_precomp=cov3D_precomp)
    return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
            "radii": radii, "alpha": alpha, "depth": depth, "contrib_num": contrib_num, "used_count": used_count,
            "proj_xy": proj_xy, "gs_depth": gs_depth}


def flashsplat_render_masks(viewpoint_camera, pc, pipe, bg_color, gt_masks, scaling_modifier=1.0, obj_num=2):
    """All object masks of ONE view (the inner loop of reference run_3d_seg.py:88-97 calls flashsplat_render once per
    mask with the same camera): preprocessing, depth sort and binning run once, only the blend is repeated per mask.
    gt_masks: (K, H, W).  Returns used_count stacked to (K, obj_num+1, P) plus the view's render / alpha / depth.
    Non-overlapping binary masks take a single blend over their merged label map."""
    from .fused_step import flash_reblend, render_raw
    if not (hasattr(pc, "flat") and pc.flat.is_cuda) or pipe.compute_cov3D_python or pipe.convert_SHs_python:
        raise RuntimeError("flashsplat_render_masks needs the flat GaussianModel on the GPU and the default pipeline flags")
    K = int(gt_masks.shape[0])
    with torch.no_grad():
        binary = gt_masks.to(torch.float32)
        inside = binary > 0
        # binary masks that do not overlap (instance masks of one image): ONE blend over the merged label map gives every
        # mask's row 1 directly and its row 0 as (total - row 1)
        if obj_num == 1 and K > 1 and bool(((binary == 0) | (binary == 1)).all()) and int(inside.sum(0).max()) <= 1:
            labels = (inside * torch.arange(1, K + 1, device=binary.device, dtype=torch.float32)[:, None, None]).sum(0)
            pkg = render_raw(viewpoint_camera, pc, bg_color, scaling_modifier, flash=dict(gt_mask=labels, num_obj=K))
            multi = pkg["used_count"]                        # (K + 1, P): row 0 = unlabelled pixels, row k = mask k - 1
            total = multi.sum(0, keepdim=True)
            ones = multi[1:]
            used = torch.stack([total - ones, ones], dim=1)  # (K, 2, P)
        else:
            pkg = render_raw(viewpoint_camera, pc, bg_color, scaling_modifier, flash=dict(gt_mask=gt_masks[0], num_obj=obj_num))
            counts = [pkg["used_count"]]
            for k in range(1, K):
                counts.append(flash_reblend(pkg, gt_masks[k], obj_num)[0])
            used = torch.stack(counts)
    return {"used_count": used, "render": pkg["render"], "alpha": pkg["alpha"], "depth": pkg["depth"],
            "radii": pkg["radii"], "visibility_filter": pkg["radii"] > 0}
