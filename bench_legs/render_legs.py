"""render / FlashSplat legs (--full): forward-only render throughput (reference render.py's use) and config C4's per-mask
contribution renders (run_3d_seg.py's inner calls), same scene as the headline."""
import time

import torch
import torch.distributed as dist

from .common import _progress


def render_mpix_per_s(args, model, cams, bg, dev, world, sync, n_r=None):
    """forward-only render throughput (reference render.py:24-35), views cycled; an untimed pass of the same length first: the
    loop keeps its n_r output images, and a first-time hipMalloc of each of them inside the timed region costs more than the
    frame it holds"""
    from w3d_amd.train import render_views
    n_r = n_r or max(4, min(args.steps, 72))
    render_views(model, [cams[i % len(cams)] for i in range(n_r)], bg)
    sync()
    r0 = time.perf_counter()
    render_views(model, [cams[i % len(cams)] for i in range(n_r)], bg)
    sync()
    r_el = torch.tensor([time.perf_counter() - r0], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(r_el, op=dist.ReduceOp.MAX)
    return round(world * n_r * args.width * args.height / 1e6 / float(r_el), 1)


def flashsplat_legs(args, model, cams, bg, dev, world, sync):
    extras = {}
    _progress("extras: FlashSplat")
    # config C4: FlashSplat per-mask contribution render (run_3d_seg.py's inner call), binary mask, same scene
    from w3d_amd.gaussian_renderer import flashsplat_render, flashsplat_render_masks
    from w3d_amd.train import PipelineParams
    yy, xx = torch.meshgrid(torch.arange(args.height, device=dev), torch.arange(args.width, device=dev), indexing="ij")
    mask = (((xx - args.width // 2) ** 2 + (yy - args.height // 2) ** 2) < (args.height // 3) ** 2).float()
    n_f = 16
    with torch.no_grad():
        # (warm-up with the loop's own holding pattern — view, running sum, next view — so that every block the loop
        #  needs exists in torch's allocator before the clock starts: the 16 timed views take ~15 ms, one first-time
        #  hipMalloc 1-2 ms)
        counts = None
        for i in range(3):
            uc = flashsplat_render(cams[i], model, PipelineParams(), bg, gt_mask=mask, obj_num=1)["used_count"]
            counts = uc if counts is None else counts + uc
        del counts, uc
        sync()
        f0 = time.perf_counter()
        counts = None
        for i in range(n_f):
            uc = flashsplat_render(cams[i % len(cams)], model, PipelineParams(), bg, gt_mask=mask, obj_num=1)["used_count"]
            counts = uc if counts is None else counts + uc
        sync()
        extras["flashsplat_views_per_s"] = round(world * n_f / (time.perf_counter() - f0), 1)
    del counts
    # run_3d_seg.py's MOST FREQUENT call (find_match :130-134, ~29 views per object mask and refine round; :362, 36 views):
    # flashsplat_render(..., used_mask=obj_used_mask) followed by alpha > 0.5 -> bounding box -> IoU against the
    # candidate masks.  The mask is applied inside the preprocess kernel; the scoring runs on the device.
    from w3d_amd.segmentation import mask_iou_device
    _progress("extras: subset renders")
    head = ((model.get_xyz.detach() - torch.tensor([0.2, -0.1, 0.3], device=dev)).norm(dim=1) < 0.06)
    cand = (torch.stack([torch.roll(mask, shifts=25 * k, dims=1) for k in range(4)]) > 0).to(torch.uint8)
    with torch.no_grad():
        for i in range(2):
            flashsplat_render(cams[i], model, PipelineParams(), bg, used_mask=head)
        sync()
        f0 = time.perf_counter()
        for i in range(args.views):
            pkg_s = flashsplat_render(cams[i % len(cams)], model, PipelineParams(), bg, used_mask=head)
            mask_iou_device(pkg_s["alpha"], cand, 0.5)
        sync()
        extras["flashsplat_subset_views_per_s"] = round(world * args.views / (time.perf_counter() - f0), 1)
        # the reference's own formulation of the same call for comparison: activated blocks gathered with the mask
        # (gaussian_renderer/__init__.py:151-156,168-170,186-187) through the drop-in rasterizer module, alpha to the host,
        # numpy threshold / bbox / IoU (run_3d_seg.py:131-163) — reached here by handing the mask over as an index tensor
        head_idx = head.nonzero(as_tuple=True)[0]
        cand_np = cand.cpu().numpy() > 0
        for i in range(2):
            flashsplat_render(cams[i], model, PipelineParams(), bg, used_mask=head_idx)
        sync()
        f0 = time.perf_counter()
        for i in range(12):
            a = flashsplat_render(cams[i % len(cams)], model, PipelineParams(), bg, used_mask=head_idx)["alpha"]
            pred = a.squeeze().detach().cpu().numpy() > 0.5
            for m_ in cand_np:
                inter, union = (m_ & pred).sum(), (m_ | pred).sum()
        sync()
        extras["flashsplat_subset_reference_formulation_views_per_s"] = round(world * 12 / (time.perf_counter() - f0), 1)
        extras["flashsplat_subset"] = {"gaussians_in_mask": int(head.sum()), "candidate_masks": 4,
                                       "loop": "flashsplat_render(used_mask) + alpha>0.5 -> bbox -> IoU, per view (host reads 13 counters)"}
        del head_idx, a
    del head, cand, pkg_s
    # ... and run_3d_seg.py's real loop shape: several object masks per view — one forward, the blend repeated per mask
    n_m = 8
    masks = torch.stack([torch.roll(mask, shifts=40 * k, dims=1) for k in range(n_m)])
    with torch.no_grad():
        flashsplat_render_masks(cams[0], model, PipelineParams(), bg, masks[:2], obj_num=1)
        sync()
        f0 = time.perf_counter()
        for i in range(4):
            uc = flashsplat_render_masks(cams[i % len(cams)], model, PipelineParams(), bg, masks, obj_num=1)["used_count"]
        sync()
        extras["flashsplat_masks_per_s_8_per_view"] = round(world * 4 * n_m / (time.perf_counter() - f0), 1)
    # non-overlapping instance masks (8 vertical stripes): one blend over the merged label map
    stripes = torch.stack([((xx >= k * args.width // n_m) & (xx < (k + 1) * args.width // n_m)).float() for k in range(n_m)])
    with torch.no_grad():
        flashsplat_render_masks(cams[0], model, PipelineParams(), bg, stripes, obj_num=1)
        sync()
        f0 = time.perf_counter()
        for i in range(4):
            uc = flashsplat_render_masks(cams[i % len(cams)], model, PipelineParams(), bg, stripes, obj_num=1)["used_count"]
        sync()
        extras["flashsplat_masks_per_s_8_disjoint_per_view"] = round(world * 4 * n_m / (time.perf_counter() - f0), 1)
    # eval_wheatgs.py's shape: ONE label image with hundreds of object ids (obj_num = max label)
    K = 300
    # (40-pixel cells: label boundaries cut through the 16x16 tiles, up to four labels per tile)
    labels = ((xx // 40) + (args.width // 40 + 1) * (yy // 40)).remainder(K + 1).float()
    with torch.no_grad():
        # (two warm-up views with the loop's own holding pattern — the previous view's 2.4-GB count matrix is still
        #  referenced while the next one is allocated — so that both blocks exist in torch's allocator before the clock
        #  starts: a first-time 2.4-GB hipMalloc inside a 4-view window costs ten times the four renders)
        for i in range(2):
            uc = flashsplat_render(cams[i], model, PipelineParams(), bg, gt_mask=labels, obj_num=K)["used_count"]
        sync()
        f0 = time.perf_counter()
        for i in range(4):
            uc = flashsplat_render(cams[i % len(cams)], model, PipelineParams(), bg, gt_mask=labels, obj_num=K)["used_count"]
        sync()
        extras["flashsplat_views_per_s_300_labels"] = round(world * 4 / (time.perf_counter() - f0), 1)
    del uc, masks, stripes, labels
    torch.cuda.empty_cache()
    return extras
