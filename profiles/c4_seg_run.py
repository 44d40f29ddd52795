#!/usr/bin/env python
"""Config C4 end to end on one MI355X: the loop structure of run_3d_seg.py on a synthetic plot with PLANTED objects.

run_3d_seg.py lifts 2-D instance masks to a 3-D labelling of the Gaussians: for one object at a time it renders every view
with that object's mask through the FlashSplat contribution rasterizer, sums the per-(label, Gaussian) contributions over
the views (:75-104), turns the sums into a membership with multi_instance_opt (:54-72), and then scores the object against
the candidate masks of other views — subset render of the member Gaussians, alpha > 0.5, bounding box, IoU (:113-175
find_match).  No wheat data ships with the reference, so here K objects ("wheat heads": all Gaussians within 9 cm of a
centre, made fairly opaque) are planted in the SURVEY section 8d scene; their per-view masks are what a perfect 2-D segmenter
would deliver (alpha > 0.5 of the object rendered alone).  The script then runs the reference's steps through this repo's
drop-ins — flashsplat_render (kernel-side accumulation over views), segmentation.multi_instance_opt,
flashsplat_render(used_mask=...) + segmentation.mask_iou_device — and reports how well the planted membership is recovered
and how fast each step runs.
    python profiles/c4_seg_run.py > gpurun_out/c4_seg_run.json
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "wheat-3dgs_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=500_000)
    ap.add_argument("--objects", type=int, default=12)
    ap.add_argument("--width", type=int, default=1600)
    ap.add_argument("--height", type=int, default=1200)
    a = ap.parse_args()
    from w3d_amd.synth import make_scene, make_cameras
    from w3d_amd.gaussian_model import GaussianModel
    from w3d_amd.gaussian_renderer import flashsplat_render
    from w3d_amd.segmentation import accumulate_counts_raw, mask_iou_device, multi_instance_opt
    from w3d_amd.train import PipelineParams
    dev = torch.device("cuda:0")
    bg = torch.zeros(3, device=dev)
    pipe = PipelineParams()
    cams = [c.to(dev) for c in make_cameras(36, a.width, a.height)]
    sc = make_scene(a.points, seed=2)
    g = torch.Generator().manual_seed(5)
    centres = torch.stack([torch.rand(a.objects, generator=g) * 2.4 - 1.2, torch.rand(a.objects, generator=g) * 1.0 - 0.5,
                           0.35 + 0.2 * torch.rand(a.objects, generator=g)], 1)      # upper half of the canopy: seen from above
    member = torch.stack([(sc.xyz - c).norm(dim=1) < 0.09 for c in centres])       # (K, P) planted membership
    member &= member.cumsum(0) == 1                                                 # (disjoint: first object wins an overlap)
    sc.opacity[member.any(0)] = 2.5
    m = GaussianModel(3, device=dev)
    m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
    m.active_sh_degree = 3
    member = member.to(dev)
    K, V = a.objects, len(cams)
    with torch.no_grad():
        # ---- the 2-D masks a perfect segmenter would deliver: each object rendered alone, alpha > 0.5
        masks = torch.zeros(V, K, a.height, a.width, dtype=torch.bool, device=dev)
        for v, cam in enumerate(cams):
            for k in range(K):
                masks[v, k] = flashsplat_render(cam, m, pipe, bg, used_mask=member[k])["alpha"][0] > 0.5
        visible_views = (masks.flatten(2).sum(2) > 50).sum(0)                      # views in which object k has a usable mask
        torch.cuda.synchronize()
        # ---- step 1 (run_3d_seg.py:75-104): contributions summed over the views, one object at a time
        t0 = time.perf_counter()
        pred = torch.zeros(K, a.points, dtype=torch.bool, device=dev)
        for k in range(K):
            counts = accumulate_counts_raw(m, cams, [masks[v, k].float() for v in range(V)], bg, obj_num=1)
            pred[k] = multi_instance_opt(counts)[1]                                # row 1: "belongs to the masked object"
        torch.cuda.synchronize()
        t_counts = time.perf_counter() - t0
        inter = (pred & member).sum(1).float()
        set_iou = (inter / (pred | member).sum(1).float().clamp_min(1)).cpu()
        recall = (inter / member.sum(1).float().clamp_min(1)).cpu()
        precision = (inter / pred.sum(1).float().clamp_min(1)).cpu()
        # ---- step 2 (:113-175 find_match): the labelled object against every candidate mask of every view
        t0 = time.perf_counter()
        hits, ious = 0, []
        for k in range(K):
            for v, cam in enumerate(cams):
                alpha = flashsplat_render(cam, m, pipe, bg, used_mask=pred[k])["alpha"]
                iou, bbox, n_pred = mask_iou_device(alpha, masks[v].to(torch.uint8), 0.5)
                if int(masks[v, k].sum()) > 50:
                    hits += int(int(iou.argmax()) == k)
                    ious.append(float(iou[k]))
        torch.cuda.synchronize()
        t_match = time.perf_counter() - t0
    n_scored = len(ious)
    print(json.dumps({
        "config": f"C4 end to end: {K} planted objects, {a.points} Gaussians, {V} views at {a.width}x{a.height}",
        "gaussians_per_object": [int(x) for x in member.sum(1).cpu()], "views_with_a_mask_per_object": [int(x) for x in visible_views.cpu()],
        "label_step": {"renders": K * V, "seconds": round(t_counts, 2), "mask_views_per_s": round(K * V / t_counts, 1),
                       "membership_iou_mean": round(float(set_iou.mean()), 4), "membership_iou_min": round(float(set_iou.min()), 4),
                       "membership_recall_mean": round(float(recall.mean()), 4),
                       "membership_precision_mean": round(float(precision.mean()), 4),
                       "note": "2-D masks cannot tell an object from what lies on the same rays in every view (the canopy under a "
                               "head seen from above): recall is the figure the planted membership can be held to"},
        "match_step": {"subset_renders": K * V, "seconds": round(t_match, 2), "views_per_s": round(K * V / t_match, 1),
                       "scored_views": n_scored, "best_candidate_is_the_object": round(hits / max(n_scored, 1), 4),
                       "mask_iou_mean": round(sum(ious) / max(n_scored, 1), 4)}}))


if __name__ == "__main__":
    main()
