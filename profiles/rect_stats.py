#!/usr/bin/env python
"""Where do a scene's tile instances come from?  For one view of bench.py's scenes: the share of the bounding-square instances
(sum over Gaussians of rect width x height) held by Gaussians whose rect exceeds 64 tiles (those carry no 64-bit tile mask:
w3d_preprocess.hip culls only rects of <= 64 tiles exactly), next to the list length the forward really produced.
    python3 profiles/rect_stats.py --scene densified [--model-file /tmp/x.pt]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "wheat-3dgs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="densified", choices=("untrained", "densified"))
    ap.add_argument("--model-file", default=None)
    a = ap.parse_args()
    args = bench.parse_defaults()
    from w3d_amd.gaussian_model import GaussianModel
    from w3d_amd.fused_step import render_raw
    dev = torch.device("cuda:0")
    bg = torch.zeros(3, device=dev)
    if a.scene == "densified":
        if a.model_file and os.path.exists(a.model_file):
            pack = torch.load(a.model_file, weights_only=False)
            cams = bench.densified_views(args, dev, bg)[0]
            model = GaussianModel(3, device=dev)
            model.restore(pack["model"], pack["opt"])
        else:
            model, opt, cams, _, _ = bench.grow_densified_model(args, dev, bg, log=bench._progress)
    else:
        sc, model, opt, cams = bench.build_scene(args, dev)
    out = []
    for ci in (0, len(cams) // 2):
        cam = cams[ci]
        with torch.no_grad():
            pkg = render_raw(cam, model, bg, flash=dict(num_obj=1))
        r = pkg["radii"].float()
        xy = pkg["proj_xy"]
        vis = r > 0
        gx, gy = (args.width + 15) // 16, (args.height + 15) // 16
        x0 = ((xy[:, 0] - r) / 16).floor().clamp(0, gx)
        x1 = ((xy[:, 0] + r + 15) / 16).floor().clamp(0, gx)
        y0 = ((xy[:, 1] - r) / 16).floor().clamp(0, gy)
        y1 = ((xy[:, 1] + r + 15) / 16).floor().clamp(0, gy)
        nt = ((x1 - x0) * (y1 - y0)) * vis
        big = nt > 64
        op = torch.sigmoid(model._opacity.detach().reshape(-1))
        hist = {str(k): int(((nt > lo) & (nt <= hi)).sum()) for k, (lo, hi) in
                {"1-4": (0, 4), "5-16": (4, 16), "17-64": (16, 64), "65-256": (64, 256), "257-1024": (256, 1024), ">1024": (1024, 1e9)}.items()}
        # what the forward actually binned: the (possibly shrunk) rect + 64-bit mask of every Gaussian
        from w3d_amd.rasterizer import debug_tile_rects
        tr = debug_tile_rects(pkg["handle"]).to(torch.int64) & 0xFFFFFFFF
        w = (tr[:, 1] & 0xFFFF) - (tr[:, 0] & 0xFFFF)
        h = (tr[:, 1] >> 16) - (tr[:, 0] >> 16)
        nt2 = (w * h) * vis
        bits = torch.zeros_like(nt2)
        for word in (tr[:, 2], tr[:, 3]):
            x = word.clone()
            for _ in range(32):
                bits += x & 1
                x >>= 1
        small = nt2 <= 64
        binned = {"entries_from_masked_rects_le_64": int((bits * small * vis).sum()), "entries_from_whole_rects_over_64": int(nt2[~small].sum()),
                  "gaussians_with_binned_rect_over_64": int((~small & vis).sum()),
                  "over_64_histogram": {"65-128": int(((nt2 > 64) & (nt2 <= 128)).sum()), "129-256": int(((nt2 > 128) & (nt2 <= 256)).sum()),
                                        "257-1024": int(((nt2 > 256) & (nt2 <= 1024)).sum()), ">1024": int((nt2 > 1024).sum())}}
        out.append({"camera": ci, "binned": binned, "gaussians": model.num_points, "visible": int(vis.sum()),
                    "square_instances": int(nt.sum()), "square_instances_from_rects_over_64": int(nt[big].sum()),
                    "gaussians_with_rect_over_64": int(big.sum()), "list_entries_after_culling": pkg["handle"]["num_rendered"],
                    "rect_size_histogram": hist, "mean_opacity_of_big": float(op[big].mean()) if big.any() else None,
                    "median_radius_px": float(r[vis].median()), "p99_radius_px": float(r[vis].quantile(0.99)) if int(vis.sum()) < 16_000_000 else None})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
