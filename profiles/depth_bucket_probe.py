#!/usr/bin/env python3
"""Bucket populations of the depth sort (w3d_binning.hip: 1024 buckets over the view's depth-key interval) on the benchmark scene:
max / mean / number of buckets beyond the LDS capacity, per camera.  Prints one JSON line per camera."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "wheat-3dgs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import bench
from w3d_amd.fused_step import render_raw
from w3d_amd.rasterizer import debug_gaussian_records

args = bench.parse_defaults()
dev = torch.device("cuda:0")
sc, model, opt, cams = bench.build_scene(args, dev)
bg = torch.zeros(3, device=dev)
for ci in (0, 5, 13, 20, 27, 35):
    with torch.no_grad():
        pkg = render_raw(cams[ci], model, bg, sync=True)
    rec = debug_gaussian_records(pkg["handle"])
    vis = pkg["radii"] > 0
    keys = rec[vis][:, 11].contiguous().view(torch.int32).to(torch.int64)
    kmin, kmax = int(keys.min()), int(keys.max())
    span = kmax - kmin
    shift = max(0, span.bit_length() - 10)
    b = torch.bincount((keys - kmin) >> shift, minlength=1024)
    print(json.dumps({"camera": ci, "visible": int(vis.sum()), "depth": [float(rec[vis][:, 11].min()), float(rec[vis][:, 11].max())], "shift": shift,
                      "buckets_used": int((b > 0).sum()), "max": int(b.max()), "mean_nonempty": float(b[b > 0].float().mean()),
                      "over_4096": int((b > 4096).sum()), "over_2048": int((b > 2048).sum()), "p99": int(b.float().quantile(0.99))}))
