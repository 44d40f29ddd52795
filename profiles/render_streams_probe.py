#!/usr/bin/env python3
"""Forward-only render rate (bench.py's render_mpix_per_s: reference render.py:24-35) against train.RENDER_STREAMS, the number of
frames in flight on separate HIP streams.  One JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "wheat-3dgs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import bench
from w3d_amd import train

args = bench.parse_defaults()
dev = torch.device("cuda:0")
sc, model, opt, cams = bench.build_scene(args, dev)
bg = torch.zeros(3, device=dev)
model.sort_spatially()
out = {}
views = [cams[i % len(cams)] for i in range(72)]
for n in ([int(a) for a in sys.argv[1:]] or (1, 2, 3, 4, 6, 8)):
    train.RENDER_STREAMS = n
    model._render_streams = None
    best = 0.0
    for rep in range(3):
        train.render_views(model, views, bg)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        train.render_views(model, views, bg)
        torch.cuda.synchronize()
        best = max(best, len(views) * args.width * args.height / 1e6 / (time.perf_counter() - t0))
    out[n] = round(best, 1)
print(json.dumps({"render_mpix_per_s_by_streams": out}))
