#!/usr/bin/env python
"""Where the blend backward's per-entry work goes (profiles/README.md): run with a library built with -DW3D_BWD_STATS
(W3D_HIP_LIB=... python profiles/walk_stats.py).  The instrumented kernel adds, per launch, into counters[8..13] of the view's
state buffer: entries staged, entries with a non-empty quadrant mask, (entry, quadrant) evaluations started, of those
the ones that reached the exponential, of those the ones some lane blended, waves.  Untrained and trained C3 scene."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "wheat-3dgs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import bench  # noqa: E402


def stats(model, cam, bg):
    from w3d_amd.fused_step import render_raw, backward_raw
    dev = model.flat.device
    with torch.no_grad():
        pkg = render_raw(cam, model, bg, sync=True)
        h = pkg["handle"]
        ctr = h["state"][:64].view(torch.int32)
        ctr[8:14] = 0
        g = torch.randn(3, cam.image_height, cam.image_width, device=dev)
        backward_raw(model, h, g)
        torch.cuda.synchronize()
        c = [int(x) & 0xFFFFFFFF for x in ctr[8:14].tolist()]
    return dict(R=h["num_rendered"], staged=c[0], entries_nonempty_mask=c[1], quadrant_evals=c[2], reached_exp=c[3],
                blended=c[4], waves=c[5])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trained-steps", type=int, default=3000)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    args = argparse.Namespace(points=2_000_000, width=1600, height=1200, views=36)
    bg = torch.zeros(3, device=dev)
    sc, model, opt, cams = bench.build_scene(args, dev)
    bench.make_ground_truth(args, cams, dev, bg)
    out = {"untrained": [stats(model, cams[i], bg) for i in (0, 18)]}
    from w3d_amd.train import Trainer
    tr = Trainer(model, cams, opt, bg, densify=False)
    for it in range(1, a.trained_steps + 1):
        tr.step(it)
    out["trained"] = [stats(model, cams[i], bg) for i in (0, 18)]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
