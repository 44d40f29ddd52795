#!/usr/bin/env python
"""Blend-backward walk statistics of one scene under the three list_share modes (library built with -DW3D_BWD_STATS:
W3D_HIP_LIB=profiles/_bin/variants/stats/libw3d_hip.so python profiles/share_stats.py --scene densified --model-file /tmp/x.pt).
Per mode and camera: list length R, entries staged by the backward's waves, entries with a non-empty quadrant mask, quadrant
evaluations started / reaching the exponential / blending, and event-timed forward / backward stage times."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "wheat-3dgs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import bench  # noqa: E402
from walk_stats import stats  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="densified")
    ap.add_argument("--model-file", required=True)
    a = ap.parse_args()
    from w3d_amd.gaussian_model import GaussianModel, OptimizationParams
    args = bench.parse_defaults()
    dev = torch.device("cuda:0")
    bg = torch.zeros(3, device=dev)
    pack = torch.load(a.model_file, weights_only=False)
    model = GaussianModel(3, device=dev)
    model.restore(pack["model"], pack.get("opt", OptimizationParams()))
    if a.scene == "densified":
        cams = bench.densified_views(args, dev, bg)[0]
    else:
        _, _, _, cams = bench.build_scene(args, dev)
    out = {}
    for share in (0, 1, 2):
        model.list_share = share
        out[f"share{share}"] = [stats(model, cams[i], bg) for i in (0, 7, 18)]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
