#!/usr/bin/env python3
"""Depth sort under a skewed depth distribution: the benchmark scene with a fraction of its Gaussians strewn far behind the slab
(a background 30-60 units away: the view's depth-key interval grows to 5 octaves, the slab keeps 1024 x 14 % of the equal-width
buckets).  Prints the depth_sort stage time (HIP events, 40 forwards) for several far fractions and the bucket populations of
the grid the library builds (tests/depth_grid_model.py restates it) next to those of equal-width buckets."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "wheat-3dgs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import bench
from w3d_amd import _lib
from w3d_amd.fused_step import render_raw
from w3d_amd.gaussian_model import GaussianModel
from w3d_amd.rasterizer import debug_gaussian_records
from w3d_amd.synth import make_scene
from depth_grid_model import grid_buckets, summary

lib = _lib.lib
lib.w3d_profile_enable.argtypes = [ctypes.c_char_p]
lib.w3d_profile_collect.argtypes = [ctypes.c_char_p, ctypes.c_uint64]
args = bench.parse_defaults()
dev = torch.device("cuda:0")
cams = [c.to(dev) for c in __import__("w3d_amd.synth", fromlist=["make_cameras"]).make_cameras(36, 1600, 1200)]
bg = torch.zeros(3, device=dev)
out = []
for far in (0.0, 0.001, 0.01, 0.1):
    sc = make_scene(2_000_000, seed=0)
    g = torch.Generator().manual_seed(1)
    sel = torch.rand(sc.P, generator=g) < far
    sc.xyz[sel, 2] = -30.0 - 30.0 * torch.rand(int(sel.sum()), generator=g)
    m = GaussianModel(3, device=dev)
    m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
    m.active_sh_degree = 3
    m.sort_spatially()
    cam = cams[0]
    with torch.no_grad():
        for _ in range(5):
            pkg = render_raw(cam, m, bg, sync=True)
        rec = debug_gaussian_records(pkg["handle"]); vis = pkg["radii"] > 0
        keys = rec[:, 11].contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
        keys[~vis] = 0xFFFFFFFF
        pop, widths, nbk = grid_buckets(keys.cpu().numpy())
        kv = keys[vis]
        kmin, kmax = int(kv.min()), int(kv.max())
        b = torch.bincount(((kv - kmin) * ((1 << 42) // (kmax - kmin + 1))) >> 32, minlength=1024)       # equal-width buckets (round 6's first grid)
        torch.cuda.synchronize()
        lib.w3d_profile_enable(b"depth_sort")
        for _ in range(40):
            render_raw(cam, m, bg, sync=True)
        torch.cuda.synchronize()
        lib.w3d_profile_enable(None)
    buf = ctypes.create_string_buffer(1 << 12)
    lib.w3d_profile_collect(buf, len(buf))
    name, cnt, ms = buf.value.decode().split()
    out.append({"far_fraction": far, "visible": int(vis.sum()), "depth_sort_ms": round(float(ms) / int(cnt), 4), "grid": summary(pop, widths),
                "equal_width": {"max": int(b.max()), "over_4096": int((b > 4096).sum()), "over_8192": int((b > 8192).sum()), "empty": int((b == 0).sum())}})
    del m
print(json.dumps(out))
