"""Differential fuzzing against the CPU oracle (round 4): random small scenes (all shapes of fuzz_culls_probe.py), activated
inputs through the drop-in module with tile_cull off.  Integer work — radii, per-tile ranges, depth-ordered lists — must be
bit-identical to the oracle's for EVERY shape; images are compared for the well-conditioned shapes only (for needles two valid fp32
evaluations of the exponent differ by tens of percent in alpha: DESIGN.md section 2.3).
  usage: python3 profiles/fuzz_oracle_probe.py [cases] [seed0]"""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "wheat-3dgs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from w3d_amd.synth import make_scene, make_cameras
from test_gpu_parity import run_hip, check_images
from util import view_inputs, make_oracle, np_inputs

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
tot_vis = tot_entries = 0
for case in range(cases):
    rs = np.random.RandomState(seed0 + case)
    g = torch.Generator().manual_seed(seed0 + case)
    P = int(rs.choice([1, 7, 64, 300, 1500]))
    W, H = int(rs.randint(8, 260)), int(rs.randint(8, 200))
    sc = make_scene(P, seed=seed0 + case, scale_mean=float(rs.choice([0.002, 0.02, 0.2])))
    kind = rs.choice(["blob", "needle", "pancake", "mixed"])
    lo, hi = math.log(1e-5), math.log(6.0)
    if kind == "needle":
        sc.scaling[:, 0] = torch.empty(P).uniform_(math.log(0.05), hi, generator=g)
        sc.scaling[:, 1:] = torch.empty(P, 2).uniform_(lo, math.log(2e-3), generator=g)
    elif kind == "pancake":
        sc.scaling[:, :2] = torch.empty(P, 2).uniform_(math.log(0.05), math.log(3.0), generator=g)
        sc.scaling[:, 2] = torch.empty(P).uniform_(lo, math.log(1e-3), generator=g)
    elif kind == "mixed":
        sc.scaling[:] = torch.empty(P, 3).uniform_(lo, hi, generator=g)
    spread = float(rs.choice([1.0, 3.0, 10.0]))
    sc.xyz[:, :2] *= spread
    sc.xyz[:, 2] += float(rs.choice([0.0, 1.0, -2.0])) * torch.rand(P, generator=g)
    sc.opacity[:] = torch.empty(P, 1).normal_(float(rs.choice([-4.0, 0.0, 4.0])), float(rs.choice([0.5, 3.0])), generator=g)
    cam = make_cameras(5, W, H)[int(rs.randint(5))]
    bg = tuple(float(x) for x in rs.choice([0.0, 0.3], 3))
    deg = int(rs.randint(4))
    d = view_inputs(sc, cam, sh_degree=deg)
    o = make_oracle(cam, bg, sh_degree=deg, nthreads=8)
    ref = o.forward(**np_inputs(d))
    out, _ = run_hip(d, cam, bg, sh_degree=deg, tile_cull=False)
    msgs = []
    tot_vis += int((ref["radii"] > 0).sum())
    tot_entries += int(out["num_rendered"])
    if not np.array_equal(out["radii"], ref["radii"]):
        msgs.append(f"radii: {int((out['radii'] != ref['radii']).sum())} differ (max |d| {np.abs(out['radii'].astype(np.int64) - ref['radii']).max()})")
    else:
        ranges, pl = o.binning()
        if not np.array_equal(out["ranges"], ranges):
            msgs.append("tile ranges differ")
        elif not np.array_equal(out["point_list"], pl):
            msgs.append(f"lists differ in {int((out['point_list'] != pl).sum())} of {pl.size} entries")
    if kind == "blob" and not msgs:
        try:
            check_images(out, ref, "")
        except AssertionError as e:
            msgs.append("images: " + str(e)[:150])
    o.free()
    if msgs:
        bad += 1
        print(f"case {seed0 + case} ({kind}, P={P}, {W}x{H}, deg {deg}, spread {spread}): " + "; ".join(msgs), flush=True)
print(f"cases {cases} from seed {seed0}: {bad} with differences ({tot_vis} visible Gaussians, {tot_entries} list entries compared)")
