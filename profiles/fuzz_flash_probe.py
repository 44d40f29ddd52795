"""Differential fuzzing of the FlashSplat forward against the CPU oracle (round 4): random small scenes, random label maps with
1 ... 9 objects (blocky, striped, random per pixel — the last puts more labels into a tile than the kernel's 4 register slots),
culls on and off alternating.  radii / proj_xy / gs_depth bit-identical, used_count within 1e-4 of its maximum, contrib_num
equal on all but threshold pixels, images within the parity bars.
  usage: python3 profiles/fuzz_flash_probe.py [cases] [seed0]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "wheat-3dgs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from w3d_amd.synth import make_scene, make_cameras
from test_gpu_parity import _settings, check_images
from util import view_inputs, make_oracle, np_inputs, rel_err
from flashsplat_rasterization import GaussianRasterizer

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda:0")
bad = 0
worst = 0.0
for case in range(cases):
    rs = np.random.RandomState(seed0 + case)
    g = torch.Generator().manual_seed(seed0 + case)
    P = int(rs.choice([1, 7, 64, 300, 1500]))
    W, H = int(rs.randint(8, 260)), int(rs.randint(8, 200))
    num_obj = int(rs.randint(1, 10))
    sc = make_scene(P, seed=seed0 + case, scale_mean=float(rs.choice([0.01, 0.03, 0.1])))
    sc.opacity[:] = torch.empty(P, 1).normal_(float(rs.choice([-3.0, 0.0, 3.0])), float(rs.choice([0.5, 2.0])), generator=g)
    cam = make_cameras(5, W, H)[int(rs.randint(5))]
    bg = (0.0, 0.0, 0.0)
    d = view_inputs(sc, cam)
    yy, xx = np.mgrid[0:H, 0:W]
    style = rs.choice(["blocks", "stripes", "noise"])
    if style == "blocks":
        mask = ((xx // int(rs.randint(3, 40)) + yy // int(rs.randint(3, 40))) % (num_obj + 1)).astype(np.float32)
    elif style == "stripes":
        mask = ((xx // int(rs.randint(1, 9))) % (num_obj + 1)).astype(np.float32)
    else:
        mask = rs.randint(0, num_obj + 1, (H, W)).astype(np.float32)
    cull = bool(case & 1)
    o = make_oracle(cam, bg, nthreads=8)
    ref = o.forward(**np_inputs(d), gt_mask=mask, num_obj=num_obj)
    rast = GaussianRasterizer(_settings(cam, bg, 3, 1.0, dev, flash=num_obj, tile_cull=cull))
    t = {k: (None if v is None else v.to(dev)) for k, v in d.items()}
    outs = rast(gt_mask=torch.as_tensor(mask, device=dev), unique_label=None, means3D=t["means3D"],
                means2D=torch.zeros(P, 3, device=dev), shs=t["shs"], colors_precomp=None, opacities=t["opacities"],
                scales=t["scales"], rotations=t["rotations"], cov3D_precomp=None)
    color, radii, depth, alpha, contrib_num, used_count, proj_xy, gs_depth = [x.cpu().numpy() for x in outs]
    msgs = []
    if not np.array_equal(radii, ref["radii"]): msgs.append("radii differ")
    if not np.array_equal(proj_xy, ref["proj_xy"]): msgs.append("proj_xy differs")
    if not np.array_equal(gs_depth, ref["gs_depth"]): msgs.append("gs_depth differs")
    e = rel_err(used_count, ref["used_count"])
    worst = max(worst, float(e))
    if e > 1e-4: msgs.append(f"used_count rel {e:.2e}")
    fr = float((contrib_num != ref["contrib_num"]).mean())
    if fr > 2e-3: msgs.append(f"contrib_num differs on {fr:.2e} of the pixels")
    try:
        check_images(dict(color=color, depth=depth, alpha=alpha), ref, "")
    except AssertionError as ex:
        msgs.append("images: " + str(ex)[:120])
    o.free()
    if msgs:
        bad += 1
        print(f"case {seed0 + case} (P={P}, {W}x{H}, {num_obj} objects, {style}, cull {cull}): " + "; ".join(msgs), flush=True)
print(f"cases {cases} from seed {seed0}: {bad} with differences; worst used_count error {worst:.1e}")
