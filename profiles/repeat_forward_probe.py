"""Bitwise repeatability of the forward (round 4): the forward has no floating-point atomics, so colour / depth / alpha / radii,
the per-tile lists and the FlashSplat outputs of the same inputs must come out IDENTICAL launch after launch — any race in the
preprocess, the depth sort, the (chunk, band) walkers or the blend shows as a differing bit.  Ragged sizes on purpose.
  usage: python3 profiles/repeat_forward_probe.py [reps]        (fill modes of tests/_poison.py through W3D_TEST_FILL)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "wheat-3dgs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import _poison
_poison.install_from_env()
from w3d_amd.synth import make_scene, make_cameras
from w3d_amd.gaussian_model import GaussianModel
from w3d_amd.fused_step import render_raw, finish

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
bad = 0
cases = [(1, 17, 16), (63, 40, 56), (65, 75, 133), (257, 203, 149), (5000, 176, 144), (6999, 208, 160), (100003, 640, 480),
         (500000, 1600, 1200), (2000000, 1600, 1200)]
for P, W, H in cases:
    cams = [c.to(dev) for c in make_cameras(3, W, H)]
    sc = make_scene(P, seed=P % 97, scale_mean=0.03 if P < 200000 else 0.006)
    m = GaussianModel(3, device=dev)
    m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
    m.active_sh_degree = 3
    bg = torch.tensor([0.2, 0.1, 0.0], device=dev)
    g = torch.Generator().manual_seed(P)
    mask = (torch.rand(H, W, generator=g) > 0.5).float().to(dev)
    n = reps if P <= 500000 else max(4, reps // 4)
    for ci, cam in enumerate(cams[:2]):
        for flash in (None, dict(num_obj=1, gt_mask=mask)):
            ref = None
            for it in range(n):
                sync = it % 2 == 0
                with torch.no_grad():
                    pkg = render_raw(cam, m, bg, sync=sync, flash=flash)
                    if not sync and not finish(pkg["handle"]):
                        pkg = render_raw(cam, m, bg, sync=True, flash=flash)
                h = pkg["handle"]
                R = int(h["num_rendered"])
                cur = dict(color=pkg["render"], depth=pkg["depth"], alpha=pkg["alpha"], radii=pkg["radii"], lists=h["point_list"][:R].clone(),
                           R=torch.tensor([R, int(h["num_visible"])]))
                if flash is not None:
                    cur.update(contrib=pkg["contrib_num"], proj=pkg["proj_xy"], gsd=pkg["gs_depth"])
                    # (used_count is summed with float atomics: compared with a tolerance)
                    cur["used"] = pkg["used_count"]
                cur = {k: v.clone() for k, v in cur.items()}
                if ref is None:
                    ref = cur
                    continue
                for k in cur:
                    if k == "used":
                        d = float((cur[k] - ref[k]).abs().max() / (ref[k].abs().max() + 1e-30))
                        ok = d <= 1e-5
                    else:
                        ok = cur[k].shape == ref[k].shape and torch.equal(cur[k], ref[k])
                        d = -1.0
                    if not ok:
                        bad += 1
                        nd = int((cur[k] != ref[k]).sum()) if cur[k].shape == ref[k].shape else -1
                        print(f"P={P} {W}x{H} cam {ci} flash={flash is not None} launch {it}: {k} differs ({nd} elements, rel {d:.2e})")
    print(f"P={P} {W}x{H}: done, anomalies so far {bad}", flush=True)
print("TOTAL", bad)
