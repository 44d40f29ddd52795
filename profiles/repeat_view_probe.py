"""Launch-to-launch repeatability of ONE view's backward (round 4): the forward + loss + gradient-writing backward of the two-rank
test's scene (5 000 Gaussians) repeated `reps` times per camera, every gradient bucket diffed against the first — an element
that moves by more than 3e-4 of its block's maximum is printed with its Gaussian.  This is what found the idle-lane race of the
per-Gaussian backward (DESIGN.md section 2.3 lessons): Gaussian P - 1 lost its gradient in 3-9 of 120 launches.
  usage: python3 profiles/repeat_view_probe.py [none|zeros|ff|nan|small|rand|unit] [reps]      (fill modes: tests/_poison.py)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "wheat-3dgs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import _poison
mode = sys.argv[1] if len(sys.argv) > 1 else ""
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
if mode and mode != "none":
    _poison.install(mode)
import test_gpu_two_ranks as T
from w3d_amd.fused_step import backward_raw, render_raw, finish
from w3d_amd.fused import l1_ssim_fwd_bwd
from w3d_amd.train import Trainer
dev = torch.device("cuda:0")
m, opt, cams = T._scene_and_cams(dev)
tr = Trainer(m, cams, opt, torch.zeros(3, device=dev), densify=False)
m.update_learning_rate(1)
sl = m.block_slices()
bad = 0
with torch.no_grad():
    for ci in range(2):
        cam = cams[tr.perm[ci % len(cams)]]
        ref = None
        for it in range(reps):
            sync = (it % 2 == 0)
            pkg = render_raw(cam, m, tr.bg, sync=sync, color_only=(it % 4 >= 2))
            if not sync and not finish(pkg["handle"]):
                print("overflow -> repeat"); pkg = render_raw(cam, m, tr.bg, sync=True)
            loss, dimg = l1_ssim_fwd_bwd(pkg["render"], cam.original_image, opt.lambda_dssim)
            gnorm, _ = backward_raw(m, pkg["handle"], dimg, want_norm=True)
            g = m.flat_grad.clone(); img = pkg["render"].clone()
            if ref is None:
                ref = (g, img, gnorm.clone()); continue
            dimg_ = float((img - ref[1]).abs().max())
            for name, (a, b) in sl.items():
                d = (g[a:b] - ref[0][a:b]).abs()
                rel = float(d.max() / (ref[0][a:b].abs().max() + 1e-30))
                # (3e-4 of the block's maximum since round 6: the tiles of a small frame run as four quadrant waves, DESIGN.md section 2.2,
                #  so a Gaussian receives four atomic partial sums per tile instead of one and the order of the float additions —
                #  free in the atomic mode, as in the reference's per-pixel atomics — moves a strongly cancelling sum by up to 1.8e-4
                #  (one Gaussian of 5 000, a recurring handful of values); a lost update, what this probe is for, is of order 1)
                if rel > 3e-4 or dimg_ > 1e-5:
                    bad += 1
                    idx = (d > 3e-4 * ref[0][a:b].abs().max()).nonzero().flatten()
                    dim = (b - a) // m.num_points
                    print(f"cam {ci} it {it} sync {sync} block {name}: rel {rel:.3e} img diff {dimg_:.2e}; {idx.numel()} elems; gaussians {sorted(set((idx // max(dim,1)).tolist()))[:12]}")
print(f"mode={mode} reps={reps} anomalies={bad}")
