#!/usr/bin/env python
"""Config C3 end to end on one MI355X: "vanilla 3DGS train, full densify to ~2M Gaussians + depth/alpha channels" at 1600x1200.

A synthetic wheat-plot scene (SURVEY.md section 8d generator; no wheat data ships with the reference) is rendered to 36 views —
30 for training, cameras 11-12 of every dozen held out, as the reference's split (scene/dataset_readers.py:181-193) — and a
model is trained against them the way train_vanilla_3dgs.py does: points -> create_from_pcd (distCUDA2 initial scales),
the reference's learning rates and loss, densify_and_prune every 100 iterations (clone / split / prune with the
reference's thresholds), an opacity reset, SH degree raised every 1000 iterations — every one of these on the HIP path:
knn grid kernel, raw-parameter forward / backward, fused loss, fused Adam, densification statistics in the backward kernel,
one-pass compaction.  The iteration counts are compressed (the reference trains 15 000 iterations with densification until
11 000; here `--iterations`, densification from 300 until 70 % of them) so that the run takes about a minute.

Prints one JSON object: Gaussian count over time, wall time per phase, PSNR on training and held-out views before / after.
    python profiles/c3_densify_run.py --iterations 6000 > gpurun_out/c3_densify_run.json
"""
import argparse
import json
import os
import sys
import time
from collections import namedtuple

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "wheat-3dgs_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402


def psnr(a, b):
    mse = float(((a - b) ** 2).mean())
    return 99.0 if mse == 0 else -10.0 * __import__("math").log10(mse)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=6000)
    ap.add_argument("--gt-points", type=int, default=600_000)
    ap.add_argument("--init-points", type=int, default=250_000)
    ap.add_argument("--width", type=int, default=1600)
    ap.add_argument("--height", type=int, default=1200)
    ap.add_argument("--grad-threshold", type=float, default=0.0002,
                    help="densify_grad_threshold (reference default 0.0002, arguments/__init__.py:88 — tuned for photographs; the "
                         "synthetic views have smoother gradients, so reaching ~2 M Gaussians takes a lower value)")
    a = ap.parse_args()
    from w3d_amd.synth import make_scene, make_cameras
    from w3d_amd.gaussian_model import GaussianModel, OptimizationParams
    from w3d_amd.train import Trainer, render_views
    dev = torch.device("cuda:0")
    bg = torch.zeros(3, device=dev)
    cams = [c.to(dev) for c in make_cameras(36, a.width, a.height)]
    gt_sc = make_scene(a.gt_points, seed=1, scale_mean=0.009)
    gt = GaussianModel(3, device=dev)
    gt.create_from_tensors(gt_sc.xyz, gt_sc.features_dc, gt_sc.features_rest, gt_sc.scaling, gt_sc.rotation, gt_sc.opacity)
    gt.active_sh_degree = 3
    for cam, img in zip(cams, render_views(gt, cams, bg)):
        cam.original_image = img.clamp(0.0, 1.0).contiguous()
    train = [c for i, c in enumerate(cams) if i % 12 < 10]
    held = [c for i, c in enumerate(cams) if i % 12 >= 10]
    # a sparse point cloud of the scene: a subset of the true positions with their base colours, jittered (what COLMAP hands over)
    g = torch.Generator().manual_seed(2)
    sel = torch.randperm(a.gt_points, generator=g)[:a.init_points]
    pts = gt_sc.xyz[sel] + 0.004 * torch.randn(a.init_points, 3, generator=g)
    col = (0.28209479177387814 * gt_sc.features_dc[sel, 0] + 0.5).clamp(0, 1)
    PCD = namedtuple("BasicPointCloud", ["points", "colors", "normals"])
    del gt
    torch.cuda.empty_cache()

    class Opt(OptimizationParams):
        iterations = a.iterations
        densify_from_iter = 300
        densify_until_iter = int(0.7 * a.iterations)
        densification_interval = 100
        opacity_reset_interval = max(1000, a.iterations // 3)
        position_lr_max_steps = a.iterations
        densify_grad_threshold = a.grad_threshold
    opt = Opt()
    m = GaussianModel(3, device=dev)
    t0 = time.perf_counter()
    m.create_from_pcd(PCD(pts.numpy(), col.numpy(), None), 1.0)
    torch.cuda.synchronize()
    t_init = time.perf_counter() - t0
    m.training_setup(opt)
    tr = Trainer(m, train, opt, bg, densify=True, cameras_extent=2.0)

    def quality(views):
        imgs = render_views(m, views, bg)
        return sum(psnr(i, v.original_image) for i, v in zip(imgs, views)) / len(views)
    q0 = (quality(train), quality(held))
    trace, losses = [], []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(1, a.iterations + 1):
        loss = tr.step(it)
        if it % 250 == 0 or it == a.iterations:
            torch.cuda.synchronize()
            trace.append({"iteration": it, "gaussians": m.num_points, "seconds": round(time.perf_counter() - t0, 2),
                          "loss": round(float(loss), 5), "sh_degree": m.active_sh_degree})
    torch.cuda.synchronize()
    t_train = time.perf_counter() - t0
    q1 = (quality(train), quality(held))
    finite = bool(torch.isfinite(m.flat).all())
    print(json.dumps({"config": "C3 end to end: densify to ~2M Gaussians at %dx%d, 30 training + 6 held-out views" % (a.width, a.height),
                      "iterations": a.iterations, "densify_grad_threshold": a.grad_threshold, "initial_points": a.init_points, "final_gaussians": m.num_points,
                      "knn_init_seconds": round(t_init, 3), "train_seconds": round(t_train, 2),
                      "iters_per_s_overall": round(a.iterations / t_train, 1),
                      "psnr_train_before_after": [round(q0[0], 2), round(q1[0], 2)],
                      "psnr_heldout_before_after": [round(q0[1], 2), round(q1[1], 2)],
                      "parameters_finite": finite, "trace": trace}))


if __name__ == "__main__":
    main()
