"""scene legs (--full): the headline step on a model GROWN by the reference's densification schedule and on a scene of opaque
surfaces trained by the reference schedule as it is."""
import ctypes
import json
import math
import os
import subprocess
import sys
import time

import torch
import torch.distributed as dist

from .common import StepMeter, _psnr_db, mean_workload, roofline_object
from .scale_model import scale_model


DENSIFIED_GT_POINTS = 600_000


def densified_views(args, dev, bg):
    """(training views, held-out views, ground-truth scene) of the densified-scene leg: 36 renders of a 600 k-Gaussian scene,
    cameras 11-12 of every dozen held out (the reference's split, scene/dataset_readers.py:181-193)."""
    from w3d_amd.synth import make_scene, make_cameras
    from w3d_amd.gaussian_model import GaussianModel
    from w3d_amd.train import render_views
    cams = [c.to(dev) for c in make_cameras(36, args.width, args.height)]
    gt_sc = make_scene(DENSIFIED_GT_POINTS, seed=1, scale_mean=0.009)
    gt = GaussianModel(3, device=dev)
    gt.create_from_tensors(gt_sc.xyz, gt_sc.features_dc, gt_sc.features_rest, gt_sc.scaling, gt_sc.rotation, gt_sc.opacity)
    gt.active_sh_degree = 3
    for cam, img in zip(cams, render_views(gt, cams, bg)):
        cam.original_image = img.clamp(0.0, 1.0).contiguous()
    del gt
    torch.cuda.empty_cache()
    return [c for i, c in enumerate(cams) if i % 12 < 10], [c for i, c in enumerate(cams) if i % 12 >= 10], gt_sc


def grow_densified_model(args, dev, bg, iterations=None, log=None):
    """Config C3's regime: a model GROWN by the reference's densification schedule instead of a random one of the final size.
    A synthetic wheat-plot scene (600 k Gaussians, SURVEY section 8d generator) is rendered to the 36 views — 30 for training,
    cameras 11-12 of every dozen held out, the reference's split (scene/dataset_readers.py:181-193); a 250 k-point cloud of
    it goes through create_from_pcd (distCUDA2 scales) and is trained the way train_vanilla_3dgs.py:55-115 does, with the
    iteration counts compressed: densify_and_prune every 100 iterations from 300 until 70 % of `iterations`, opacity reset
    every max(1000, iterations/3), SH degree raised every 1000 — every one of these on the HIP path.
    Returns (model, opt, train_cams, held_cams, report)."""
    from collections import namedtuple
    from w3d_amd.gaussian_model import GaussianModel, OptimizationParams
    from w3d_amd.train import Trainer, render_views
    iterations = iterations or args.densify_iterations
    gt_points, init_points = DENSIFIED_GT_POINTS, 250_000
    train, held, gt_sc = densified_views(args, dev, bg)
    g = torch.Generator().manual_seed(2)
    sel = torch.randperm(gt_points, generator=g)[:init_points]
    pts = gt_sc.xyz[sel] + 0.004 * torch.randn(init_points, 3, generator=g)
    col = (0.28209479177387814 * gt_sc.features_dc[sel, 0] + 0.5).clamp(0, 1)
    PCD = namedtuple("BasicPointCloud", ["points", "colors", "normals"])

    opt = OptimizationParams()                # (instance attributes override the class defaults)
    opt.iterations = iterations
    opt.densify_from_iter = 300
    opt.densify_until_iter = int(0.7 * iterations)
    opt.densification_interval = 100
    opt.opacity_reset_interval = max(1000, iterations // 3)
    opt.position_lr_max_steps = iterations
    opt.densify_grad_threshold = args.densify_grad_threshold
    m = GaussianModel(3, device=dev)
    m.create_from_pcd(PCD(pts.numpy(), col.numpy(), None), 1.0)
    m.training_setup(opt)
    tr = Trainer(m, train, opt, bg, densify=True, cameras_extent=2.0, spatial_order=not args.no_spatial_order)

    def quality(views):
        return sum(_psnr_db(i, v.original_image) for i, v in zip(render_views(m, views, bg), views)) / len(views)
    q0 = (quality(train), quality(held))
    trace = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(1, iterations + 1):
        tr.step(it)
        if it % 500 == 0 or it == iterations:
            torch.cuda.synchronize()
            trace.append([it, m.num_points, round(time.perf_counter() - t0, 2)])
            if log:
                log(f"densified scene: iteration {it}, {m.num_points} Gaussians")
    torch.cuda.synchronize()
    t_train = time.perf_counter() - t0
    q1 = (quality(train), quality(held))
    report = {"schedule": {"iterations": iterations, "densify_from_iter": opt.densify_from_iter,
                           "densify_until_iter": opt.densify_until_iter, "densification_interval": opt.densification_interval,
                           "opacity_reset_interval": opt.opacity_reset_interval,
                           "densify_grad_threshold": opt.densify_grad_threshold,
                           "reference_densify_grad_threshold": 0.0002, "initial_points": init_points,
                           "views": "30 training + 6 held out of 36"},
              "gaussians": m.num_points, "peak_gaussians": max(t[1] for t in trace), "train_seconds": round(t_train, 2),
              "iters_per_s_overall": round(iterations / t_train, 1),
              "psnr_train_db_before_after": [round(q0[0], 2), round(q1[0], 2)],
              "psnr_heldout_db_before_after": [round(q0[1], 2), round(q1[1], 2)],
              "parameters_finite": bool(torch.isfinite(m.flat).all()), "trace_iteration_gaussians_seconds": trace}
    return m, opt, train, held, report


def densified_scene(args, dev, bg, log, with_scale_model=False):
    """The headline measurement on the densified model: --steps fixed-P steps (no densification inside the timed region,
    iteration numbers continue after the schedule), with its own dominant-kernel roofline, stage table and workload."""
    from w3d_amd.train import Trainer
    m, opt, train, held, rep = grow_densified_model(args, dev, bg, log=log)
    tr = Trainer(m, train, opt, bg, densify=False, spatial_order=not args.no_spatial_order)
    meter = StepMeter(tr, 1, dev)
    it = opt.iterations
    opt.iterations = 10 ** 9                       # (the step past `iterations` skips the optimizer: keep stepping)
    opt.densify_until_iter = 10 ** 9                # statistics tracked as in the headline; Trainer(densify=False) keeps P fixed
    for _ in range(max(5, args.warmup)):
        it += 1
        tr.step(it)
    meas = meter.measure(args.steps, it, args.profile, args.all_stages)
    ws = mean_workload(m, train, bg, dev)
    P, HW = m.num_points, args.width * args.height
    ips = args.steps / meas["elapsed"]
    rep.update(value=round(ips, 3), ms_per_step=round(1e3 * meas["elapsed"] / args.steps, 4), steps=args.steps,
               stage_ms=meas["stage_ms"], visible_per_view=int(ws["V"]), tile_instances_per_view=int(ws["R"]),
               walked_instances_per_view=int(ws["R_walk"]), mean_contributors_per_pixel=round(ws["mean_contrib"], 2), mean_last_contributor_list_position=round(ws["mean_last"], 2),
               final_loss=round(float(tr.last["loss"]), 6),
               roofline=roofline_object(meas, P, ws, HW, True, ips, "densified"))
    if with_scale_model:
        log("scale model: densified scene")
        try:
            rep["scale_model"], _ = scale_model(m, opt, train, bg, dev, meas["it"] + 64)
        except Exception as e:
            rep["scale_model"] = {"error": repr(e)}
    return rep, m


# ------------------------------------------------------------------------------------------------ opaque-surface scene
def opaque_views(args, dev, bg):
    """(training views, held-out views, ground-truth scene, ground-truth workload) of the opaque-surface leg"""
    from w3d_amd.gaussian_model import GaussianModel
    from w3d_amd.synth import make_cameras, make_opaque_scene
    from w3d_amd.train import render_views
    cams = [c.to(dev) for c in make_cameras(36, args.width, args.height)]
    # (sized like the benchmark: ~1.9 M opaque Gaussians in the ground truth, a 1 M-point cloud to start from)
    gt_sc = make_opaque_scene(seed=3, ground=1_400_000, heads=8000, per_head=50, per_stem=15)
    gt = GaussianModel(3, device=dev)
    gt.create_from_tensors(gt_sc.xyz, gt_sc.features_dc, gt_sc.features_rest, gt_sc.scaling, gt_sc.rotation, gt_sc.opacity)
    gt.active_sh_degree = 3
    for cam, img in zip(cams, render_views(gt, cams, bg)):
        cam.original_image = img.clamp(0.0, 1.0).contiguous()
    gt_ws = mean_workload(gt, cams, bg, dev)
    del gt
    torch.cuda.empty_cache()
    return [c for i, c in enumerate(cams) if i % 12 < 10], [c for i, c in enumerate(cams) if i % 12 >= 10], gt_sc, gt_ws


def opaque_scene(args, dev, bg, log, with_scale_model=False, return_model=False):
    """A trained scene that SATURATES like a photographed one.  The benchmark scene is a random translucent slab (its pixels
    saturate after 15 % of their lists, 107 k of 1.2 M visible Gaussians get a gradient) and the densified leg fits renders of
    such a slab (724 contributors per pixel at the end); a real 3DGS model is made of opaque surfaces, tens of contributors per
    pixel.  Here the ground truth is synth.make_opaque_scene — a sheet of opaque ground discs with ears on stems — seen by the
    same 36 cameras (30 training, 6 held out); a 1 M-point cloud of it goes through create_from_pcd and the REFERENCE schedule
    as it is (arguments/__init__.py:73-89: 15 000 iterations, densify_and_prune every 100 from 500 to 11 000 at
    densify_grad_threshold 2e-4, opacity reset every 3 000, SH degree up every 1 000, position_lr_max_steps 30 000) to whatever
    size that reaches (`as_trained`); then the model is padded to ~--points Gaussians by rounds of densify_and_prune — the
    reference's own clone / split rule with the threshold at the quantile of the accumulated gradient norms that closes the
    gap —, settled for 600 steps, and measured again (`padded`): BASELINE.json's size with a converged scene's walk statistics."""
    from collections import namedtuple
    from w3d_amd.gaussian_model import GaussianModel, OptimizationParams
    from w3d_amd.rasterizer import list_share_of
    from w3d_amd.synth import make_cameras, make_opaque_scene
    from w3d_amd.train import Trainer, render_views
    iterations = args.opaque_iterations
    train, held, gt_sc, gt_ws = opaque_views(args, dev, bg)
    g = torch.Generator().manual_seed(4)
    init_points = 1_000_000
    sel = torch.randperm(gt_sc.P, generator=g)[:init_points]
    pts = gt_sc.xyz[sel] + 0.0015 * torch.randn(init_points, 3, generator=g)
    col = (0.28209479177387814 * gt_sc.features_dc[sel, 0] + 0.5).clamp(0, 1)
    PCD = namedtuple("BasicPointCloud", ["points", "colors", "normals"])
    opt = OptimizationParams()                # the reference's defaults, unchanged except the iteration count when shortened
    opt.iterations = iterations
    if iterations < 15_000:                   # (a shortened run keeps the proportions of the schedule)
        opt.densify_until_iter = int(iterations * 11_000 / 15_000)
    m = GaussianModel(3, device=dev)
    m.create_from_pcd(PCD(pts.numpy(), col.numpy(), None), 1.0)
    m.training_setup(opt)
    tr = Trainer(m, train, opt, bg, densify=True, cameras_extent=2.0, spatial_order=not args.no_spatial_order)

    def quality(views):
        return sum(_psnr_db(i, v.original_image) for i, v in zip(render_views(m, views, bg), views)) / len(views)
    q0 = (quality(train), quality(held))
    trace = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(1, iterations + 1):
        tr.step(it)
        if it % 1000 == 0 or it == iterations:
            torch.cuda.synchronize()
            trace.append([it, m.num_points, round(time.perf_counter() - t0, 2)])
            log(f"opaque scene: iteration {it}, {m.num_points} Gaussians")
    torch.cuda.synchronize()
    t_train = time.perf_counter() - t0
    q1 = (quality(train), quality(held))
    rep = {"ground_truth": {"gaussians": gt_sc.P, "mean_contributors_per_pixel": round(gt_ws["mean_contrib"], 2),
                            "mean_last_contributor_list_position": round(gt_ws["mean_last"], 2),
                            "what": "synth.make_opaque_scene(seed=3, ground=1.4 M discs, 8000 ears x 50 on stems x 15)"},
           "schedule": {"iterations": iterations, "densify_from_iter": opt.densify_from_iter, "densify_until_iter": opt.densify_until_iter,
                        "densification_interval": opt.densification_interval, "opacity_reset_interval": opt.opacity_reset_interval,
                        "densify_grad_threshold": opt.densify_grad_threshold, "position_lr_max_steps": opt.position_lr_max_steps,
                        "initial_points": init_points, "views": "30 training + 6 held out of 36",
                        "reference": "arguments/__init__.py:73-89 as it is" if iterations == 15_000 else "arguments/__init__.py:73-89, shortened"},
           "train_seconds": round(t_train, 2), "iters_per_s_overall": round(iterations / t_train, 1),
           "psnr_train_db_before_after": [round(q0[0], 2), round(q1[0], 2)],
           "psnr_heldout_db_before_after": [round(q0[1], 2), round(q1[1], 2)],
           "parameters_finite": bool(torch.isfinite(m.flat).all()), "trace_iteration_gaussians_seconds": trace}

    def measure(tag, it):
        t = Trainer(m, train, opt, bg, densify=False, spatial_order=not args.no_spatial_order)
        meter = StepMeter(t, 1, dev)
        for _ in range(max(5, args.warmup)):
            it += 1
            t.step(it)
        meas = meter.measure(args.steps, it, args.profile, args.all_stages)
        ws = mean_workload(m, train, bg, dev)
        P, HW = m.num_points, args.width * args.height
        ips = args.steps / meas["elapsed"]
        out = dict(gaussians=P, value=round(ips, 3), ms_per_step=round(1e3 * meas["elapsed"] / args.steps, 4), steps=args.steps,
                   stage_ms=meas["stage_ms"], visible_per_view=int(ws["V"]), tile_instances_per_view=int(ws["R"]),
                   walked_instances_per_view=int(ws["R_walk"]), mean_contributors_per_pixel=round(ws["mean_contrib"], 2), mean_last_contributor_list_position=round(ws["mean_last"], 2),
                   list_share=list_share_of(m), walk_fraction=None if t.share_rho is None else round(t.share_rho, 3),
                   roofline=roofline_object(meas, P, ws, HW, True, ips, "opaque_" + tag))
        return out, meas["it"]
    it = iterations
    opt.iterations = opt.densify_until_iter = 10 ** 9        # keep stepping and tracking statistics; Trainer(densify=False) keeps P
    log("opaque scene: measuring as trained")
    rep["as_trained"], it = measure("as_trained", it)
    # ---- pad to --points with the reference's own clone / split rule
    target = args.points
    if m.num_points < 0.97 * target:
        t = Trainer(m, train, opt, bg, densify=False, spatial_order=not args.no_spatial_order)
        rounds = []
        for rnd in range(4):                       # (a Gaussian is cloned / split once per round: the gap may take several)
            if m.num_points >= 0.97 * target:
                break
            m._reset_stats()
            for _ in range(150):                   # five views of every training camera: fresh densification statistics
                it += 1
                t.step(it)
            grads = (m.xyz_gradient_accum / m.denom).nan_to_num_(0.0).reshape(-1)
            k = min(target - m.num_points, int((grads > 0).sum()) - 1)
            if k < 1:
                break
            thr = float(torch.topk(grads, k).values[-1])
            torch.manual_seed(99 + rnd)
            before = m.num_points
            m.densify_and_prune(thr, 0.005, 2.0, None)
            rounds.append({"max_grad": float(f"{thr:.3e}"), "from": before, "to": m.num_points})
        for _ in range(600):
            it += 1
            t.step(it)
        q2 = (quality(train), quality(held))
        log(f"opaque scene: padded to {m.num_points} Gaussians, measuring")
        rep["padded"], it = measure("padded", it)
        rep["padded"].update(how="rounds of densify_and_prune(max_grad = the quantile of the mean gradient norms of 150 steps that closes "
                                 "the gap to --points, min_opacity 0.005, no size threshold), then 600 steps at fixed size",
                             rounds=rounds, psnr_train_db=round(q2[0], 2), psnr_heldout_db=round(q2[1], 2))
        if with_scale_model:
            log("scale model: opaque scene (padded)")
            try:
                rep["scale_model"], _ = scale_model(m, opt, train, bg, dev, it + 64)
            except Exception as e:
                rep["scale_model"] = {"error": repr(e)}
    if return_model:
        return rep, m, opt, train, it
    return rep

