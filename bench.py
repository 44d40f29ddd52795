#!/usr/bin/env python
"""bench.py — the headline measurement: train-step iters/sec (and forward render Mpix/sec) at
2M Gaussians, 1600x1200 (BASELINE.json `metric`, config C3) on synthetic data.

A step is the loop body of reference train_vanilla_3dgs.py:55-115 (render, 0.8*L1+0.2*(1-SSIM),
backward, densification statistics, Adam step, zero_grad) at fixed P (no densify/prune inside the
timed region).  With N>1 GPUs every rank renders a different camera per step (view-parallel,
weak scaling) and the 59xP gradient bucket is all-reduced over RCCL; `value` counts views/sec.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "wheat-3dgs_amd"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--points", type=int, default=2_000_000)
    ap.add_argument("--width", type=int, default=1600)
    ap.add_argument("--height", type=int, default=1200)
    ap.add_argument("--views", type=int, default=36)
    ap.add_argument("--profile", default="render_bwd", help="stage timed with HIP events for the roofline object")
    ap.add_argument("--all-stages", action="store_true", help="also print per-stage event times to stderr")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--torch-loss", action="store_true", help="use the PyTorch conv2d SSIM instead of the fused kernel")
    ap.add_argument("--autograd-path", action="store_true",
                    help="run the drop-in render()+autograd step instead of the fused raw-parameter step")
    return ap.parse_args()


def build_scene(args, dev):
    from w3d_amd.synth import make_scene, make_cameras
    from w3d_amd.gaussian_model import GaussianModel, OptimizationParams
    sc = make_scene(args.points, seed=0)
    model = GaussianModel(3, device=dev)
    model.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
    model.active_sh_degree = 3
    opt = OptimizationParams()
    model.training_setup(opt)
    cams = [c.to(dev) for c in make_cameras(args.views, args.width, args.height)]
    return sc, model, opt, cams


def make_ground_truth(args, cams, dev, bg):
    """GT image of each view = render of a DIFFERENT seed's scene + noise, so the loss gradient is dense."""
    from w3d_amd.synth import make_scene
    from w3d_amd.gaussian_model import GaussianModel
    from w3d_amd.train import render_views
    sc = make_scene(max(args.points // 4, 1000), seed=1, scale_mean=0.009)
    gt_model = GaussianModel(3, device=dev)
    gt_model.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
    gt_model.active_sh_degree = 3
    g = torch.Generator(device="cpu").manual_seed(7)
    for cam, img in zip(cams, render_views(gt_model, cams, bg)):
        noise = 0.03 * torch.randn(img.shape, generator=g).to(dev)
        cam.original_image = (img + noise).clamp(0.0, 1.0).contiguous()
    del gt_model
    torch.cuda.empty_cache()


def workload_stats(model, cam, bg, dev):
    """Measured V, R and R_walk (entries the reverse walk must visit) of one view."""
    from w3d_amd.rasterizer import _forward_impl, debug_pixel_state
    from w3d_amd.gaussian_renderer import _settings
    from w3d_amd.rasterizer import GaussianRasterizationSettings
    with torch.no_grad():
        s = _settings(GaussianRasterizationSettings, cam, model, bg, 1.0, False)
        _, radii, _, _, saved, _ = _forward_impl(s, model.get_xyz, model.get_features, None, model.get_opacity,
                                                 model.get_scaling, model.get_rotation, None)
        _, nc = debug_pixel_state(saved)
        H, W = nc.shape
        gy, gx = (H + 15) // 16, (W + 15) // 16
        pad = torch.zeros(gy * 16, gx * 16, dtype=torch.int64, device=dev)
        pad[:H, :W] = nc.to(torch.int64)
        r_walk = int(pad.view(gy, 16, gx, 16).amax(dim=(1, 3)).sum())
        return dict(V=saved["num_visible"], R=saved["num_rendered"], R_walk=r_walk,
                    mean_contrib=float(nc.float().mean()))


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary
    (profiles/rNN/pmc_hbm_traffic.csv: FETCH_SIZE x2 + WRITE_SIZE, collected in separate --pmc passes);
    None when no summary exists.  The counters cannot be read live from inside the process."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_hbm_traffic.csv")))
    if not files:
        return None
    best = None
    for r in csv.DictReader(open(files[-1])):
        if kernel in r["kernel"] and "<true>" not in r["kernel"].replace("<true, true>", ""):
            mib = float(r["hbm_read_MiB_corrected_x2"]) + float(r["hbm_write_MiB"])
            best = max(best or 0.0, mib)
    return None if best is None else int(best * 1024 * 1024)


def cpu_baseline(args):
    """The oracle (kind "port") timed on this box's host cores on a bounded sample: ONE full
    train-step's rasterizer work (forward + backward of one 1600x1200 view of the 2M scene)."""
    import numpy as np
    from util import view_inputs, make_oracle, np_inputs
    from w3d_amd.synth import make_scene, make_cameras
    cores = os.cpu_count() or 1
    sc = make_scene(args.points, seed=0)
    cam = make_cameras(args.views, args.width, args.height)[0]
    d = np_inputs(view_inputs(sc, cam))
    o = make_oracle(cam, (0.0, 0.0, 0.0), nthreads=cores)
    gc = np.random.RandomState(0).randn(3, args.height, args.width).astype(np.float32)
    t0 = time.perf_counter()
    o.forward(**d)
    t1 = time.perf_counter()
    o.backward(gc, None, None)
    t2 = time.perf_counter()
    o.free()
    return {"value": round(1.0 / (t2 - t0), 5), "unit": "iters/s", "cores": cores, "kind": "port",
            "sample": f"1 step (oracle forward {t1 - t0:.2f}s + backward {t2 - t1:.2f}s; rasterizer only, no loss/Adam) "
                      f"of the same {args.points}-Gaussian {args.width}x{args.height} view, OpenMP over tiles",
            "render_mpix_per_s": round(args.width * args.height / 1e6 / (t1 - t0), 4)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the rasterizer has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force_dist = os.environ.get("W3D_FORCE_DIST", "0") == "1"      # 1-rank RCCL group: exercises the exchange path
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from w3d_amd import _lib
    from w3d_amd.train import Trainer, render_views
    from w3d_amd.loss import photometric_loss, photometric_loss_torch

    bg = torch.zeros(3, device=dev)
    sc, model, opt, cams = build_scene(args, dev)
    make_ground_truth(args, cams, dev, bg)
    loss_fn = photometric_loss_torch if args.torch_loss else photometric_loss
    trainer = Trainer(model, cams, opt, bg, densify=False, loss_fn=loss_fn, fused=False if args.autograd_path else None,
                      force_exchange=force_dist)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    it = 0
    for _ in range(args.warmup):
        it += 1
        trainer.step(it)
    _lib.lib.w3d_profile_enable.argtypes = [ctypes.c_char_p]
    _lib.lib.w3d_profile_collect.argtypes = [ctypes.c_char_p, ctypes.c_uint64]
    prof_sel = b"*" if args.all_stages else args.profile.encode()
    sync()
    _lib.lib.w3d_profile_enable(prof_sel)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        it += 1
        trainer.step(it)
    sync()
    t1 = time.perf_counter()
    _lib.lib.w3d_profile_enable(None)
    buf = ctypes.create_string_buffer(1 << 16)
    _lib.lib.w3d_profile_collect(buf, len(buf))
    stages = {}
    for line in buf.value.decode().splitlines():
        name, cnt, ms = line.split()
        stages[name] = (int(cnt), float(ms))
    elapsed = torch.tensor([t1 - t0], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed)
    final_loss = float(trainer.last["loss"])

    # per-stage times, OUTSIDE the timed region (an event pair around every stage costs ~4 % of the step): 20 more steps
    stage_ms = {}
    if not args.all_stages:
        sync()
        _lib.lib.w3d_profile_enable(b"*")
        for _ in range(20):
            it += 1
            trainer.step(it)
        sync()
        _lib.lib.w3d_profile_enable(None)
        buf2 = ctypes.create_string_buffer(1 << 16)
        _lib.lib.w3d_profile_collect(buf2, len(buf2))
        for line in buf2.value.decode().splitlines():
            name, cnt, ms = line.split()
            if int(cnt) > 0:
                stage_ms[name] = round(float(ms) / int(cnt), 4)
    else:
        stage_ms = {k: round(ms / c, 4) for k, (c, ms) in stages.items() if c > 0}

    # forward-only render throughput (reference render.py's use), same scene, views cycled
    n_r = max(4, min(args.steps, 36))
    render_views(model, cams[:2], bg)
    sync()
    r0 = time.perf_counter()
    render_views(model, [cams[i % len(cams)] for i in range(n_r)], bg)
    sync()
    r1 = time.perf_counter()
    r_el = torch.tensor([r1 - r0], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(r_el, op=dist.ReduceOp.MAX)
    mpix = world * n_r * args.width * args.height / 1e6 / float(r_el)

    # config C4: FlashSplat per-mask contribution render (run_3d_seg.py's inner call), binary mask, same scene
    from w3d_amd.gaussian_renderer import flashsplat_render
    from w3d_amd.train import PipelineParams
    yy, xx = torch.meshgrid(torch.arange(args.height, device=dev), torch.arange(args.width, device=dev), indexing="ij")
    mask = (((xx - args.width // 2) ** 2 + (yy - args.height // 2) ** 2) < (args.height // 3) ** 2).float()
    n_f = 8
    with torch.no_grad():
        flashsplat_render(cams[0], model, PipelineParams(), bg, gt_mask=mask, obj_num=1)
        sync()
        f0 = time.perf_counter()
        counts = None
        for i in range(n_f):
            uc = flashsplat_render(cams[i % len(cams)], model, PipelineParams(), bg, gt_mask=mask, obj_num=1)["used_count"]
            counts = uc if counts is None else counts + uc
        sync()
        f1 = time.perf_counter()
    flash_vps = world * n_f / (f1 - f0)
    del counts
    # ... and run_3d_seg.py's real loop shape: several object masks per view — one forward, the blend repeated per mask
    from w3d_amd.gaussian_renderer import flashsplat_render_masks
    n_m = 8
    masks = torch.stack([torch.roll(mask, shifts=40 * k, dims=1) for k in range(n_m)])
    with torch.no_grad():
        flashsplat_render_masks(cams[0], model, PipelineParams(), bg, masks[:2], obj_num=1)
        sync()
        f0 = time.perf_counter()
        for i in range(4):
            uc = flashsplat_render_masks(cams[i % len(cams)], model, PipelineParams(), bg, masks, obj_num=1)["used_count"]
        sync()
        f1 = time.perf_counter()
    flash_mps = world * 4 * n_m / (f1 - f0)
    # non-overlapping instance masks (8 vertical stripes): one blend over the merged label map
    stripes = torch.stack([((xx >= k * args.width // n_m) & (xx < (k + 1) * args.width // n_m)).float() for k in range(n_m)])
    with torch.no_grad():
        flashsplat_render_masks(cams[0], model, PipelineParams(), bg, stripes, obj_num=1)
        sync()
        f0 = time.perf_counter()
        for i in range(4):
            uc = flashsplat_render_masks(cams[i % len(cams)], model, PipelineParams(), bg, stripes, obj_num=1)["used_count"]
        sync()
        f1 = time.perf_counter()
    flash_mps_disjoint = world * 4 * n_m / (f1 - f0)
    del uc, masks, stripes

    if rank == 0:
        ws = [workload_stats(model, cams[i], bg, dev) for i in (0, len(cams) // 2)]
        V = sum(w["V"] for w in ws) / len(ws)
        R = sum(w["R"] for w in ws) / len(ws)
        Rw = sum(w["R_walk"] for w in ws) / len(ws)
        HW = args.width * args.height
        P = args.points
        # algorithmic bytes per launch of the dominant kernel:
        #   blend backward: 84 B per tile instance the reverse walk visits (44 B gather + 40 B record update)
        #                   + 20 B per pixel (dL/dpixel 12 + final_T 4 + n_contrib 4)
        algo = {"render_bwd": 84.0 * Rw + 20.0 * HW, "render_fwd": 48.0 * Rw + 36.0 * HW,
                "preprocess_fwd": 236.0 * P + 56.0 * V, "preprocess_bwd": 236.0 * P + 64.0 * V + 252.0 * P,
                "fill_lists": 4.0 * R + 8.0 * V, "depth_sort": 4 * 16.0 * P, "tile_count_scan": 8.0 * V}
        roof = None
        if args.profile in stages and stages[args.profile][0] > 0:
            cnt, ms = stages[args.profile]
            avg_ms = ms / cnt
            ach = algo.get(args.profile, 0.0) / (avg_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": args.profile, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": pmc_traffic(args.profile + "_kernel"),
                    "avg_launch_ms": round(avg_ms, 4), "launches": cnt,
                    "algorithmic_bytes_per_launch": int(algo.get(args.profile, 0.0))}
        if args.all_stages:
            for k, (c, ms) in sorted(stages.items(), key=lambda kv: -kv[1][1]):
                print(f"[stage] {k:18s} {c:5d} launches  avg {ms / c:8.4f} ms", file=sys.stderr)
        out = {
            "metric": "train iters/sec @ 2M Gaussians, 1600x1200 (render+loss+backward+Adam); render Mpix/sec alongside",
            "value": round(world * args.steps / elapsed, 3), "unit": "iters/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "render_mpix_per_s": round(mpix, 1), "flashsplat_views_per_s": round(flash_vps, 1),
            "flashsplat_masks_per_s_8_per_view": round(flash_mps, 1),
            "flashsplat_masks_per_s_8_disjoint_per_view": round(flash_mps_disjoint, 1),
            "config": {"workload": f"C3: plot-shaped synthetic scene, {P} Gaussians, SH degree 3, "
                                   f"{args.width}x{args.height}, {args.views} overhead cameras, depth+alpha channels",
                       "points": P, "image": [args.width, args.height], "views_per_step": world,
                       "parallelism": f"view-parallel dp{world}" if world > 1 else "single GPU",
                       "visible_per_view": int(V), "tile_instances_per_view": int(R), "walked_instances_per_view": int(Rw),
                       "loss": "torch conv2d" if args.torch_loss else "fused HIP L1+SSIM",
                       "step": "fused raw-parameter kernels (no autograd)" if trainer.fused else "drop-in render() + autograd",
                       "final_loss": round(final_loss, 6)},
            "roofline": roof,
            "stage_ms": stage_ms,
        }
        # the HBM-bound kernel of the step next to the (VALU-bound) dominant one: per-Gaussian backward + Adam + statistics
        fused_adam = trainer.fused and trainer.fused_adam and world == 1 and not force_dist
        if fused_adam and "preprocess_bwd" in stage_ms:
            b = 6 * 236.0 * P + 104.0 * V          # parameters + both moments read and written; 2-D gradient records read
            ach = b / (stage_ms["preprocess_bwd"] * 1e-3) / 1e9
            out["roofline_hbm_kernel"] = {"bound": "hbm", "kernel": "preprocess_bwd+adam", "achieved": round(ach, 1),
                                          "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                                          "traffic": pmc_traffic("preprocess_bwd_kernel"),
                                          "avg_launch_ms": stage_ms["preprocess_bwd"], "algorithmic_bytes_per_launch": int(b)}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
