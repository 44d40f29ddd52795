#!/usr/bin/env python
"""bench.py — the headline measurement: train-step iters/sec (and forward render Mpix/sec) at
2M Gaussians, 1600x1200 (BASELINE.json `metric`, config C3) on synthetic data.

A step is the loop body of reference train_vanilla_3dgs.py:55-115 (render, 0.8*L1+0.2*(1-SSIM),
backward, densification statistics, Adam step, zero_grad) at fixed P (no densify/prune inside the
timed region).  With N>1 GPUs every rank renders a different camera per step (view-parallel,
weak scaling) and the ranks exchange gradients over RCCL; `value` counts views/sec of the whole job.

  python bench.py --gpus N --steps K --warmup W
      N > 1 without a torch.distributed environment: this process starts N ranks itself
      (python -m torch.distributed.run, one per GPU) BEFORE touching the GPU and relays their output.
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
      the ranks read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.

Prints ONE JSON line on rank 0.  `value` is the fused raw-parameter step (the repo's own trainer);
`dropin_iters_per_s` is the UNMODIFIED reference loop (train_vanilla_3dgs.py:16-18,55-115 statement by
statement: render() -> l1_loss / ssim -> loss.backward() -> loss.item() -> max_radii2D / add_densification_stats
-> optimizer.step() -> zero_grad(set_to_none=True)) under the import redirect w3d_amd.dropin.install(), and
`modules_only_iters_per_s` the same script without it (only the rasterizer packages swapped); `trained_scene` repeats the headline measurement after 3000 more training steps (fixed P).
"""
import argparse
import ctypes
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "wheat-3dgs_amd"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable
# fp32 vector peak of the chip: 256 CUs x 4 SIMD-32 x 2.4 GHz x 2 flop (MI355X_MICROARCH.md 'Peak FP32 (vector)')
# = one wave64 VALU instruction per 2 cycles per SIMD.  The VALU roofline of the blend kernels prices every issued
# wave64 VALU instruction as 128 flop-equivalents against it (the measured sustainable rate is in
# profiles/r02/valu_microbench.json and replaces the spec figure when present).
VALU_SPEC_TFLOPS = 157.3
FLOP_PER_VALU_INSTR = 128.0


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--points", type=int, default=2_000_000)
    ap.add_argument("--width", type=int, default=1600)
    ap.add_argument("--height", type=int, default=1200)
    ap.add_argument("--views", type=int, default=36)
    ap.add_argument("--profile", default="auto",
                    help="stage timed with HIP events INSIDE the timed region for the `roofline` object; auto (default): the "
                         "longest stage of the scene being measured, found by a short untimed probe with every stage timed")
    ap.add_argument("--all-stages", action="store_true", help="also print per-stage event times to stderr")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--torch-loss", action="store_true", help="use the PyTorch conv2d SSIM instead of the fused kernel")
    ap.add_argument("--autograd-path", action="store_true",
                    help="headline loop = Trainer.step through render()+autograd instead of the fused raw-parameter step")
    ap.add_argument("--exchange", default="auto", choices=("auto", "rows", "lowrank", "dense"),
                    help="view-parallel exchange (N > 1); auto: a few untimed steps of 'rows' and of 'lowrank' after the warm-up, "
                         "the faster one (max over ranks) runs the timed steps")
    ap.add_argument("--dropin-steps", type=int, default=-1, help="steps of the reference-loop measurement (-1: = --steps, 0: skip)")
    ap.add_argument("--trained-steps", type=int, default=3000,
                    help="extra training steps before the second (trained-scene) measurement; 0: skip")
    ap.add_argument("--dropin-only", action="store_true",
                    help="run only the reference-loop measurement (for rocprofv3 --kernel-trace of that loop) and print its JSON")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip render / FlashSplat / drop-in / modules-only / trained-scene / densified-scene measurements")
    ap.add_argument("--densify-iterations", type=int, default=8000,
                    help="iterations of the compressed C3 schedule that grows the densified scene (densified_scene leg); 0: skip")
    ap.add_argument("--densify-grad-threshold", type=float, default=1.5e-5,
                    help="densify_grad_threshold of that schedule (reference default 2e-4, arguments/__init__.py:88, tuned for "
                         "photographs: the smooth synthetic views only grow to ~0.26 M Gaussians with it; stated in the line)")
    ap.add_argument("--opaque-iterations", type=int, default=15_000,
                    help="iterations of the reference schedule that trains the opaque-surface scene (opaque_scene leg; 15000 = "
                         "arguments/__init__.py:73-89 as it is); 0: skip")
    ap.add_argument("--densified-only", action="store_true",
                    help="of the extra measurements keep only the densified-scene one")
    ap.add_argument("--modules-only-steps", type=int, default=-1,
                    help="steps of the INTEGRATION.md section 1 measurement (rasterizer modules only; -1: = min(--steps, 60), 0: skip)")
    ap.add_argument("--trained-only", action="store_true",
                    help="of the extra measurements keep only the trained-scene one (kernel A/B runs: profiles/ab_variants.sh)")
    ap.add_argument("--torch-restatement", default=None, help=argparse.SUPPRESS)      # child mode of the cpu_baseline leg
    ap.add_argument("--no-spatial-order", action="store_true",
                    help="keep the Gaussians in the order the scene was created in (default: Trainer(spatial_order=True), the "
                         "model is stored in Morton order of the positions and re-sorted during densification)")
    ap.add_argument("--no-scale-model", action="store_true",
                    help="skip the N-GPU prediction from single-GPU measurements (per-camera step times, exchange machinery on a 1-rank group)")
    ap.add_argument("--force-dist", action="store_true", help="1-rank RCCL group: exercises the exchange path on one GPU")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous / reporting test on CPU over gloo with a stub step (no kernels, no GPU); "
                         "the line says so in `data`")
    return ap.parse_args(argv)


def parse_defaults():
    """the default arguments (profiles/scene_step.py builds bench.py's scenes with them)"""
    return parse([])


# ------------------------------------------------------------------------------------------------ launch
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args):
    """--gpus N > 1 outside a torch.distributed environment: start the N ranks as a fresh child BEFORE anything here
    has touched the GPU (device_count() does not initialise it) and exit with the child's status."""
    if not args.dry_run:
        have = torch.cuda.device_count()
        if have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} requested but only {have} GPU(s) are visible", file=sys.stderr)
            sys.exit(2)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    sys.exit(subprocess.call(cmd, env=env))


def dist_env(args):
    """(world, rank, local) from the torch.distributed environment; launches the ranks first if there is none."""
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            launch_ranks(args)
        return 1, 0, 0
    world = int(os.environ["WORLD_SIZE"])
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU", file=sys.stderr)
        sys.exit(2)
    return world, int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def dry_run(args, world, rank):
    """The launch / rendezvous / max-over-ranks / one-line protocol with a stub step on CPU tensors over gloo."""
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    bucket = torch.ones(1 << 16)

    def step():
        if world > 1:
            dist.all_reduce(bucket)
        bucket.mul_(1.0 / max(world, 1))
    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "train iters/sec (DRY RUN: stub step, no kernels)", "value": round(world * args.steps / float(el), 3),
                          "unit": "iters/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(1e3 * float(el) / args.steps, 4), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "dry-run (launcher test on CPU/gloo, nothing measured)",
                          "config": {"workload": "stub", "views_per_step": world}, "roofline": None, "cpu_baseline": None}))
    if world > 1:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ scene
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
        # (one list per tile, whatever list_share the trainer currently runs: R and the walk lengths are then the culled
        #  per-tile figures, comparable between scenes and rounds)
        s = _settings(GaussianRasterizationSettings, cam, model, bg, 1.0, False)._replace(list_share=0)
        _, radii, _, _, saved, _ = _forward_impl(s, model.get_xyz, model.get_features, None, model.get_opacity,
                                                 model.get_scaling, model.get_rotation, None)
        _, nc = debug_pixel_state(saved)
        H, W = nc.shape
        gy, gx = (H + 15) // 16, (W + 15) // 16
        pad = torch.zeros(gy * 16, gx * 16, dtype=torch.int64, device=dev)
        pad[:H, :W] = nc.to(torch.int64)
        r_walk = int(pad.view(gy, 16, gx, 16).amax(dim=(1, 3)).sum())
        # contributors = entries a pixel actually BLENDS (alpha >= 1/255, before it saturates): the FlashSplat forward counts
        # them (contrib_num); n_contrib above is the list POSITION of the last one — every entry of the tile's list in front of
        # it counts there, whether it touches the pixel or not
        from w3d_amd.rasterizer import FlashSplatRasterizationSettings
        sf = FlashSplatRasterizationSettings(*s[:12], mask_grad=False, num_obj=1, tile_cull=True, deterministic=False, list_share=0)
        ex = _forward_impl(sf, model.get_xyz, model.get_features, None, model.get_opacity, model.get_scaling, model.get_rotation,
                           None, flash=dict(gt_mask=None, num_obj=1))[5]
        return dict(V=saved["num_visible"], R=saved["num_rendered"], R_walk=r_walk,
                    mean_last=float(nc.float().mean()), mean_contrib=float(ex[0].float().mean()))


def _newest(pattern):
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", pattern)))
    return files[-1] if files else None


# kernels of every stage of the step (names as rocprofv3 prints them, template arguments included where they matter)
STAGE_KERNELS = {
    "preprocess_fwd": ("preprocess_fwd_kernel",),
    "depth_sort": ("radix_hist_kernel", "radix_rowscan_kernel", "radix_scatter_kernel", "onesweep_", "depth_"),
    "tile_count_scan": ("chunk_walk_kernel<0", "seg_sum_kernel", "tile_scan_kernel", "chunk_off_kernel", "tile_count_", "band_"),
    "fill_lists": ("chunk_walk_kernel<1", "fill_"),
    "render_fwd": ("render_fwd_kernel", "tile_order_kernel"),
    "loss": ("ssim_pass_a", "ssim_pass_b", "loss_finalize"),
    "render_bwd": ("render_bwd_kernel", "zero_visible_records_kernel", "det_gather_kernel"),
    "preprocess_bwd": ("preprocess_bwd_kernel",),
}


def _scene_csv(prefix, scene):
    """newest profiles/rNN/<prefix>_<scene>.csv (per-step sums inside the marker window of profiles/scene_step.py)"""
    return _newest(f"{prefix}_{scene}.csv")


def pmc_traffic(stage, scene="untrained"):
    """HBM-side bytes ONE STEP of `scene` moves in `stage`: the SUM over every kernel of the stage (STAGE_KERNELS) and over
    all its launches in a step, from the newest committed profiles/rNN/pmc_hbm_traffic_<scene>.csv (FETCH_SIZE x2 +
    WRITE_SIZE collected in separate rocprofv3 --pmc passes over profiles/scene_step.py, whose K steps sit between two
    marker kernels; profiles/summarize_pmc.py --window).  (bytes, [kernel rows]) or (None, None) when no summary exists —
    the counters cannot be read live from inside the process."""
    import csv
    f = _scene_csv("pmc_hbm_traffic", scene)
    if not f:
        return None, None
    tot, used = 0.0, []
    for r in csv.DictReader(open(f)):
        if any(r["kernel"].startswith(k) for k in STAGE_KERNELS.get(stage, (stage,))):
            mib = float(r["hbm_read_MiB_corrected_x2"]) + float(r["hbm_write_MiB"])
            tot += mib
            used.append({"kernel": r["kernel"], "launches_per_step": float(r["launches_per_step"]), "MiB_per_step": round(mib, 2)})
    return (int(tot * 1024 * 1024), used) if used else (None, None)


def valu_instructions(kernel, scene="untrained"):
    """wave64 VALU instructions `kernel` issues per step of `scene` (SQ_INSTS_VALU summed over its launches inside the marker
    window, newest committed profiles/rNN/sq_counters_<scene>.csv); (count, file) or (None, None)."""
    import csv
    f = _scene_csv("sq_counters", scene)
    if not f:
        return None, None
    for r in csv.DictReader(open(f)):
        if r["kernel"].startswith(kernel) and "<true>" not in r["kernel"]:
            return float(r["SQ_INSTS_VALU"]), os.path.relpath(f, ROOT)
    return None, None


def valu_peak():
    """Sustainable wave64 VALU issue rate of the chip measured by profiles/valu_microbench.hip (v_fma_f32, best over the
    waves-per-SIMD settings), as TFLOP/s-equivalents (x128); falls back to the spec fp32 vector peak."""
    f = _newest("valu_microbench.json")
    if f:
        try:
            res = json.load(open(f))["results"]
            rate = max(r["wave_instr_per_s"] for r in res if r["op"] == "v_fma_f32")
            return rate * FLOP_PER_VALU_INSTR / 1e12, os.path.relpath(f, ROOT)
        except Exception:
            pass
    return VALU_SPEC_TFLOPS, "spec (MI355X_MICROARCH.md, Peak FP32 vector)"


_REAL_STDOUT = None


def _stdout_to_stderr():
    """Everything any library prints to stdout from here on (RCCL's version banner, for one) goes to stderr; the ONE JSON
    line is written to the real stdout by _emit()."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def _emit(line):
    sys.stdout.flush()
    if _REAL_STDOUT is None:
        print(line, flush=True)
    else:
        os.write(_REAL_STDOUT, (line + "\n").encode())


def _progress(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def _median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def cpu_baseline(args, own_view0=None):
    """own_view0: (colour, depth, alpha, gt) numpy images of camera 0 of the benchmark scene rendered by the HIP path BEFORE
    any training step — compared with the oracle's render of the same view (the `psnr` entry of the result: BASELINE.json's
    "PSNR vs ref", reference utils/image_utils.py:17-19).
    BASELINE.md section 4: the oracle (kind "port": this repo's C restatement of the rasterizer, OpenMP over tiles, all host
    cores) timed on this box beside the GPU number, with time.perf_counter:
      * `value`: ONE view of the benchmark's own C3 workload — rasterizer forward + backward on a fixed dL/dcolor (the
        rasterizer's share of the bracket of train_vanilla_3dgs.py:56,82; the loss is NOT in it: `bracket` says so) — a
        bounded sample, 1 warm-up + 3 timed iterations, median;
      * `c1`: config C1 (10 k Gaussians, 400x300), 3 warm-up + 10 timed iterations, median, cameras cycled — the C oracle
        (rasterizer only) and, next to it, the PyTorch restatement (oracle.torch_render, float32, 16 threads) with the full
        bracket: render + 0.8*L1 + 0.2*(1-SSIM) + backward by autograd."""
    import numpy as np
    from util import view_inputs, make_oracle, np_inputs
    from w3d_amd.synth import make_scene, make_cameras
    from w3d_amd.loss import photometric_loss_torch
    cores = os.cpu_count() or 1
    _progress("cpu_baseline: C3 sample")

    first_view = {}

    def c_oracle_protocol(P, width, height, warm, timed, seed, nthreads=None):
        nthreads = cores if nthreads is None else nthreads
        sc = make_scene(P, seed=seed, **({} if P >= 100_000 else {"scale_mean": 0.012}))
        cams = make_cameras(args.views, width, height)
        gc = np.random.RandomState(0).randn(3, height, width).astype(np.float32)
        fwd, step = [], []
        for i in range(warm + timed):
            cam = cams[i % len(cams)]                       # cameras cycled
            d = np_inputs(view_inputs(sc, cam))
            o = make_oracle(cam, (0.0, 0.0, 0.0), nthreads=nthreads)
            t0 = time.perf_counter()
            ref = o.forward(**d)
            t1 = time.perf_counter()
            if i == 0 and (P, width, height) not in first_view:
                first_view[(P, width, height)] = {k: ref[k].copy() for k in ("color", "depth", "alpha", "radii")}
            gref = o.backward(gc, None, None)
            t2 = time.perf_counter()
            if i == 0 and "gref" not in first_view[(P, width, height)] and P == args.points:
                first_view[(P, width, height)].update(gref={k: (None if v is None else np.array(v)) for k, v in gref.items()}, d=d, cam=cam)
            o.free()
            if i >= warm:
                fwd.append(t1 - t0)
                step.append(t2 - t0)
        return _median(fwd), _median(step)

    f3, s3 = c_oracle_protocol(args.points, args.width, args.height, 1, 3, 0)
    gc3 = np.random.RandomState(0).randn(3, args.height, args.width).astype(np.float32)
    _progress("cpu_baseline: C1, C oracle")
    # C1 has 475 tiles (the oracle's OpenMP loop runs over tiles): with one thread per host core of a 256-core box it times the
    # fork / join and the atomics, not the rasterizer.  A short sweep picks the thread count; the protocol runs with it and says so.
    sweep = {}
    for nt in sorted({n for n in (8, 32, 128, cores) if n <= cores}):
        sweep[nt] = c_oracle_protocol(10_000, 400, 300, 1, 3, 4, nthreads=nt)[1]
    # (the short sweep is noisy on a box whose other cores are busy: the protocol runs with its two best counts, the better one is quoted)
    best = None
    for nt in sorted(sweep, key=sweep.get)[:2]:
        f, t = c_oracle_protocol(10_000, 400, 300, 3, 10, 4, nthreads=nt)
        if best is None or t < best[2]:
            best = (nt, f, t)
    c1_threads, f1, s1 = best
    out = {"value": round(1.0 / s3, 5), "unit": "iters/s", "cores": cores, "kind": "port",
           "bracket": "rasterizer forward + backward only (no loss, no Adam)",
           "sample": f"one {args.points}-Gaussian {args.width}x{args.height} view of the benchmark scene through the C oracle (OpenMP "
                     f"over tiles, {cores} threads), fixed dL/dcolor; 1 warm-up + 3 timed iterations, median {s3:.2f} s "
                     f"(forward {f3:.2f} s)",
           "render_mpix_per_s": round(args.width * args.height / 1e6 / f3, 4),
           "c1": {"workload": "C1: 10000 Gaussians, 400x300", "protocol": "3 warm-up + 10 timed, median, cameras cycled, seed 4",
                  "c_oracle_iters_per_s": round(1.0 / s1, 3), "c_oracle_render_mpix_per_s": round(0.12 / f1, 3),
                  "c_oracle_threads": c1_threads,
                  "c_oracle_thread_sweep_iters_per_s": {str(k): round(1.0 / v, 3) for k, v in sweep.items()},
                  "c_oracle_bracket": "rasterizer forward + backward only"}}
    if own_view0 is not None:
        # "PSNR vs ref": the HIP render of camera 0 of the benchmark scene against the oracle's render of the same inputs
        # (psnr of reference utils/image_utils.py:17-19: 20 log10(1 / sqrt(mse)), per image here)
        from util import psnr as _psnr
        ref0 = first_view[(args.points, args.width, args.height)]
        own_c, own_d, own_a, gt0 = own_view0[:4]
        pg_own, pg_ref = _psnr(own_c, gt0), _psnr(ref0["color"], gt0)
        dmax, amax = float(ref0["depth"].max()) or 1.0, 1.0
        out["psnr"] = {"view": "camera 0 of the benchmark scene, initial parameters", "own_vs_gt_db": round(pg_own, 6),
                       "oracle_vs_gt_db": round(pg_ref, 6), "delta_db": float(f"{pg_own - pg_ref:.3e}"),
                       "own_vs_oracle_db": {"color": round(_psnr(own_c, ref0["color"]), 2),
                                            "depth": round(_psnr(own_d / dmax, ref0["depth"] / dmax), 2),
                                            "alpha": round(_psnr(own_a / amax, ref0["alpha"] / amax), 2)},
                       "bar_db": 1e-3, "formula": "utils/image_utils.py:17-19"}
    if own_view0 is not None and len(own_view0) > 4:
        try:
            _progress("cpu_baseline: parity tail (oracle backward, second oracle run with the other fp32 roundings)")
            out["parity_tail"] = parity_tail(args, first_view[(args.points, args.width, args.height)], own_view0[4], own_view0[5], gc3, cores)
        except Exception as e:      # never take the line down
            out["parity_tail"] = {"error": repr(e)}
    # the PyTorch restatement at C1 (BASELINE.md section 4 names it): float32 torch_render, loss, backward by autograd
    # the PyTorch restatement at C1 runs in CHILD processes with a time limit each: with one thread per core of a 256-core box a
    # single view of its per-tile Python loop did not finish in 20 minutes (every one of its thousands of small ops forks and
    # joins all threads)
    def torch_protocol(nthreads, warm, timed, limit):
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--torch-restatement", f"{nthreads},{warm},{timed},{args.views}"],
                               capture_output=True, text=True, timeout=limit)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            return json.loads(line[-1]) if r.returncode == 0 and line else {"error": (r.stderr or "no output")[-300:]}
        except subprocess.TimeoutExpired:
            return {"timeout_s": limit}
    try:
        _progress("cpu_baseline: C1, PyTorch restatement")
        c1 = out["c1"]
        c1["torch_restatement_bracket"] = "render + 0.8*L1+0.2*(1-SSIM) + backward (train_vanilla_3dgs.py:56,82)"
        # n = all host cores (stated), 3 warm-up + 10 timed, median, cameras cycled — if ONE view with that
        # many threads fits ~9 s; the 16-thread figure (thousands of tiny ops per view: more threads mostly add fork / join
        # time) beside it under the same rule
        runs = {}
        for nt in ([cores] if cores <= 16 else [cores, 16]):
            probe = torch_protocol(nt, 0, 1, 14)
            if "step_s" not in probe:
                runs[nt] = {"threads": nt, "protocol": "not completed: one view (plus ~5 s of imports) did not finish in 14 s", **probe}
                continue
            full = 13 * probe["step_s"] <= 45.0
            r = torch_protocol(nt, *((3, 10) if full else (1, 3)), 120)
            if "step_s" not in r:
                r = probe
                full = None
            runs[nt] = {"threads": nt, "iters_per_s": round(1.0 / r["step_s"], 3), "render_mpix_per_s": round(0.12 / r["fwd_s"], 3),
                        "protocol": ("one view" if full is None else "3 warm-up + 10 timed" if full else "1 warm-up + 3 timed") +
                                    ", median, cameras cycled"}
        c1["torch_restatement_all_cores"] = runs[cores]
        if 16 in runs and cores > 16:
            c1["torch_restatement_16_threads"] = runs[16]
        best = max((r for r in runs.values() if "iters_per_s" in r), key=lambda r: r["iters_per_s"], default=None)
        if best is not None:
            c1.update(torch_restatement_iters_per_s=best["iters_per_s"], torch_restatement_render_mpix_per_s=best["render_mpix_per_s"],
                      torch_restatement_threads=best["threads"], torch_restatement_protocol=best["protocol"])
    except Exception as e:      # the baseline leg must never take the bench line down
        out["c1"]["torch_restatement_error"] = repr(e)
    return out


def torch_restatement_child(spec):
    """bench.py --torch-restatement threads,warm,timed,views: config C1 through oracle.torch_render + loss + autograd backward on the
    CPU; prints {"fwd_s", "step_s"} (medians).  Never touches the GPU."""
    from util import view_inputs
    from oracle.oracle import torch_render
    from w3d_amd.loss import photometric_loss_torch
    from w3d_amd.synth import make_scene, make_cameras
    nthreads, warm, timed, views = (int(x) for x in spec.split(","))
    torch.set_num_threads(nthreads)
    sc = make_scene(10_000, seed=4, scale_mean=0.012)
    cams = make_cameras(views, 400, 300)
    gt = torch.rand(3, 300, 400, generator=torch.Generator().manual_seed(3))
    fwd, step = [], []
    for i in range(warm + timed):
        cam = cams[i % len(cams)]
        d = {k: (None if v is None else v.clone().requires_grad_(True)) for k, v in view_inputs(sc, cam).items()}
        t0 = time.perf_counter()
        c = torch_render(300, 400, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.zeros(3), cam.world_view_transform,
                         cam.full_proj_transform, cam.camera_center, sh_degree=3, **d)[0]
        t1 = time.perf_counter()
        photometric_loss_torch(c, gt, 0.2).backward()
        t2 = time.perf_counter()
        if i >= warm:
            fwd.append(t1 - t0)
            step.append(t2 - t0)
    print(json.dumps({"fwd_s": _median(fwd), "step_s": _median(step)}))



def parity_tail(args, first, own, sc, gc, cores):
    """north_star: "densification-grad norms within 1e-4 of the reference".  The HIP gradients of camera 0 (initial parameters,
    dL/dcolor ~ N(0,1) seed 0) against the oracle's — per Gaussian, relative to that Gaussian's own gradient, over the
    Gaussians that have one: p50 / p99 / p99.9 / max and the number beyond 1e-4, for the densification norm
    ||means2D.grad[:, :2]|| and every parameter block (the oracle's gradients chained through exp / sigmoid / normalize in
    float64) — and beside each the SAME statistics between two runs of the oracle itself: the second with the other legal fp32
    roundings (exp2f, fp32 accumulation, FMA-contracted exponent, the other form of the suffix recurrence —
    w3do_set_exp_mode(15)) on activations moved by one ulp, i.e. what any other faithful fp32 build of the reference's
    rasterizer may differ from it by.  The fields of profiles/r04/fullsize_parity.jsonl (tests/test_gpu_fullsize.py)."""
    import numpy as np
    from oracle.oracle import COracle
    from util import densify_norm_error, flip_pixels, gradient_stats, make_oracle, raw_grads_from_oracle
    t0 = time.perf_counter()
    ref_radii, gref, d, cam = first["radii"], first["gref"], first["d"], first["cam"]
    vis = ref_radii > 0
    want = raw_grads_from_oracle(gref, sc)
    n_ref = np.linalg.norm(gref["means2D"][:, :2].astype(np.float64), axis=1)
    rng = np.random.RandomState(11)
    d_probe = dict(d)
    for k in ("scales", "rotations", "opacities"):
        a = d[k]
        d_probe[k] = np.nextafter(a, np.where(rng.rand(*a.shape) < 0.5, -np.inf, np.inf).astype(np.float32)).astype(np.float32)
    COracle.set_exp_mode(15)
    try:
        o = make_oracle(cam, (0.0, 0.0, 0.0), nthreads=cores)
        ref1 = o.forward(**d_probe)
        gref1 = o.backward(gc, None, None)
        o.free()
    finally:
        COracle.set_exp_mode(0)
    want1 = raw_grads_from_oracle(gref1, sc)
    n1 = np.linalg.norm(gref1["means2D"][:, :2].astype(np.float64), axis=1)
    own_vis = own["radii"] > 0
    keep = lambda st: {k: (float(f"{v:.3e}") if isinstance(v, float) else v) for k, v in st.items() if k != "worst_mixed"}  # noqa: E731
    hip = {"densify_norm": keep(densify_norm_error(own["densify_norm"], n_ref, vis))}
    spread = {"densify_norm": keep(densify_norm_error(n1, n_ref, vis))}
    hip.update({k: keep(v) for k, v in gradient_stats({k: own[k] for k in want}, want, vis).items()})
    spread.update({k: keep(v) for k, v in gradient_stats(want1, want, vis).items()})
    return {"view": "camera 0 of the benchmark scene, initial parameters, dL/dcolor ~ N(0,1) seed 0",
            "statistic": "per Gaussian: max_d |g - g_ref| / max_d |g_ref| over the Gaussians whose reference gradient is not zero; "
                         "outliers = Gaussians beyond 1e-4",
            "bar": 1e-4, "gaussians_with_a_gradient": hip["densify_norm"]["n"],
            "radii_differing": int((own["radii"] != ref_radii).sum()), "visibility_differs": bool((own_vis != vis).any()),
            "hip_vs_oracle": hip, "oracle_vs_oracle_other_fp32_roundings": spread,
            "flip_pixels_oracle_vs_oracle": flip_pixels(ref1, {k: first[k] for k in ("color", "alpha")}),
            "seconds": round(time.perf_counter() - t0, 1)}


# ------------------------------------------------------------------------------------------------ drop-in loop
def time_standin(args, sc, cams, bg, dev, perm, hook, n, warm):
    """tests/standin_checkout/train_loop.py — the import lines and the loop body of reference train_vanilla_3dgs.py:16-18,55-115,
    statement by statement, starting from a checkpoint 13-tuple as --start_checkpoint does (:38-40) — timed as a whole
    (loss.item() and the boolean-mask statistics lines, i.e. the reference loop's host syncs, included).  The GPU box has no
    reference checkout, so the script imports its GaussianModel / render / l1_loss / ssim from the stand-in modules of the same
    names (tests/standin_checkout/README.md: six nn.Parameters with torch activations, torch.optim.Adam over six groups,
    render() marshalling into `diff_gaussian_rasterization`, conv2d SSIM).  hook=False: as it is — only the rasterizer packages
    are this repo's (INTEGRATION.md section 1 without the redirect).  hook=True: under w3d_amd.dropin.install() — the same
    unmodified script and modules, the four names redirected to this repo's fast path.  A tuple: only those modules."""
    from util import standin_checkout, checkpoint_tuple
    from w3d_amd.gaussian_model import OptimizationParams
    from w3d_amd.train import PipelineParams
    opt, pipe = OptimizationParams(), PipelineParams()
    with standin_checkout(hook) as loop:
        owners = {k: getattr(loop, k).__module__ for k in ("GaussianModel", "render", "l1_loss", "ssim")}
        g, _ = loop.training(checkpoint_tuple(sc, device=dev), opt, pipe, cams, bg, perm, 1, warm)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop.training(None, opt, pipe, cams, bg, perm, 1 + warm, n, gaussians=g)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # the reference's own timer (TensorBoard `iter_time`, train_vanilla_3dgs.py:56,82,149): CUDA/HIP events around
        # render + loss + backward — a separate short run, so the event pairs do not sit in the timed loop above
        ev = []
        loop.training(None, opt, pipe, cams, bg, perm, 1 + warm + n, min(n, 36), gaussians=g, iter_events=ev)
        torch.cuda.synchronize()
        iter_ms = sorted(a.elapsed_time(b) for a, b in ev)
        del g
    torch.cuda.empty_cache()
    return {"iters_per_s": round(n / dt, 2), "ms_per_step": round(1e3 * dt / n, 4), "steps": n,
            "iter_time_ms_median": round(iter_ms[len(iter_ms) // 2], 4), "resolved": owners}


def time_dropin(args, sc, cams, bg, dev, perm, n=None):
    """The unmodified loop script under the import redirect (w3d_amd.dropin.install())."""
    if n is None:
        n = args.steps if args.dropin_steps < 0 else args.dropin_steps
    if n <= 0:
        return None
    # (W3D_SPATIAL_ORDER: the documented switch of the redirect's GaussianModel, INTEGRATION.md section 1 — the model the script
    #  restores from its checkpoint is put into Morton order, as Trainer(spatial_order=True) does for the fused step)
    prev = os.environ.get("W3D_SPATIAL_ORDER")
    if not args.no_spatial_order:
        os.environ["W3D_SPATIAL_ORDER"] = "2"
    try:
        out = time_standin(args, sc, cams, bg, dev, perm, True, n, max(3, min(args.warmup, 10)))
    finally:
        if prev is None:
            os.environ.pop("W3D_SPATIAL_ORDER", None)
        else:
            os.environ["W3D_SPATIAL_ORDER"] = prev
    out["spatial_order"] = not args.no_spatial_order
    assert all(v.startswith("w3d_amd.") for v in out["resolved"].values()), out["resolved"]
    out["iter_time"] = "HIP events around render + loss + backward, the bracket of train_vanilla_3dgs.py:56,82 (no optimizer step)"
    return out



# ------------------------------------------------------------------------------------------------ roofline helpers
SPATIAL_ORDER = True        # (main() clears it under --no-spatial-order)


def kernel_bytes(P, V, R, Rw, HW, fused_adam):
    """Algorithmic HBM bytes per launch of every stage (what the stage must read and write once)."""
    return {
        # parameters read; packed per-visible records written.  In Morton order the culled Gaussians come in runs and their
        # 180-B SH rows are not requested at all: 56 B of geometry for everyone, the SH rows of the visible
        "preprocess_fwd": (56.0 * P + 180.0 * V + 64.0 * V) if SPATIAL_ORDER else (236.0 * P + 64.0 * V),
        # first pass reads P (key, id) pairs, the others V; the last writes 24-B records from a 16-B rect/mask gather
        "depth_sort": 8.0 * P + 8.0 * V + 2 * 16.0 * V + (8.0 + 16.0 + 24.0) * V,
        "tile_count_scan": 24.0 * V,                           # the records, once
        "fill_lists": 24.0 * V + 4.0 * R,                      # the records once + the lists
        "render_fwd": 48.0 * Rw + 36.0 * HW,                   # 4-B id + 44-B gather per walked instance; image + aux
        "loss": 2 * 12.0 * HW + 12.0 * HW,                     # image + gt read, gradient written
        "render_bwd": 84.0 * Rw + 20.0 * HW,                   # gather + one 40-B record update; dL/dpixel + aux
        # fused Adam: parameters + both moments read and written, 2-D records read / otherwise gradients written
        "preprocess_bwd": (6 * 236.0 * P + 104.0 * V) if fused_adam else (236.0 * P + 64.0 * V + 252.0 * P),
    }


def kernel_table(stage_ms, kb, scene):
    """Per stage: event-timed ms, algorithmic bytes, achieved GB/s and fraction of the HBM peak, and the PMC-counter traffic of
    the stage on THIS scene — summed over all its kernels and launches (pmc_traffic) — with its ratio to the algorithmic bytes
    (wasted re-reads show up there)."""
    rows = []
    for k, ms in sorted(stage_ms.items(), key=lambda kv: -kv[1]):
        if k not in kb:
            continue
        gbs = kb[k] / (ms * 1e-3) / 1e9
        tr, used = pmc_traffic(k, scene)
        rows.append({"stage": k, "ms": ms, "algorithmic_bytes": int(kb[k]), "achieved_GBps": round(gbs, 1),
                     "hbm_frac": round(gbs / HBM_PEAK_GBS, 4), "pmc_traffic_bytes": tr,
                     "traffic_ratio": None if not tr else round(tr / kb[k], 2),
                     "pmc_GBps": None if not tr else round(tr / (ms * 1e-3) / 1e9, 1), "pmc_kernels": used})
    return rows


def step_roofline(P, V, R, HW, it_per_s, stage_ms, kb):
    """SURVEY.md section 8(d) / BASELINE.md section 5 as written: (B_f + B_b + B_adam) * iters/s / peak with the MEASURED
    V and R, and the same with this design's own byte count (sum of the stages' algorithmic bytes — fusion removed the
    gradient round trip and the 64-bit key sort, so it is smaller)."""
    B_f = 236.0 * P + 56.0 * V + 80.0 * R + 28.0 * HW
    B_b = 484.0 * P + 80.0 * V + 84.0 * R + 20.0 * HW
    B_adam = 1652.0 * P
    tot = B_f + B_b + B_adam
    own = sum(kb[k] for k in kb if k in stage_ms)
    return {"formula": "B_f + B_b + B_adam, B_f = 236P + 56V + 80R + 28HW, B_b = 484P + 80V + 84R + 20HW, B_adam = 1652P",
            "P": P, "V": int(V), "R": int(R), "HW": HW, "B_f": int(B_f), "B_b": int(B_b), "B_adam": int(B_adam),
            "algorithmic_bytes": int(tot), "achieved_GBps": round(tot * it_per_s / 1e9, 1),
            "frac": round(tot * it_per_s / 1e9 / HBM_PEAK_GBS, 4),
            "design_bytes": int(own), "design_achieved_GBps": round(own * it_per_s / 1e9, 1),
            "design_frac": round(own * it_per_s / 1e9 / HBM_PEAK_GBS, 4)}


VALU_BOUND_STAGES = {"render_bwd": "render_bwd_kernel", "render_fwd": "render_fwd_kernel"}


def valu_object(stage, ms, scene):
    """`stage` (a blend kernel) against the VALU issue roof: wave64 VALU instructions per launch from
    the committed SQ counter summary of THIS scene x 128 flop-equivalents / the measured launch time."""
    n_valu, src = valu_instructions(VALU_BOUND_STAGES[stage], scene)
    if not n_valu:
        return None
    peak, peak_src = valu_peak()
    ach = n_valu * FLOP_PER_VALU_INSTR / (ms * 1e-3) / 1e12
    return {"bound": "valu", "kernel": stage, "achieved": round(ach, 2), "peak": VALU_SPEC_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / VALU_SPEC_TFLOPS, 4), "peak_measured": round(peak, 1), "frac_of_measured": round(ach / peak, 4),
            "avg_launch_ms": round(ms, 4), "valu_wave_instr_per_launch": int(n_valu), "valu_instr_source": src,
            "peak_source": peak_src, "flop_equiv_per_wave_instr": FLOP_PER_VALU_INSTR}


def roofline_object(meas, P, ws, HW, fused_adam, it_per_s, scene):
    """The `roofline` object of one scene: its DOMINANT stage (the longest one, found by the probe; timed with HIP events on
    its launch stream inside the timed region) against the roof that bounds it — HBM for the per-Gaussian and binning
    stages, VALU issue for the blend kernels (their HBM view is kept beside it) — plus the whole-step formula of SURVEY
    section 8(d) and the per-stage table."""
    V, R, Rw = ws["V"], ws["R"], ws["R_walk"]
    kb = kernel_bytes(P, V, R, Rw, HW, fused_adam)
    dom = meas["dominant"]
    if dom not in kb or not meas.get("live") or meas["live"][0] <= 0:
        return None
    cnt, ms = meas["live"]
    avg_ms = ms / cnt
    hbm_ach = kb[dom] / (avg_ms * 1e-3) / 1e9
    tr, used = pmc_traffic(dom, scene)
    label = "preprocess_bwd+adam" if (dom == "preprocess_bwd" and fused_adam) else dom
    roof = {"bound": "hbm", "kernel": label, "achieved": round(hbm_ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(hbm_ach / HBM_PEAK_GBS, 4), "traffic": tr, "traffic_ratio": None if not tr else round(tr / kb[dom], 2),
            "avg_launch_ms": round(avg_ms, 4), "launches": cnt, "algorithmic_bytes_per_launch": int(kb[dom]),
            "scene": scene, "chosen": meas["chosen_by"], "probe_stage_ms": meas["probe_ms"]}
    if dom in VALU_BOUND_STAGES:
        vo = valu_object(dom, avg_ms, scene)
        if vo is not None:
            hbm_view = {k: roof[k] for k in ("achieved", "peak", "unit", "frac", "traffic", "traffic_ratio",
                                             "algorithmic_bytes_per_launch")}
            roof.update(vo)
            roof["kernel"] = label
            roof["hbm_view"] = hbm_view
        else:
            roof["note"] = ("a blend kernel: VALU-issue-bound (DESIGN.md section 2.1); no SQ counter summary of this scene is "
                            "committed, so only its HBM view is given")
    roof["step"] = step_roofline(P, V, R, HW, it_per_s, meas["stage_ms"], kb)
    roof["kernels"] = kernel_table(meas["stage_ms"], kb, scene)
    return roof


class StepMeter:
    """Trainer steps between barrier + synchronize brackets (max over ranks), with the library's per-stage event timing."""

    def __init__(self, trainer, world, dev):
        from w3d_amd import _lib
        self.trainer, self.world, self.dev, self.lib = trainer, world, dev, _lib.lib
        self.lib.w3d_profile_enable.argtypes = [ctypes.c_char_p]
        self.lib.w3d_profile_collect.argtypes = [ctypes.c_char_p, ctypes.c_uint64]

    def sync(self):
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(self, n_steps, it, prof_sel):
        """n_steps trainer steps; returns (seconds, it, {stage: (launches, total ms)})."""
        self.sync()
        self.lib.w3d_profile_enable(prof_sel)
        t0 = time.perf_counter()
        for _ in range(n_steps):
            it += 1
            self.trainer.step(it)
        self.sync()
        t1 = time.perf_counter()
        self.lib.w3d_profile_enable(None)
        buf = ctypes.create_string_buffer(1 << 16)
        self.lib.w3d_profile_collect(buf, len(buf))
        stages = {}
        for line in buf.value.decode().splitlines():
            name, cnt, ms = line.split()
            stages[name] = (int(cnt), float(ms))
        el = torch.tensor([t1 - t0], device=self.dev, dtype=torch.float64)
        if self.world > 1:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return float(el), it, stages

    def measure(self, n_steps, it, profile="auto", all_stages=False, probe_steps=10, stage_steps=20):
        """The measurement protocol of one scene: (1) an untimed probe with every stage timed finds the dominant stage;
        (2) n_steps timed steps with ONLY that stage's events inside the timed region (an event pair around every stage costs
        ~4 % of the step); (3) stage_steps more steps with every stage timed, for the per-stage table."""
        _, it, pr = self.timed(probe_steps, it, b"*")
        probe_ms = {k: round(ms / c, 4) for k, (c, ms) in pr.items() if c > 0}
        known = kernel_bytes(1, 1, 1, 1, 1, True)
        if profile == "auto":
            cand = {k: v for k, v in probe_ms.items() if k in known}
            dominant = max(cand, key=cand.get) if cand else "preprocess_bwd"
            chosen_by = f"longest stage of a {probe_steps}-step probe with every stage timed"
        else:
            dominant, chosen_by = profile, "--profile"
        el, it, st = self.timed(n_steps, it, b"*" if all_stages else dominant.encode())
        if all_stages:
            stage_ms = {k: round(ms / c, 4) for k, (c, ms) in st.items() if c > 0}
        else:
            _, it, st2 = self.timed(stage_steps, it, b"*")
            stage_ms = {k: round(ms / c, 4) for k, (c, ms) in st2.items() if c > 0}
        return {"elapsed": el, "it": it, "dominant": dominant, "chosen_by": chosen_by, "probe_ms": probe_ms,
                "live": st.get(dominant), "stages": st, "stage_ms": stage_ms}


# ------------------------------------------------------------------------------------------------ scaling model (N = 1 runs)
XGMI_LINKS, XGMI_LINK_GBS = 7, 153.0        # MI355X_MICROARCH.md: 7 point-to-point links per GPU, ~153 GB/s each


def per_view_ms(trainer, it, rounds=2):
    """GPU time of the fused step per CAMERA (HIP events around every step; `rounds` steps per camera, the FASTER one: a
    one-off stall — a list buffer that a view outgrows is re-allocated and the view repeated, 50-100 ms once — is not what
    the view costs in a run)."""
    n = len(trainer.cameras)
    ev = []
    for _ in range(rounds * n):
        it += 1
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        trainer.step(it)
        b.record()
        ev.append((trainer.perm[((it - 1) * trainer.world + trainer.rank) % n], a, b))
    torch.cuda.synchronize()
    acc = {}
    for cam, a, b in ev:
        acc.setdefault(cam, []).append(a.elapsed_time(b))
    return {c: min(v) for c, v in acc.items()}, it


def scale_model(model, opt, cams, bg, dev, it, steps=30):
    """What an N-GPU view-parallel run of THIS scene should do, from single-GPU measurements — written down before the first
    multi-GPU run so that the run can falsify it (the builder has never had more than one GPU).
      T_N = straggler(N) * mean view time + machinery + wire,   speed-up = N * T_1 / T_N
    * view times: the fused step per camera (36 cameras, HIP events); straggler(N) = mean over the schedule's groups of N
      cameras (Trainer.camera_for) of the slowest view / mean view;
    * machinery: what the exchange kernels cost with no wire at all — the same trainer on a 1-rank RCCL group, rows form and
      low-rank form (pack / index / rows_adam, or the separate optimizer passes) against the single-GPU fused step;
    * wire: bytes a rank RECEIVES per step in each form (rows: (N-1) * 64 B * rows per view, low-rank: (N-1) * (12 + 88/N) * P)
      over the stated aggregate inbound rate — nothing overlaps it in the model (the sparse form's
      collectives sit between the per-Gaussian backward and the replicated optimizer)."""
    from w3d_amd.train import Trainer
    P = model.num_points
    out = {"gaussians": P}
    single = Trainer(model, cams, opt, bg, densify=False, spatial_order=SPATIAL_ORDER)
    for _ in range(8):
        it += 1
        single.step(it)
    views, it = per_view_ms(single, it)
    ms = [views[c] for c in sorted(views)]
    mean = sum(ms) / len(ms)
    out["view_ms"] = {"mean": round(mean, 4), "min": round(min(ms), 4), "max": round(max(ms), 4),
                      "p90": round(sorted(ms)[int(0.9 * (len(ms) - 1))], 4), "cameras": len(ms)}
    strag = {}
    for N in (2, 4, 8):
        groups = [[views[single.perm[(g * N + r) % len(cams)]] for r in range(N)] for g in range(len(cams))]
        strag[N] = sum(max(g) for g in groups) / len(groups) / mean
    out["straggler_factor"] = {str(N): round(v, 4) for N, v in strag.items()}
    # machinery: 1-rank RCCL group (no wire)
    mach, rows_per_view = {}, None
    try:
        if not dist.is_initialized():
            # an in-process store: no TCP rendezvous (on one box the c10d TCP store spent 3 minutes in reverse-lookups of a
            # hostname that does not resolve), and RCCL's own bootstrap kept on the loopback interface
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
            dist.init_process_group("nccl", store=dist.HashStore(), rank=0, world_size=1, device_id=dev)
        for mode in ("rows", "lowrank"):
            tr = Trainer(model, cams, opt, bg, densify=False, force_exchange=True, exchange=mode, spatial_order=SPATIAL_ORDER)
            for _ in range(10):
                it += 1
                tr.step(it)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                it += 1
                tr.step(it)
            torch.cuda.synchronize()
            mach[mode] = 1e3 * (time.perf_counter() - t0) / steps
            if mode == "rows":
                rows_per_view = max(tr._rows_recent) if tr._rows_recent else None
                out["rows_form_steps"] = dict(tr.exchange_used)
            del tr
        for _ in range(4):
            it += 1
            single.step(it)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            it += 1
            single.step(it)
        torch.cuda.synchronize()
        base = 1e3 * (time.perf_counter() - t0) / steps
        out["machinery_ms"] = {"single_gpu_step": round(base, 4), "rows": round(mach["rows"], 4), "lowrank": round(mach["lowrank"], 4),
                               "how": "same trainer on a 1-rank RCCL group (no wire time), host-timed over %d steps" % steps}
    except Exception as e:
        out["machinery_error"] = repr(e)
        base, mach = mean, {}
    out["rows_per_view_max"] = rows_per_view
    pred = {}
    for rate in (350.0, 700.0):
        for N in (2, 4, 8):
            forms = {}
            if rows_per_view is not None and "rows" in mach and rows_per_view <= (12 + 88.0 / N) / 64.0 * P:
                forms["rows"] = ((N - 1) * 64.0 * rows_per_view, mach["rows"] - base)
            if "lowrank" in mach:
                forms["lowrank"] = ((N - 1) * (12.0 + 88.0 / N) * P, mach["lowrank"] - base)
            best = None
            for form, (nbytes, extra) in forms.items():
                t = strag[N] * mean + max(extra, 0.0) + 1e3 * nbytes / (rate * 1e9)
                if best is None or t < best[1]:
                    best = (form, t, nbytes)
            if best is not None:
                pred[f"{int(rate)}GBps_N{N}"] = {"form": best[0], "ms_per_step": round(best[1], 4), "bytes_in_per_rank": int(best[2]),
                                                 "speedup": round(N * mean / best[1], 3), "efficiency": round(mean / best[1], 4)}
    out["prediction"] = pred
    out["assumptions"] = (f"aggregate inbound xGMI rate per GPU as stated in each key (peak {XGMI_LINKS} x {XGMI_LINK_GBS:.0f} = "
                          f"{XGMI_LINKS * XGMI_LINK_GBS:.0f} GB/s); wire time not overlapped; N views per step drawn by Trainer.camera_for; "
                          "weak scaling (one view per rank and step)")
    return out, it


def mean_workload(model, cams, bg, dev):
    ws = [workload_stats(model, cams[i], bg, dev) for i in (0, len(cams) // 2)]
    return {k: sum(w[k] for w in ws) / len(ws) for k in ("V", "R", "R_walk", "mean_contrib", "mean_last")}


# ------------------------------------------------------------------------------------------------ densified scene
def _psnr_db(a, b):
    """reference utils/image_utils.py:17-19 on one image"""
    mse = float(((a - b) ** 2).mean())
    return 99.0 if mse == 0 else 20.0 * math.log10(1.0 / math.sqrt(mse))


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


# ------------------------------------------------------------------------------------------------ modules-only loop
def time_modules_only(args, sc, cams, bg, dev, perm):
    """The same loop script WITHOUT the redirect — INTEGRATION.md section 1's first step alone: only the three rasterizer
    packages on the path are this repo's; the model (six nn.Parameters, torch activations, six-group torch.optim.Adam), render()'s
    marshalling and the conv2d SSIM are the checkout's own Python — then with the redirect on for the loss module only, for the
    model + render modules only, and (time_dropin) for all of them."""
    n = min(args.steps, 60) if args.modules_only_steps < 0 else args.modules_only_steps
    if n <= 0:
        return None
    w = max(3, min(args.warmup, 6))
    base = time_standin(args, sc, cams, bg, dev, perm, False, n, w)
    assert not any(v.startswith("w3d_amd.") for v in base["resolved"].values()), base["resolved"]
    out = dict(base, what="tests/standin_checkout/train_loop.py as it is, no import redirect: only diff_gaussian_rasterization is this "
                          "repo's; six nn.Parameters with torch exp / sigmoid / normalize / cat, torch.optim.Adam (6 groups), torch "
                          "conv2d SSIM, the reference loop's host syncs")
    # which of the other swaps buys what (same script, the redirect switched on for one part at a time)
    try:
        out["redirect_loss_module_only_iters_per_s"] = time_standin(args, sc, cams, bg, dev, perm, ("utils.loss_utils",), n, w)["iters_per_s"]
        out["redirect_model_and_render_only_iters_per_s"] = time_standin(args, sc, cams, bg, dev, perm,
                                                                         ("scene.gaussian_model", "gaussian_renderer"), n, w)["iters_per_s"]
    except Exception as e:
        out["breakdown_error"] = repr(e)
    return out


# ------------------------------------------------------------------------------------------------ exchange
def exchange_bandwidth(model, world, dev, rows=0):
    """The step's gradient collectives alone (N > 1): achieved bus bandwidth per GPU.
    Low-rank form — all-gather of the (P,3) colour gradients: every rank receives (world-1)*12P bytes; all-reduce of the
    11-float geometry span: ring model 2*(world-1)/world * 44P bytes per rank.  Sparse form (rows > 0: the largest per-view
    row count of the last step) — all-gather of `rows` 64-byte rows per rank: (world-1)*64*rows bytes received."""
    P = model.num_points
    d = torch.randn(P, 3, device=dev)
    d_all = torch.empty(world, P, 3, device=dev)
    geo = torch.randn(11 * P, device=dev)
    out = {}
    cases = [("all_gather_dcolor", lambda: dist.all_gather_into_tensor(d_all.view(-1), d.view(-1)), (world - 1) * 12.0 * P),
             ("all_reduce_geometry", lambda: dist.all_reduce(geo), 2.0 * (world - 1) / world * 44.0 * P)]
    if rows > 0:
        r_own = torch.randn(rows, 16, device=dev)
        r_all = torch.empty(world, rows, 16, device=dev)
        cases.append(("all_gather_rows", lambda: dist.all_gather_into_tensor(r_all.view(-1), r_own.view(-1)), (world - 1) * 64.0 * rows))
    for name, fn, nbytes in cases:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        dt = torch.tensor([(time.perf_counter() - t0) / 10], device=dev, dtype=torch.float64)
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        out[name] = {"ms": round(1e3 * float(dt), 4), "bus_GBps_per_gpu": round(nbytes / float(dt) / 1e9, 1)}
    return out


def replicas_identical(model, world, dev):
    """Every rank must hold bit-identical parameters (nothing re-synchronises them): compare a checksum of the bits."""
    bits = model.flat.detach().view(torch.int32).to(torch.int64)
    s = torch.stack([bits.sum(), (bits * (torch.arange(bits.numel(), device=dev) % 8191 + 1)).sum()])
    allsums = [torch.zeros_like(s) for _ in range(world)]
    dist.all_gather(allsums, s)
    return all(bool(torch.equal(allsums[0], x)) for x in allsums)


# ------------------------------------------------------------------------------------------------ main
def main():
    args = parse()
    if args.torch_restatement:
        return torch_restatement_child(args.torch_restatement)
    global SPATIAL_ORDER
    SPATIAL_ORDER = not args.no_spatial_order
    world, rank, local = dist_env(args)
    if args.dry_run:
        return dry_run(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the rasterizer has no CPU fallback")
    _stdout_to_stderr()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force_dist = args.force_dist
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from w3d_amd.train import Trainer, render_views
    from w3d_amd.loss import photometric_loss, photometric_loss_torch

    bg = torch.zeros(3, device=dev)
    sc, model, opt, cams = build_scene(args, dev)
    make_ground_truth(args, cams, dev, bg)
    if args.dropin_only:
        del model
        g = torch.Generator(device="cpu").manual_seed(0)
        _emit(json.dumps({"dropin": time_dropin(args, sc, cams, bg, dev, torch.randperm(len(cams), generator=g).tolist())}))
        return
    # (before the Trainer puts the model into Morton order: the oracle leg builds the same scene in the order it is created in)
    # camera 0 with the INITIAL parameters, for "PSNR vs ref" (compared with the oracle's render in the cpu_baseline leg)
    own_view0 = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from w3d_amd.fused_step import render_raw
        import numpy as np
        from w3d_amd.fused_step import backward_raw
        with torch.no_grad():
            r0 = render_raw(cams[0], model, bg)
            # ... and the backward of that view on the cpu_baseline leg's fixed dL/dcolor (N(0,1), seed 0): gradients of every
            # parameter block and the densification norm, held against the oracle's there (`parity_tail`)
            gc0 = torch.from_numpy(np.random.RandomState(0).randn(3, args.height, args.width).astype(np.float32)).to(dev)
            gn0, _ = backward_raw(model, r0["handle"], gc0, want_norm=True)
            own_grads0 = {k: model.grad_view(k).detach().cpu().numpy().copy() for k in ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")}
            own_grads0["densify_norm"] = gn0.cpu().numpy().astype(np.float64)
            own_grads0["radii"] = r0["radii"].cpu().numpy()
            model.flat_grad.zero_()
        own_view0 = tuple(t.detach().cpu().numpy() for t in (r0["render"], r0["depth"], r0["alpha"], cams[0].original_image)) + (own_grads0, sc)
        del r0, gc0, gn0

    loss_fn = photometric_loss_torch if args.torch_loss else photometric_loss
    trainer = Trainer(model, cams, opt, bg, densify=False, loss_fn=loss_fn, fused=False if args.autograd_path else None,
                      force_exchange=force_dist, exchange="rows" if args.exchange == "auto" else args.exchange,
                      spatial_order=not args.no_spatial_order)
    meter = StepMeter(trainer, world, dev)
    sync = meter.sync
    _progress("warm-up")
    it = 0
    for _ in range(args.warmup):
        it += 1
        trainer.step(it)
    # N > 1: nothing re-synchronises the replicas, so the exchange must keep them bit-identical.  Checked on the real links
    # before anything is timed; if the low-rank exchange (replicated optimizer) fails the check, the replicas are
    # re-synchronised from rank 0 and the run falls back to the dense exchange (reduce-scatter, sharded Adam, all-gather)
    selfcheck = None
    autotune = None
    multi = world > 1 or force_dist        # (--force-dist: the N > 1 code below runs on a 1-rank group, so that one GPU can test it)
    if multi and trainer.fused and args.exchange == "auto":
        autotune = {}
        # (rows: one 32-bit view mask per Gaussian; lowrank_early: the low-rank form with its colour-gradient all-gather issued
        #  between the blend backward and the per-Gaussian backward, so that 12 of its 23 bytes per Gaussian travel under compute)
        for mode in (("rows", "lowrank", "lowrank_early") if world <= 32 else ("lowrank", "lowrank_early")):
            trainer.exchange_mode, trainer.early_gather = mode.split("_")[0], mode.endswith("_early")
            for _ in range(2):
                it += 1
                trainer.step(it)
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(6):
                it += 1
                trainer.step(it)
            torch.cuda.synchronize()
            dt = torch.tensor([(time.perf_counter() - t0) / 6], device=dev, dtype=torch.float64)
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)                   # the same number, hence the same choice, on every rank
            autotune[mode] = round(1e3 * float(dt), 4)
        best = min(autotune, key=autotune.get)
        trainer.exchange_mode, trainer.early_gather = best.split("_")[0], best.endswith("_early")
    if multi:
        selfcheck = {"mode_requested": trainer.exchange_mode if trainer.fused else "dense",
                     "replicas_identical_after_warmup": replicas_identical(model, world, dev)}
        if not selfcheck["replicas_identical_after_warmup"]:
            for buf in (model.flat_store, model.optimizer.exp_avg, model.optimizer.exp_avg_sq, model.xyz_gradient_accum,
                        model.denom, model.max_radii2D):
                dist.broadcast(buf, 0)
            if trainer.fused and trainer.exchange_mode in ("lowrank", "rows"):
                trainer.exchange_mode = "dense"
                selfcheck["fell_back_to"] = "dense"
            for _ in range(max(2, args.warmup // 2)):
                it += 1
                trainer.step(it)
            selfcheck["replicas_identical_after_fallback"] = replicas_identical(model, world, dev)
    _progress("timed steps")
    meas = meter.measure(args.steps, it, args.profile, args.all_stages)
    elapsed, it, stages, stage_ms = meas["elapsed"], meas["it"], meas["stages"], meas["stage_ms"]
    final_loss = float(trainer.last["loss"])

    extras_on = not args.no_extras and not args.trained_only and not args.densified_only
    is_fused = bool(trainer.fused)
    extras = {}
    exchange = None
    if multi:
        exchange = exchange_bandwidth(model, world, dev, rows=max(getattr(trainer, "last_row_counts", None) or [0]))
        exchange["mode"] = (trainer.exchange_mode + ("_early" if trainer.exchange_mode == "lowrank" and trainer.early_gather else "")) \
            if trainer.fused else "dense"
        exchange["selfcheck"] = selfcheck
        if autotune is not None:
            exchange["autotune_ms_per_step"] = autotune
        if exchange["mode"] == "rows":
            exchange["rows"] = {"steps_by_form": dict(trainer.exchange_used), "rows_per_view_last_step": getattr(trainer, "last_row_counts", None),
                                "row_bytes": 64, "break_even_rows": trainer.rows_limit(model.num_points)}
        exchange["replicas_identical_after_timed_steps"] = replicas_identical(model, world, dev)
        # one boolean for the driver: the replicas were bit-identical after the warm-up (or after the dense fallback) AND
        # after the timed steps — nothing re-synchronises them, so this is the proof that the exchange is correct on the links
        exchange["selfcheck_ok"] = bool((selfcheck["replicas_identical_after_warmup"] or
                                         selfcheck.get("replicas_identical_after_fallback", False)) and
                                        exchange["replicas_identical_after_timed_steps"])
    if extras_on:
        _progress("extras: render / FlashSplat")
        # forward-only render throughput (reference render.py's use), same scene, views cycled
        n_r = max(4, min(args.steps, 72))
        # (an untimed pass of the same length first: the loop keeps its n_r output images, and a first-time hipMalloc of each
        #  of them inside the timed region costs more than the frame it holds)
        render_views(model, [cams[i % len(cams)] for i in range(n_r)], bg)
        sync()
        r0 = time.perf_counter()
        render_views(model, [cams[i % len(cams)] for i in range(n_r)], bg)
        sync()
        r_el = torch.tensor([time.perf_counter() - r0], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(r_el, op=dist.ReduceOp.MAX)
        extras["render_mpix_per_s"] = round(world * n_r * args.width * args.height / 1e6 / float(r_el), 1)

        # config C4: FlashSplat per-mask contribution render (run_3d_seg.py's inner call), binary mask, same scene
        from w3d_amd.gaussian_renderer import flashsplat_render, flashsplat_render_masks
        from w3d_amd.train import PipelineParams
        yy, xx = torch.meshgrid(torch.arange(args.height, device=dev), torch.arange(args.width, device=dev), indexing="ij")
        mask = (((xx - args.width // 2) ** 2 + (yy - args.height // 2) ** 2) < (args.height // 3) ** 2).float()
        n_f = 16
        with torch.no_grad():
            # (warm-up with the loop's own holding pattern — view, running sum, next view — so that every block the loop
            #  needs exists in torch's allocator before the clock starts: the 16 timed views take ~15 ms, one first-time
            #  hipMalloc 1-2 ms)
            counts = None
            for i in range(3):
                uc = flashsplat_render(cams[i], model, PipelineParams(), bg, gt_mask=mask, obj_num=1)["used_count"]
                counts = uc if counts is None else counts + uc
            del counts, uc
            sync()
            f0 = time.perf_counter()
            counts = None
            for i in range(n_f):
                uc = flashsplat_render(cams[i % len(cams)], model, PipelineParams(), bg, gt_mask=mask, obj_num=1)["used_count"]
                counts = uc if counts is None else counts + uc
            sync()
            extras["flashsplat_views_per_s"] = round(world * n_f / (time.perf_counter() - f0), 1)
        del counts
        # run_3d_seg.py's MOST FREQUENT call (find_match :130-134, ~29 views per object mask and refine round; :362, 36 views):
        # flashsplat_render(..., used_mask=obj_used_mask) followed by alpha > 0.5 -> bounding box -> IoU against the
        # candidate masks.  The mask is applied inside the preprocess kernel; the scoring runs on the device.
        from w3d_amd.segmentation import mask_iou_device
        _progress("extras: subset renders")
        head = ((model.get_xyz.detach() - torch.tensor([0.2, -0.1, 0.3], device=dev)).norm(dim=1) < 0.06)
        cand = (torch.stack([torch.roll(mask, shifts=25 * k, dims=1) for k in range(4)]) > 0).to(torch.uint8)
        with torch.no_grad():
            for i in range(2):
                flashsplat_render(cams[i], model, PipelineParams(), bg, used_mask=head)
            sync()
            f0 = time.perf_counter()
            for i in range(args.views):
                pkg_s = flashsplat_render(cams[i % len(cams)], model, PipelineParams(), bg, used_mask=head)
                mask_iou_device(pkg_s["alpha"], cand, 0.5)
            sync()
            extras["flashsplat_subset_views_per_s"] = round(world * args.views / (time.perf_counter() - f0), 1)
            # the reference's own formulation of the same call for comparison: activated blocks gathered with the mask
            # (gaussian_renderer/__init__.py:151-156,168-170,186-187) through the drop-in rasterizer module, alpha to the host,
            # numpy threshold / bbox / IoU (run_3d_seg.py:131-163) — reached here by handing the mask over as an index tensor
            head_idx = head.nonzero(as_tuple=True)[0]
            cand_np = cand.cpu().numpy() > 0
            for i in range(2):
                flashsplat_render(cams[i], model, PipelineParams(), bg, used_mask=head_idx)
            sync()
            f0 = time.perf_counter()
            for i in range(12):
                a = flashsplat_render(cams[i % len(cams)], model, PipelineParams(), bg, used_mask=head_idx)["alpha"]
                pred = a.squeeze().detach().cpu().numpy() > 0.5
                for m_ in cand_np:
                    inter, union = (m_ & pred).sum(), (m_ | pred).sum()
            sync()
            extras["flashsplat_subset_reference_formulation_views_per_s"] = round(world * 12 / (time.perf_counter() - f0), 1)
            extras["flashsplat_subset"] = {"gaussians_in_mask": int(head.sum()), "candidate_masks": 4,
                                           "loop": "flashsplat_render(used_mask) + alpha>0.5 -> bbox -> IoU, per view (host reads 13 counters)"}
            del head_idx, a
        del head, cand, pkg_s
        # ... and run_3d_seg.py's real loop shape: several object masks per view — one forward, the blend repeated per mask
        n_m = 8
        masks = torch.stack([torch.roll(mask, shifts=40 * k, dims=1) for k in range(n_m)])
        with torch.no_grad():
            flashsplat_render_masks(cams[0], model, PipelineParams(), bg, masks[:2], obj_num=1)
            sync()
            f0 = time.perf_counter()
            for i in range(4):
                uc = flashsplat_render_masks(cams[i % len(cams)], model, PipelineParams(), bg, masks, obj_num=1)["used_count"]
            sync()
            extras["flashsplat_masks_per_s_8_per_view"] = round(world * 4 * n_m / (time.perf_counter() - f0), 1)
        # non-overlapping instance masks (8 vertical stripes): one blend over the merged label map
        stripes = torch.stack([((xx >= k * args.width // n_m) & (xx < (k + 1) * args.width // n_m)).float() for k in range(n_m)])
        with torch.no_grad():
            flashsplat_render_masks(cams[0], model, PipelineParams(), bg, stripes, obj_num=1)
            sync()
            f0 = time.perf_counter()
            for i in range(4):
                uc = flashsplat_render_masks(cams[i % len(cams)], model, PipelineParams(), bg, stripes, obj_num=1)["used_count"]
            sync()
            extras["flashsplat_masks_per_s_8_disjoint_per_view"] = round(world * 4 * n_m / (time.perf_counter() - f0), 1)
        # eval_wheatgs.py's shape: ONE label image with hundreds of object ids (obj_num = max label)
        K = 300
        # (40-pixel cells: label boundaries cut through the 16x16 tiles, up to four labels per tile)
        labels = ((xx // 40) + (args.width // 40 + 1) * (yy // 40)).remainder(K + 1).float()
        with torch.no_grad():
            # (two warm-up views with the loop's own holding pattern — the previous view's 2.4-GB count matrix is still
            #  referenced while the next one is allocated — so that both blocks exist in torch's allocator before the clock
            #  starts: a first-time 2.4-GB hipMalloc inside a 4-view window costs ten times the four renders)
            for i in range(2):
                uc = flashsplat_render(cams[i], model, PipelineParams(), bg, gt_mask=labels, obj_num=K)["used_count"]
            sync()
            f0 = time.perf_counter()
            for i in range(4):
                uc = flashsplat_render(cams[i % len(cams)], model, PipelineParams(), bg, gt_mask=labels, obj_num=K)["used_count"]
            sync()
            extras["flashsplat_views_per_s_300_labels"] = round(world * 4 / (time.perf_counter() - f0), 1)
        del uc, masks, stripes, labels
        torch.cuda.empty_cache()

    P, HW = args.points, args.width * args.height
    single = world == 1 and not force_dist
    fused_adam = trainer.fused and trainer.fused_adam and single
    ws = mean_workload(model, cams, bg, dev) if rank == 0 else None

    # the UNMODIFIED reference loop body on the drop-in modules (single GPU: the reference is single-GPU), and the same
    # loop with ONLY the rasterizer module swapped (INTEGRATION.md section 1 as written)
    dropin = modules_only = None
    if extras_on and single:
        _progress("drop-in loop")
        dropin = time_dropin(args, sc, cams, bg, dev, trainer.perm)
        _progress("modules-only loop")
        try:
            modules_only = time_modules_only(args, sc, cams, bg, dev, trainer.perm)
        except Exception as e:                     # an extra must never take the bench line down
            modules_only = {"error": repr(e)}

    # what N GPUs should do with this scene, predicted from single-GPU measurements (scale_model)
    scale = None
    do_scale = extras_on and single and trainer.fused and rank == 0 and not args.no_scale_model
    if do_scale:
        _progress("scale model: untrained scene")
        try:
            _stdout_to_stderr()
            sm, it = scale_model(model, opt, cams, bg, dev, it)
            scale = {"untrained": sm}
        except Exception as e:
            scale = {"error": repr(e)}

    # the same measurement on a TRAINED scene: the fit lowers opacities and lengthens the per-tile walks
    trained = None
    if not args.no_extras and not args.densified_only and args.trained_steps > 0 and trainer.fused:
        _progress("trained scene")
        for _ in range(args.trained_steps):
            it += 1
            trainer.step(it)
        tm = meter.measure(args.steps, it, args.profile, args.all_stages)
        it = tm["it"]
        trained = {"value": round(world * args.steps / tm["elapsed"], 3), "ms_per_step": round(1e3 * tm["elapsed"] / args.steps, 4),
                   "after_steps": it - args.steps, "final_loss": round(float(trainer.last["loss"]), 6), "stage_ms": tm["stage_ms"]}
        if rank == 0:
            w2 = mean_workload(model, cams, bg, dev)
            trained.update(visible_per_view=int(w2["V"]), tile_instances_per_view=int(w2["R"]),
                           walked_instances_per_view=int(w2["R_walk"]),
                           mean_contributors_per_pixel=round(w2["mean_contrib"], 2), mean_last_contributor_list_position=round(w2["mean_last"], 2),
                           roofline=roofline_object(tm, P, w2, HW, fused_adam, args.steps / tm["elapsed"], "trained"))
        if do_scale and scale is not None and "error" not in scale:
            _progress("scale model: trained scene")
            try:
                scale["trained"], it = scale_model(model, opt, cams, bg, dev, it)
            except Exception as e:
                scale["trained"] = {"error": repr(e)}

    # ... and on a DENSIFIED one: a model grown to ~2 M Gaussians by the reference's schedule (config C3's regime)
    densified = None
    if not args.no_extras and not args.trained_only and args.densify_iterations > 0 and single and is_fused and rank == 0:
        _progress("densified scene")
        del trainer, meter
        model = None
        torch.cuda.empty_cache()
        try:
            densified, _m = densified_scene(args, dev, bg, _progress, with_scale_model=do_scale)
            del _m
            if scale is not None and "scale_model" in densified:
                scale["densified"] = densified.pop("scale_model")
        except Exception as e:
            densified = {"error": repr(e)}
        torch.cuda.empty_cache()

    # ... and on a scene of OPAQUE surfaces trained by the reference's schedule as it is (what a photographed plot converges to)
    opaque = None
    if not args.no_extras and not args.trained_only and not args.densified_only and args.opaque_iterations > 0 and single and is_fused \
            and rank == 0:
        _progress("opaque scene")
        try:
            opaque = opaque_scene(args, dev, bg, _progress, with_scale_model=do_scale)
            if scale is not None and "scale_model" in opaque:
                scale["opaque_padded"] = opaque.pop("scale_model")
        except Exception as e:
            opaque = {"error": repr(e)}
        torch.cuda.empty_cache()

    if rank == 0:
        it_per_s = args.steps / elapsed          # per GPU (weak scaling: every rank runs this step)
        roof = roofline_object(meas, P, ws, HW, fused_adam, it_per_s, "untrained")
        # the blend backward's own roof is VALU issue, not HBM: kept beside the dominant kernel's object
        valu = valu_object("render_bwd", stage_ms["render_bwd"], "untrained") if "render_bwd" in stage_ms else None
        if args.all_stages:
            for k, (c, ms) in sorted(stages.items(), key=lambda kv: -kv[1][1]):
                print(f"[stage] {k:18s} {c:5d} launches  avg {ms / c:8.4f} ms", file=sys.stderr)
        step_desc = "fused raw-parameter kernels (no autograd)" if is_fused else "render() + autograd (Trainer.step)"
        out = {
            "metric": "train iters/sec @ 2M Gaussians, 1600x1200 (render+loss+backward+Adam); render Mpix/sec and PSNR vs the oracle alongside",
            "value": round(world * args.steps / elapsed, 3), "unit": "iters/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            # the same step on the scene after `trained_steps` more training steps (opacities dropped, walks 2-3x longer) and on a
            # model GROWN to ~2 M Gaussians by the reference's densification schedule: the regimes a real run spends its time in
            "trained_value": None if trained is None else trained["value"],
            "trained_ms_per_step": None if trained is None else trained["ms_per_step"],
            "densified_value": None if not densified or "value" not in densified else densified["value"],
            "densified_ms_per_step": None if not densified or "value" not in densified else densified["ms_per_step"],
            # ... and a scene of opaque surfaces trained by the reference's schedule as it is, padded to --points (opaque_scene)
            "opaque_value": None if not opaque or "padded" not in opaque else opaque["padded"]["value"],
            "opaque_ms_per_step": None if not opaque or "padded" not in opaque else opaque["padded"]["ms_per_step"],
            "dropin_iters_per_s": None if dropin is None else dropin["iters_per_s"],
            "modules_only_iters_per_s": None if not modules_only or "iters_per_s" not in modules_only else modules_only["iters_per_s"],
            "config": {"workload": f"C3: plot-shaped synthetic scene, {P} Gaussians, SH degree 3, "
                                   f"{args.width}x{args.height}, {args.views} overhead cameras, depth+alpha channels",
                       "points": P, "image": [args.width, args.height], "views_per_step": world,
                       "parallelism": f"view-parallel dp{world}" if world > 1 else "single GPU",
                       "visible_per_view": int(ws["V"]), "tile_instances_per_view": int(ws["R"]),
                       "walked_instances_per_view": int(ws["R_walk"]),
                       "mean_contributors_per_pixel": round(ws["mean_contrib"], 2),
                       "mean_last_contributor_list_position": round(ws["mean_last"], 2),
                       "loss": "torch conv2d" if args.torch_loss else "fused HIP L1+SSIM",
                       "step": step_desc,
                       "storage_order": "Morton order of the positions (Trainer(spatial_order=True))" if SPATIAL_ORDER
                                        else "as created (random)",
                       "final_loss": round(final_loss, 6)},
            "roofline": roof,
            "roofline_valu_kernel": valu,
            "stage_ms": stage_ms,
        }
        out.update(extras)
        if dropin is not None:
            out["dropin"] = dict(dropin, loop="tests/standin_checkout/train_loop.py (import lines + loop body of reference "
                                              "train_vanilla_3dgs.py:16-18,55-115) UNMODIFIED under w3d_amd.dropin.install(): render / "
                                              "GaussianModel / l1_loss / ssim redirected to this repo; loss.item() and the "
                                              "boolean-mask statistics lines (host syncs of the reference loop) included")
        if modules_only is not None:
            out["modules_only"] = modules_only
            if dropin is not None and "iters_per_s" in modules_only:
                out["modules_only"]["with_redirect_iters_per_s"] = dropin["iters_per_s"]
                out["modules_only"]["redirect_speedup"] = round(dropin["iters_per_s"] / modules_only["iters_per_s"], 2)
        if trained is not None:
            out["trained_scene"] = trained
        if densified is not None:
            out["densified_scene"] = densified
        if opaque is not None:
            out["opaque_scene"] = opaque
        if exchange is not None:
            out["exchange"] = exchange
        if scale is not None:
            out["scale_model"] = scale
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, own_view0)
            out["psnr"] = out["cpu_baseline"].pop("psnr", None)
            out["parity_tail"] = out["cpu_baseline"].pop("parity_tail", None)
        else:
            out["cpu_baseline"] = None
            out["psnr"] = None
        if densified and "psnr_train_db_before_after" in densified:
            out["psnr"] = dict(out["psnr"] or {}, densified_run_train_db_before_after=densified["psnr_train_db_before_after"],
                               densified_run_heldout_db_before_after=densified["psnr_heldout_db_before_after"])
        _emit(json.dumps(out))
    if world > 1 or force_dist:
        dist.barrier()
    if dist.is_initialized():            # (N = 1: scale_model's 1-rank group)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
