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

Prints ONE COMPACT JSON line (< 4 KB) on rank 0: metric / value / ms_per_step / config / roofline / cpu_baseline / psnr /
parity (+ exchange when N > 1) — everything the driver parses.  The full record (per-stage tables, probe timings, ...) is
written to `bench_detail.json` next to this file (`detail_file` in the line; --detail-file to move it).

The default run measures the headline step, the forward-only render rate and a bounded cpu_baseline / parity sample and
nothing else.  `--full` adds the legs under bench_legs/ (FlashSplat renders, the unmodified reference loop under the import
redirect, the trained / densified / opaque scenes, the N-GPU scale model, config C1 on the host) to the detail file; their
headline scalars also appear in the line.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from bench_legs import common  # noqa: E402  (puts wheat-3dgs_amd/ and tests/ on sys.path)
from bench_legs.common import (HBM_PEAK_GBS, StepMeter, _progress, build_scene, make_ground_truth, mean_workload,  # noqa: E402,F401
                               roofline_object, valu_object, workload_stats)


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
    ap.add_argument("--full", action="store_true",
                    help="also run the legs under bench_legs/ (FlashSplat / drop-in / modules-only / trained, densified and opaque "
                         "scenes / scale model / config C1 on the host); they go to the detail file")
    ap.add_argument("--no-extras", action="store_true", help="(kept for old command lines: the default already runs no extra leg)")
    ap.add_argument("--detail-file", default=os.path.join(ROOT, "bench_detail.json"),
                    help="where the full record goes ('' or 'none': nowhere)")
    ap.add_argument("--densify-iterations", type=int, default=8000,
                    help="iterations of the compressed C3 schedule that grows the densified scene (densified_scene leg); 0: skip")
    ap.add_argument("--densify-grad-threshold", type=float, default=1.5e-5,
                    help="densify_grad_threshold of that schedule (reference default 2e-4, arguments/__init__.py:88, tuned for "
                         "photographs: the smooth synthetic views only grow to ~0.26 M Gaussians with it; stated in the line)")
    ap.add_argument("--opaque-iterations", type=int, default=15_000,
                    help="iterations of the reference schedule that trains the opaque-surface scene (opaque_scene leg; 15000 = "
                         "arguments/__init__.py:73-89 as it is); 0: skip")
    ap.add_argument("--densified-only", action="store_true",
                    help="of the extra measurements run only the densified-scene one (implies --full for that leg)")
    ap.add_argument("--modules-only-steps", type=int, default=-1,
                    help="steps of the INTEGRATION.md section 1 measurement (rasterizer modules only; -1: = min(--steps, 60), 0: skip)")
    ap.add_argument("--trained-only", action="store_true",
                    help="of the extra measurements run only the trained-scene one (implies --full for that leg; kernel A/B runs: "
                         "profiles/ab_variants.sh)")
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




# ------------------------------------------------------------------------------------------------ the compact line
LINE_LIMIT = 4096          # the driver parses a bounded tail of stdout (round 5's 38 KB line was cut: BENCH_r05.parsed = null)
PARITY_STATEMENT = "p99 <= 1e-4; tail attributed; reference-CUDA parity unpinned"


def _pick(d, keys):
    return None if not isinstance(d, dict) else {k: d[k] for k in keys if k in d}


def compact_line(full):
    """The ONE line the driver parses, cut out of the full record `full` (what --detail-file receives): the contract's fields,
    the dominant kernel's roofline, the cpu baseline, PSNR vs the oracle and the parity statistics — under LINE_LIMIT bytes
    whatever the legs added to the record (tests/test_bench_line.py)."""
    out = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    for k in ("render_mpix_per_s", "trained_value", "densified_value", "opaque_value", "dropin_iters_per_s",
              "modules_only_iters_per_s", "flashsplat_views_per_s"):
        if full.get(k) is not None:
            out[k] = full[k]
    cfg = full.get("config") or {}
    out["config"] = {"workload": cfg.get("workload"), "points": cfg.get("points"), "image": cfg.get("image"),
                     "views_per_step": cfg.get("views_per_step"), "parallelism": cfg.get("parallelism"),
                     "V": cfg.get("visible_per_view"), "R": cfg.get("tile_instances_per_view"),
                     "R_walk": cfg.get("walked_instances_per_view"), "step": cfg.get("step"), "loss": cfg.get("loss")}
    roof = full.get("roofline")
    if roof:
        r = _pick(roof, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_ratio", "avg_launch_ms",
                         "launches", "algorithmic_bytes_per_launch", "scene", "peak_measured", "frac_of_measured"))
        if "hbm_view" in roof:
            r["hbm_view"] = _pick(roof["hbm_view"], ("achieved", "frac", "traffic_ratio"))
        st = roof.get("step")
        if st:
            r["step"] = _pick(st, ("algorithmic_bytes", "achieved_GBps", "frac", "design_bytes", "design_frac"))
        out["roofline"] = r
    else:
        out["roofline"] = None
    out["stage_ms"] = full.get("stage_ms")
    cb = full.get("cpu_baseline")
    out["cpu_baseline"] = None if not cb else _pick(cb, ("value", "unit", "cores", "kind", "bracket", "sample", "render_mpix_per_s"))
    ps = full.get("psnr")
    out["psnr"] = None if not ps else _pick(ps, ("delta_db", "own_vs_gt_db", "oracle_vs_gt_db", "bar_db"))
    pt = full.get("parity_tail")
    if pt and "hip_vs_oracle" in pt:
        dn = pt["hip_vs_oracle"]["densify_norm"]
        par = {"densify_norm": _pick(dn, ("n", "p50", "p99", "p999", "max", "outliers")), "bar": pt.get("bar"),
               "radii_differing": pt.get("radii_differing")}
        at = pt.get("attribution")
        if at:
            par.update(unattributed_outliers=at["unattributed_outliers"], flipped_pixels=at["flipped_pixels"],
                       flipped_not_on_a_threshold=at["flipped_not_on_a_threshold"],
                       outliers_blended_at_a_flipped_pixel=at["beyond_1e4_blended_at_a_flipped_pixel"])
        par["statement"] = PARITY_STATEMENT
        out["parity"] = par
    elif pt:
        out["parity"] = {"error": str(pt.get("error"))[:200], "statement": PARITY_STATEMENT}
    ex = full.get("exchange")
    if ex:
        out["exchange"] = ex
    out["detail_file"] = full.get("detail_file")
    line = json.dumps(out)
    if len(line) >= LINE_LIMIT:            # (cannot happen with the fields above unless `exchange` grows: drop the optional parts)
        for k in ("stage_ms", "exchange"):
            if k in out and len(line) >= LINE_LIMIT:
                out[k] = {"see": "detail_file"} if k == "stage_ms" else _pick(out[k], ("mode", "selfcheck_ok", "autotune_ms_per_step"))
                line = json.dumps(out)
    assert len(line) < LINE_LIMIT, len(line)
    return line


def write_detail(path, full):
    if not path or path.lower() == "none":
        return None
    try:
        with open(path, "w") as f:
            json.dump(full, f, indent=1)
        return os.path.relpath(path, ROOT) if os.path.abspath(path).startswith(ROOT + os.sep) else path
    except OSError as e:             # a read-only checkout must not cost the line
        _progress(f"detail file not written: {e!r}")
        return None


# ------------------------------------------------------------------------------------------------ main
def capture_view0(args, model, cams, bg, dev, sc):
    """camera 0 with the INITIAL parameters through the HIP path, before any training step and before the Trainer puts the model
    into Morton order: images (for "PSNR vs ref"), gradients of every parameter block + the densification norm on the
    cpu_baseline leg's fixed dL/dcolor (N(0,1), seed 0), radii and the final transmittance (for the parity object) — all
    compared with the oracle's on the same inputs in bench_legs/cpu_baseline.py."""
    import numpy as np
    from w3d_amd.fused_step import backward_raw, render_raw
    from w3d_amd.rasterizer import debug_pixel_state
    with torch.no_grad():
        r0 = render_raw(cams[0], model, bg)
        gc0 = torch.from_numpy(np.random.RandomState(0).randn(3, args.height, args.width).astype(np.float32)).to(dev)
        final_T = debug_pixel_state(r0["handle"])[0].cpu().numpy()
        gn0, _ = backward_raw(model, r0["handle"], gc0, want_norm=True)
        own = {k: model.grad_view(k).detach().cpu().numpy().copy() for k in ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")}
        own["densify_norm"] = gn0.cpu().numpy().astype(np.float64)
        own["radii"] = r0["radii"].cpu().numpy()
        own["final_T"] = final_T
        model.flat_grad.zero_()
    return tuple(t.detach().cpu().numpy() for t in (r0["render"], r0["depth"], r0["alpha"], cams[0].original_image)) + (own, sc)


def main():
    args = parse()
    if args.torch_restatement:
        from bench_legs.cpu_baseline import torch_restatement_child
        return torch_restatement_child(args.torch_restatement)
    common.SPATIAL_ORDER = not args.no_spatial_order
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
    from w3d_amd.train import Trainer
    from w3d_amd.loss import photometric_loss, photometric_loss_torch

    bg = torch.zeros(3, device=dev)
    sc, model, opt, cams = build_scene(args, dev)
    make_ground_truth(args, cams, dev, bg)
    if args.dropin_only:
        from bench_legs.dropin import time_dropin
        del model
        g = torch.Generator(device="cpu").manual_seed(0)
        _emit(json.dumps({"dropin": time_dropin(args, sc, cams, bg, dev, torch.randperm(len(cams), generator=g).tolist())}))
        return
    own_view0 = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        own_view0 = capture_view0(args, model, cams, bg, dev, sc)

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
    if multi:
        from bench_legs.exchange import exchange_bandwidth, replicas_identical
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

    # which legs run: none by default; --full all of them; --trained-only / --densified-only exactly that one
    only = args.trained_only or args.densified_only
    legs_on = args.full and not only
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
    # the other half of BASELINE.json's metric: forward-only render Mpix/s (reference render.py:24-35), same scene, views cycled
    from bench_legs.render_legs import flashsplat_legs, render_mpix_per_s
    _progress("forward-only render")
    extras["render_mpix_per_s"] = render_mpix_per_s(args, model, cams, bg, dev, world, sync)
    if legs_on:
        extras.update(flashsplat_legs(args, model, cams, bg, dev, world, sync))

    P, HW = args.points, args.width * args.height
    single = world == 1 and not force_dist
    fused_adam = trainer.fused and trainer.fused_adam and single
    ws = mean_workload(model, cams, bg, dev) if rank == 0 else None

    # the UNMODIFIED reference loop body on the drop-in modules (single GPU: the reference is single-GPU), and the same
    # loop with ONLY the rasterizer module swapped (INTEGRATION.md section 1 as written)
    dropin = modules_only = None
    if legs_on and single:
        from bench_legs.dropin import time_dropin, time_modules_only
        _progress("drop-in loop")
        dropin = time_dropin(args, sc, cams, bg, dev, trainer.perm)
        _progress("modules-only loop")
        try:
            modules_only = time_modules_only(args, sc, cams, bg, dev, trainer.perm)
        except Exception as e:                     # a leg must never take the bench line down
            modules_only = {"error": repr(e)}

    # what N GPUs should do with this scene, predicted from single-GPU measurements (scale_model)
    scale = None
    do_scale = legs_on and single and trainer.fused and rank == 0 and not args.no_scale_model
    if do_scale:
        from bench_legs.scale_model import scale_model
        _progress("scale model: untrained scene")
        try:
            _stdout_to_stderr()
            sm, it = scale_model(model, opt, cams, bg, dev, it)
            scale = {"untrained": sm}
        except Exception as e:
            scale = {"error": repr(e)}

    # the same measurement on a TRAINED scene: the fit lowers opacities and lengthens the per-tile walks
    trained = None
    if (legs_on or args.trained_only) and args.trained_steps > 0 and trainer.fused:
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
    if (legs_on or args.densified_only) and args.densify_iterations > 0 and single and is_fused and rank == 0:
        from bench_legs.scenes import densified_scene
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
    if legs_on and args.opaque_iterations > 0 and single and is_fused and rank == 0:
        from bench_legs.scenes import opaque_scene
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
            # (--full) the same step on the scene after `trained_steps` more training steps (opacities dropped, walks 2-3x longer) and
            # on a model GROWN to ~2 M Gaussians by the reference's densification schedule: the regimes a real run spends its time in
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
                       "storage_order": "Morton order of the positions (Trainer(spatial_order=True))" if common.SPATIAL_ORDER
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
            from bench_legs.cpu_baseline import cpu_baseline
            out["cpu_baseline"] = cpu_baseline(args, own_view0, full=legs_on)
            out["psnr"] = out["cpu_baseline"].pop("psnr", None)
            out["parity_tail"] = out["cpu_baseline"].pop("parity_tail", None)
        else:
            out["cpu_baseline"] = None
            out["psnr"] = None
        if densified and "psnr_train_db_before_after" in densified:
            out["psnr"] = dict(out["psnr"] or {}, densified_run_train_db_before_after=densified["psnr_train_db_before_after"],
                               densified_run_heldout_db_before_after=densified["psnr_heldout_db_before_after"])
        out["detail_file"] = None if not args.detail_file or args.detail_file.lower() == "none" else \
            (os.path.relpath(args.detail_file, ROOT) if os.path.abspath(args.detail_file).startswith(ROOT + os.sep) else args.detail_file)
        if write_detail(args.detail_file, out) is None:
            out["detail_file"] = None
        _emit(compact_line(out))
    if world > 1 or force_dist:
        dist.barrier()
    if dist.is_initialized():            # (N = 1 --full: scale_model's 1-rank group)
        dist.destroy_process_group()


# (profiles/*.py and tests build the benchmark's scenes through this module)
def __getattr__(name):
    import importlib
    for mod in ("bench_legs.scenes", "bench_legs.scale_model", "bench_legs.dropin", "bench_legs.cpu_baseline", "bench_legs.exchange",
                "bench_legs.render_legs"):
        m = importlib.import_module(mod)
        if hasattr(m, name):
            return getattr(m, name)
    raise AttributeError(name)


if __name__ == "__main__":
    main()
