#!/usr/bin/env python
"""K fused training steps of ONE of bench.py's scenes between two marker kernels, for rocprofv3 counter passes.

    python3 profiles/scene_step.py --scene untrained|trained|densified --steps 6 [--model-file /tmp/x.pt]

The scene is built exactly as bench.py builds it (same generator, cameras, ground truth; `trained` = 3000 more steps at
fixed P, `densified` = bench.grow_densified_model).  With --model-file the prepared model (GaussianModel.capture()) is
saved there on the first run and restored on later ones, so the counter passes do not repeat the preparation under the
profiler.  The K measured steps sit between two launches of a marker kernel (torch.lgamma on 1 element — nothing else in
the process launches it): profiles/summarize_pmc.py --window / summarize_sq.py --window sum every kernel's counters over
the dispatches between the markers and divide by K -> per-step, per-kernel figures that bench.py adds up per stage."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "wheat-3dgs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="untrained", choices=("untrained", "trained", "densified", "opaque"))
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=40, help="steps before the window (>= 36 so that every camera's walk hint exists)")
    ap.add_argument("--model-file", default=None)
    ap.add_argument("--trained-steps", type=int, default=3000)
    ap.add_argument("--freeze", action="store_true",
                    help="with --report: the parameters do NOT change — every iteration is forward + loss + gradient-writing backward "
                         "(no optimizer step), so variants that alter results are still timed on identical work")
    ap.add_argument("--walk-stats", action="store_true",
                    help="print the distribution of the tiles' walk lengths (the blend backward's work per wave) per camera and what "
                         "the one-wave-per-tile schedule loses to its longest tiles: list-scheduling makespan of the 8 XCD ranges on "
                         "512 wave slots each, tiles longest first, against the perfectly divisible bound; the same with every tile "
                         "above `split` x the bound cut into two half-tile waves costing 0.6 of the tile each")
    ap.add_argument("--bucket-stats", action="store_true",
                    help="populations of the depth sort's 1024 buckets (w3d_binning.hip) on six cameras of the scene: max, p99, buckets "
                         "beyond the LDS capacity of the in-bucket sort")
    ap.add_argument("--report", action="store_true",
                    help="instead of the marker window: time --steps steps (bench.StepMeter: probe, timed, per-stage events) and print "
                         "one JSON line {scene, iters_per_s, ms_per_step, stage_ms} — the A/B harness of profiles/ab_scenes.sh")
    a = ap.parse_args()
    args = bench.parse_defaults()
    from w3d_amd.gaussian_model import GaussianModel, OptimizationParams
    from w3d_amd.train import Trainer
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    bg = torch.zeros(3, device=dev)
    have = a.model_file and os.path.exists(a.model_file)
    if a.scene == "opaque":             # bench.opaque_scene: trained by the reference schedule as it is, padded to --points
        if have:
            pack = torch.load(a.model_file, weights_only=False)
            cams = bench.opaque_views(args, dev, bg)[0]
            opt = pack["opt"]
            model = GaussianModel(3, device=dev)
            model.restore(pack["model"], opt)
            it = pack["it"]
        else:
            _, model, opt, cams, it = bench.opaque_scene(args, dev, bg, bench._progress, return_model=True)
            if a.model_file:
                torch.save({"model": model.capture(), "opt": opt, "it": it}, a.model_file)
        opt.iterations, opt.densify_until_iter = 10 ** 9, 10 ** 9
    elif a.scene == "densified":
        if have:
            pack = torch.load(a.model_file, weights_only=False)
            cams = bench.densified_views(args, dev, bg)[0]
            opt = pack["opt"]
            model = GaussianModel(3, device=dev)
            model.restore(pack["model"], opt)
            it = pack["it"]
        else:
            model, opt, cams, _, _ = bench.grow_densified_model(args, dev, bg, log=bench._progress)
            it = opt.iterations
            if a.model_file:
                torch.save({"model": model.capture(), "opt": opt, "it": it}, a.model_file)
        opt.iterations, opt.densify_until_iter = 10 ** 9, 10 ** 9
    else:
        sc, model, opt, cams = bench.build_scene(args, dev)
        bench.make_ground_truth(args, cams, dev, bg)
        it = 0
        if a.scene == "trained":
            if have:
                pack = torch.load(a.model_file, weights_only=False)
                model = GaussianModel(3, device=dev)
                model.restore(pack["model"], OptimizationParams())
                it = pack["it"]
            else:
                tr0 = Trainer(model, cams, opt, bg, densify=False, spatial_order=not args.no_spatial_order)
                for _ in range(a.trained_steps):
                    it += 1
                    tr0.step(it)
                if a.model_file:
                    torch.save({"model": model.capture(), "it": it}, a.model_file)
    trainer = Trainer(model, cams, opt, bg, densify=False, spatial_order=not args.no_spatial_order)   # (bench.py's storage order)
    for _ in range(0 if (a.report and a.freeze) else a.warmup):      # (--freeze: the variant under test never trains the model)
        it += 1
        trainer.step(it)
    if a.bucket_stats:
        import json
        from w3d_amd.fused_step import render_raw
        from w3d_amd.rasterizer import debug_gaussian_records
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
        from depth_grid_model import grid_buckets, summary
        out = []
        for ci in (0, 7, 14, 21, 28, 35):
            cam = cams[ci % len(cams)]
            with torch.no_grad():
                pkg = render_raw(cam, model, bg, sync=True)
            rec = debug_gaussian_records(pkg["handle"])
            vis = pkg["radii"] > 0
            keys = rec[:, 11].contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
            keys[~vis] = 0xFFFFFFFF
            pop, widths, nbk = grid_buckets(keys.cpu().numpy())
            out.append({"camera": ci, "visible": int(vis.sum()), "buckets": nbk, **summary(pop, widths)})
        print(json.dumps({"scene": a.scene, "gaussians": model.num_points, "depth_buckets": out}))
        return
    if a.walk_stats:
        import heapq
        import json

        def makespan(jobs, slots):
            h = [0.0] * slots
            heapq.heapify(h)
            for j in sorted(jobs, reverse=True):
                heapq.heappush(h, heapq.heappop(h) + j)
            return max(h)
        out = []
        for ci in (0, 7, 14, 21, 28, 35):
            cam = cams[ci % len(cams)]
            w = cam.world_view_transform._w3d_tile_walk.cpu().numpy().astype(float) + 12.0     # (+ a wave's fixed cost in entries)
            T = len(w)
            per = (T + 7) // 8
            res = {"camera": ci, "tiles": T, "walk_sum": int(w.sum()), "mean": round(w.mean(), 1), "p99": int(sorted(w)[int(0.99 * T)]), "max": int(w.max())}
            bound = max(w.sum() / 4096.0, 1.0)
            for name, split in (("one_wave_per_tile", None), ("split_2.0", 2.0), ("split_1.5", 1.5), ("split_1.0", 1.0)):
                ms = 0.0
                for x in range(8):
                    jobs = []
                    for t in w[x * per:(x + 1) * per]:
                        if split is not None and t > split * bound:
                            jobs += [0.6 * t, 0.6 * t]
                        else:
                            jobs.append(t)
                    ms = max(ms, makespan(jobs, 512))
                res[name] = round(ms / bound, 3)
            out.append(res)
        print(json.dumps({"scene": a.scene, "gaussians": model.num_points, "schedule_over_divisible_bound": out}))
        return
    if a.report and a.freeze:
        import ctypes
        import json
        import time
        from w3d_amd import _lib
        from w3d_amd.fused import l1_ssim_fwd_bwd
        from w3d_amd.fused_step import backward_raw, finish, render_raw
        lib = _lib.lib
        lib.w3d_profile_enable.argtypes = [ctypes.c_char_p]
        lib.w3d_profile_collect.argtypes = [ctypes.c_char_p, ctypes.c_uint64]

        def one(i):
            cam = cams[i % len(cams)]
            with torch.no_grad():
                pkg = render_raw(cam, model, bg, sync=False)
                loss, dimg = l1_ssim_fwd_bwd(pkg["render"], cam.original_image, 0.2)
                backward_raw(model, pkg["handle"], dimg)
                finish(pkg["handle"])
        for i in range(40):
            one(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            one(i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        lib.w3d_profile_enable(b"*")
        for i in range(36):
            one(i)
        torch.cuda.synchronize()
        lib.w3d_profile_enable(None)
        buf = ctypes.create_string_buffer(1 << 16)
        lib.w3d_profile_collect(buf, len(buf))
        st = {ln.split()[0]: round(float(ln.split()[2]) / int(ln.split()[1]), 4) for ln in buf.value.decode().splitlines()}
        print(json.dumps({"scene": a.scene, "lib": os.environ.get("W3D_HIP_LIB", "product"), "gaussians": model.num_points,
                          "iters_per_s": round(a.steps / dt, 2), "ms_per_step": round(1e3 * dt / a.steps, 4), "stage_ms": st}))
        return
    if a.report:
        import json
        meter = bench.StepMeter(trainer, 1, dev)
        meas = meter.measure(a.steps, it, "auto", False, probe_steps=6, stage_steps=20)
        print(json.dumps({"scene": a.scene, "lib": os.environ.get("W3D_HIP_LIB", "product"), "gaussians": model.num_points,
                          "iters_per_s": round(a.steps / meas["elapsed"], 2), "ms_per_step": round(1e3 * meas["elapsed"] / a.steps, 4),
                          "stage_ms": meas["stage_ms"]}))
        return
    mark = torch.ones(1, device=dev)
    torch.cuda.synchronize()
    torch.lgamma(mark)                      # ---- window opens
    for _ in range(a.steps):
        it += 1
        trainer.step(it)
    torch.lgamma(mark)                      # ---- window closes
    torch.cuda.synchronize()
    print(f"scene_step: {a.scene}, {model.num_points} Gaussians, {a.steps} steps in the window", file=sys.stderr)


if __name__ == "__main__":
    main()
