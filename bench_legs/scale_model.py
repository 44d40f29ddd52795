"""scale-model leg (--full): what an N-GPU view-parallel run should do, predicted from single-GPU measurements."""
import ctypes
import json
import math
import os
import subprocess
import sys
import time

import torch
import torch.distributed as dist

from . import common


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
    single = Trainer(model, cams, opt, bg, densify=False, spatial_order=common.SPATIAL_ORDER)
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
            tr = Trainer(model, cams, opt, bg, densify=False, force_exchange=True, exchange=mode, spatial_order=common.SPATIAL_ORDER)
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

