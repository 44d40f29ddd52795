#!/usr/bin/env python
"""Is the forward-only loop (render Mpix/s) bound by the host or by the GPU?  Enqueues N raw-parameter forwards back to
back without waiting for their counters and reports (a) the host time to enqueue one frame, (b) the GPU time per frame with
1 .. 4 frames in flight on separate HIP streams.
    python profiles/render_host_probe.py > gpurun_out/render_host_probe.json
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "wheat-3dgs_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    from w3d_amd.synth import make_scene, make_cameras
    from w3d_amd.gaussian_model import GaussianModel
    from w3d_amd.fused_step import render_raw, finish
    import w3d_amd.train as T
    dev = torch.device("cuda:0")
    W, H, P = 1600, 1200, 2_000_000
    cams = [c.to(dev) for c in make_cameras(36, W, H)]
    sc = make_scene(P, seed=1)
    m = GaussianModel(3, device=dev)
    m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
    m.active_sh_degree = 3
    bg = torch.zeros(3, device=dev)
    out = {}
    with torch.no_grad():
        for c in cams:                                   # warm: capacities, per-camera hints, allocator
            finish(render_raw(c, m, bg, sync=False)["handle"])
        torch.cuda.synchronize()
        n = 72
        t0 = time.perf_counter()
        keep = [render_raw(cams[i % 36], m, bg, sync=False) for i in range(n)]
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        out["enqueue_only"] = {"host_ms_per_frame": round((t1 - t0) / n * 1e3, 4), "total_ms_per_frame": round((t2 - t0) / n * 1e3, 4)}
        del keep
        for ns in (1, 2, 3, 4):
            T.RENDER_STREAMS = ns
            T.render_views(m, cams[:4], bg)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                T.render_views(m, [cams[i % 36] for i in range(n)], bg)
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / n)
            out[f"render_views_{ns}_streams"] = {"ms_per_frame": round(best * 1e3, 4), "mpix_per_s": round(W * H / 1e6 / best, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
