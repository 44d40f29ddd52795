#!/usr/bin/env python
"""Can an HBM-bound sweep hide under the VALU-bound blend backward?  Times, on the benchmark scene (2 M Gaussians, 1600x1200):
  a) the blend backward alone (w3d_backward_blend_dcolor: render_bwd + a 25-us extraction kernel),
  b) an Adam sweep over 0.39 x 59 x P elements alone (the optimizer traffic of the culled Gaussians of one view),
  c) a then b on one stream, d) a and b on two streams.
    python profiles/overlap_probe.py > gpurun_out/overlap_probe.json
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
    from w3d_amd.gaussian_model import GaussianModel, OptimizationParams
    from w3d_amd.fused_step import render_raw, finish, backward_blend_dcolor
    import w3d_amd.fused  # noqa: F401  (sets the argtypes of w3d_adam_step)
    from w3d_amd._lib import check, lib, ptr, stream_ptr
    dev = torch.device("cuda:0")
    W, H, P = 1600, 1200, 2_000_000
    trained = len(sys.argv) > 1 and sys.argv[1] == "lowopacity"
    cams = [c.to(dev) for c in make_cameras(36, W, H)]
    sc = make_scene(P, seed=1)
    if trained:
        sc.opacity.fill_(-2.0)            # low opacities: long walks, as on a trained scene
    m = GaussianModel(3, device=dev)
    m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
    m.active_sh_degree = 3
    m.training_setup(OptimizationParams())
    bg = torch.zeros(3, device=dev)
    n = int(0.39 * 59 * P) // 4 * 4
    bufs = [torch.randn(n, device=dev) for _ in range(4)]
    bufs[3].abs_()
    side = torch.cuda.Stream(device=dev)
    out = {"elements_in_sweep": n}
    with torch.no_grad():
        pkg = render_raw(cams[3], m, bg, sync=True)
        dimg = torch.randn(3, H, W, device=dev) * 1e-3

        def blend():
            backward_blend_dcolor(m, pkg["handle"], dimg)

        def sweep():
            check(lib.w3d_adam_step(n, ptr(bufs[0]), ptr(bufs[1]), ptr(bufs[2]), ptr(bufs[3]), 1e-4, 0.9, 0.999, 1e-15, 0.5, 0.5, 0,
                                    stream_ptr(dev)))

        def timed(fn, reps=20):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return round((time.perf_counter() - t0) / reps * 1e3, 4)

        def both_serial():
            blend()
            sweep()

        pool = [torch.cuda.Stream(device=dev) for _ in range(6)]     # consecutive streams of torch's pool

        def overlapped(sa, sb):
            """blend on stream sa, sweep on stream sb (None = the current stream), both released together"""
            cur = torch.cuda.current_stream(dev)
            for st in (sa, sb):
                if st is not None:
                    st.wait_stream(cur)
            with torch.cuda.stream(sa if sa is not None else cur):
                blend()
            with torch.cuda.stream(sb if sb is not None else cur):
                sweep()
            for st in (sa, sb):
                if st is not None:
                    cur.wait_stream(st)
        out["blend_backward_ms"] = timed(blend)
        out["sweep_ms"] = timed(sweep)
        out["serial_ms"] = timed(both_serial)
        # HIP streams share a handful of hardware queues: whether two of them overlap depends on WHICH two (profiles/README.md)
        out["two_streams_ms"] = {}
        for name, (ia, ib) in {"current+pool0": (None, 0), "current+pool1": (None, 1), "current+pool2": (None, 2), "current+pool3": (None, 3),
                               "pool0+pool1": (0, 1), "pool1+pool2": (1, 2), "pool2+pool3": (2, 3), "pool0+pool2": (0, 2),
                               "pool0+pool3": (0, 3), "pool4+pool5": (4, 5)}.items():
            sa = None if ia is None else pool[ia]
            out["two_streams_ms"][name] = timed(lambda: overlapped(sa, pool[ib]))
        out["scene"] = "low opacity (long walks)" if trained else "benchmark scene"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
