#!/usr/bin/env python
"""One view of a scene whose flat parameter buffer is LARGER THAN 2^31 floats (default 40 M Gaussians x 59 = 2.36e9) against the
CPU oracle: every index computation of the kernels beyond 32 bits, list buffers of 10^8 entries, the compaction pass and the
optimizer sweep at that size.  Needs ~60 GB of HBM and ~80 GB of host memory (refuses to start with less); the oracle takes a
few minutes on all host cores.
    python3 profiles/big_scene_probe.py [P] [--no-oracle]      -> one JSON line"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "wheat-3dgs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def mem_available_gb():
    for ln in open("/proc/meminfo"):
        if ln.startswith("MemAvailable"):
            return int(ln.split()[1]) / 1e6
    return 0.0


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 40_000_000
    with_oracle = "--no-oracle" not in sys.argv
    need = 2.0 * P * 236 / 1e9 * (4 if with_oracle else 1.5)
    if mem_available_gb() < need:
        print(json.dumps({"skipped": f"host memory {mem_available_gb():.0f} GB < {need:.0f} GB"}))
        return
    from w3d_amd.fused_step import backward_raw, render_raw
    from w3d_amd.gaussian_model import GaussianModel, OptimizationParams
    from w3d_amd.synth import make_cameras, make_scene
    from util import gradient_stats, make_oracle, np_inputs, psnr, raw_grads_from_oracle, view_inputs
    W, H = 1600, 1200
    dev = torch.device("cuda:0")
    t0 = time.perf_counter()
    # (the benchmark's slab with 20x the Gaussians: scale them down so that the lists stay of the benchmark's order)
    sc = make_scene(P, seed=0, scale_mean=0.006 * (2_000_000 / P) ** (1 / 3))
    cam = make_cameras(36, W, H)[0]
    out = {"P": P, "flat_floats": 59 * P, "beyond_2^31_floats": 59 * P > 2 ** 31, "scene_seconds": round(time.perf_counter() - t0, 1)}
    m = GaussianModel(3, device=dev)
    m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
    m.active_sh_degree = 3
    opt = OptimizationParams()
    m.training_setup(opt)
    bg = torch.zeros(3, device=dev)
    gc = np.random.RandomState(3).randn(3, H, W).astype(np.float32)
    camd = cam.to(dev)
    with torch.no_grad():
        pkg = render_raw(camd, m, bg, sync=True)
        gnorm, _ = backward_raw(m, pkg["handle"], torch.as_tensor(gc, device=dev), want_norm=True)
    torch.cuda.synchronize()
    own = dict(color=pkg["render"].cpu().numpy(), depth=pkg["depth"].cpu().numpy(), alpha=pkg["alpha"].cpu().numpy())
    radii = pkg["radii"].cpu().numpy()
    got = {k: m.grad_view(k).detach().cpu().numpy().copy() for k in ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")}
    out.update(visible=int((radii > 0).sum()), num_rendered=int(pkg["handle"]["num_rendered"]),
               hbm_allocated_gb=round(torch.cuda.max_memory_allocated() / 1e9, 1))
    del pkg
    m.flat_grad.zero_()
    if with_oracle:
        t0 = time.perf_counter()
        d = np_inputs(view_inputs(sc, cam))
        o = make_oracle(cam, (0.0, 0.0, 0.0), nthreads=os.cpu_count() or 1)
        ref = o.forward(**d)
        t1 = time.perf_counter()
        gref = o.backward(gc, None, None)
        t2 = time.perf_counter()
        o.free()
        out["oracle_seconds"] = {"inputs": round(t1 - t0, 1), "backward": round(t2 - t1, 1)}
        vis = ref["radii"] > 0
        bad = radii != ref["radii"]
        out["radii"] = {"differing": int(bad.sum()), "max_abs": int(np.abs(radii[bad] - ref["radii"][bad]).max(initial=0)),
                        "visibility_same": bool(np.array_equal(radii > 0, vis))}
        imgs = {}
        for k in ("color", "depth", "alpha"):
            a, b = own[k], ref[k]
            scale = max(1.0, float(np.abs(b).max()))
            diff = np.abs(a - b)
            imgs[k] = {"max": float(diff.max()), "frac_gt_2e-4": float((diff > 2e-4 * scale).mean()), "psnr_db": round(psnr(a / scale, b / scale), 1)}
        out["images"] = imgs
        want = raw_grads_from_oracle(gref, sc)
        out["grads"] = {k: {kk: (float(vv) if isinstance(vv, (float, np.floating)) else vv) for kk, vv in st.items()}
                        for k, st in gradient_stats(got, want, vis).items()}
        n_ref = np.linalg.norm(gref["means2D"][:, :2].astype(np.float64), axis=1)
        n_own = gnorm.cpu().numpy().astype(np.float64)
        sel = vis & (n_ref > 0)
        e = np.abs(n_own[sel] - n_ref[sel]) / n_ref[sel]
        out["densify_norm"] = {"n": int(sel.sum()), "p50": float(np.quantile(e, 0.5)), "p99": float(np.quantile(e, 0.99)),
                               "p999": float(np.quantile(e, 0.999)), "culled_zero": bool(np.all(n_own[~vis] == 0))}
        del d, ref, gref, want
    # the rest of the step at this size: the Trainer's fused step (loss, backward + Adam in one kernel), a re-sort (the
    # densification compaction pass over parameters and both moments), a step again
    from w3d_amd.train import Trainer
    torch.cuda.synchronize()
    camd.original_image = torch.rand(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
    tr = Trainer(m, [camd], opt, bg, densify=False, spatial_order=False)
    p0 = m.flat[:: 1 << 20].clone()
    t0 = time.perf_counter()
    l1 = float(tr.step(1))
    torch.cuda.synchronize()
    out["fused_step_ms_first_call"] = round(1e3 * (time.perf_counter() - t0), 2)
    out["adam_moved_parameters"] = bool((m.flat[:: 1 << 20] != p0).any())
    xyz_before = m.get_xyz.detach().clone()
    t0 = time.perf_counter()
    perm = m.sort_spatially()
    torch.cuda.synchronize()
    out["sort_spatially_ms"] = round(1e3 * (time.perf_counter() - t0), 2)
    out["sort_moved_rows_consistently"] = bool(torch.equal(m.get_xyz.detach(), xyz_before[perm]))
    del xyz_before, perm
    tr.step(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(3, 8):
        l2 = float(tr.step(it))
    torch.cuda.synchronize()
    out["fused_step_ms"] = round(1e3 * (time.perf_counter() - t0) / 5, 2)
    out["loss_first_last"] = [round(l1, 6), round(l2, 6)]
    out["finite_after"] = bool(torch.isfinite(m.flat[:: 4097]).all() and torch.isfinite(tr.last["image"]).all())
    out["hbm_peak_gb"] = round(torch.cuda.max_memory_allocated() / 1e9, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
