"""Is every cull exact?  (round 4)  The footprint culls — the rect shrunk to the ellipse's own extent, the per-tile mask, the
per-quadrant masks of the blend kernels — may only drop (Gaussian, tile / quadrant) instances that contribute NOTHING, under the
blend's own fp32 evaluation of the exponent.  Stress scene: 20 000 needles (one axis 0.05 ... 8 scene units, the others 1e-5 ...
2e-3: conic condition up to 1e7), many centred outside the frame.  The images with tile_cull on and off — and, when
W3D_HIP_LIB_NOQUAD names a library built with -DW3D_NO_QUADMASK, with and without the quadrant masks — must be IDENTICAL bit for bit.
  usage: python3 profiles/cull_exactness_probe.py [W H seeds]     (a child process renders with the other library)"""
import os, sys, math, subprocess, pickle
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "wheat-3dgs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch


from util import needle_scene


def render_all(W, H, seeds, culls):
    from w3d_amd.synth import make_cameras
    from test_gpu_parity import run_hip
    from util import view_inputs
    out = {}
    for seed in range(seeds):
        sc = needle_scene(seed)
        for ci, cam in enumerate(make_cameras(3, W, H)):
            d = view_inputs(sc, cam)
            for cull in culls:
                o, _ = run_hip(d, cam, (0.0, 0.0, 0.0), tile_cull=cull)
                out[(seed, ci, cull)] = {k: o[k] for k in ("color", "depth", "alpha", "radii", "num_rendered")}
    return out


def diff(a, b, tag):
    bad = 0
    for k in ("color", "depth", "alpha", "radii"):
        if not np.array_equal(a[k], b[k]):
            df = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64))
            bad += 1
            print(f"{tag} {k}: {int((df > 0).sum())} elements differ, max {df.max():.3e}")
    return bad


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        W, H, seeds = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
        pickle.dump(render_all(W, H, seeds, (False,)), open(sys.argv[5], "wb"))
        sys.exit(0)
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 800
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 608
    seeds = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    mine = render_all(W, H, seeds, (False, True))
    tot = 0
    for seed in range(seeds):
        for ci in range(3):
            a, b = mine[(seed, ci, False)], mine[(seed, ci, True)]
            tot += diff(a, b, f"seed {seed} cam {ci} cull on/off")
            print(f"seed {seed} cam {ci}: list entries without / with the culls {a['num_rendered']} / {b['num_rendered']}", flush=True)
    other = os.environ.get("W3D_HIP_LIB_NOQUAD")
    if other:
        tmp = os.path.join(ROOT, "gpurun_out", "_noquad.pkl")
        os.makedirs(os.path.dirname(tmp), exist_ok=True)
        env = dict(os.environ, W3D_HIP_LIB=other)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(W), str(H), str(seeds), tmp], env=env, check=True)
        theirs = pickle.load(open(tmp, "rb"))
        os.remove(tmp)
        for key, b in theirs.items():
            tot += diff(mine[key], b, f"seed {key[0]} cam {key[1]} quadrant masks on/off")
    print("TOTAL", tot)
