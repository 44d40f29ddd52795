"""Differential fuzzing of the backward against the CPU oracle (round 4): random small scenes of well-conditioned shapes (blobs,
mild ellipsoids), random SH degree, background, frame size and camera, with and without depth / alpha image gradients, the
deterministic and the atomic backward alternating, culls on and off alternating.  Every gradient block within 2e-4 of its maximum
(the bar of tests/test_gpu_parity.py), gradients of culled Gaussians exactly zero.
  usage: python3 profiles/fuzz_grads_probe.py [cases] [seed0]"""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "wheat-3dgs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from w3d_amd.synth import make_scene, make_cameras
from test_gpu_parity import _settings
from util import view_inputs, make_oracle, np_inputs

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda:0")
bad = 0
explained = 0
worst = {}
for case in range(cases):
    rs = np.random.RandomState(seed0 + case)
    g = torch.Generator().manual_seed(seed0 + case)
    P = int(rs.choice([1, 7, 64, 300, 1500]))
    W, H = int(rs.randint(8, 260)), int(rs.randint(8, 200))
    sc = make_scene(P, seed=seed0 + case, scale_mean=float(rs.choice([0.01, 0.03, 0.1])))
    if rs.rand() < 0.5:      # mild ellipsoids: axis ratios up to 8
        sc.scaling[:] = sc.scaling[:, :1] + torch.empty(P, 3).uniform_(-1.0, 1.0, generator=g)
    sc.opacity[:] = torch.empty(P, 1).normal_(float(rs.choice([-3.0, 0.0, 3.0])), float(rs.choice([0.5, 2.0])), generator=g)
    cam = make_cameras(5, W, H)[int(rs.randint(5))]
    bg = tuple(float(x) for x in rs.choice([0.0, 0.3], 3))
    deg = int(rs.randint(4))
    cull, det, da = bool(case & 1), bool(case & 2), bool(case & 4)
    d = view_inputs(sc, cam, sh_degree=deg)
    o = make_oracle(cam, bg, sh_degree=deg, nthreads=8)
    ref = o.forward(**np_inputs(d))
    gc = rs.randn(3, H, W).astype(np.float32)
    gd = rs.randn(1, H, W).astype(np.float32) if da else None
    ga = rs.randn(1, H, W).astype(np.float32) if da else None
    gref = o.backward(gc, gd, ga)
    from diff_gaussian_rasterization import GaussianRasterizer
    t = {k: (None if v is None else v.to(dev).requires_grad_(True)) for k, v in d.items()}
    means2D = torch.zeros_like(t["means3D"], requires_grad=True)
    st = _settings(cam, bg, deg, 1.0, dev, tile_cull=cull)._replace(deterministic=det)
    color, radii, depth, alpha = GaussianRasterizer(raster_settings=st)(
        means3D=t["means3D"], means2D=means2D, shs=t["shs"], colors_precomp=None, opacities=t["opacities"],
        scales=t["scales"], rotations=t["rotations"], cov3D_precomp=None)
    loss = (color * torch.from_numpy(gc).to(dev)).sum()
    if da:
        loss = loss + (depth * torch.from_numpy(gd).to(dev)).sum() + (alpha * torch.from_numpy(ga).to(dev)).sum()
    loss.backward()
    got = {k: (None if v is None or v.grad is None else v.grad.cpu().numpy()) for k, v in t.items()}
    got["means2D"] = means2D.grad.cpu().numpy()
    vis = ref["radii"] > 0
    msgs = []
    if not np.array_equal(radii.cpu().numpy(), ref["radii"]):
        msgs.append("radii differ")
    for k, r in gref.items():
        if r is None or got.get(k) is None:
            continue
        a, b = np.asarray(got[k], np.float64).reshape(np.asarray(r).shape), np.asarray(r, np.float64)
        if not np.all(a.reshape(P, -1)[~vis] == 0):
            msgs.append(f"grad {k}: non-zero on a culled Gaussian")
        e = np.abs(a - b).max() / (np.abs(b).max() + 1e-30)
        worst[k] = max(worst.get(k, 0.0), float(e))
        if e > 2e-4:
            msgs.append(f"grad {k}: rel {e:.2e}")
    if msgs:
        # a pixel whose pair sits ON a threshold (alpha 1/255, T 1e-4) flips its contributor set with the last bit of the exponent
        # and moves one Gaussian's gradient by a pixel's worth: explained if every flipped pixel is one the oracle calls fragile
        dc = np.abs(color.detach().cpu().numpy() - ref["color"]).max(0) > 5e-5
        dc |= np.abs(alpha.detach().cpu().numpy()[0] - ref["alpha"][0]) > 5e-5
        frag = o.fragile_pixels(1e-3).reshape(H, W).astype(bool)
        n_flip, n_unexpl = int(dc.sum()), int((dc & ~frag).sum())
        msgs.append(f"[{n_flip} flipped pixels, {n_unexpl} of them not fragile]")
        if n_flip > 0 and n_unexpl == 0:
            explained += 1
    o.free()
    if msgs:
        bad += 1
        print(f"case {seed0 + case} (P={P}, {W}x{H}, deg {deg}, cull {cull}, det {det}, depth/alpha grads {da}): " + "; ".join(msgs), flush=True)
print(f"cases {cases} from seed {seed0}: {bad} with differences ({explained} of them explained by threshold flips); worst per block: " + ", ".join(f"{k} {v:.1e}" for k, v in worst.items()))
