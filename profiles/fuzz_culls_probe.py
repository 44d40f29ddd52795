"""Differential fuzzing of the culls (round 4): random scenes — blobs, needles, pancakes, huge and tiny Gaussians, opacities
from 1/255 to saturated, centres inside / outside / behind the frame, random frame sizes and cameras — rendered with tile_cull
off and on through the drop-in module, backward in the DETERMINISTIC mode (per-Gaussian sums over its tiles in tile order: a
dropped tile only removes an exact zero from the sum).  Images AND gradients must be bit-identical; no oracle involved, so
hundreds of cases run in seconds.  (With the atomic backward the geometry gradients of needles differ by tens of percent
between ANY two runs — their chain through the conic cancels to the last bits of the sums — which is not a cull effect.)
  usage: python3 profiles/fuzz_culls_probe.py [cases] [seed0]"""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "wheat-3dgs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from w3d_amd.synth import make_scene, make_cameras
from test_gpu_parity import _settings
from util import view_inputs



def run(d, cam, bg, cull, grads):
    from diff_gaussian_rasterization import GaussianRasterizer
    dev = torch.device("cuda:0")
    t = {k: (None if v is None else v.to(dev).requires_grad_(True)) for k, v in d.items()}
    means2D = torch.zeros_like(t["means3D"], requires_grad=True)
    st = _settings(cam, bg, 3, 1.0, dev, tile_cull=cull)._replace(deterministic=True)
    color, radii, depth, alpha = GaussianRasterizer(raster_settings=st)(
        means3D=t["means3D"], means2D=means2D, shs=t["shs"], colors_precomp=None, opacities=t["opacities"],
        scales=t["scales"], rotations=t["rotations"], cov3D_precomp=None)
    saved = color.grad_fn.saved
    out = dict(color=color.detach().cpu().numpy(), radii=radii.cpu().numpy(), depth=depth.detach().cpu().numpy(),
               alpha=alpha.detach().cpu().numpy(), num_rendered=saved["num_rendered"])
    gc, gd, ga = grads
    loss = (color * gc.to(dev)).sum()
    if gd is not None:
        loss = loss + (depth * gd.to(dev)).sum() + (alpha * ga.to(dev)).sum()
    loss.backward()
    g = {k: (None if v is None or v.grad is None else v.grad.cpu().numpy()) for k, v in t.items()}
    g["means2D"] = means2D.grad.cpu().numpy()
    return out, g


cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
for case in range(cases):
    rs = np.random.RandomState(seed0 + case)
    g = torch.Generator().manual_seed(seed0 + case)
    P = int(rs.choice([1, 7, 64, 300, 2000, 8000]))
    W, H = int(rs.randint(8, 700)), int(rs.randint(8, 500))
    sc = make_scene(P, seed=seed0 + case, scale_mean=float(rs.choice([0.002, 0.02, 0.2])))
    kind = rs.choice(["blob", "needle", "pancake", "mixed"])
    lo, hi = math.log(1e-5), math.log(6.0)
    if kind == "needle":
        sc.scaling[:, 0] = torch.empty(P).uniform_(math.log(0.05), hi, generator=g)
        sc.scaling[:, 1:] = torch.empty(P, 2).uniform_(lo, math.log(2e-3), generator=g)
    elif kind == "pancake":
        sc.scaling[:, :2] = torch.empty(P, 2).uniform_(math.log(0.05), math.log(3.0), generator=g)
        sc.scaling[:, 2] = torch.empty(P).uniform_(lo, math.log(1e-3), generator=g)
    elif kind == "mixed":
        sc.scaling[:] = torch.empty(P, 3).uniform_(lo, hi, generator=g)
    spread = float(rs.choice([1.0, 3.0, 10.0]))
    sc.xyz[:, :2] *= spread
    sc.xyz[:, 2] += float(rs.choice([0.0, 1.0, -2.0])) * torch.rand(P, generator=g)
    sc.opacity[:] = torch.empty(P, 1).normal_(float(rs.choice([-4.0, 0.0, 4.0])), float(rs.choice([0.5, 3.0])), generator=g)
    cam = make_cameras(5, W, H)[int(rs.randint(5))]
    bg = tuple(float(x) for x in rs.choice([0.0, 0.3], 3))
    d = view_inputs(sc, cam)
    gc = torch.from_numpy(rs.randn(3, H, W).astype(np.float32))
    gd = torch.from_numpy(rs.randn(1, H, W).astype(np.float32)) if case % 2 else None
    ga = torch.from_numpy(rs.randn(1, H, W).astype(np.float32)) if case % 2 else None
    a, ga_ = run(d, cam, bg, False, (gc, gd, ga))
    b, gb_ = run(d, cam, bg, True, (gc, gd, ga))
    msgs = []
    for k in ("color", "depth", "alpha", "radii"):
        if not np.array_equal(a[k], b[k]):
            msgs.append(f"{k}: {int((a[k] != b[k]).sum())} elements differ, max {np.abs(a[k].astype(np.float64) - b[k]).max():.2e}")
    for k, ra in ga_.items():
        if ra is None or gb_.get(k) is None:
            continue
        ra, rb = np.asarray(ra), np.asarray(gb_[k])
        if not np.array_equal(ra, rb, equal_nan=True):
            fin = np.isfinite(ra) & np.isfinite(rb)
            e = np.abs(ra[fin].astype(np.float64) - rb[fin]).max() / (np.abs(ra[fin]).max() + 1e-30) if fin.any() else float("nan")
            msgs.append(f"grad {k}: {int((ra != rb).sum())} elements differ, rel {e:.2e}")
    if msgs:
        bad += 1
        print(f"case {seed0 + case} ({kind}, P={P}, {W}x{H}, spread {spread}): " + "; ".join(msgs), flush=True)
print(f"cases {cases} from seed {seed0}: {bad} with differences; last R {a['num_rendered']} -> {b['num_rendered']}")
