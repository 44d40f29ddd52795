"""Differential fuzzing of the shared lists (round 5): the random scenes of fuzz_culls_probe.py — blobs, needles, pancakes, mixed;
random frame sizes (ragged list cells), cameras, backgrounds, opacities — rendered through the drop-in module with
list_share 0, 1 and 2 (tile_cull on, atomic backward).  Colour, depth, alpha, radii, final_T and the FlashSplat contributor counts
must be BIT-IDENTICAL in the three modes (a tile reads another list, it never blends anything else); for the well-conditioned
shapes the gradients agree to float-atomic noise (1e-3 of every block's maximum).
  usage: python3 profiles/fuzz_share_probe.py [cases] [seed0]"""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "wheat-3dgs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from w3d_amd.synth import make_scene, make_cameras
from test_gpu_parity import _settings
from util import view_inputs


def run(d, cam, bg, share, gc):
    from diff_gaussian_rasterization import GaussianRasterizer
    from flashsplat_rasterization import GaussianRasterizer as Flash
    from w3d_amd.rasterizer import debug_pixel_state
    dev = torch.device("cuda:0")
    t = {k: (None if v is None else v.to(dev).requires_grad_(True)) for k, v in d.items()}
    means2D = torch.zeros_like(t["means3D"], requires_grad=True)
    st = _settings(cam, bg, 3, 1.0, dev, tile_cull=True, list_share=share)
    color, radii, depth, alpha = GaussianRasterizer(raster_settings=st)(
        means3D=t["means3D"], means2D=means2D, shs=t["shs"], colors_precomp=None, opacities=t["opacities"],
        scales=t["scales"], rotations=t["rotations"], cov3D_precomp=None)
    saved = color.grad_fn.saved
    out = dict(color=color.detach().cpu().numpy(), radii=radii.cpu().numpy(), depth=depth.detach().cpu().numpy(),
               alpha=alpha.detach().cpu().numpy(), final_T=debug_pixel_state(saved)[0].cpu().numpy(), num_rendered=saved["num_rendered"])
    (color * gc.to(dev)).sum().backward()
    g = {k: v.grad.cpu().numpy() for k, v in t.items() if v is not None and v.grad is not None}
    g["means2D"] = means2D.grad.cpu().numpy()
    fs = _settings(cam, bg, 3, 1.0, dev, flash=1, tile_cull=True, list_share=share)
    with torch.no_grad():
        H, W = cam.image_height, cam.image_width
        mask = (torch.arange(W, device=dev)[None, :] + torch.arange(H, device=dev)[:, None]) % 7 < 3
        fo = Flash(fs)(means3D=t["means3D"].detach(), means2D=None, gt_mask=mask.float(), shs=t["shs"].detach(), colors_precomp=None,
                       opacities=t["opacities"].detach(), scales=t["scales"].detach(), rotations=t["rotations"].detach(), cov3D_precomp=None)
    out["contrib_num"] = fo[4].cpu().numpy()
    out["used_count"] = fo[5].cpu().numpy()
    return out, g


cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
shrink = []
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
    a, ga = run(d, cam, bg, 0, gc)
    msgs = []
    for share in (1, 2):
        b, gb = run(d, cam, bg, share, gc)
        if a["num_rendered"]:
            shrink.append(b["num_rendered"] / a["num_rendered"])
        for k in ("color", "depth", "alpha", "radii", "final_T", "contrib_num"):
            if not np.array_equal(a[k], b[k]):
                msgs.append(f"share {share} {k}: {int((a[k] != b[k]).sum())} elements differ, max {np.abs(a[k].astype(np.float64) - b[k]).max():.2e}")
        uc = np.abs(a["used_count"] - b["used_count"]).max() / (np.abs(a["used_count"]).max() + 1e-30)
        if uc > 1e-5:
            msgs.append(f"share {share} used_count rel {uc:.2e}")
        if kind == "blob":
            for k, ra in ga.items():
                rb = gb[k]
                fin = np.isfinite(ra) & np.isfinite(rb)
                e = np.abs(ra[fin].astype(np.float64) - rb[fin]).max() / (np.abs(ra[fin]).max() + 1e-30) if fin.any() else 0.0
                if e > 1e-3:
                    msgs.append(f"share {share} grad {k}: rel {e:.2e}")
    if msgs:
        bad += 1
        print(f"case {seed0 + case} ({kind}, P={P}, {W}x{H}, spread {spread}): " + "; ".join(msgs), flush=True)
print(f"cases {cases} from seed {seed0}: {bad} with differences; mean list length vs one list per tile {np.mean(shrink) if shrink else 1.0:.3f}")
