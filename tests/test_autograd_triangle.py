"""The per-Gaussian stages (EWA projection, conic, SH colour and their backward: SURVEY.md rows A3 / A7) against something
that is NOT their own text: `oracle.torch_render`, a float64 PyTorch restatement whose backward is torch.autograd — no
hand-written derivative anywhere.  The HIP kernels and the C oracle both implement the published explicit backward; this
closes the triangle  HIP <-> C oracle <-> autograd  at 5 000 Gaussians / 160x120 on a scene that contains what the small
consistency test (tests/test_oracle_consistency.py, 150 Gaussians) leaves out: Gaussians whose view-space position is
FoV-CLAMPED (|x/z| > 1.3 tan(fov/2): the clamp contributes no gradient) and Gaussians whose opacity exceeds the 0.99 ALPHA
CAP (the cap passes gradient straight through).  torch_render mirrors exactly those two conventions of the published
backward (its docstring); everything else is plain autograd.

CPU (`-m "not gpu"`): the C oracle's explicit backward against autograd.  GPU: the HIP path — activated-parameter drop-in
module and raw-parameter kernels — against autograd.
"""
import math

import numpy as np
import pytest
import torch

from util import view_inputs, make_oracle, np_inputs
from w3d_amd.synth import make_scene, make_cameras

W, H, P = 160, 120, 5000


def special_scene(seed=17):
    """make_scene + 300 large Gaussians placed 1.35-1.7 half-widths off the optical axis (FoV-clamped, yet reaching into the
    image) + 200 Gaussians with opacity logits 5-8 (sigmoid > 0.993: above the 0.99 cap near their centres)."""
    sc = make_scene(P, seed=seed, scale_mean=0.03)
    cam = make_cameras(6, W, H)[2]
    g = torch.Generator().manual_seed(seed)
    tfx, tfy = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
    n_cl = 300
    z = 1.6 + 1.2 * torch.rand(n_cl, generator=g)
    u = (1.35 + 0.35 * torch.rand(n_cl, generator=g)) * torch.where(torch.rand(n_cl, generator=g) < 0.5, -1.0, 1.0)
    v = -0.7 + 1.4 * torch.rand(n_cl, generator=g)
    horiz = torch.rand(n_cl, generator=g) < 0.6
    xv = torch.where(horiz, u, v) * z * tfx
    yv = torch.where(horiz, v, u) * z * tfy
    pv = torch.stack([xv, yv, z, torch.ones(n_cl)], 1).double()
    pw = (pv @ torch.linalg.inv(cam.world_view_transform.double()))[:, :3].float()
    sc.xyz[:n_cl] = pw
    sc.scaling[:n_cl] = math.log(0.22) + 0.2 * torch.randn(n_cl, 3, generator=g)
    sc.opacity[:n_cl] = 0.5 * torch.randn(n_cl, 1, generator=g)
    sc.opacity[n_cl:n_cl + 200] = 5.0 + 3.0 * torch.rand(200, 1, generator=g)
    return sc, cam


def autograd_reference(d, cam, bg, gc, gd=None, ga=None):
    from oracle.oracle import torch_render
    dt = torch.float64
    t = {k: (None if v is None else v.to(dt).requires_grad_(True)) for k, v in d.items()}
    m2d = torch.zeros(d["means3D"].shape[0], 3, dtype=dt, requires_grad=True)
    c, r, dep, a = torch_render(cam.image_height, cam.image_width, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5),
                                torch.tensor(bg), cam.world_view_transform, cam.full_proj_transform, cam.camera_center,
                                means2D=m2d, sh_degree=3, scale_modifier=1.0, **t)
    loss = (c * torch.tensor(gc, dtype=dt)).sum()
    if gd is not None:
        loss = loss + (dep * torch.tensor(gd, dtype=dt)).sum() + (a * torch.tensor(ga, dtype=dt)).sum()
    loss.backward()
    g = {k: (None if v is None else v.grad.numpy()) for k, v in t.items()}
    g["means2D"] = m2d.grad.numpy()
    return dict(color=c.detach().numpy(), depth=dep.detach().numpy(), alpha=a.detach().numpy(), radii=r.numpy()), g


def clamp_and_cap_census(d, cam, radii):
    """How many visible Gaussians are FoV-clamped / above the alpha cap (the cases this test exists for)."""
    pv = torch.cat([d["means3D"].double(), torch.ones(len(radii), 1, dtype=torch.float64)], 1) @ cam.world_view_transform.double()
    tfx, tfy = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
    clamped = ((pv[:, 0] / pv[:, 2]).abs() > 1.3 * tfx) | ((pv[:, 1] / pv[:, 2]).abs() > 1.3 * tfy)
    vis = torch.from_numpy(np.asarray(radii) > 0)
    return int((clamped & vis).sum()), int(((d["opacities"].reshape(-1) > 0.99) & vis).sum())


def per_gaussian_rel(got, ref):
    got, ref = np.asarray(got, np.float64).reshape(len(ref), -1), np.asarray(ref, np.float64).reshape(len(ref), -1)
    mag = np.abs(ref).max(1)
    nz = mag > 0
    floor = 1e-3 * np.median(mag[nz])
    return (np.abs(got - ref).max(1) / (mag + floor))[nz]


def check_against_autograd(out, g, ref, gref, tag, marker):
    """Images: all but a handful of pixels within 2e-5.  Gradients per Gaussian, relative to that Gaussian's own gradient.
    fp32 against float64 cannot agree on every threshold decision: where a pixel's transmittance differs by more than 0.1 %
    (one (pixel, Gaussian) pair fell on the other side of alpha >= 1/255 or T < 1e-4: the smallest possible change of a
    contributor set is 0.39 %), every Gaussian blended at that pixel is set aside (`marker` = the C oracle's
    contributors_of on the same inputs — bookkeeping, no formula) and only counted.  Over all the others: median <= 1e-5,
    p99 <= 1e-4 (the north_star bar), p99.9 <= 1e-3, nothing beyond 1e-2 — fp32 terms summed with random-sign image
    gradients against float64."""
    np.testing.assert_array_equal(out["radii"], ref["radii"])
    for k in ("color", "depth", "alpha"):
        diff = np.abs(out[k] - ref[k]) / max(1.0, float(np.abs(ref[k]).max()))
        assert (diff > 2e-5).mean() <= 2e-3, f"{tag}{k}: {(diff > 2e-5).mean():.2e} of the pixels differ by more than 2e-5"
        assert np.median(diff) <= 1e-6, f"{tag}{k}: median difference {np.median(diff):.2e}"
    # (the fp32 side's own final transmittance, not 1 - alpha: that difference has no digits left where T ~ 1e-4)
    T_ref, T_own = 1.0 - ref["alpha"][0].astype(np.float64), np.asarray(out["final_T"], np.float64).reshape(T_shape(ref))
    flipped = np.abs(T_own - T_ref) > 1e-3 * np.maximum(T_ref, 1e-5)
    assert flipped.mean() <= 1e-3, f"{tag}{int(flipped.sum())} pixels with a different contributor set"
    set_aside = marker(flipped)
    assert set_aside.mean() <= 0.05, f"{tag}{int(set_aside.sum())} Gaussians touch a flipped pixel"
    stats = {}
    for k, want in gref.items():
        if want is None:
            continue
        got = np.asarray(g[k], np.float64).reshape(len(want), -1)
        w = np.asarray(want, np.float64).reshape(len(want), -1)
        mag = np.abs(w).max(1)
        nz = mag > 0
        e = np.abs(got - w).max(1) / (mag + 1e-3 * np.median(mag[nz]))
        assert np.all(got[~nz] == 0) or k in ("means3D", "xyz"), f"{tag}grad {k}: gradient where autograd has none"
        c = e[nz & ~set_aside]
        stats[k] = dict(n=int(c.size), p50=float(np.median(c)), p99=float(np.percentile(c, 99)), p999=float(np.percentile(c, 99.9)),
                        max=float(c.max()), set_aside=int((nz & set_aside).sum()))
        assert stats[k]["p50"] <= 1e-5, f"{tag}grad {k}: median per-Gaussian rel err {stats[k]['p50']:.2e}"
        assert stats[k]["p99"] <= 1e-4, f"{tag}grad {k}: p99 per-Gaussian rel err {stats[k]['p99']:.2e}"
        assert stats[k]["p999"] <= 1e-3, f"{tag}grad {k}: p99.9 per-Gaussian rel err {stats[k]['p999']:.2e}"
        assert stats[k]["max"] <= 1e-2, f"{tag}grad {k}: worst per-Gaussian rel err {stats[k]['max']:.2e}"
    return stats


def T_shape(ref):
    return ref["alpha"][0].shape


def _marker(cam, d, bg):
    """contributors_of of a C-oracle forward on the same inputs (which Gaussians are blended at a given set of pixels)."""
    o = make_oracle(cam, bg, nthreads=8)
    o.forward(**np_inputs(d))
    return o.contributors_of


def _setup():
    sc, cam = special_scene()
    d = view_inputs(sc, cam)
    # image gradients shaped like a photometric loss's: spatially correlated (3-px blur of white noise, unit variance) plus
    # 10 % white noise — pure white noise makes every per-Gaussian sum a random walk of cancelling terms, which measures
    # fp32 summation noise against float64 rather than the formulas
    from scipy.ndimage import gaussian_filter
    rng = np.random.RandomState(5)

    def field(c, amp):
        f = np.stack([gaussian_filter(rng.randn(H, W), 3.0) for _ in range(c)])
        return (amp * (f / f.std() + 0.1 * rng.randn(c, H, W))).astype(np.float32)
    return sc, cam, d, field(3, 1.0), field(1, 0.1), field(1, 0.1)


_ref_cache = {}


def _reference(with_da):
    if with_da not in _ref_cache:
        sc, cam, d, gc, gd, ga = _setup()
        _ref_cache[with_da] = autograd_reference(d, cam, (0.2, 0.1, 0.3), gc, gd if with_da else None, ga if with_da else None)
    return _ref_cache[with_da]


@pytest.mark.parametrize("with_da", [False, True])
def test_c_oracle_explicit_backward_equals_autograd_on_clamped_and_capped_gaussians(with_da):
    sc, cam, d, gc, gd, ga = _setup()
    bg = (0.2, 0.1, 0.3)
    ref, gref = _reference(with_da)
    n_clamped, n_capped = clamp_and_cap_census(d, cam, ref["radii"])
    assert n_clamped >= 100 and n_capped >= 100, (n_clamped, n_capped)
    o = make_oracle(cam, bg, nthreads=8)
    out = o.forward(**np_inputs(d))
    g = o.backward(gc, gd if with_da else None, ga if with_da else None)
    g["opacities"] = g["opacities"].reshape(-1)
    out["final_T"] = o.pixel_state()[0]
    check_against_autograd(out, g, ref, gref, "[C oracle vs autograd] ", o.contributors_of)
    o.free()


@pytest.mark.gpu
@pytest.mark.parametrize("with_da", [False, True])
def test_hip_dropin_module_equals_autograd(with_da):
    """diff_gaussian_rasterization.GaussianRasterizer (activated inputs) forward + backward against float64 autograd."""
    from test_gpu_parity import run_hip
    sc, cam, d, gc, gd, ga = _setup()
    ref, gref = _reference(with_da)
    out, g = run_hip(d, cam, (0.2, 0.1, 0.3), grads=(gc, gd if with_da else None, ga if with_da else None))
    g["opacities"] = g["opacities"].reshape(-1)
    check_against_autograd(out, g, ref, gref, "[HIP drop-in vs autograd] ", _marker(cam, d, (0.2, 0.1, 0.3)))


@pytest.mark.gpu
def test_hip_raw_parameter_kernels_equal_autograd_through_the_activations():
    """The raw-parameter kernels (exp / sigmoid / normalize chained inside) against autograd through torch's own
    activations in float64 — the path bench.py times."""
    from oracle.oracle import torch_render
    from w3d_amd.fused_step import render_raw, backward_raw
    from w3d_amd.gaussian_model import GaussianModel
    sc, cam, d_act, gc, gd, ga = _setup()
    dev = torch.device("cuda:0")
    dt = torch.float64
    raw = {k: getattr(sc, k).to(dt).requires_grad_(True) for k in ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation")}
    m2d = torch.zeros(P, 3, dtype=dt, requires_grad=True)
    bg = (0.2, 0.1, 0.3)
    c, r, dep, a = torch_render(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.tensor(bg), cam.world_view_transform,
                                cam.full_proj_transform, cam.camera_center, means3D=raw["xyz"],
                                opacities=torch.sigmoid(raw["opacity"]), means2D=m2d,
                                shs=torch.cat([raw["features_dc"], raw["features_rest"]], 1), scales=torch.exp(raw["scaling"]),
                                rotations=torch.nn.functional.normalize(raw["rotation"]), sh_degree=3)
    ((c * torch.tensor(gc, dtype=dt)).sum() + (dep * torch.tensor(gd, dtype=dt)).sum() + (a * torch.tensor(ga, dtype=dt)).sum()).backward()
    m = GaussianModel(3, device=dev)
    m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
    m.active_sh_degree = 3
    pkg = render_raw(cam.to(dev), m, torch.tensor(bg, device=dev), sync=True)
    _, hip_m2d = backward_raw(m, pkg["handle"], torch.as_tensor(gc, device=dev), torch.as_tensor(gd, device=dev),
                              torch.as_tensor(ga, device=dev), want_means2D=True)
    radii = pkg["radii"].cpu().numpy()
    assert (radii != r.numpy()).sum() <= 2 and np.array_equal(radii > 0, r.numpy() > 0)
    from w3d_amd.rasterizer import debug_pixel_state
    out = dict(color=pkg["render"].cpu().numpy(), depth=pkg["depth"].cpu().numpy(), alpha=pkg["alpha"].cpu().numpy(), radii=r.numpy(),
               final_T=debug_pixel_state(pkg["handle"])[0].cpu().numpy())
    ref = dict(color=c.detach().numpy(), depth=dep.detach().numpy(), alpha=a.detach().numpy(), radii=r.numpy())
    names = dict(xyz="xyz", f_dc="features_dc", f_rest="features_rest", opacity="opacity", scaling="scaling", rotation="rotation")
    g = {k: m.grad_view(k).detach().cpu().numpy() for k in names}
    gref = {k: raw[v].grad.numpy() for k, v in names.items()}
    g["means2D"], gref["means2D"] = hip_m2d.cpu().numpy(), m2d.grad.numpy()
    check_against_autograd(out, g, ref, gref, "[HIP raw vs autograd] ", _marker(cam, d_act, bg))
