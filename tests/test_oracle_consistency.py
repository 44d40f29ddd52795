"""CPU: the two oracle restatements against each other, plus invariants of the algorithm
(SURVEY.md §4 implication (2)).  The explicit backward of the C oracle — the formulas a kernel
implements — must equal torch.autograd through the differentiable restatement (float64)."""
import math

import numpy as np
import pytest
import torch

from oracle.oracle import COracle, torch_render, knn_dist2
from util import view_inputs, make_oracle, np_inputs, rel_err
from w3d_amd.synth import small_test_scene


def _autograd_reference(d, cam, bg, deg, mod, gc, gd, ga):
    dt = torch.float64
    t = {k: (None if v is None else v.to(dt).requires_grad_(True)) for k, v in d.items()}
    m2d = torch.zeros(d["means3D"].shape[0], 3, dtype=dt, requires_grad=True)
    c, r, dep, a = torch_render(cam.image_height, cam.image_width, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5),
                                torch.tensor(bg), cam.world_view_transform, cam.full_proj_transform,
                                cam.camera_center, means2D=m2d, sh_degree=deg, scale_modifier=mod, **t)
    loss = (c * torch.tensor(gc, dtype=dt)).sum()
    if gd is not None:
        loss = loss + (dep * torch.tensor(gd, dtype=dt)).sum() + (a * torch.tensor(ga, dtype=dt)).sum()
    loss.backward()
    g = {k: (None if v is None else v.grad.numpy()) for k, v in t.items()}
    g["means2D"] = m2d.grad.numpy()
    return dict(color=c.detach().numpy(), depth=dep.detach().numpy(), alpha=a.detach().numpy(), radii=r.numpy()), g


@pytest.mark.parametrize("deg,pc,pcov,bg,mod,da", [
    (3, False, False, (0.1, 0.2, 0.3), 1.0, True),
    (0, False, False, (1.0, 1.0, 1.0), 1.0, False),
    (2, True, True, (0.0, 0.0, 0.0), 0.8, True),
    (1, False, True, (0.0, 0.0, 0.0), 1.0, False),
])
def test_c_oracle_backward_equals_autograd(deg, pc, pcov, bg, mod, da):
    sc, cams = small_test_scene(P=150, W=48, H=32, seed=deg + 1)
    cam = cams[1]
    d = view_inputs(sc, cam, sh_degree=deg, precomp_color=pc, precomp_cov=pcov, scale_modifier=mod)
    H, W = cam.image_height, cam.image_width
    rng = np.random.RandomState(4)
    gc = rng.randn(3, H, W).astype(np.float32)
    gd = rng.randn(1, H, W).astype(np.float32) if da else None
    ga = rng.randn(1, H, W).astype(np.float32) if da else None
    o = make_oracle(cam, bg, sh_degree=deg, scale_modifier=mod)
    out = o.forward(**np_inputs(d))
    g = o.backward(gc, gd, ga)
    ref, gref = _autograd_reference(d, cam, bg, deg, mod, gc, gd, ga)
    np.testing.assert_array_equal(out["radii"], ref["radii"])
    for k in ("color", "depth", "alpha"):
        assert np.abs(out[k] - ref[k]).max() <= 5e-6 * max(1.0, np.abs(ref[k]).max()), k
    for k, want in gref.items():
        if want is None:
            continue
        got = g[k] if k != "opacities" else g[k].reshape(want.shape)
        assert rel_err(got, want) <= 2e-5, f"grad {k}: {rel_err(got, want):.2e}"


def test_invariants():
    sc, cams = small_test_scene(P=300, W=80, H=64, seed=8)
    cam = cams[0]
    d = np_inputs(view_inputs(sc, cam))
    o = make_oracle(cam, (1.0, 1.0, 1.0))
    out = o.forward(**d)
    ft, nc = o.pixel_state()
    # alpha == 1 - final_T ; colour = sum(c a T) + T * bg (white bg, colour >= alpha-weighted black part)
    assert np.abs(out["alpha"][0] - (1.0 - ft)).max() <= 1e-5
    o0 = make_oracle(cam, (0.0, 0.0, 0.0))
    out0 = o0.forward(**d)
    assert np.abs(out["color"] - (out0["color"] + ft[None])).max() <= 1e-6
    # permutation invariance: shuffling the Gaussians changes nothing but the ids
    perm = np.random.RandomState(0).permutation(300)
    dp = {k: (None if v is None else v[perm]) for k, v in d.items()}
    outp = make_oracle(cam, (0.0, 0.0, 0.0)).forward(**dp)
    np.testing.assert_array_equal(outp["radii"], out0["radii"][perm])
    assert np.abs(outp["color"] - out0["color"]).max() <= 1e-5
    # per-tile lists are depth-sorted with ties by index
    ranges, pl = o0.binning()
    depth = o0.geom()["depth"]
    for b, e in ranges:
        seg = pl[b:e]
        key = depth[seg].view(np.uint32).astype(np.int64) * (1 << 32) + seg.astype(np.int64)
        assert (np.diff(key) > 0).all()
    # radii > 0 <=> listed in at least one tile
    assert set(np.unique(pl).tolist()) == set(np.nonzero(out0["radii"] > 0)[0].tolist())


def test_error_behaviour_and_flash_shapes():
    sc, cams = small_test_scene(P=40, W=32, H=32, seed=2)
    cam = cams[0]
    d = np_inputs(view_inputs(sc, cam))
    o = make_oracle(cam, (0, 0, 0))
    with pytest.raises(Exception):
        o.forward(d["means3D"], d["opacities"], shs=d["shs"], colors_precomp=np.zeros((40, 3), np.float32),
                  scales=d["scales"], rotations=d["rotations"])
    with pytest.raises(Exception):
        o.forward(d["means3D"], d["opacities"], shs=d["shs"])
    mask = np.zeros((32, 32), np.float32)
    mask[8:20, 8:20] = 1
    out = o.forward(**d, gt_mask=mask, num_obj=1)
    assert out["used_count"].shape == (2, 40)
    # weights scattered to the two labels add up to the alpha image
    assert abs(out["used_count"].sum() - out["alpha"].sum()) <= 1e-3
    assert abs(out["used_count"][1].sum() - out["alpha"][0][8:20, 8:20].sum()) <= 1e-3


def test_knn_oracle():
    pts = np.array([[0, 0, 0], [1, 0, 0], [0, 2, 0], [0, 0, 3], [5, 5, 5]], np.float32)
    got = knn_dist2(pts)
    assert got[0] == pytest.approx((1 + 4 + 9) / 3)
    assert got[1] == pytest.approx((1 + 5 + 10) / 3)
    assert knn_dist2(pts[:1])[0] == 0
