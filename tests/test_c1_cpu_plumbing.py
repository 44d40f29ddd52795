"""BASELINE.json config C1: "plot_461 vanilla 3DGS, 10k Gaussians, 400x300, PyTorch-CPU render path (plumbing, no GPU)".
The reference has no CPU render path (every allocation is device="cuda", gaussian_renderer/__init__.py:30), so the CPU leg
of this configuration is the oracle: one forward + backward of a 10 k-Gaussian / 400x300 synthetic view, checked through
the invariants the algorithm guarantees, and the product path's refusal to run without a GPU."""
import numpy as np
import pytest
import torch

from util import view_inputs, make_oracle, np_inputs
from w3d_amd.synth import make_scene, make_cameras


def test_c1_oracle_forward_backward_invariants():
    P, W, H = 10_000, 400, 300
    sc = make_scene(P, seed=4, scale_mean=0.012)
    cam = make_cameras(36, W, H)[3]
    bg = np.array([0.25, 0.5, 0.75], np.float32)
    d = np_inputs(view_inputs(sc, cam))
    o = make_oracle(cam, bg, nthreads=4)
    ref = o.forward(**d)
    ft, nc = o.pixel_state()
    vis = ref["radii"] > 0
    assert 0.3 * P < vis.sum() <= P and o.num_rendered() > P
    # alpha = 1 - final transmittance; colour = blended + T * background; depth not normalised
    assert np.abs(ref["alpha"][0] + ft - 1.0).max() <= 1e-5
    blended = ref["color"] - ft[None] * bg[:, None, None]
    assert blended.min() >= -1e-5 and np.isfinite(ref["color"]).all() and ref["depth"].min() >= 0
    assert (nc[ref["alpha"][0] > 0] > 0).all() and (nc[ref["alpha"][0] == 0] == 0).all()
    # per-tile lists: ascending (depth bits, index), ranges partition the list
    ranges, pl = o.binning()
    g = o.geom()
    assert ranges[0, 0] == 0 and ranges[-1, 1] == len(pl) and (ranges[1:, 0] == ranges[:-1, 1]).all()
    for t in range(0, len(ranges), 37):
        ids = pl[ranges[t, 0]:ranges[t, 1]].astype(np.int64)
        key = g["depth"][ids].view(np.uint32).astype(np.int64) * (1 << 32) + ids
        assert (np.diff(key) > 0).all()
    # backward: linear in the image gradient, zero on culled Gaussians, z-column of means2D zero
    gc = np.random.RandomState(0).randn(3, H, W).astype(np.float32)
    g1 = o.backward(gc, None, None)
    g2 = o.backward(2.0 * gc, None, None)
    for k in ("means3D", "means2D", "shs", "opacities", "scales", "rotations"):
        assert np.allclose(g2[k], 2.0 * g1[k], rtol=1e-4, atol=1e-6 * np.abs(g1[k]).max())
        assert (g1[k][~vis] == 0).all()
    assert (g1["means2D"][:, 2] == 0).all()
    o.free()


def test_c1_product_path_refuses_the_cpu():
    """No CPU fallback: the drop-in modules fail loudly on CPU tensors instead of computing somewhere else."""
    from w3d_amd.gaussian_model import GaussianModel
    from w3d_amd.gaussian_renderer import render
    from w3d_amd.train import PipelineParams
    sc = make_scene(100, seed=1)
    cam = make_cameras(2, 64, 48)[0]
    m = GaussianModel(3, device="cpu")
    m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
    with pytest.raises(RuntimeError, match="no CPU path"):
        render(cam, m, PipelineParams(), torch.zeros(3))
