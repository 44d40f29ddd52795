"""-m gpu: the storage order of the Gaussians (GaussianModel.sort_spatially / Trainer(spatial_order=True)) changes nothing the
rasterizer computes — images bit for bit, gradients and training up to the order of the float atomics — and survives the
training schedule: densifications re-sort every so often, the last one once more, moments and statistics stay with their rows."""
import numpy as np
import pytest
import torch

from test_gpu_list_share import _scene, _model

pytestmark = pytest.mark.gpu


def _neighbour_step(m):
    x = m.get_xyz.detach()
    return float((x[1:] - x[:-1]).norm(dim=1).mean())


def test_render_and_gradients_do_not_depend_on_the_storage_order():
    from w3d_amd.fused_step import backward_raw, render_raw
    # few enough Gaussians that no two depths tie (seeded scene): every output bit for bit
    sc, cams, bg = _scene(P=2_500, n_cams=3, scale=0.04)
    a, _ = _model(sc, None)
    b, _ = _model(sc, None)
    perm = b.sort_spatially()
    with torch.no_grad():
        for cam in cams:
            ra, rb = render_raw(cam, a, bg), render_raw(cam, b, bg)
            for k in ("render", "depth", "alpha"):
                assert torch.equal(ra[k], rb[k]), k
            assert torch.equal(ra["radii"][perm], rb["radii"])
    sc, cams, bg = _scene(P=30_000, n_cams=3)
    a, _ = _model(sc, None)
    b, _ = _model(sc, None)
    perm = b.sort_spatially()
    assert not torch.equal(perm, torch.arange(sc.P, device=perm.device))
    g = torch.Generator().manual_seed(0)
    for cam in cams:
        dimg = torch.randn(3, cam.image_height, cam.image_width, generator=g).cuda()
        with torch.no_grad():
            ra, rb = render_raw(cam, a, bg), render_raw(cam, b, bg)
            # same lists, same blending order — except between Gaussians of EXACTLY equal depth, whose order is the order of
            # their rows (the reference's contract: ties by index); among 30 000 fp32 depths a few dozen pairs tie, and where
            # two of a pair overlap a handful of pixels see them swapped
            for k in ("render", "depth", "alpha"):
                d = (ra[k] - rb[k]).abs()
                assert float((d > 1e-6).float().mean()) <= 1e-3 and float(d.max()) <= 0.2, (k, float(d.max()))
            assert float((ra["render"] != rb["render"]).float().mean()) <= 5e-3
            assert torch.equal(ra["radii"][perm], rb["radii"])
            na, _ = backward_raw(a, ra["handle"], dimg, want_norm=True)
            nb, _ = backward_raw(b, rb["handle"], dimg, want_norm=True)

        def differing(x, y):          # fraction of Gaussians whose values differ by more than 1e-4 of the largest one
            x, y = x.reshape(x.shape[0], -1), y.reshape(y.shape[0], -1)
            return float((((x - y).abs().max(dim=1).values) > 1e-4 * float(x.abs().max())).float().mean())
        assert differing(na[perm], nb) <= 2e-3
        for k in ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"):
            # (float atomics in another order; the Gaussians of a swapped pair and those behind them in its pixels)
            assert differing(a.grad_view(k).detach()[perm], b.grad_view(k).detach()) <= 2e-3, k
        a.flat_grad.zero_()
        b.flat_grad.zero_()


def test_training_is_the_same_in_morton_order():
    from w3d_amd.train import Trainer
    sc, cams, bg = _scene()
    runs = []
    for order in (False, True):
        m, opt = _model(sc, 1)
        tr = Trainer(m, cams, opt, bg, densify=False, spatial_order=order)
        losses = np.array([float(tr.step(it)) for it in range(1, 31)])
        runs.append((losses, m, tr))
    (l0, m0, t0), (l1, m1, t1) = runs
    assert t0.initial_perm is None and t1.initial_perm is not None and _neighbour_step(m1) < 0.4 * _neighbour_step(m0)
    assert l0[0] > l0[-1] and np.allclose(l0, l1, rtol=2e-4, atol=2e-6)
    perm = t1.initial_perm
    for k in ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"):
        d = (m0._p[k].detach()[perm] - m1._p[k].detach()).abs().reshape(-1).cpu().numpy()
        assert np.quantile(d, 0.999) <= 2e-2 and (d > 1e-4).mean() <= 2e-2, (k, float(np.quantile(d, 0.999)))
    assert torch.equal(m0.denom[perm], m1.denom)                                 # visibility counts: exact
    assert float((m0.max_radii2D[perm] - m1.max_radii2D).abs().max()) <= 1.0
    e = (m0.xyz_gradient_accum[perm] - m1.xyz_gradient_accum).abs().max() / m0.xyz_gradient_accum.abs().max()
    assert float(e) <= 2e-3


def test_the_order_survives_the_densification_schedule():
    from w3d_amd.gaussian_model import OptimizationParams
    from w3d_amd.train import Trainer, render_views

    class Opt(OptimizationParams):
        densify_from_iter = 20
        densification_interval = 15
        opacity_reset_interval = 70
        densify_until_iter = 100
        densify_grad_threshold = 0.00002
    sc, cams, bg = _scene(P=20_000)
    m, _ = _model(sc, None)
    opt = Opt()
    m.training_setup(opt)
    tr = Trainer(m, cams, opt, bg, densify=True, cameras_extent=2.0, spatial_order=True)
    tr.SPATIAL_ORDER_EVERY = 3
    m.spatial_order_every = 3
    random_step = float((sc.xyz[1:] - sc.xyz[:-1]).norm(dim=1).mean())
    first = None
    sizes = []
    for it in range(1, 121):
        loss = float(tr.step(it))
        first = loss if first is None else first
        if it in (30, 45, 60, 75, 90, 100):
            sizes.append((it, m.num_points, _neighbour_step(m)))
    assert m._densify_calls == 5 and m.num_points > sc.P            # iterations 30, 45, 60, 75, 90; re-sorted at the 1st and 4th
    by_it = {it: (n, s) for it, n, s in sizes}
    assert by_it[30][1] < 0.4 * random_step and by_it[75][1] < 0.4 * random_step
    # the rounds between append their clones and children behind the sorted part: less coherent, never random
    assert by_it[60][1] < 0.8 * random_step
    # ... and the end of densification (iteration 100) puts everything back into place
    assert by_it[100][1] < 0.4 * random_step and by_it[100][1] <= by_it[90][1]
    assert bool(torch.isfinite(m.flat).all()) and loss < first
    P = m.num_points
    assert m.xyz_gradient_accum.shape == (P, 1) and m.denom.shape == (P, 1) and m.max_radii2D.shape == (P,)
    assert m.optimizer.exp_avg.numel() == m.flat.numel()
    with torch.no_grad():
        img = render_views(m, cams[:1], bg)[0]
    assert bool(torch.isfinite(img).all())
