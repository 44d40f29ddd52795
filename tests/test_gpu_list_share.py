"""-m gpu: w3d_view.list_share (one depth-ordered list per 16x16 tile / per 32x16 pair / per 32x32 block) through the
raw-parameter paths — training, FlashSplat, subset renders, reblend — and the Trainer's choice of the mode from the walked
fraction it measures.  (The activated-parameter API under the three modes against the oracle: tests/test_gpu_parity.py.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(P=40_000, W=320, H=240, seed=2, scale=0.012, n_cams=6):
    from w3d_amd.fused_step import render_raw
    from w3d_amd.gaussian_model import GaussianModel
    from w3d_amd.synth import make_scene, make_cameras
    dev = torch.device("cuda")
    cams = [c.to(dev) for c in make_cameras(n_cams, W, H)]
    bg = torch.zeros(3, device=dev)
    gt = make_scene(P, seed=seed + 1, scale_mean=scale)
    gm = GaussianModel(3)
    gm.create_from_tensors(gt.xyz, gt.features_dc, gt.features_rest, gt.scaling, gt.rotation, gt.opacity)
    gm.active_sh_degree = 3
    with torch.no_grad():
        for c in cams:
            c.original_image = render_raw(c, gm, bg)["render"].clamp(0, 1).clone()
    return make_scene(P, seed=seed, scale_mean=scale), cams, bg


def _model(sc, share):
    from w3d_amd.gaussian_model import GaussianModel, OptimizationParams
    m = GaussianModel(3)
    m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
    m.active_sh_degree = 3
    m.list_share = share
    opt = OptimizationParams()
    m.training_setup(opt)
    return m, opt


def test_training_is_the_same_under_every_list_share_and_the_trainer_picks_one():
    from w3d_amd.rasterizer import list_share_of
    from w3d_amd.train import Trainer
    sc, cams, bg = _scene()
    runs = {}
    for share in (0, 1, 2, None):
        m, opt = _model(sc, share)
        tr = Trainer(m, cams, opt, bg, densify=False)
        tr.SHARE_PROBE_EVERY = 8
        losses = [float(tr.step(it)) for it in range(1, 41)]
        runs[share] = (np.array(losses), m.flat.detach().cpu().numpy(), m.denom.cpu().numpy(), m.max_radii2D.cpu().numpy(), tr, m)
    base = runs[0]
    assert base[0][0] > base[0][-1]
    for share in (1, 2, None):
        r = runs[share]
        # float atomics: a near-zero gradient's sign may differ between two runs (an Adam step of 2 lr); everything else agrees
        assert np.allclose(r[0], base[0], rtol=2e-4, atol=2e-6), (share, r[0][-3:], base[0][-3:])
        assert np.array_equal(r[2], base[2]) and np.abs(r[3] - base[3]).max() <= 1.0, share
        d = np.abs(r[1] - base[1])
        assert np.quantile(d, 0.999) <= 2e-2 and (d > 1e-4).mean() <= 2e-2, (share, float(np.quantile(d, 0.999)), float((d > 1e-4).mean()))
    tr, m = runs[None][4], runs[None][5]
    assert m.list_share is None and m._list_share_chosen in (0, 1, 2) and tr.share_rho is not None and 0.0 < tr.share_rho <= 1.0
    assert list_share_of(m) == m._list_share_chosen
    from w3d_amd.rasterizer import SHARE_HYST, SHARE_RHO
    want = 2 if tr.share_rho < SHARE_RHO[0] - SHARE_HYST else 1 if SHARE_RHO[0] + SHARE_HYST < tr.share_rho < SHARE_RHO[1] - SHARE_HYST \
        else 0 if tr.share_rho > SHARE_RHO[1] + SHARE_HYST else None
    assert want is None or m._list_share_chosen == want, (tr.share_rho, m._list_share_chosen)
    # a caller's explicit setting is left alone
    assert runs[2][5].list_share == 2 and runs[2][5]._list_share_chosen is None and runs[2][4].share_rho is None


def test_flashsplat_subset_and_reblend_under_shared_lists():
    from w3d_amd.gaussian_renderer import flashsplat_render, flashsplat_render_masks
    from w3d_amd.train import PipelineParams
    sc, cams, bg = _scene(P=30_000, n_cams=3)
    H, W = cams[0].image_height, cams[0].image_width
    yy, xx = torch.meshgrid(torch.arange(H, device="cuda"), torch.arange(W, device="cuda"), indexing="ij")
    labels = ((xx // 40 + 3 * (yy // 50)) % 5).float()                  # 5 labels, several per tile
    masks = torch.stack([(labels == k).float() for k in (1, 2, 3)])
    pipe = PipelineParams()
    out = {}
    with torch.no_grad():
        for share in (0, 1, 2):
            m, _ = _model(sc, share)
            head = (m.get_xyz.detach() - torch.tensor([0.1, 0.0, 0.3], device="cuda")).norm(dim=1) < 0.25
            a = flashsplat_render(cams[1], m, pipe, bg, gt_mask=labels, obj_num=4)
            b = flashsplat_render(cams[2], m, pipe, bg, used_mask=head, gt_mask=(labels > 1).float(), obj_num=1)
            c = flashsplat_render_masks(cams[0], m, pipe, bg, masks, obj_num=1)
            out[share] = (a, b, c)
    a0, b0, c0 = out[0]
    for share in (1, 2):
        a, b, c = out[share]
        for k in ("render", "alpha", "depth", "contrib_num", "radii", "proj_xy", "gs_depth"):
            assert torch.equal(a[k], a0[k]), (share, k)
        for k in ("render", "alpha", "depth", "radii"):
            assert torch.equal(b[k], b0[k]) and torch.equal(c[k], c0[k]), (share, k)
        for x, x0, name in ((a["used_count"], a0["used_count"], "labels"), (b["used_count"], b0["used_count"], "subset"),
                            (c["used_count"], c0["used_count"], "masks")):
            err = float((x - x0).abs().max() / x0.abs().max().clamp_min(1e-30))
            assert err <= 2e-6, (share, name, err)                        # (float atomics: order of the additions only)
