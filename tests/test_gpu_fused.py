"""-m gpu: the fused "next-row" kernels (N1 loss, N2 Adam) and one training step, against the
torch restatements of the reference formulas (fp32 tolerance stated per check)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(3, 40, 56), (3, 75, 133), (3, 300, 400), (1, 17, 16)])
def test_fused_l1_ssim_matches_torch(shape):
    from w3d_amd.loss import photometric_loss, photometric_loss_torch
    g = torch.Generator().manual_seed(shape[1])
    gt = torch.rand(*shape, generator=g)
    img = (gt + 0.1 * torch.randn(*shape, generator=g)).clamp(0, 1)
    img[:, :3, :5] = gt[:, :3, :5]                      # exact zeros of |x-y|: sign(0) = 0
    a = img.clone().requires_grad_(True)
    ref = photometric_loss_torch(a, gt, 0.2)
    ref.backward()
    b = img.cuda().requires_grad_(True)
    out = photometric_loss(b, gt.cuda(), 0.2)
    out.backward()
    assert abs(float(out) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref)))
    gerr = (b.grad.cpu() - a.grad).abs().max() / a.grad.abs().max()
    assert float(gerr) <= 2e-5, f"loss gradient rel err {float(gerr):.2e}"


def test_fused_loss_against_golden():
    """tests/golden/loss.npz holds l1 / ssim computed by the reference's own utils/loss_utils.py."""
    import os
    from w3d_amd.loss import photometric_loss
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "loss.npz"))
    img1, img2 = torch.tensor(z["img1"]).cuda(), torch.tensor(z["img2"]).cuda()
    want = 0.8 * float(z["l1"]) + 0.2 * (1.0 - float(z["ssim"]))
    got = float(photometric_loss(img1, img2, 0.2))
    assert abs(got - want) <= 2e-6


@pytest.mark.parametrize("n,offset", [(1000, 0), (4099, 3), (5, 1), (1 << 20, 2)])
def test_fused_adam_matches_torch(n, offset):
    from w3d_amd.fused import adam_step
    g = torch.Generator().manual_seed(n)
    base = [torch.randn(n + 8, generator=g) for _ in range(3)] + [torch.rand(n + 8, generator=g)]
    p0, g0, m0, v0 = [t[offset:offset + n].clone() for t in base]
    p = torch.nn.Parameter(p0.clone())
    optim = torch.optim.Adam([p], lr=0.01, eps=1e-15)
    # prime torch's state with the same moments and step count
    p.grad = g0.clone()
    optim.step()
    st = optim.state[p]
    st["exp_avg"].copy_(m0)
    st["exp_avg_sq"].copy_(v0)
    with torch.no_grad():
        p.copy_(p0)
    optim.step()                                     # step 2 with the primed state
    dev = [t.cuda() for t in base]
    pc, gc, mc, vc = [t[offset:offset + n] for t in dev]
    b1, b2 = 0.9, 0.999
    adam_step(pc, gc, mc, vc, 0.01, b1, b2, 1e-15, 1 - b1 ** 2, 1 - b2 ** 2, zero_grad=True)
    assert torch.allclose(pc.cpu(), p.detach(), rtol=2e-5, atol=2e-6)
    assert float((mc.cpu() - st["exp_avg"]).abs().max()) <= 1e-6
    assert float((vc.cpu() - st["exp_avg_sq"]).abs().max()) <= 1e-6
    assert float(gc.abs().max()) == 0
    # neighbours outside the slice untouched
    assert torch.equal(dev[0][:offset].cpu(), base[0][:offset]) and torch.equal(dev[0][offset + n:].cpu(), base[0][offset + n:])


def test_training_step_decreases_loss_and_tracks_stats():
    from w3d_amd.synth import make_scene, make_cameras
    from w3d_amd.gaussian_model import GaussianModel, OptimizationParams
    from w3d_amd.train import Trainer, render_views
    dev = torch.device("cuda:0")
    W, H = 160, 120
    cams = [c.to(dev) for c in make_cameras(6, W, H)]
    bg = torch.zeros(3, device=dev)
    target = make_scene(4000, seed=5, scale_mean=0.03)
    tm = GaussianModel(3, device=dev)
    tm.create_from_tensors(target.xyz, target.features_dc, target.features_rest, target.scaling, target.rotation, target.opacity)
    tm.active_sh_degree = 3
    for cam, img in zip(cams, render_views(tm, cams, bg)):
        cam.original_image = img.clamp(0, 1)
    sc = make_scene(4000, seed=6, scale_mean=0.03)
    m = GaussianModel(3, device=dev)
    m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
    m.active_sh_degree = 3
    opt = OptimizationParams()
    m.training_setup(opt)
    tr = Trainer(m, cams, opt, bg, densify=False)
    losses = [float(tr.step(i + 1)) for i in range(60)]
    assert np.mean(losses[-10:]) < 0.9 * np.mean(losses[:10])
    assert float(m.denom.max()) > 0 and float(m.xyz_gradient_accum.max()) > 0 and float(m.max_radii2D.max()) > 0
    assert tr.fused                                     # default on the GPU: raw-parameter kernels, no autograd
    assert torch.isfinite(m.flat).all()


@pytest.mark.parametrize("deg,W,H", [(2, 200, 152), (0, 176, 144), (1, 203, 149), (3, 160, 128)])
def test_fused_raw_step_equals_autograd_step(deg, W, H):
    """The fused raw-parameter path and the drop-in render()+autograd path are the same arithmetic:
    same image, same gradient bucket, same densification statistics, same parameters after Adam."""
    from w3d_amd.synth import make_scene, make_cameras
    from w3d_amd.gaussian_model import GaussianModel, OptimizationParams
    from w3d_amd.train import Trainer
    dev = torch.device("cuda:0")
    cams = [c.to(dev) for c in make_cameras(4, W, H)]
    g = torch.Generator().manual_seed(0)
    for cam in cams:
        cam.original_image = torch.rand(3, H, W, generator=g).to(dev)
    bg = torch.tensor([0.1, 0.0, 0.2], device=dev)
    sc = make_scene(6000, seed=9, scale_mean=0.02)
    import w3d_amd.gaussian_renderer as gr
    gr.RAW_AUTOGRAD = False              # the drop-in side marshals activated tensors, as the reference's render() does
    models, grads, images = [], [], []
    for fused in (False, True):
        m = GaussianModel(3, device=dev)
        m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
        m.active_sh_degree = deg
        opt = OptimizationParams()
        m.training_setup(opt)
        # (spatial_order=False: the spy below reads the gradient bucket before optimizer.step(); after a re-sort the parameters are
        #  new nn.Parameters whose first autograd .grad is a fresh tensor that step() copies into the bucket — as after a densification)
        tr = Trainer(m, cams, opt, bg, densify=False, fused=fused, spatial_order=False)
        tr.fused_adam = False            # the gradient bucket is compared below: keep the optimizer a separate sweep
        assert tr.fused == fused
        # peek at the gradient bucket of the first step before Adam consumes it
        orig = m.optimizer.step
        def spy(*a, _m=m, **k):
            grads.append(_m.flat_grad.clone())
            return orig(*a, **k)
        m.optimizer.step = spy
        l0 = float(tr.step(1))
        images.append(tr.last["image"].clone())
        # statistics after ONE step are compared strictly; later steps only loosely (Adam turns a last-bit sign
        # difference of a near-zero gradient into a 2*lr difference, after which the two runs drift apart)
        stats1 = (m.denom.clone(), m.xyz_gradient_accum.clone(), m.max_radii2D.clone())
        for it in range(2, 6):
            tr.step(it)
        models.append((m, l0, stats1))
    gr.RAW_AUTOGRAD = True
    (ma, la, sa), (mb, lb, sb) = models
    assert abs(la - lb) <= 1e-6
    assert float((images[0] - images[1]).abs().max()) <= 2e-5       # torch vs in-kernel exp/sigmoid/normalize
    ga, gb = grads[0], grads[5]          # first step of each flavour (5 steps each)
    for name, (a, b) in ma.block_slices().items():
        ref = ga[a:b]
        d = (gb[a:b] - ref).abs()
        err = float(d.max() / (ref.abs().max() + 1e-20))
        if err > 2e-4:                   # say which elements, should this ever trip: a single Gaussian or a whole block?
            bad = (d > 2e-4 * ref.abs().max()).nonzero().flatten()[:8].tolist()
            detail = [(i, float(ref[i]), float(gb[a:b][i])) for i in bad]
            raise AssertionError(f"{name}: rel err {err:.2e}, {int((d > 2e-4 * ref.abs().max()).sum())} of {b - a} "
                                 f"elements beyond the bound, first (index, autograd, fused): {detail}")
    # (a near-zero gradient whose sign differs in the last bit moves that parameter by 2*lr under Adam:
    #  bound the bulk tightly and the maximum by the largest learning rate)
    diff = (ma.flat - mb.flat).abs()
    assert float((diff > 1e-4).float().mean()) <= 1e-3 and float(diff.max()) <= 0.25
    assert torch.equal(sa[0], sb[0]) and torch.equal(sa[2], sb[2])
    assert float((sa[1] - sb[1]).abs().max() / sa[1].abs().max()) <= 2e-4
    assert float((ma.denom != mb.denom).float().mean()) <= 1e-2


@pytest.mark.parametrize("fused_adam", [False, True])
def test_speculative_list_capacity_overflow_is_repeated(fused_adam):
    """(fused_adam: the backward kernel that applies the optimizer must update NOTHING when the lists overflowed.)
    The fused forward sizes the per-tile list buffer from previous views and never syncs with the host;
    when the guess is too small the view must be repeated, with the same result as a run whose guess held."""
    from w3d_amd import fused_step
    from w3d_amd.synth import make_scene, make_cameras
    from w3d_amd.gaussian_model import GaussianModel, OptimizationParams
    from w3d_amd.train import Trainer
    dev = torch.device("cuda:0")
    W, H = 160, 120
    cams = [c.to(dev) for c in make_cameras(4, W, H)]
    g = torch.Generator().manual_seed(1)
    for cam in cams:
        cam.original_image = torch.rand(3, H, W, generator=g).to(dev)
    bg = torch.zeros(3, device=dev)
    sc = make_scene(5000, seed=3, scale_mean=0.03)
    finals, retries = [], []
    for sabotage in (False, True):
        m = GaussianModel(3, device=dev)
        m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
        m.active_sh_degree = 3
        opt = OptimizationParams()
        m.training_setup(opt)
        tr = Trainer(m, cams, opt, bg, densify=False)
        tr.fused_adam = fused_adam
        calls = {"n": 0}
        orig = fused_step.render_raw

        def counting(*a, **k):
            calls["n"] += 1
            return orig(*a, **k)
        fused_step.render_raw = counting
        try:
            for it in range(1, 5):
                fused_step.list_capacity(m, H, W).known = 100 if sabotage else 10_000_000      # far too small / generous
                tr.step(it)
        finally:
            fused_step.render_raw = orig
        finals.append(m.flat.clone())
        retries.append(calls["n"])
    assert retries[0] == 4 and retries[1] == 8          # every sabotaged view was rendered twice
    # same parameters after 4 Adam steps (float-atomic ordering noise can flip the sign of a near-zero
    # gradient, which Adam turns into a 2*lr difference on that element: bound the bulk tightly, the max loosely)
    diff = (finals[0] - finals[1]).abs()
    assert float((diff > 1e-5).float().mean()) <= 1e-3 and float(diff.max()) <= 0.2
    assert float(m.denom.max()) > 0


def test_training_with_densification_prune_and_opacity_reset():
    """The whole step incl. the episodic host logic (rows A10-A11): clone / split / prune rebuild the flat
    buffers and the Adam moments, opacity reset clears its moments; the kernels keep running on the new P."""
    from w3d_amd.synth import make_scene, make_cameras
    from w3d_amd.gaussian_model import GaussianModel, OptimizationParams
    from w3d_amd.train import Trainer, render_views
    dev = torch.device("cuda:0")
    W, H = 160, 120
    cams = [c.to(dev) for c in make_cameras(6, W, H)]
    bg = torch.zeros(3, device=dev)
    target = make_scene(3000, seed=15, scale_mean=0.04)
    tm = GaussianModel(3, device=dev)
    tm.create_from_tensors(target.xyz, target.features_dc, target.features_rest, target.scaling, target.rotation, target.opacity)
    tm.active_sh_degree = 3
    for cam, img in zip(cams, render_views(tm, cams, bg)):
        cam.original_image = img.clamp(0, 1)
    sc = make_scene(1500, seed=16, scale_mean=0.04)
    m = GaussianModel(3, device=dev)
    m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
    m.active_sh_degree = 3

    class Opt(OptimizationParams):
        densify_from_iter = 10
        densification_interval = 10
        opacity_reset_interval = 45
        densify_until_iter = 70
        densify_grad_threshold = 0.00005
    opt = Opt()
    m.training_setup(opt)
    tr = Trainer(m, cams, opt, bg, densify=True, cameras_extent=2.0)
    sizes, losses = [], []
    for it in range(1, 91):
        losses.append(float(tr.step(it)))
        sizes.append(m.num_points)
    assert len(set(sizes)) > 3                       # P changed several times (clone/split/prune)
    assert 59 * m.num_points <= m.flat.numel() <= 59 * m.num_points + 15 and m.optimizer.exp_avg.numel() == m.flat.numel()
    assert m.xyz_gradient_accum.shape[0] == m.num_points and m.max_radii2D.shape[0] == m.num_points
    assert torch.isfinite(m.flat).all() and all(np.isfinite(losses))
    assert min(losses[30:44]) < np.mean(losses[:5])        # learning happens between the opacity resets (45, 90)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_densify_compaction_kernel_matches_reference_golden(tag, monkeypatch):
    """csrc/w3d_densify.hip against tests/golden/densify.npz (the reference's own densify_and_prune, CPU): the random
    split samples are taken from the CPU stream the fixture was made with, everything else runs on the GPU.  Rows, order,
    Adam moments, which_object and statistics must match bit for bit; the split children's positions (a bmm on the GPU)
    to 1e-6."""
    import os
    from test_host_logic import _model_from_golden, G
    z = np.load(os.path.join(G, "densify.npz"))
    m = _model_from_golden(z, tag, device="cuda:0")
    max_grad, min_opacity, extent, mss, seed = [float(x) for x in z[f"{tag}_args"]]
    real_normal = torch.normal

    def cpu_stream_normal(mean, std, **kw):
        return real_normal(mean=mean.cpu(), std=std.cpu(), **kw).to(std.device)
    monkeypatch.setattr(torch, "normal", cpu_stream_normal)
    torch.manual_seed(int(seed))
    P0 = m.num_points
    m.densify_and_prune(max_grad, min_opacity, extent, None if mss < 0 else mss)
    assert m.num_points == z[f"{tag}_post_xyz"].shape[0] != P0
    mom = m.optimizer.moments()
    for n in ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"):
        got, ref = m._p[n].detach().cpu().numpy(), z[f"{tag}_post_{n}"]
        if n == "xyz":
            np.testing.assert_allclose(got, ref, rtol=0, atol=1e-6)
            assert (got != ref).mean() < 0.2          # only split children may differ at all
        elif n == "scaling":
            np.testing.assert_allclose(got, ref, rtol=0, atol=1e-6)     # log() of the children's scales on the GPU
        else:
            assert np.array_equal(got, ref), n
        assert np.array_equal(mom[n][0].cpu().numpy(), z[f"{tag}_post_m_{n}"]), n
        assert np.array_equal(mom[n][1].cpu().numpy(), z[f"{tag}_post_v_{n}"]), n
    assert np.array_equal(m._which_object.cpu().numpy(), z[f"{tag}_post_which_object"])
    assert float(m.xyz_gradient_accum.abs().max()) == 0 and float(m.max_radii2D.abs().max()) == 0
    # prune_points keeps the survivors' statistics and moments
    m.max_radii2D = torch.arange(m.num_points, device="cuda:0", dtype=torch.float32)
    before = m.flat.clone()
    sl = m.block_slices()["f_rest"]
    mask = torch.zeros(m.num_points, dtype=torch.bool, device="cuda:0")
    mask[1::3] = True
    Pb = m.num_points
    m.prune_points(mask)
    keep = (~mask).nonzero().squeeze(1)
    assert torch.equal(m.max_radii2D, keep.float())
    assert torch.equal(m._p["f_rest"].detach().reshape(-1, 45), before[sl[0]:sl[1]].view(Pb, 45)[keep])


@pytest.mark.parametrize("P", [7000, 4999])      # 4999: odd, so the SH blocks are not 16-B aligned (dword fallback paths)
def test_fused_adam_backward_equals_separate_adam_sweep(P):
    """w3d_backward_raw_adam (optimizer applied by the backward kernel, no gradient bucket) against w3d_backward_raw +
    w3d_adam_step: same parameters and moments after one step (up to float-atomic ordering noise in the gradients), same
    statistics, step counter advanced, and the gradient bucket never written."""
    from w3d_amd.synth import make_scene, make_cameras
    from w3d_amd.gaussian_model import GaussianModel, OptimizationParams
    from w3d_amd.train import Trainer
    dev = torch.device("cuda:0")
    W, H = 208, 160
    cams = [c.to(dev) for c in make_cameras(4, W, H)]
    g = torch.Generator().manual_seed(2)
    for cam in cams:
        cam.original_image = torch.rand(3, H, W, generator=g).to(dev)
    bg = torch.tensor([0.0, 0.1, 0.2], device=dev)
    sc = make_scene(P, seed=11, scale_mean=0.02)
    out = []
    for fused_adam in (False, True):
        m = GaussianModel(3, device=dev)
        m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
        m.active_sh_degree = 3
        opt = OptimizationParams()
        m.training_setup(opt)
        tr = Trainer(m, cams, opt, bg, densify=False)
        tr.fused_adam = fused_adam
        m.flat_grad.fill_(123.0)         # (after the Trainer: its spatial_order re-sort rebinds the model's buffers)
        p0 = m.flat.clone()
        tr.step(1)
        one = (m.flat.clone(), m.optimizer.exp_avg.clone(), m.optimizer.exp_avg_sq.clone(), m.xyz_gradient_accum.clone(),
               m.denom.clone(), m.max_radii2D.clone())
        if fused_adam:
            assert float((m.flat_grad - 123.0).abs().max()) == 0          # bucket untouched
        for it in range(2, 5):
            tr.step(it)
        assert m.optimizer.step_count == 4
        out.append((one, m.flat.clone(), p0))
    (a1, a4, p0), (b1, b4, _) = out
    # moments after one step are (1-b1) g and (1-b2) g^2: compare them like gradients
    for k, tol in ((1, 2e-4), (2, 4e-4)):
        for name, (lo, hi) in m.block_slices().items():
            ref = a1[k][lo:hi]
            err = float((b1[k][lo:hi] - ref).abs().max() / (ref.abs().max() + 1e-30))
            assert err <= tol, f"moment {k} of {name}: rel err {err:.2e}"
    # the first Adam step moves every parameter with a non-zero gradient by ~lr: both flavours moved the same way
    moved = (a1[0] - p0).abs() > 0
    assert float(moved.float().mean()) > 0.3
    d1 = (a1[0] - b1[0]).abs()
    assert float((d1 > 1e-6).float().mean()) <= 2e-3 and float(d1.max()) <= 0.11
    assert torch.equal(a1[4], b1[4]) and torch.equal(a1[5], b1[5])
    assert float((a1[3] - b1[3]).abs().max() / a1[3].abs().max()) <= 2e-4
    d4 = (a4 - b4).abs()
    assert float((d4 > 1e-4).float().mean()) <= 2e-3 and float(d4.max()) <= 0.25


def test_flashsplat_raw_fast_path_equals_drop_in_path():
    """flashsplat_render under no_grad on a flat GaussianModel takes the raw-parameter forward (no activation / cat kernels);
    with autograd enabled it goes through the drop-in FlashSplatRasterizer.  Same 10-key dict, same integers, same counts."""
    from w3d_amd.synth import make_scene, make_cameras
    from w3d_amd.gaussian_model import GaussianModel
    from w3d_amd.gaussian_renderer import flashsplat_render
    from w3d_amd.train import PipelineParams
    dev = torch.device("cuda:0")
    W, H = 200, 136
    cam = make_cameras(3, W, H)[1].to(dev)
    sc = make_scene(5000, seed=21, scale_mean=0.03)
    m = GaussianModel(3, device=dev)
    m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
    m.active_sh_degree = 3
    bg = torch.tensor([0.2, 0.1, 0.0], device=dev)
    yy, xx = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
    labels = ((xx // 50) % 3).float()                    # labels 0..2 -> obj_num = 2, rows 0..2
    slow = flashsplat_render(cam, m, PipelineParams(), bg, gt_mask=labels, obj_num=2)
    with torch.no_grad():
        fast = flashsplat_render(cam, m, PipelineParams(), bg, gt_mask=labels, obj_num=2)
    assert set(fast.keys()) == set(slow.keys())
    assert torch.equal(fast["radii"], slow["radii"]) and torch.equal(fast["visibility_filter"], slow["visibility_filter"])
    assert fast["used_count"].shape == (3, m.num_points)
    assert float((fast["contrib_num"] != slow["contrib_num"]).float().mean()) <= 1e-3
    for k in ("render", "alpha", "depth", "proj_xy", "gs_depth"):
        assert float((fast[k] - slow[k]).abs().max()) <= 5e-5 * max(1.0, float(slow[k].abs().max())), k
    uc_f, uc_s = fast["used_count"], slow["used_count"]
    assert float((uc_f - uc_s).abs().max() / uc_s.abs().max()) <= 1e-4
    assert float(uc_s.sum()) > 0


def test_flashsplat_masks_of_one_view_reuse_the_forward():
    """flashsplat_render_masks (one forward, the blend repeated per mask) == flashsplat_render called once per mask."""
    from w3d_amd.synth import make_scene, make_cameras
    from w3d_amd.gaussian_model import GaussianModel
    from w3d_amd.gaussian_renderer import flashsplat_render, flashsplat_render_masks
    from w3d_amd.train import PipelineParams
    dev = torch.device("cuda:0")
    W, H = 200, 136
    cam = make_cameras(3, W, H)[2].to(dev)
    sc = make_scene(5000, seed=22, scale_mean=0.03)
    m = GaussianModel(3, device=dev)
    m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
    m.active_sh_degree = 3
    bg = torch.zeros(3, device=dev)
    yy, xx = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
    masks = torch.stack([(xx < 60).float(), ((xx - 100) ** 2 + (yy - 70) ** 2 < 900).float(), (yy > 100).float()])
    # masks 0 and 1 overlap nowhere, mask 2 overlaps both: first the general path (one blend per mask) ...
    out = flashsplat_render_masks(cam, m, PipelineParams(), bg, masks, obj_num=1)
    assert out["used_count"].shape == (3, 2, m.num_points)
    # ... then the disjoint pair through the single merged-label blend
    pair = flashsplat_render_masks(cam, m, PipelineParams(), bg, masks[:2], obj_num=1)
    assert pair["used_count"].shape == (2, 2, m.num_points)
    err = float((pair["used_count"] - out["used_count"][:2]).abs().max() / out["used_count"].abs().max())
    assert err <= 1e-5, err
    with torch.no_grad():
        for k in range(3):
            ref = flashsplat_render(cam, m, PipelineParams(), bg, gt_mask=masks[k], obj_num=1)
            err = float((out["used_count"][k] - ref["used_count"]).abs().max() / ref["used_count"].abs().max())
            assert err <= 1e-5, (k, err)
            assert torch.equal(out["render"], ref["render"])
    # every blended weight lands in exactly one of the two rows: the row sum does not depend on the mask
    tot = out["used_count"].sum(1)
    assert float((tot[0] - tot[1]).abs().max() / tot[0].abs().max()) <= 1e-5


def test_render_under_no_grad_takes_the_raw_forward_and_matches():
    from w3d_amd.synth import make_scene, make_cameras
    from w3d_amd.gaussian_model import GaussianModel
    from w3d_amd.gaussian_renderer import render
    from w3d_amd.train import PipelineParams
    dev = torch.device("cuda:0")
    cam = make_cameras(3, 176, 120)[0].to(dev)
    sc = make_scene(4000, seed=23, scale_mean=0.03)
    m = GaussianModel(3, device=dev)
    m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
    m.active_sh_degree = 2
    bg = torch.tensor([0.3, 0.2, 0.1], device=dev)
    slow = render(cam, m, PipelineParams(), bg, scaling_modifier=0.9)
    assert slow["render"].requires_grad
    with torch.no_grad():
        fast = render(cam, m, PipelineParams(), bg, scaling_modifier=0.9)
    assert set(fast.keys()) == set(slow.keys())
    assert torch.equal(fast["radii"], slow["radii"])
    for k in ("render", "depth", "alpha"):
        assert float((fast[k] - slow[k].detach()).abs().max()) <= 5e-5 * max(1.0, float(slow[k].abs().max())), k


def test_training_recovers_a_perturbed_scene():
    """End to end: a scene rendered to 8 ground-truth views, its parameters perturbed, 300 fused steps (fused Adam, densification
    every 50 iterations) — the PSNR must rise by >= 10 dB on the training views and >= 5 dB on a held-out view."""
    from w3d_amd.synth import make_scene, make_cameras
    from w3d_amd.gaussian_model import GaussianModel, OptimizationParams
    from w3d_amd.train import Trainer, render_views
    from util import psnr
    dev = torch.device("cuda:0")
    W, H = 320, 240
    cams = [c.to(dev) for c in make_cameras(9, W, H)]
    bg = torch.zeros(3, device=dev)
    sc = make_scene(12000, seed=41, scale_mean=0.02)
    gt = GaussianModel(3, device=dev)
    gt.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
    gt.active_sh_degree = 3
    for cam, img in zip(cams, render_views(gt, cams, bg)):
        cam.original_image = img.clamp(0, 1)
    g = torch.Generator().manual_seed(5)
    m = GaussianModel(3, device=dev)
    m.create_from_tensors(sc.xyz + 0.01 * torch.randn(sc.xyz.shape, generator=g), sc.features_dc + 0.4 * torch.randn(sc.features_dc.shape, generator=g),
                          sc.features_rest * 0.0, sc.scaling + 0.15 * torch.randn(sc.scaling.shape, generator=g),
                          sc.rotation + 0.1 * torch.randn(sc.rotation.shape, generator=g),
                          sc.opacity + 0.5 * torch.randn(sc.opacity.shape, generator=g))
    m.active_sh_degree = 3

    class Opt(OptimizationParams):
        densify_from_iter = 50
        densification_interval = 50
        opacity_reset_interval = 100000
        densify_until_iter = 250
        position_lr_init = 0.00016 * 5
    opt = Opt()
    m.training_setup(opt)
    train, held_out = cams[:8], cams[8]
    tr = Trainer(m, train, opt, bg, densify=True, cameras_extent=2.0)
    assert tr.fused and tr.fused_adam

    def quality():
        imgs = render_views(m, cams, bg)
        ps = [psnr(i.clamp(0, 1).cpu().numpy(), c.original_image.cpu().numpy()) for i, c in zip(imgs, cams)]
        return float(np.mean(ps[:8])), float(ps[8])
    p_train0, p_held0 = quality()
    for it in range(1, 301):
        tr.step(it)
    p_train1, p_held1 = quality()
    print(f"PSNR train {p_train0:.2f} -> {p_train1:.2f} dB, held-out {p_held0:.2f} -> {p_held1:.2f} dB, P {sc.xyz.shape[0]} -> {m.num_points}")
    assert p_train1 >= p_train0 + 10.0, (p_train0, p_train1)      # measured: 19.1 -> 36.9 dB
    assert p_held1 >= p_held0 + 5.0, (p_held0, p_held1)          # measured: 19.0 -> 27.9 dB
    assert torch.isfinite(m.flat).all()


@pytest.mark.parametrize("shape", [(3, 97, 131), (3, 160, 208), (1, 33, 40)])
def test_reference_loss_lines_run_as_one_fused_pair(shape):
    """train_vanilla_3dgs.py:77-79 as written — Ll1 = l1_loss(image, gt); loss = (1-l)*Ll1 + l*(1-ssim(image, gt)) — goes
    through ONE autograd node (loss._FusedLossPair): values and dL/dimage equal the torch restatement of the reference."""
    from w3d_amd import loss as L
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    gt = torch.rand(*shape, generator=g).to(dev)
    base = (gt.cpu() + 0.1 * torch.randn(*shape, generator=g)).clamp(0, 1).to(dev)
    lam = 0.2
    img = base.clone().requires_grad_(True)
    Ll1 = L.l1_loss(img, gt)
    assert L._tls.pending is not None and type(Ll1.grad_fn).__name__.startswith("_FusedLossPair")
    s = L.ssim(img, gt)
    assert L._tls.pending is None and s.grad_fn is Ll1.grad_fn            # the same node: no second pass A
    loss = (1.0 - lam) * Ll1 + lam * (1.0 - s)
    loss.backward()
    # (the torch restatement on the CPU, like the other loss tests: its GPU convolutions go through MIOpen, whose backward once
    #  aborted the whole suite run on a fresh box)
    ref_img = base.cpu().clone().requires_grad_(True)
    ref_l1 = torch.abs(ref_img - gt.cpu()).mean()
    ref_s = L.ssim_torch(ref_img, gt.cpu())
    ref = (1.0 - lam) * ref_l1 + lam * (1.0 - ref_s)
    ref.backward()
    assert abs(float(Ll1) - float(ref_l1)) <= 2e-6 and abs(float(s) - float(ref_s)) <= 2e-6
    assert abs(float(loss) - float(ref)) <= 2e-6
    err = float((img.grad.cpu() - ref_img.grad).abs().max() / ref_img.grad.abs().max())
    assert err <= 2e-4, err
    # and it equals the single-call fused loss bit for bit in value, to rounding in the gradient
    img2 = base.clone().requires_grad_(True)
    l2 = L.photometric_loss(img2, gt, lam)
    l2.backward()
    assert abs(float(l2) - float(loss)) <= 1e-6
    assert float((img2.grad - img.grad).abs().max() / img.grad.abs().max()) <= 1e-5
    # only one of the two outputs used: the other weight is zero
    img3 = base.clone().requires_grad_(True)
    (3.0 * L.l1_loss(img3, gt)).backward()
    r3 = base.clone().requires_grad_(True)
    (3.0 * torch.abs(r3 - gt).mean()).backward()
    assert float((img3.grad - r3.grad).abs().max()) <= 1e-7 * 3.0
    # a pending SSIM is only handed to a call on the SAME tensors; ssim() on other images runs its own kernel pair
    img4 = base.clone().requires_grad_(True)
    L.l1_loss(img4, gt)
    other = (base * 0.5).clone().requires_grad_(True)
    s_other = L.ssim(other, gt)
    assert abs(float(s_other) - float(L.ssim_torch(other.detach().cpu(), gt.cpu()))) <= 2e-6
    assert L._tls.pending is None
    # without grad (evaluation code) l1_loss is the plain torch expression
    with torch.no_grad():
        assert L.l1_loss(base, gt).grad_fn is None and L._tls.pending is None


def test_add_densification_stats_kernel_matches_the_reference_statements():
    """scene/gaussian_model.py:461-463 on a boolean filter: one kernel, same values as the masked torch statements."""
    from w3d_amd.gaussian_model import GaussianModel
    from w3d_amd.synth import make_scene
    dev = torch.device("cuda:0")
    sc = make_scene(5000, seed=3)
    m = GaussianModel(3, device=dev)
    m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
    from w3d_amd.gaussian_model import OptimizationParams
    m.training_setup(OptimizationParams())
    g = torch.Generator().manual_seed(1)
    vp = torch.zeros(5000, 3, device=dev, requires_grad=True)
    vp.grad = torch.randn(5000, 3, generator=g).to(dev)
    filt = (torch.rand(5000, generator=g) > 0.4).to(dev)
    acc0 = torch.rand(5000, 1, generator=g).to(dev)
    den0 = torch.randint(0, 5, (5000, 1), generator=g).float().to(dev)
    m.xyz_gradient_accum.copy_(acc0); m.denom.copy_(den0)
    m.add_densification_stats(vp, filt)
    ref_acc, ref_den = acc0.clone(), den0.clone()
    ref_acc[filt] += torch.norm(vp.grad[filt, :2], dim=-1, keepdim=True)
    ref_den[filt] += 1
    assert torch.equal(m.denom, ref_den)
    assert float((m.xyz_gradient_accum - ref_acc).abs().max()) <= 1e-6
    assert torch.equal(m.xyz_gradient_accum[~filt], acc0[~filt])


@pytest.mark.parametrize("P", [1, 63, 65, 4099, 6001])
def test_raw_forward_with_ragged_point_counts(P):
    """The raw-parameter preprocess fetches the SH rows of a wave's 64 Gaussians as one span through LDS: point counts that
    leave a partial last wave (and 1-3 floats behind its last whole 16-B chunk) must give the drop-in module's image."""
    from w3d_amd.synth import make_scene, make_cameras
    from w3d_amd.gaussian_model import GaussianModel
    from w3d_amd.gaussian_renderer import render
    from w3d_amd.train import PipelineParams
    import w3d_amd.gaussian_renderer as gr
    dev = torch.device("cuda:0")
    cam = make_cameras(3, 208, 160)[1].to(dev)
    sc = make_scene(P, seed=P, scale_mean=0.05)
    m = GaussianModel(3, device=dev)
    m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
    m.active_sh_degree = 3
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    with torch.no_grad():
        raw = render(cam, m, PipelineParams(), bg)                    # no grad: the raw-parameter forward
    gr.RAW_AUTOGRAD = False
    try:
        ref = render(cam, m, PipelineParams(), bg)                    # activated tensors through the drop-in module
    finally:
        gr.RAW_AUTOGRAD = True
    assert int((raw["radii"] > 0).sum()) > 0
    assert torch.equal(raw["radii"] > 0, ref["radii"] > 0)
    # (in-kernel exp / sigmoid / normalize vs torch's: inputs differ in the last bit, so a pixel where some alpha sits on
    #  the 1/255 threshold may flip — a wrong SH row would move whole footprints by O(0.1))
    for k in ("render", "depth"):
        d = (raw[k] - ref[k]).abs() / max(1.0, float(ref[k].abs().max()))
        assert float((d > 2e-5).float().mean()) <= 2e-4 and float(d.max()) <= 5e-3, (k, float(d.max()))


def test_two_views_in_one_backward_pass_add_up():
    """A multi-view loss: two render() calls on the SAME model feed one loss.backward().  Both autograd nodes run before any
    AccumulateGrad and both see .grad None; only the first may write the flat bucket directly, the other must hand autograd a
    temporary — the parameters end up with the SUM of the two views' gradients (a second direct node would overwrite the
    bucket and the engine would add the same views to themselves: twice the second view, the first lost), and the
    reference's `optimizer.step()` then steps on that sum."""
    from w3d_amd.synth import make_scene, make_cameras
    from w3d_amd.gaussian_model import GaussianModel, OptimizationParams
    from w3d_amd.gaussian_renderer import render
    from w3d_amd.train import PipelineParams
    dev = torch.device("cuda:0")
    W, H = 176, 128
    cams = [c.to(dev) for c in make_cameras(4, W, H)]
    sc = make_scene(5000, seed=31, scale_mean=0.03)
    bg = torch.tensor([0.1, 0.2, 0.0], device=dev)
    g = torch.Generator().manual_seed(8)
    wts = [torch.randn(3, H, W, generator=g).to(dev) for _ in range(2)]

    def model():
        m = GaussianModel(3, device=dev)
        m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
        m.active_sh_degree = 3
        m.deterministic = True              # (bit-reproducible gradients: the comparison below can be tight)
        m.training_setup(OptimizationParams())
        m.optimizer.zero_grad(set_to_none=True)
        return m
    singles = []
    for v in range(2):
        m = model()
        (render(cams[v], m, PipelineParams(), bg)["render"] * wts[v]).sum().backward()
        singles.append(torch.cat([p.grad.reshape(-1) for p in m._p.values()]).clone())
        assert all(p.grad.data_ptr() == m.grad_view(n).data_ptr() for n, p in m._p.items())      # adopted without a copy
    want = singles[0] + singles[1]
    m = model()
    loss = sum((render(cams[v], m, PipelineParams(), bg)["render"] * wts[v]).sum() for v in range(2))
    loss.backward()
    got = torch.cat([p.grad.reshape(-1) for p in m._p.values()])
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 1e-6 * scale, float((got - want).abs().max()) / scale
    assert float((got - 2 * singles[1]).abs().max()) > 1e-3 * scale       # (what the unguarded path produced)
    # ... and the drop-in optimizer.step() uses that sum, wherever autograd left it
    ref = model()
    ref.flat_grad.copy_(want)
    ref.optimizer.step(respect_none_grads=False)
    m.optimizer.step()
    assert float((m.flat - ref.flat).abs().max()) <= 1e-6
    assert m.optimizer.steps == ref.optimizer.steps == {n: 1 for n in m.optimizer.steps}
    # a fresh pass after zero_grad claims the bucket again
    m.optimizer.zero_grad(set_to_none=True)
    (render(cams[2], m, PipelineParams(), bg)["render"] * wts[0]).sum().backward()
    assert all(p.grad.data_ptr() == m.grad_view(n).data_ptr() for n, p in m._p.items())


def test_kept_backward_scratch_is_handed_back_clean_and_changes_no_gradient(monkeypatch):
    """w3d_view.records_kept_clean: a model keeps its backward scratch from call to call, the blend backward skips its zeroing
    pass and the per-Gaussian backward writes zeros over every record it consumed.  Invariant: after EVERY backward flavour
    (gradient-writing, fused Adam, low-rank in one call and in two, alternating views with different visible sets) the buffer
    is all zero bits again; and the gradients equal those of the same calls on a fresh scratch zeroed by the library.  A
    backward abandoned half way (first half of the two-call form only) leaves the buffer dirty and marked so: the next call
    zero-fills it."""
    import w3d_amd.fused_step as FS
    from w3d_amd.synth import make_scene, make_cameras
    from w3d_amd.gaussian_model import GaussianModel, OptimizationParams
    dev = torch.device("cuda:0")
    W, H = 208, 160
    cams = [c.to(dev) for c in make_cameras(5, W, H)]
    sc = make_scene(7001, seed=41, scale_mean=0.03)
    bg = torch.zeros(3, device=dev)
    g = torch.Generator().manual_seed(3)
    dimgs = [torch.randn(3, H, W, generator=g).to(dev) * 1e-2 for _ in cams]

    def model():
        m = GaussianModel(3, device=dev)
        m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
        m.active_sh_degree = 3
        m.training_setup(OptimizationParams())
        return m

    def clean(m):
        k = m._w3d_bwd_scratch
        return k.clean and int(torch.count_nonzero(k.buf)) == 0

    def run(m, kept):
        if not kept:     # the library zeroes a fresh buffer per call (records_kept_clean = 0)
            monkeypatch.setattr(FS, "backward_scratch", lambda view, P, pl, d, owner=None: real(view, P, pl, d, None))
        else:
            monkeypatch.setattr(FS, "backward_scratch", real)
        out = []
        with torch.no_grad():
            for i, cam in enumerate(cams):
                pkg = FS.render_raw(cam, m, bg, sync=True)
                if i % 3 == 0:
                    gn, m2d = FS.backward_raw(m, pkg["handle"], dimgs[i], want_norm=True, want_means2D=True)
                    out.append((m.flat_grad.clone(), gn.clone(), m2d.clone()))
                elif i % 3 == 1:
                    gn, dcol = FS.backward_raw_lowrank(m, pkg["handle"], dimgs[i])
                    out.append((m.flat_grad.clone(), gn.clone(), dcol.clone()))
                else:
                    early = FS.backward_blend_dcolor(m, pkg["handle"], dimgs[i])
                    if kept:
                        assert not m._w3d_bwd_scratch.clean            # between the halves: dirty, and marked so
                    gn, _ = FS.backward_raw_lowrank(m, pkg["handle"], None)
                    out.append((m.flat_grad.clone(), gn.clone(), early.clone()))
                if kept:
                    assert clean(m), i
        return out
    real = FS.backward_scratch
    a, b = run(model(), True), run(model(), False)
    for i, (xa, xb) in enumerate(zip(a, b)):
        for ta, tb in zip(xa, xb):
            scale = float(tb.abs().max()) + 1e-30
            assert float((ta - tb).abs().max()) <= 2e-5 * scale, (i, float((ta - tb).abs().max()) / scale)
    # abandoned first half -> dirty; the next complete backward zero-fills, computes the same gradient and leaves it clean
    monkeypatch.setattr(FS, "backward_scratch", real)
    m = model()
    with torch.no_grad():
        pkg = FS.render_raw(cams[0], m, bg, sync=True)
        FS.backward_blend_dcolor(m, pkg["handle"], dimgs[0] * 3.0)
        assert not m._w3d_bwd_scratch.clean and int(torch.count_nonzero(m._w3d_bwd_scratch.buf)) > 0
        pkg = FS.render_raw(cams[0], m, bg, sync=True)
        gn, m2d = FS.backward_raw(m, pkg["handle"], dimgs[0], want_norm=True, want_means2D=True)
        assert clean(m)
        scale = float(a[0][0].abs().max())
        assert float((m.flat_grad - a[0][0]).abs().max()) <= 2e-5 * scale
        # ... and the fused-Adam flavour (reads the same records) hands it back clean as well
        pkg = FS.render_raw(cams[1], m, bg, sync=True)
        FS.backward_raw_adam(m, pkg["handle"], dimgs[1])
        assert clean(m)
        # another backward of the same model between the two halves wipes the first half's records: the second half must
        # refuse instead of returning zero geometry gradients
        pkg_a = FS.render_raw(cams[2], m, bg, sync=True)
        FS.backward_blend_dcolor(m, pkg_a["handle"], dimgs[2])
        pkg_b = FS.render_raw(cams[3], m, bg, sync=True)
        FS.backward_raw(m, pkg_b["handle"], dimgs[3])
        with pytest.raises(RuntimeError, match="between backward_blend_dcolor"):
            FS.backward_raw_lowrank(m, pkg_a["handle"], None)
        assert clean(m)


def test_color_only_forward_equals_the_full_forward():
    """w3d_forward_stage2 with out_depth = out_alpha = NULL (render_raw(color_only=True), what Trainer.step_fused asks for): the
    colour image, the per-pixel state kept for the backward and the gradients are bit-identical to the full forward's; asking
    for FlashSplat outputs without the two images is refused."""
    import ctypes
    import w3d_amd.fused_step as FS
    from w3d_amd._lib import lib, ptr, stream_ptr
    from w3d_amd.synth import make_scene, make_cameras
    from w3d_amd.gaussian_model import GaussianModel, OptimizationParams
    from w3d_amd.rasterizer import debug_pixel_state
    dev = torch.device("cuda:0")
    W, H = 203, 149
    cam = make_cameras(2, W, H)[1].to(dev)
    sc = make_scene(9000, seed=12, scale_mean=0.03)
    bg = torch.tensor([0.2, 0.0, 0.1], device=dev)
    m = GaussianModel(3, device=dev)
    m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
    m.active_sh_degree = 3
    m.deterministic = True
    m.training_setup(OptimizationParams())
    dimg = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
    with torch.no_grad():
        full = FS.render_raw(cam, m, bg)
        FS.backward_raw(m, full["handle"], dimg)
        g_full = m.flat_grad.clone()
        co = FS.render_raw(cam, m, bg, color_only=True)
        assert co["depth"] is None and co["alpha"] is None and torch.equal(co["render"], full["render"])
        sa = debug_pixel_state(dict(view=full["handle"]["view"], state=full["handle"]["state"], P=full["handle"]["P"]))
        sb = debug_pixel_state(dict(view=co["handle"]["view"], state=co["handle"]["state"], P=co["handle"]["P"]))
        assert torch.equal(sa[0], sb[0]) and torch.equal(sa[1], sb[1])
        FS.backward_raw(m, co["handle"], dimg)
        assert torch.equal(m.flat_grad, g_full)
        # FlashSplat outputs need the two images: refused on the host side of the ABI
        h = co["handle"]
        cn = torch.empty(H, W, dtype=torch.int32, device=dev)
        sb_, tb_ = ctypes.c_uint64(), ctypes.c_uint64()
        lib.w3d_forward_sizes(h["P"], H, W, ctypes.byref(sb_), ctypes.byref(tb_))
        scratch = torch.empty(tb_.value, dtype=torch.uint8, device=dev)
        rc = lib.w3d_forward_stage2(ctypes.byref(h["view"].c), h["P"], ptr(h["state"]), ptr(scratch), ptr(h["point_list"]),
                                    ctypes.c_uint64(h["capacity"]), ptr(co["render"]), None, None, None, 1, None, ptr(cn), None, None,
                                    stream_ptr(dev))
        assert rc != 0 and b"both or neither" in lib.w3d_last_error()


def test_last_gaussian_of_a_ragged_workgroup():
    """P is not a multiple of the per-Gaussian backward's 256-lane workgroups: the idle lanes of the last workgroup stay in
    the kernel for its barriers (they alias Gaussian P - 1) and must not act on it.  Round 4: under records_kept_clean such a
    lane could zero the record of P - 1 before its owner had read it — Gaussian P - 1 then lost its whole gradient for that
    view, about once in 40 launches, which surfaced as three unrelated-looking one-off failures of the full suite.  The same
    view's backward is repeated and the last Gaussian's gradient must come out the same every time."""
    from w3d_amd.synth import make_scene, make_cameras
    from w3d_amd.gaussian_model import GaussianModel, OptimizationParams
    from w3d_amd.fused_step import backward_raw, backward_raw_adam, render_raw
    from w3d_amd.fused import l1_ssim_fwd_bwd
    dev = torch.device("cuda:0")
    # the last workgroup holds 136 Gaussians: P - 1 shares its wave with 7 others (whose loads go to eight different places),
    # the last wave is idle — 64 lanes on ONE address, the wave that ran ahead and cleared the record
    W, H, P = 176, 144, 19 * 256 + 136
    cam = make_cameras(3, W, H)[1].to(dev)
    cam.original_image = torch.rand(3, H, W, generator=torch.Generator().manual_seed(3)).to(dev)
    sc = make_scene(P, seed=21, scale_mean=0.03)
    sc.xyz[-1] = sc.xyz.mean(0)              # the last Gaussian sits in the middle of the scene: visible, with a gradient
    sc.opacity[-1] = 2.0
    m = GaussianModel(3, device=dev)
    m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
    m.active_sh_degree = 3
    opt = OptimizationParams()
    m.training_setup(opt)
    bg = torch.zeros(3, device=dev)
    rows = []
    with torch.no_grad():
        for it in range(400):
            pkg = render_raw(cam, m, bg, sync=True, color_only=True)
            _, dimg = l1_ssim_fwd_bwd(pkg["render"], cam.original_image, opt.lambda_dssim)
            backward_raw(m, pkg["handle"], dimg)
            rows.append(torch.cat([m.flat_grad[a:b].view(P, -1)[-1] for a, b in m.block_slices().values()]).clone())
    rows = torch.stack(rows)
    assert float(rows[0].abs().max()) > 0, "the last Gaussian has no gradient in this view: the test tests nothing"
    spread = float((rows - rows[0]).abs().max() / rows[0].abs().max())
    assert spread <= 1e-4, f"the last Gaussian's gradient varies from launch to launch: {spread:.2e} of its largest component"
    # ... and the fused backward + Adam kernel: the first moment the first step leaves behind is 0.1 x that gradient
    for it in range(50):
        m2 = GaussianModel(3, device=dev)
        m2.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
        m2.active_sh_degree = 3
        m2.training_setup(opt)
        m2.update_learning_rate(1)
        with torch.no_grad():
            pkg = render_raw(cam, m2, bg, sync=True, color_only=True)
            _, dimg = l1_ssim_fwd_bwd(pkg["render"], cam.original_image, opt.lambda_dssim)
            backward_raw_adam(m2, pkg["handle"], dimg, want_norm=False, update_stats=False)
        mom = torch.cat([m2.optimizer.exp_avg[a:b].view(P, -1)[-1] for a, b in m2.block_slices().values()])
        err = float((mom - 0.1 * rows[0]).abs().max() / (0.1 * rows[0].abs().max()))
        assert err <= 1e-4, f"fused backward + Adam, launch {it}: first moment of the last Gaussian off by {err:.2e}"
