"""Deterministic backward mode (w3d_view.deterministic, SURVEY.md section 5 sanitizer row): every (tile, Gaussian)
contribution goes to the slot of its list entry and is added per Gaussian in tile order instead of by float atomics.
Checked here: same gradients as the default mode up to the order of the additions, bit-identical from run to run, and
"same parameters after k optimizer steps" WITHOUT the 2*lr escape hatch the atomic mode needs (tests/test_gpu_fused.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


# The switch is an attribute of the model (raw-parameter path) / a field of the settings tuple (drop-in module): there is no
# process-wide mode.
DET = dict(deterministic=True)


def _setup(P=6000, W=208, H=160, seed=9, n_cams=4):
    from w3d_amd.synth import make_scene, make_cameras
    dev = torch.device("cuda:0")
    cams = [c.to(dev) for c in make_cameras(n_cams, W, H)]
    g = torch.Generator().manual_seed(seed)
    for cam in cams:
        cam.original_image = torch.rand(3, H, W, generator=g).to(dev)
    return dev, cams, make_scene(P, seed=seed, scale_mean=0.02)


def _model(sc, dev, deterministic=False):
    from w3d_amd.gaussian_model import GaussianModel, OptimizationParams
    m = GaussianModel(3, device=dev)
    m.deterministic = deterministic
    m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
    m.active_sh_degree = 3
    opt = OptimizationParams()
    m.training_setup(opt)
    return m, opt


def _raw_grads(m, cam, bg, dL, with_da=False):
    from w3d_amd.fused_step import render_raw, finish, backward_raw
    pkg = render_raw(cam, m, bg, 1.0, sync=True)
    assert finish(pkg["handle"])
    H, W = dL.shape[1:]
    dd = torch.full((1, H, W), 1e-3, device=dL.device) if with_da else None
    da = torch.full((1, H, W), -2e-3, device=dL.device) if with_da else None
    gn, m2d = backward_raw(m, pkg["handle"], dL, dd, da, want_means2D=True)
    return m.flat_grad.clone(), m2d.clone()


@pytest.mark.parametrize("with_da", [False, True])
def test_deterministic_backward_matches_default_and_repeats_bit_for_bit(with_da):
    dev, cams, sc = _setup()
    bg = torch.tensor([0.1, 0.0, 0.2], device=dev)
    m, _ = _model(sc, dev)
    dL = torch.randn(3, 160, 208, generator=torch.Generator().manual_seed(3)).to(dev) * 1e-3
    ref, ref2d = _raw_grads(m, cams[1], bg, dL, with_da)
    m.deterministic = True
    a, a2d = _raw_grads(m, cams[1], bg, dL, with_da)
    b, b2d = _raw_grads(m, cams[1], bg, dL, with_da)
    m.deterministic = False
    assert torch.equal(a, b) and torch.equal(a2d, b2d)
    assert float(ref.abs().max()) > 0
    for name, (lo, hi) in m.block_slices().items():
        r = ref[lo:hi]
        err = float((a[lo:hi] - r).abs().max() / (r.abs().max() + 1e-30))
        assert err <= 2e-5, f"{name}: {err:.2e}"           # only the order of the fp32 additions differs
    assert float((a2d - ref2d).abs().max() / ref2d.abs().max()) <= 2e-5
    # culled Gaussians: exactly zero in both modes
    from w3d_amd.fused_step import render_raw
    vis = render_raw(cams[1], m, bg, 1.0, sync=True)["radii"] > 0
    lo, hi = m.block_slices()["xyz"]
    assert float(a[lo:hi].view(-1, 3)[~vis].abs().max()) == 0


def test_deterministic_backward_through_the_dropin_module():
    from w3d_amd.gaussian_renderer import render
    import w3d_amd.gaussian_renderer as gr
    from w3d_amd.train import PipelineParams
    dev, cams, sc = _setup(P=3000)
    bg = torch.tensor([0.0, 0.1, 0.0], device=dev)
    res = []
    for raw in (False, True, False, True):
        gr.RAW_AUTOGRAD = raw
        try:
            m, _ = _model(sc, dev, **DET)
            img = render(cams[0], m, PipelineParams(), bg)["render"]
            (img * torch.linspace(0, 1, img.numel(), device=dev).view_as(img)).sum().backward()
            res.append(torch.cat([p.grad.reshape(-1) for p in m._p.values()]))
        finally:
            gr.RAW_AUTOGRAD = True
    assert torch.equal(res[0], res[2]) and torch.equal(res[1], res[3])


def _train(sc, cams, bg, dev, steps, **kw):
    from w3d_amd.train import Trainer
    m, opt = _model(sc, dev, **DET)
    tr = Trainer(m, cams, opt, bg, densify=False, **kw)
    for k, v in kw.pop("attrs", {}).items():
        setattr(tr, k, v)
    losses = [float(tr.step(it)) for it in range(1, steps + 1)]
    return m, losses


def test_k_steps_repeat_bit_for_bit():
    dev, cams, sc = _setup(P=8000)
    bg = torch.tensor([0.2, 0.1, 0.0], device=dev)
    runs = []
    for _ in range(2):
        m, losses = _train(sc, cams, bg, dev, 6)
        runs.append((m.flat.clone(), m.optimizer.exp_avg.clone(), m.optimizer.exp_avg_sq.clone(),
                     m.xyz_gradient_accum.clone(), losses))
    for x, y in zip(runs[0][:4], runs[1][:4]):
        assert torch.equal(x, y)
    assert runs[0][4] == runs[1][4]


def test_fused_adam_equals_backward_plus_sweep_without_escape_hatch():
    """In the atomic mode this comparison needs 'max diff <= 0.25' (a last-bit sign flip of a ~0 gradient becomes 2*lr under
    Adam).  With deterministic gradients the two optimizer placements see the SAME gradient bits."""
    from w3d_amd.train import Trainer
    dev, cams, sc = _setup(P=8000, seed=11)
    bg = torch.tensor([0.0, 0.1, 0.2], device=dev)
    out = []
    for fused_adam in (False, True):
        m, opt = _model(sc, dev, **DET)
        tr = Trainer(m, cams, opt, bg, densify=False)
        tr.fused_adam = fused_adam
        for it in range(1, 5):
            tr.step(it)
        out.append((m.flat.clone(), m.optimizer.exp_avg.clone(), m.optimizer.exp_avg_sq.clone()))
    for x, y in zip(out[0], out[1]):
        assert torch.equal(x, y)


def test_results_do_not_depend_on_the_tile_walk_hint():
    """w3d_view.tile_walk_hint only orders the blend kernels' tiles (longest walk first within each XCD): the images are
    bit-identical whatever it holds — zeros (first render of a camera), the previous render's lengths, or garbage — and in the
    deterministic mode so are the gradients."""
    from w3d_amd.fused_step import render_raw, backward_raw, finish
    dev, cams, sc = _setup(P=9000, W=400, H=304)            # 25 x 19 tiles: every XCD gets a handful
    bg = torch.tensor([0.05, 0.0, 0.1], device=dev)
    m, _ = _model(sc, dev, **DET)
    cam = cams[2]
    dL = torch.randn(3, 304, 400, generator=torch.Generator().manual_seed(4)).to(dev) * 1e-3
    outs = []
    for fill in ("fresh", "previous", "garbage", "descending"):
        hint = getattr(cam.world_view_transform, "_w3d_tile_walk", None)
        if fill == "fresh":
            assert hint is None
        elif fill == "previous":
            assert hint is not None and int(hint.max()) > 0          # the first render left its walk lengths behind
        elif fill == "garbage":
            hint.copy_(torch.randint(0, 2 ** 31 - 1, hint.shape, generator=torch.Generator().manual_seed(1)).to(dev))
        else:
            hint.copy_(torch.arange(hint.numel(), 0, -1, device=dev, dtype=torch.int32))
        pkg = render_raw(cam, m, bg, 1.0, sync=True)
        assert finish(pkg["handle"])
        backward_raw(m, pkg["handle"], dL)
        outs.append((pkg["render"].clone(), pkg["depth"].clone(), pkg["alpha"].clone(), pkg["radii"].clone(), m.flat_grad.clone()))
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert torch.equal(a, b)
    assert float(outs[0][4].abs().max()) > 0
