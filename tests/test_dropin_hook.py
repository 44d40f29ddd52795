"""CPU: w3d_amd.dropin's import redirect.

* against the REAL reference checkout where it exists (/root/reference: this container only — nothing of it travels to the
  GPU box): `train_vanilla_3dgs.py`, `run_3d_seg.py` and `render.py` are imported UNMODIFIED in fresh interpreters under each
  of the three ways of switching the redirect on, and the names they bound must be this repo's while every module still comes
  from the checkout;
* on the stand-in checkout (tests/standin_checkout) everywhere: redirected / partially redirected / not redirected;
* the host-side surface the reference's scripts use on the model they get (checkpoint 13-tuple, optimizer groups, deepcopy +
  prune_points(during_training=False) of run_3d_seg.py:327-346) on CPU tensors.
"""
import copy
import json
import os
import subprocess
import sys
import textwrap

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "wheat-3dgs_amd")
REF = "/root/reference"
needs_ref = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "train_vanilla_3dgs.py")),
                               reason="the reference checkout exists in the build container only")

# Third-party modules the reference imports and this image lacks (wandb, plyfile, ffmpeg, torchvision, shapely, ...) are
# replaced by inert stubs — found by retrying on ModuleNotFoundError, never for a module that lives in the checkout.
PROBE = textwrap.dedent('''
    import importlib, json, os, sys, types
    REF = %(ref)r
    class _Anything(types.ModuleType):
        __path__ = []
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            m = _Anything(self.__name__ + "." + k); sys.modules[m.__name__] = m
            return m
        def __call__(self, *a, **k):
            return None
    stubbed = []
    def imp(name):
        while True:
            try:
                return importlib.import_module(name)
            except ModuleNotFoundError as e:
                top = e.name.split(".")[0]
                assert not os.path.exists(os.path.join(REF, top)) and not os.path.exists(os.path.join(REF, top + ".py")), e
                sys.modules[e.name] = _Anything(e.name); stubbed.append(e.name)
                for n in list(sys.modules):
                    f = getattr(sys.modules[n], "__file__", None) or ""
                    if f.startswith(REF):
                        del sys.modules[n]
    %(activate)s
    T, S, R = imp("train_vanilla_3dgs"), imp("run_3d_seg"), imp("render")
    import scene, scene.gaussian_model as G, utils.loss_utils as L, gaussian_renderer as GR, utils.general_utils as U
    from w3d_amd import dropin
    own = lambda o: (getattr(o, "__module__", "") or "")
    print("RESULT " + json.dumps({
        "train": {k: own(getattr(T, k)) for k in ("GaussianModel", "render", "l1_loss", "ssim", "Scene", "psnr")},
        "seg": {k: own(getattr(S, k)) for k in ("GaussianModel", "flashsplat_render", "Scene", "multi_instance_opt")},
        "render": {k: own(getattr(R, k)) for k in ("GaussianModel", "render")},
        "scene_init": own(scene.GaussianModel),
        "files": {m.__name__: m.__file__ for m in (T, S, R, scene, G, L, GR, U)},
        "kept": {"BasicPointCloud": own(G.BasicPointCloud), "l2_loss": own(L.l2_loss), "reference_model": own(G._reference_GaussianModel),
                 "reference_render": own(GR._reference_render), "reference_ssim": own(L._reference_ssim)},
        "distCUDA2": own(G.distCUDA2), "rasterizer": own(GR.GaussianRasterizer), "flash": own(GR.FlashSplat_GaussianRasterizer),
        "status": dropin.status(), "stubbed": stubbed}))
''')


def _check_reference_result(r):
    assert r["train"] == {"GaussianModel": "w3d_amd.gaussian_model", "render": "w3d_amd.gaussian_renderer",
                          "l1_loss": "w3d_amd.loss", "ssim": "w3d_amd.loss", "Scene": "scene", "psnr": "utils.image_utils"}
    assert r["seg"] == {"GaussianModel": "w3d_amd.gaussian_model", "flashsplat_render": "w3d_amd.gaussian_renderer",
                        "Scene": "scene", "multi_instance_opt": "run_3d_seg"}
    assert r["render"] == {"GaussianModel": "w3d_amd.gaussian_model", "render": "w3d_amd.gaussian_renderer"}
    assert r["scene_init"] == "w3d_amd.gaussian_model"
    assert all(f.startswith(REF + "/") for f in r["files"].values()), r["files"]        # every module is still the checkout's
    assert r["kept"] == {"BasicPointCloud": "utils.graphics_utils", "l2_loss": "utils.loss_utils",
                         "reference_model": "scene.gaussian_model", "reference_render": "gaussian_renderer",
                         "reference_ssim": "utils.loss_utils"}
    assert r["distCUDA2"] == "w3d_amd.rasterizer" and r["rasterizer"] == "w3d_amd.rasterizer" and r["flash"] == "w3d_amd.rasterizer"
    assert all(v is True for v in r["status"].values()), r["status"]
    assert not any(s.split(".")[0] in ("scene", "utils", "gaussian_renderer", "arguments", "w3d_amd", "torch") for s in r["stubbed"])


def _run(code, env_extra=None, argv=(), cwd=None):
    env = dict(os.environ)
    env.pop("PYTHONPATH", None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, *argv] if argv else [sys.executable, "-c", code], env=env, cwd=cwd, capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):]), p


@needs_ref
def test_reference_scripts_resolve_to_this_repo_with_install():
    """`import w3d_amd.dropin; w3d_amd.dropin.install()` ahead of the checkout's imports."""
    act = f"sys.path[:0] = [REF, {PKG!r}]\nimport w3d_amd.dropin as d\nd.install()"
    r, _ = _run(PROBE % {"ref": REF, "activate": act})
    _check_reference_result(r)


@needs_ref
def test_reference_scripts_resolve_to_this_repo_with_sitecustomize():
    """PYTHONPATH=<repo>/wheat-3dgs_amd/dropin_site:<repo>/wheat-3dgs_amd — no line of Python added anywhere."""
    act = "sys.path.insert(0, REF)\nassert 'w3d_amd.dropin' in sys.modules, 'sitecustomize did not run'"
    r, _ = _run(PROBE % {"ref": REF, "activate": act},
                {"PYTHONPATH": os.pathsep.join([os.path.join(PKG, "dropin_site"), PKG])}, cwd=REF)
    _check_reference_result(r)


@needs_ref
def test_reference_scripts_resolve_to_this_repo_with_dash_m(tmp_path):
    """python -m w3d_amd.dropin <script>: the script runs as __main__ with its directory first on sys.path."""
    script = tmp_path / "probe_main.py"
    act = "assert __name__ == '__main__' and sys.argv[1:] == ['--flag', '7'], (__name__, sys.argv)\nsys.path.insert(0, REF)"
    script.write_text(PROBE % {"ref": REF, "activate": act})
    r, p = _run(None, {"PYTHONPATH": PKG}, argv=["-m", "w3d_amd.dropin", str(script), "--flag", "7"], cwd=REF)
    _check_reference_result(r)
    assert "[w3d_amd.dropin] redirect installed" in p.stderr


@needs_ref
def test_without_the_redirect_the_reference_keeps_its_own_python():
    """Control: only the rasterizer packages on the path -> the model, render() and the loss are the checkout's."""
    act = f"sys.path[:0] = [REF, {PKG!r}]"
    code = (PROBE % {"ref": REF, "activate": act}).replace('"reference_model": own(G._reference_GaussianModel),', "") \
        .replace('"reference_render": own(GR._reference_render), "reference_ssim": own(L._reference_ssim)', "")
    r, _ = _run(code)
    assert r["train"]["GaussianModel"] == "scene.gaussian_model" and r["train"]["render"] == "gaussian_renderer"
    assert r["train"]["l1_loss"] == "utils.loss_utils" and r["seg"]["flashsplat_render"] == "gaussian_renderer"
    assert r["rasterizer"] == "w3d_amd.rasterizer" and r["distCUDA2"] == "w3d_amd.rasterizer"
    assert all(v is False for v in r["status"].values())


# ------------------------------------------------------------------ stand-in checkout (runs everywhere)
def _owners(mod):
    return {k: getattr(mod, k).__module__ for k in ("GaussianModel", "render", "l1_loss", "ssim")}


def test_standin_checkout_redirected_partially_and_not():
    from util import standin_checkout
    from w3d_amd import dropin
    with standin_checkout(False) as m:
        assert _owners(m) == {"GaussianModel": "scene.gaussian_model", "render": "gaussian_renderer",
                              "l1_loss": "utils.loss_utils", "ssim": "utils.loss_utils"}
        assert not dropin.installed()
    with standin_checkout(True) as m:
        assert _owners(m) == {"GaussianModel": "w3d_amd.gaussian_model", "render": "w3d_amd.gaussian_renderer",
                              "l1_loss": "w3d_amd.loss", "ssim": "w3d_amd.loss"}
        import utils.loss_utils as L
        assert L.l2_loss.__module__ == "utils.loss_utils" and L._reference_ssim.__module__ == "utils.loss_utils"
        assert all(dropin.status().values())
    with standin_checkout(("utils.loss_utils",)) as m:
        assert _owners(m) == {"GaussianModel": "scene.gaussian_model", "render": "gaussian_renderer",
                              "l1_loss": "w3d_amd.loss", "ssim": "w3d_amd.loss"}
    with standin_checkout(("scene.gaussian_model", "gaussian_renderer")) as m:
        assert _owners(m) == {"GaussianModel": "w3d_amd.gaussian_model", "render": "w3d_amd.gaussian_renderer",
                              "l1_loss": "utils.loss_utils", "ssim": "utils.loss_utils"}
    assert not dropin.installed() and "train_loop" not in sys.modules and "scene" not in sys.modules
    with pytest.raises(ValueError):
        dropin.install(only=("utils.general_utils",))


def test_late_install_patches_in_place_and_warns():
    from util import standin_checkout
    from w3d_amd import dropin
    with standin_checkout(False) as m:
        with pytest.warns(UserWarning, match="imported before the redirect"):
            dropin.install()
        import utils.loss_utils as L
        assert L.ssim.__module__ == "w3d_amd.loss"            # the module is patched ...
        assert m.ssim.__module__ == "utils.loss_utils"        # ... a name already bound elsewhere is not (documented)


def test_host_surface_the_reference_scripts_use_on_the_model():
    """What train_vanilla_3dgs.py / run_3d_seg.py do with the object `GaussianModel(sh_degree)` gives them, on CPU tensors:
    restore() from a checkpoint 13-tuple (:38-40), optimizer groups by name, capture() / torch.save (:117-119), deepcopy +
    prune_points(mask=..., during_training=False) + save_ply (run_3d_seg.py:327-347)."""
    from util import checkpoint_tuple
    from w3d_amd.gaussian_model import GaussianModel, OptimizationParams
    from w3d_amd.synth import small_test_scene
    sc, _ = small_test_scene(P=37, W=32, H=32, seed=3)
    opt = OptimizationParams()
    m = GaussianModel(3, device="cpu")
    m.restore(checkpoint_tuple(sc, device="cpu"), opt)
    assert m.active_sh_degree == 3 and len(m.get_xyz) == 37 and m._xyz.shape == (37, 3) and m._features_rest.shape == (37, 15, 3)
    assert [g["name"] for g in m.optimizer.param_groups] == ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"]
    assert m.optimizer.param_groups[0]["params"][0] is m._xyz
    cap = m.capture()
    assert len(cap) == 13 and set(cap[11]) == {"state", "param_groups"} and cap[12] == 1.0
    m2 = GaussianModel(3, device="cpu")
    m2.restore(cap, opt)
    assert torch.equal(m2.flat, m.flat)
    # run_3d_seg.py:326-347
    m._which_object[:11] = 4
    obj = copy.deepcopy(m)
    assert obj._xyz.untyped_storage().data_ptr() == obj.flat.untyped_storage().data_ptr() != m.flat.untyped_storage().data_ptr()
    assert obj.optimizer.model is obj and torch.equal(obj.flat, m.flat)
    obj.prune_points(mask=torch.flatten(obj.get_which_object.detach() != 4), during_training=False)
    assert len(obj.get_xyz) == 11 and len(m.get_xyz) == 37
    assert torch.equal(obj._xyz, m._xyz[:11]) and int((obj.get_which_object == 4).sum()) == 11
    with torch.no_grad():
        obj._xyz.add_(1.0)                                     # the parameters of the copy are views of ITS flat buffer
    a, b = obj.block_slices()["xyz"]
    assert torch.equal(obj.flat[a:b].view(11, 3), obj._xyz.detach()) and not torch.equal(m._xyz[:11], obj._xyz)


# ------------------------------------------------------------------ the reference's own GaussianModel beside ours (CPU)
SIDE_BY_SIDE = textwrap.dedent('''
    import importlib, json, os, sys, types
    import numpy as np, torch
    REF, PKG, ROOT = %(ref)r, %(pkg)r, %(root)r
    # "cuda" allocations of the reference land on the CPU (as in tests/golden/make_golden.py)
    def _cpuify(fn):
        def w(*a, **k):
            if "device" in k and str(k["device"]).startswith("cuda"):
                k["device"] = "cpu"
            return fn(*a, **k)
        return w
    for n in ("zeros", "ones", "zeros_like", "tensor", "empty", "rand"):
        setattr(torch, n, _cpuify(getattr(torch, n)))
    torch.Tensor.cuda = lambda self, *a, **k: self
    class _Anything(types.ModuleType):
        __path__ = []
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            m = _Anything(self.__name__ + "." + k); sys.modules[m.__name__] = m
            return m
        def __call__(self, *a, **k):
            return None
    def imp(name):
        while True:
            try:
                return importlib.import_module(name)
            except ModuleNotFoundError as e:
                top = e.name.split(".")[0]
                assert not os.path.exists(os.path.join(REF, top)) and not os.path.exists(os.path.join(REF, top + ".py")), e
                sys.modules[e.name] = _Anything(e.name)
                for n in list(sys.modules):
                    if (getattr(sys.modules[n], "__file__", None) or "").startswith(REF):
                        del sys.modules[n]
    sys.path[:0] = [REF, PKG, ROOT, os.path.join(ROOT, "tests")]
    import w3d_amd.dropin as d
    d.install()
    import cpu_twins                # (our class on CPU tensors: the torch stand-ins of its kernels, tests/cpu_twins.py)
    cpu_twins.install()
    G = imp("scene.gaussian_model")
    A = imp("arguments")
    from argparse import ArgumentParser
    from oracle.oracle import knn_dist2
    import w3d_amd.rasterizer as wr
    knn = lambda pts: torch.from_numpy(knn_dist2(pts.detach().cpu().numpy().astype(np.float32), nthreads=4))
    G.distCUDA2 = knn              # the reference module's global (its CUDA kNN) ...
    wr.dist2_knn3 = knn            # ... and ours (GPU only): the same exact 3-NN on the CPU for both
    Ref, Ours = G._reference_GaussianModel, G.GaussianModel
    opt = A.OptimizationParams(ArgumentParser())          # the reference's own defaults object
    rs = np.random.RandomState(0)
    P = 300
    pcd = G.BasicPointCloud(points=rs.rand(P, 3).astype(np.float32), colors=rs.rand(P, 3).astype(np.float32), normals=np.zeros((P, 3), np.float32))
    ref, ours = Ref(3), Ours(3, device="cpu")
    ref.create_from_pcd(pcd, 2.5); ours.create_from_pcd(pcd, 2.5)
    ref.training_setup(opt); ours.training_setup(opt)
    names = ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity")
    res = {"create": {n: float((getattr(ref, n) - getattr(ours, n)).abs().max()) for n in names},
           "groups_ref": [g["name"] for g in ref.optimizer.param_groups], "groups_ours": [g["name"] for g in ours.optimizer.param_groups],
           "lr0": [[g["lr"] for g in ref.optimizer.param_groups], [g["lr"] for g in ours.optimizer.param_groups]]}
    g = torch.Generator().manual_seed(1)
    lrs, step_err = [], 0.0
    for it in range(1, 7):
        lrs.append((ref.update_learning_rate(it), ours.update_learning_rate(it)))
        for n in names:
            gr = torch.randn(getattr(ref, n).shape, generator=g) * 1e-3
            getattr(ref, n).grad = gr.clone()
            p = getattr(ours, n)
            p.grad = gr.clone()
        vs = torch.zeros(P, 3); vs.grad = torch.randn(P, 3, generator=g) * 1e-3
        vis = torch.rand(P, generator=g) > 0.3
        ref.add_densification_stats(vs, vis); ours.add_densification_stats(vs, vis)
        if it == 4:
            ref.reset_opacity(); ours.reset_opacity()
        ref.optimizer.step(); ours.optimizer.step()
        ref.optimizer.zero_grad(set_to_none=True); ours.optimizer.zero_grad(set_to_none=True)
        step_err = max(step_err, max(float((getattr(ref, n).detach() - getattr(ours, n).detach()).abs().max()) for n in names))
    res["lr"] = lrs
    res["step_err"] = step_err
    res["stats"] = [float((ref.xyz_gradient_accum - ours.xyz_gradient_accum).abs().max()), float((ref.denom - ours.denom).abs().max())]
    # checkpoints cross the two classes in both directions (train_vanilla_3dgs.py:38-40,117-119)
    cap_ref, cap_ours = ref.capture(), ours.capture()
    res["tuple_lens"] = [len(cap_ref), len(cap_ours)]
    a = Ours(3, device="cpu"); a.restore(cap_ref, opt)
    b = Ref(3); b.restore(cap_ours, opt)
    res["ref_to_ours"] = max(float((getattr(a, n).detach() - getattr(ref, n).detach()).abs().max()) for n in names)
    res["ours_to_ref"] = max(float((getattr(b, n).detach() - getattr(ours, n).detach()).abs().max()) for n in names)
    # ... and keep stepping identically after the switch
    for m_ in (a, b, ref, ours):
        for n in names:
            getattr(m_, n).grad = torch.full(getattr(m_, n).shape, 2e-3)
        m_.optimizer.step()
    res["after_switch"] = [max(float((getattr(a, n).detach() - getattr(ref, n).detach()).abs().max()) for n in names),
                           max(float((getattr(b, n).detach() - getattr(ours, n).detach()).abs().max()) for n in names)]
    res["moments"] = float((a.optimizer.moments()["xyz"][0] - ref.optimizer.state[ref._xyz]["exp_avg"]).abs().max())
    # densify_and_prune (scene/gaussian_model.py:445-459) on both, twice, with steps in between: same rows in the same order
    dens = []
    for rnd in range(2):
        for m_ in (ref, ours):
            gg = torch.Generator().manual_seed(50 + rnd)
            n = m_.get_xyz.shape[0]
            m_.xyz_gradient_accum = torch.rand(n, 1, generator=gg) * 6e-4
            m_.denom = torch.ones(n, 1)
            m_.max_radii2D = torch.rand(n, generator=gg) * 30
            with torch.no_grad():
                m_._opacity[::7] = -8.0                      # some below min_opacity
            torch.manual_seed(77 + rnd)                      # the split samples (torch.normal) draw from the global generator
            m_.densify_and_prune(0.0002, 0.005, 1.0, 20 if rnd else None)
        same = ref.get_xyz.shape[0] == ours.get_xyz.shape[0]
        err = max(float((getattr(ref, n).detach() - getattr(ours, n).detach()).abs().max()) for n in names) if same else -1.0
        mom = float((ours.optimizer.moments()["scaling"][1] - ref.optimizer.state[ref._scaling]["exp_avg_sq"]).abs().max()) if same else -1.0
        dens.append([int(ref.get_xyz.shape[0]), int(ours.get_xyz.shape[0]), err, mom,
                     float((ref.max_radii2D - ours.max_radii2D).abs().max()) if same else -1.0])
        for m_ in (ref, ours):
            for n in names:
                p_ = getattr(m_, n)
                p_.grad = torch.full(p_.shape, 1e-3)
            m_.optimizer.step()
            m_.optimizer.zero_grad(set_to_none=True)
    res["densify"] = dens
    # reset_label (scene/gaussian_model.py:465-506: run_3d_seg.py:326 labels every identified object through it): random label
    # histories and masks, the sequence of return values and the label vectors must be the same
    import contextlib, io
    rl = {"returns_equal": True, "labels_equal": True, "merged": 0, "new_inside_old": 0, "plain": 0}
    for case in range(40):
        gg = torch.Generator().manual_seed(900 + case)
        n = ref.get_xyz.shape[0]
        wo = torch.zeros(n, 1, dtype=torch.int)
        for obj in range(1, 5):                                # earlier objects: random blobs of indices
            lo = int(torch.randint(0, n - 40, (1,), generator=gg))
            wo[lo:lo + int(torch.randint(5, 40, (1,), generator=gg))] = obj
        ref._which_object, ours._which_object = wo.clone(), wo.clone()
        for obj in range(5, 9):
            lo = int(torch.randint(0, n - 60, (1,), generator=gg))
            mask = torch.zeros(n, dtype=torch.bool)
            mask[lo:lo + int(torch.randint(3, 60, (1,), generator=gg))] = True
            mask &= torch.rand(n, generator=gg) < float(torch.rand(1, generator=gg)) * 0.5 + 0.5
            with contextlib.redirect_stdout(io.StringIO()):
                ra = ref.reset_label(obj_used_mask=mask, set_which_object_to=obj)
            rb = ours.reset_label(obj_used_mask=mask, set_which_object_to=obj)
            rl["returns_equal"] &= (ra == rb)
            rl["labels_equal"] &= bool(torch.equal(ref._which_object.reshape(-1), ours._which_object.reshape(-1)))
            rl["merged" if ra is not None else "plain"] += 1
    res["reset_label"] = rl
    res["after_densify"] = max(float((getattr(ref, n).detach() - getattr(ours, n).detach()).abs().max()) for n in names)
    print("RESULT " + json.dumps(res))
''')


@needs_ref
def test_the_references_own_gaussian_model_beside_ours():
    """scene.gaussian_model.GaussianModel AS SHIPPED BY THE REFERENCE (kept as `_reference_GaussianModel` by the redirect) and
    this repo's class through the same calls on the CPU: create_from_pcd, training_setup with the reference's own
    OptimizationParams object, six iterations of update_learning_rate / add_densification_stats / optimizer.step / zero_grad with
    an opacity reset in the middle, checkpoints restored across the two classes in both directions and one more step, then two live
    rounds of densify_and_prune (clone, split with the same random samples, prune by opacity / size) with optimizer steps between."""
    r, _ = _run(SIDE_BY_SIDE % {"ref": REF, "pkg": PKG, "root": ROOT})
    assert all(v == 0.0 for v in r["create"].values()), r["create"]                   # identical initialisation, bit for bit
    assert r["groups_ref"] == r["groups_ours"] == ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"]
    assert r["lr0"][0] == pytest.approx(r["lr0"][1], rel=1e-12)
    assert all(a == pytest.approx(b, rel=1e-12) for a, b in r["lr"])
    assert r["step_err"] <= 1e-6 and r["stats"][0] <= 1e-9 and r["stats"][1] == 0.0, r
    assert r["tuple_lens"] == [13, 13] and r["ref_to_ours"] == 0.0 and r["ours_to_ref"] == 0.0
    assert max(r["after_switch"]) <= 1e-6 and r["moments"] <= 1e-9, r
    for n_ref, n_ours, err, mom, rad in r["densify"]:           # two live densify_and_prune rounds: same rows, same order
        assert n_ref == n_ours > 300 and err <= 2e-6 and mom <= 1e-9 and rad == 0.0, r["densify"]
    assert r["after_densify"] <= 2e-6
    rl = r["reset_label"]                                       # 160 labelling calls, merges into earlier objects among them
    assert rl["returns_equal"] and rl["labels_equal"] and rl["merged"] >= 3 and rl["plain"] >= 50, rl
