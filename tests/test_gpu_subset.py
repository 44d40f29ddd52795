"""-m gpu: flashsplat_render(..., used_mask=obj_used_mask) — the most frequent rasterizer call of run_3d_seg.py
(:130-134 find_match, ~29 views per object mask and refine round; :362, 36 views per mask) — through the raw-parameter fast
path with the mask applied INSIDE the preprocess kernel (w3d_forward_stage1_raw_subset), against

  * the reference's own formulation: the activated parameter blocks gathered with the mask and handed to the drop-in
    rasterizer module (reference gaussian_renderer/__init__.py:151-156,168-170,186-187), and
  * the CPU oracle run on the row subset,

small and at config C4's size (500 k Gaussians, 1600x1200).  Bars: radii exact (raw-parameter caveat of
tests/test_gpu_fullsize.py::check_radii_raw), alpha / colour / depth |dPSNR| <= 1e-3 dB, the `alpha > 0.5` prediction of
find_match identical up to pixels sitting on the threshold, every per-Gaussian output in the reference's SUBSET indexing.
"""
import numpy as np
import pytest
import torch

from util import view_inputs, make_oracle, np_inputs
from w3d_amd.synth import small_test_scene, make_scene, make_cameras

pytestmark = pytest.mark.gpu
KEYS = ("render", "viewspace_points", "visibility_filter", "radii", "alpha", "depth", "contrib_num", "used_count",
        "proj_xy", "gs_depth")


def _model(sc, dev):
    from w3d_amd.gaussian_model import GaussianModel
    m = GaussianModel(3, device=dev)
    m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
    m.active_sh_degree = 3
    return m


def _subset_inputs(sc, cam, sel):
    d = view_inputs(sc, cam, device="cuda:0")      # (torch's device activations: the raw-parameter kernels reproduce their bits)
    return {k: (None if v is None else v[sel].contiguous()) for k, v in d.items()}


def _images(pkg):
    return dict(color=pkg["render"].cpu().numpy(), depth=pkg["depth"].cpu().numpy(), alpha=pkg["alpha"].cpu().numpy())


def test_subset_render_small_matches_reference_formulation_and_oracle():
    from w3d_amd.gaussian_renderer import flashsplat_render
    from w3d_amd.train import PipelineParams
    from test_gpu_parity import check_images
    dev = torch.device("cuda:0")
    P, W, H = 900, 112, 80
    sc, cams = small_test_scene(P=P, W=W, H=H, seed=33)
    m = _model(sc, dev)
    bg = torch.tensor([0.0, 0.0, 0.0], device=dev)
    rng = np.random.RandomState(4)
    for case, sel_np in (("third", np.arange(P) % 3 == 0), ("random", rng.rand(P) < 0.4), ("none", np.zeros(P, bool)),
                         ("all", np.ones(P, bool)), ("one", np.arange(P) == 17)):
        sel = torch.from_numpy(sel_np)
        used = sel.to(dev)
        cam = cams[1].to(dev)
        with torch.no_grad():
            fast = flashsplat_render(cam, m, PipelineParams(), bg, used_mask=used)
            assert hasattr(used, "_w3d_rows")           # the kernel-side path ran (it leaves its cached row list on the mask)
            # an index tensor selects the same rows but is not the fast path's input type: the reference's formulation,
            # gathered activated blocks through the drop-in module
            idx = used.nonzero(as_tuple=True)[0]
            slow = flashsplat_render(cam, m, PipelineParams(), bg, used_mask=idx)
            assert not hasattr(idx, "_w3d_rows")        # ... and this one went through the gathered blocks
        n = int(sel_np.sum())
        assert set(fast) == set(KEYS) == set(slow), case
        for k in KEYS:
            assert fast[k].shape == slow[k].shape and fast[k].dtype == slow[k].dtype, (case, k, fast[k].shape, slow[k].shape)
        assert fast["radii"].shape == (n,) and fast["used_count"].shape == (3, n) and fast["proj_xy"].shape == (n, 2)
        assert fast["viewspace_points"].shape == (P, 3)
        assert torch.equal(fast["radii"], slow["radii"]), case
        assert torch.equal(fast["visibility_filter"], slow["visibility_filter"])
        assert torch.allclose(fast["proj_xy"], slow["proj_xy"], rtol=0, atol=2e-4) and \
            torch.allclose(fast["gs_depth"], slow["gs_depth"], rtol=1e-6, atol=0)
        assert n == 0 or float(fast["used_count"].abs().max()) == 0            # no gt_mask: nothing is scattered
        if n == 0:
            assert float(fast["alpha"].abs().max()) == 0 and float(fast["render"].abs().max()) == 0
            continue
        o = make_oracle(cams[1], (0.0, 0.0, 0.0))
        ref = o.forward(**np_inputs(_subset_inputs(sc, cams[1], sel)))
        o.free()
        np.testing.assert_array_equal(fast["radii"].cpu().numpy(), ref["radii"])
        check_images(_images(fast), ref, f"[subset {case}, fast] ")
        check_images(_images(slow), ref, f"[subset {case}, gathered] ")
        assert abs(int((fast["alpha"] > 0.5).sum()) - int((ref["alpha"] > 0.5).sum())) <= 2


def test_subset_render_with_a_label_mask_scatters_in_subset_indexing():
    """gt_mask and used_mask together (the API allows it): used_count rows follow the subset."""
    from w3d_amd.gaussian_renderer import flashsplat_render
    from w3d_amd.train import PipelineParams
    dev = torch.device("cuda:0")
    P, W, H = 700, 96, 80
    sc, cams = small_test_scene(P=P, W=W, H=H, seed=12)
    m = _model(sc, dev)
    sel = torch.from_numpy(np.random.RandomState(1).rand(P) < 0.5)
    labels = (np.random.RandomState(2).rand(H, W) < 0.4).astype(np.float32)
    with torch.no_grad():
        pkg = flashsplat_render(cams[0].to(dev), m, PipelineParams(), torch.zeros(3, device=dev),
                                gt_mask=torch.as_tensor(labels, device=dev), used_mask=sel.to(dev), obj_num=1)
    o = make_oracle(cams[0], (0.0, 0.0, 0.0))
    ref = o.forward(**np_inputs(_subset_inputs(sc, cams[0], sel)), gt_mask=labels, num_obj=1)
    o.free()
    uc = pkg["used_count"].cpu().numpy()
    assert uc.shape == ref["used_count"].shape == (2, int(sel.sum()))
    assert np.abs(uc - ref["used_count"]).max() <= 1e-4 * ref["used_count"].max()
    np.testing.assert_array_equal(pkg["radii"].cpu().numpy(), ref["radii"])


def test_row_list_follows_in_place_changes_of_the_mask():
    from w3d_amd.gaussian_renderer import _subset_rows
    dev = torch.device("cuda:0")
    mask = torch.zeros(100, dtype=torch.bool, device=dev)
    mask[::10] = True
    r1 = _subset_rows(mask)
    assert r1.tolist() == list(range(0, 100, 10)) and _subset_rows(mask) is r1       # cached on the tensor
    mask[5] = True                                                                    # in-place change: version bump
    assert _subset_rows(mask).tolist() == sorted(list(range(0, 100, 10)) + [5])


@pytest.mark.parametrize("kind", ["wheat_head", "third"])
def test_subset_render_c4_size_against_oracle(kind):
    """Config C4's size.  `wheat_head`: the Gaussians inside a 10-cm ball (what one object mask of run_3d_seg.py selects —
    a few thousand of the scene); `third`: every third Gaussian (the list machinery at scale)."""
    from w3d_amd.gaussian_renderer import flashsplat_render
    from w3d_amd.segmentation import mask_iou_device
    from w3d_amd.train import PipelineParams
    from test_gpu_fullsize import check_images_fullsize, check_radii_raw, NTHREADS, W, H
    dev = torch.device("cuda:0")
    P = 500_000
    sc = make_scene(P, seed=2)
    cams = make_cameras(36, W, H)
    m = _model(sc, dev)
    bg = torch.zeros(3, device=dev)
    if kind == "wheat_head":
        centre = torch.tensor([0.2, -0.1, 0.3])
        sel = ((sc.xyz - centre).norm(dim=1) < 0.1)
        # make the object opaque enough to segment: find_match thresholds alpha at 0.5
        with torch.no_grad():
            m._p["opacity"][sel.to(dev)] = 3.0
        sc.opacity[sel] = 3.0
    else:
        sel = torch.arange(P) % 3 == 0
    n = int(sel.sum())
    assert 300 < n < P
    used = sel.to(dev)
    for vi in (0, 7, 23):
        cam = cams[vi]
        with torch.no_grad():
            pkg = flashsplat_render(cam.to(dev), m, PipelineParams(), bg, used_mask=used)
            assert hasattr(used, "_w3d_rows")           # (kernel-side subset path)
            pkg2 = flashsplat_render(cam.to(dev), m, PipelineParams(), bg, used_mask=used)     # speculative list size now
        assert torch.equal(pkg["alpha"], pkg2["alpha"]) and torch.equal(pkg["radii"], pkg2["radii"])
        o = make_oracle(cam, (0.0, 0.0, 0.0), nthreads=NTHREADS)
        ref = o.forward(**np_inputs(_subset_inputs(sc, cam, sel)))
        o.free()
        assert pkg["radii"].shape == (n,)
        check_radii_raw(pkg["radii"].cpu().numpy(), ref["radii"], f"[C4 subset {kind} view {vi}] ")
        check_images_fullsize(_images(pkg), ref, f"[C4 subset {kind} view {vi}] ")
        # find_match's prediction (run_3d_seg.py:131-134): alpha > 0.5, its bounding box and pixel count — on the device
        _, bbox, n_pred = mask_iou_device(pkg["alpha"], None, 0.5)
        want = ref["alpha"][0] > 0.5
        on_edge = int((np.abs(ref["alpha"][0] - 0.5) < 1e-5).sum())
        assert abs(n_pred - int(want.sum())) <= on_edge, (n_pred, int(want.sum()), on_edge)
        if on_edge == 0 and want.any():
            ys, xs = np.nonzero(want)
            assert bbox == (int(xs.min()), int(ys.min()), int(xs.max()), int(ys.max()))
        if kind == "wheat_head":
            assert want.sum() > 0
