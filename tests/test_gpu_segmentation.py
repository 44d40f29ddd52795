"""-m gpu: SURVEY.md §8f row N4 on the device — the per-view mask work of run_3d_seg.py / eval_wheatgs.py around the
FlashSplat render (binarise, alpha > 0.5 -> bounding box -> IoU), multi_instance_opt on GPU tensors against the
reference's golden vector, PLY round trip of a GPU model, counts accumulated inside the kernel, and tiles with more labels
than the scatter's fast path holds."""
import os

import numpy as np
import pytest
import torch

from util import view_inputs, make_oracle, np_inputs
from w3d_amd.synth import make_scene, make_cameras, small_test_scene

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _ref_binarize(a):
    """utils/wheatgs_utils.py:14-37: PILtoTorch (normalised to [0,1]) then > 0 / any over channels."""
    a = np.asarray(a, np.float32)
    return (a > 0).astype(np.float32) if a.ndim == 2 else (a > 0).any(axis=2).astype(np.float32)


def test_binarize_mask_on_the_device():
    from w3d_amd.segmentation import binarize_mask_device
    rng = np.random.RandomState(0)
    for shape in ((75, 133), (48, 64, 3), (1200, 1600), (30, 20, 1)):
        a = (rng.rand(*shape) < 0.3) * rng.randint(1, 256, size=shape)
        a = a.astype(np.uint8)
        got = binarize_mask_device(a).cpu().numpy()
        want = _ref_binarize(a if a.ndim == 2 or a.shape[2] != 1 else a[:, :, 0])
        assert got.shape == want.shape and np.array_equal(got, want)
        assert set(np.unique(got)) <= {0.0, 1.0}
    with pytest.raises(ValueError):
        binarize_mask_device(np.zeros((4, 4, 2), np.uint8))


def test_mask_iou_and_bbox_on_the_device():
    """against utils/wheatgs_utils.py:45-53 get_bbox_from_mask and :94-103 calculate_seg_iou, restated in numpy"""
    from w3d_amd.segmentation import mask_iou_device
    rng = np.random.RandomState(1)
    H, W, K = 301, 517, 9
    alpha = rng.rand(H, W).astype(np.float32)
    alpha[:40] = 0.0
    alpha[:, 500:] = 0.2
    masks = rng.rand(K, H, W) < np.linspace(0.05, 0.9, K)[:, None, None]
    masks[3] = False                                    # an empty mask: union = pred, IoU 0
    iou, bbox, n_pred = mask_iou_device(torch.from_numpy(alpha).cuda()[None], torch.from_numpy(masks).cuda())
    pred = alpha > 0.5
    ys, xs = np.nonzero(pred)
    assert bbox == (xs.min(), ys.min(), xs.max(), ys.max()) and n_pred == int(pred.sum())
    for k in range(K):
        u = np.logical_or(masks[k], pred).sum()
        want = np.logical_and(masks[k], pred).sum() / u if u > 0 else 0.0
        assert abs(float(iou[k]) - want) < 1e-12
    # nothing above the threshold: no box, every IoU is |empty| / |mask|
    iou0, bbox0, n0 = mask_iou_device(torch.zeros(H, W, device="cuda"), torch.from_numpy(masks).cuda())
    assert bbox0 is None and n0 == 0 and float(iou0.max()) == 0.0
    iou1, bbox1, _ = mask_iou_device(torch.ones(7, 5, device="cuda"))      # K = 0
    assert iou1.numel() == 0 and bbox1 == (0, 0, 4, 6)


def test_mask_work_matches_the_references_own_helpers():
    """tests/golden/masks.npz holds what utils/wheatgs_utils.py ITSELF returns (PILtoTorch + binarize_mask; alpha > 0.5 ->
    get_bbox_from_mask, calculate_seg_iou — tests/golden/make_golden_masks.py imports them): the device kernels must agree
    exactly (bbox, pixel count, binarised masks) and to 1e-12 (IoU)."""
    from w3d_amd.segmentation import binarize_mask_device, mask_iou_device
    z = np.load(os.path.join(GOLD, "masks.npz"))
    for i in range(3):
        got = binarize_mask_device(z[f"pixels{i}"]).cpu().numpy()
        assert got.shape == z[f"binary{i}"].shape and np.array_equal(got, z[f"binary{i}"]), i
    iou, bbox, n_pred = mask_iou_device(torch.from_numpy(z["alpha"]).cuda(), torch.from_numpy(z["masks"]).cuda())
    assert bbox == tuple(int(v) for v in z["bbox"]) and n_pred == int(z["n_pred"])
    assert np.abs(iou.numpy() - z["iou"]).max() < 1e-12
    assert bool(z["bbox_empty_is_none"]) and float(z["iou_empty_union"]) == 0.0
    iou0, bbox0, n0 = mask_iou_device(torch.zeros(4, 4, device="cuda"), torch.zeros(1, 4, 4, dtype=torch.bool, device="cuda"))
    assert bbox0 is None and n0 == 0 and float(iou0[0]) == 0.0           # the same two edge cases on the device


def test_multi_instance_opt_on_gpu_tensors_matches_the_reference_golden():
    from w3d_amd.segmentation import multi_instance_opt
    z = np.load(os.path.join(GOLD, "multi_instance_opt.npz"))
    for counts, gamma, labels in (("counts2", 0.0, "labels2"), ("countsK", 0.0, "labelsK"), ("countsK", 0.2, "labelsK_g")):
        got = multi_instance_opt(torch.from_numpy(z[counts]).cuda(), gamma)
        assert got.is_cuda and np.array_equal(got.cpu().numpy(), z[labels].astype(bool)), (counts, gamma)


def test_ply_round_trip_of_a_gpu_model(tmp_path):
    from w3d_amd.gaussian_model import GaussianModel
    sc = make_scene(1234, seed=3)
    m = GaussianModel(3, device="cuda")
    m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
    m._which_object[::7] = 5
    path = os.path.join(tmp_path, "point_cloud.ply")
    m.save_ply(path)
    m2 = GaussianModel(3, device="cuda")
    m2.load_ply(path)
    assert m2.flat.is_cuda and m2.num_points == 1234 and m2.active_sh_degree == 3
    assert torch.equal(m2.flat, m.flat)                 # every parameter bit for bit
    assert torch.equal(m2.get_which_object, m.get_which_object)
    # the attribute order / channel-major SH layout of reference scene/gaussian_model.py:196-237
    head = open(path, "rb").read(4000).split(b"end_header")[0].decode()
    props = [ln.split()[-1] for ln in head.splitlines() if ln.startswith("property")]
    assert props[:6] == ["x", "y", "z", "nx", "ny", "nz"] and props[6:9] == ["f_dc_0", "f_dc_1", "f_dc_2"]
    assert props[9] == "f_rest_0" and props[-1] == "which_object" and len(props) == 63


def test_counts_accumulated_inside_the_kernel_equal_the_sum_of_views():
    from w3d_amd.gaussian_model import GaussianModel
    from w3d_amd.gaussian_renderer import flashsplat_render
    from w3d_amd.segmentation import accumulate_counts, accumulate_counts_raw
    from w3d_amd.train import PipelineParams
    dev = torch.device("cuda:0")
    W, H, P, K = 208, 160, 6000, 3
    sc = make_scene(P, seed=8, scale_mean=0.03)
    cams = [c.to(dev) for c in make_cameras(5, W, H)]
    m = GaussianModel(3, device=dev)
    m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
    m.active_sh_degree = 3
    yy, xx = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
    masks = [(((xx // 37) + (yy // 29) + i) % (K + 1)).float() for i in range(len(cams))]
    bg = torch.zeros(3, device=dev)
    a = accumulate_counts(lambda cam, mk: flashsplat_render(cam, m, PipelineParams(), bg, gt_mask=mk, obj_num=K), cams, masks, K)
    b = accumulate_counts_raw(m, cams, masks, bg, K)
    assert a.shape == b.shape == (K + 1, P)
    assert float((a - b).abs().max()) <= 1e-5 * float(a.max())       # (float atomics: the order of the additions differs)


def test_tiles_with_more_labels_than_the_fast_path_holds():
    """A label image that changes every 3 pixels puts ~25 labels into every tile: the scatter's fallback (one wave reduction
    per label and entry) and the 1..4-label row-sum path must both reproduce the oracle."""
    from flashsplat_rasterization import GaussianRasterizer
    from test_gpu_parity import _settings
    from util import rel_err
    dev = torch.device("cuda:0")
    P, W, H = 500, 80, 64
    sc, cams = small_test_scene(P=P, W=W, H=H, seed=33)
    cam, bg = cams[0], (0.0, 0.0, 0.0)
    d = view_inputs(sc, cam)
    yy, xx = np.mgrid[0:H, 0:W]
    for K, mask in ((30, ((xx // 3) + 5 * (yy // 3)) % 31), (3, (xx // 10 + yy // 20) % 4), (2, (xx > 37).astype(int) * 2)):
        mask = mask.astype(np.float32)
        o = make_oracle(cam, bg)
        ref = o.forward(**np_inputs(d), gt_mask=mask, num_obj=K)
        o.free()
        rast = GaussianRasterizer(_settings(cam, bg, 3, 1.0, dev, flash=K))
        t = {k: (None if v is None else v.to(dev)) for k, v in d.items()}
        outs = rast(gt_mask=torch.as_tensor(mask, device=dev), unique_label=None, means3D=t["means3D"],
                    means2D=torch.zeros(P, 3, device=dev), shs=t["shs"], colors_precomp=None, opacities=t["opacities"],
                    scales=t["scales"], rotations=t["rotations"], cov3D_precomp=None)
        used_count = outs[5].cpu().numpy()
        assert used_count.shape == (K + 1, P)
        e = rel_err(used_count, ref["used_count"])
        assert e <= 1e-4, f"K={K}: used_count rel err {e:.2e}"
        assert abs(used_count.sum() - outs[3].sum().item()) <= 1e-3 * outs[3].sum().item()
