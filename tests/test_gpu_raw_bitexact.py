"""-m gpu: the fused raw-parameter path (what bench.py times) must hand the rasterizer the very bits the reference formulation
produces — torch.exp / torch.sigmoid / F.normalize on the device (scene/gaussian_model.py:33-41,101-121) followed by the
rasterizer on the activated tensors (gaussian_renderer/__init__.py:57-84).  At C3 and C2: radii, tile ranges, lists, the
per-Gaussian records, the images and the per-pixel state of w3d_forward_stage1_raw + stage2 are array_equal to those of the
activated API fed model.get_scaling / get_rotation / get_opacity computed by torch on the same device."""
import math

import numpy as np
import pytest
import torch

from w3d_amd.synth import make_scene, make_cameras

pytestmark = pytest.mark.gpu
W, H = 1600, 1200


def _model(sc, dev):
    from w3d_amd.gaussian_model import GaussianModel
    m = GaussianModel(3, device=dev)
    m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
    m.active_sh_degree = 3
    return m


def raw_and_activated(m, cam, bg, list_share=0):
    from w3d_amd.fused_step import render_raw
    from w3d_amd.rasterizer import (GaussianRasterizationSettings, _forward_impl, debug_gaussian_records, debug_pixel_state,
                                    debug_tile_ranges)
    m.list_share = list_share
    with torch.no_grad():
        pkg = render_raw(cam, m, bg, sync=True)
        h = pkg["handle"]
        s = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), bg, 1.0, cam.world_view_transform,
                                          cam.full_proj_transform, m.active_sh_degree, cam.camera_center, False, False, True, False,
                                          list_share)
        color, radii, depth, alpha, saved, _ = _forward_impl(s, m.get_xyz, m.get_features, None, m.get_opacity, m.get_scaling,
                                                             m.get_rotation, None)
    raw = dict(radii=pkg["radii"], color=pkg["render"], depth=pkg["depth"], alpha=pkg["alpha"], R=h["num_rendered"],
               ranges=debug_tile_ranges(h), plist=h["point_list"], rec=debug_gaussian_records(h), px=debug_pixel_state(h))
    act = dict(radii=radii, color=color, depth=depth, alpha=alpha, R=saved["num_rendered"], ranges=debug_tile_ranges(saved),
               plist=saved["point_list"], rec=debug_gaussian_records(saved), px=debug_pixel_state(saved))
    return raw, act


@pytest.mark.parametrize("P,cam_index,name", [(2_000_000, 0, "C3"), (500_000, 5, "C2")])
@pytest.mark.parametrize("list_share", [0, 2])
def test_raw_path_integers_equal_the_activated_api_on_torch_activations(P, cam_index, name, list_share):
    dev = torch.device("cuda:0")
    sc = make_scene(P, seed=0)
    cam = make_cameras(36, W, H)[cam_index].to(dev)
    m = _model(sc, dev)
    raw, act = raw_and_activated(m, cam, torch.zeros(3, device=dev), list_share)
    tag = f"[{name} list_share={list_share}] "
    vis = act["radii"] > 0
    assert 0.3 * P < int(vis.sum()) < P
    assert torch.equal(raw["radii"], act["radii"]), f"{tag}{int((raw['radii'] != act['radii']).sum())} radii differ"
    assert raw["R"] == act["R"] and raw["R"] > P
    assert torch.equal(raw["ranges"], act["ranges"]), f"{tag}tile ranges differ"
    assert torch.equal(raw["plist"][:raw["R"]], act["plist"][:act["R"]]), f"{tag}lists differ"
    # the per-Gaussian records of the visible Gaussians, bit for bit (pixel centre, rect, conic, opacity, colour, depth key)
    a, b = raw["rec"][vis].view(torch.int32), act["rec"][vis].view(torch.int32)
    assert torch.equal(a, b), f"{tag}{int((a != b).any(1).sum())} of {int(vis.sum())} records differ; columns {(a != b).any(0).nonzero().flatten().tolist()}"
    for k in ("color", "depth", "alpha"):
        assert torch.equal(raw[k].view(torch.int32), act[k].view(torch.int32)), f"{tag}{k} differs"
    assert torch.equal(raw["px"][0].view(torch.int32), act["px"][0].view(torch.int32)) and torch.equal(raw["px"][1], act["px"][1])


def test_activations_bit_for_bit_on_adversarial_values():
    """the three activations alone, through the records of a tiny render: quaternions over 60 binades (torch clamps the norm at
    1e-12), scales and logits over the whole range a trained model visits; every Gaussian placed in front of the camera."""
    dev = torch.device("cuda:0")
    n = 200_000
    g = torch.Generator().manual_seed(5)
    sc = make_scene(n, seed=3)
    sc.rotation = torch.randn(n, 4, generator=g) * torch.exp2(torch.randint(-30, 30, (n, 1), generator=g).float())
    sc.rotation[::1000] = 0.0                                   # a zero quaternion: x / max(0, 1e-12) = 0
    sc.scaling = (torch.rand(n, 3, generator=g) * 9.0 - 10.0)   # log-scales in [-10, -1]
    sc.opacity = torch.randn(n, 1, generator=g) * 6.0           # logits far into both tails
    cam = make_cameras(36, W, H)[3].to(dev)
    m = _model(sc, dev)
    raw, act = raw_and_activated(m, cam, torch.zeros(3, device=dev))
    vis = act["radii"] > 0
    assert int(vis.sum()) > 0.2 * n
    assert torch.equal(raw["radii"], act["radii"])
    a, b = raw["rec"][vis].view(torch.int32), act["rec"][vis].view(torch.int32)
    assert torch.equal(a, b), f"{int((a != b).any(1).sum())} records differ; columns {(a != b).any(0).nonzero().flatten().tolist()}"
    assert torch.equal(raw["color"].view(torch.int32), act["color"].view(torch.int32))


@pytest.mark.gpu
@pytest.mark.parametrize("far", [0.0, 0.002])
def test_depth_grid_restatement_equals_the_kernels_grid(far):
    """tests/depth_grid_model.py — which the sort tests use to assert that a scene reaches the large-bucket paths — against the grid
    the library really built (w3d_debug_depth_buckets): bucket populations, bucket intervals and the bucket count agree exactly, on
    an even view and on one whose depth interval a far background stretches."""
    import numpy as np
    from depth_grid_model import grid_buckets
    from w3d_amd.fused_step import render_raw
    from w3d_amd.gaussian_model import GaussianModel
    from w3d_amd.rasterizer import debug_depth_buckets, debug_gaussian_records
    from w3d_amd.synth import make_scene, make_cameras
    dev = torch.device("cuda:0")
    sc = make_scene(300_000, seed=3)
    if far:
        g = torch.Generator().manual_seed(2)
        sel = torch.rand(sc.P, generator=g) < far
        sc.xyz[sel, 2] = -30.0 - 30.0 * torch.rand(int(sel.sum()), generator=g)
    m = GaussianModel(3, device=dev)
    m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
    m.active_sh_degree = 3
    m.sort_spatially()
    m.debug_keep_scratch = True
    cam = make_cameras(6, 640, 480)[1].to(dev)
    with torch.no_grad():
        pkg = render_raw(cam, m, torch.zeros(3, device=dev), sync=True)
    h = pkg["handle"]
    bstart, brange = debug_depth_buckets(h)
    vis = (pkg["radii"] > 0).cpu().numpy()
    keys = debug_gaussian_records(h)[:, 11].contiguous().view(torch.int32).cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    keys[~vis] = 0xFFFFFFFF
    pop, widths, nbk = grid_buckets(keys)
    assert bstart[1024] == vis.sum() and vis.sum() > 100_000
    assert np.array_equal(np.diff(bstart), pop)
    assert np.array_equal(brange[:, 1], widths) and (brange[nbk:] == 0).all()
    assert np.array_equal(brange[:nbk, 0], np.concatenate([[0], np.cumsum(widths[:nbk])[:-1]]))
