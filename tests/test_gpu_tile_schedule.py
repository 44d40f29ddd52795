"""-m gpu: the blend kernels' block -> (tile, part) schedule (w3d_render.hip tile_schedule_kernel) as an
object of its own.  Whatever the walk lengths it is built from — zeros, uniform, heavy-tailed, one giant tile, values that
overflow 32-bit sums, garbage — the schedule must be a PARTITION of the frame: every tile covered exactly once, either whole or
by part-waves whose quadrant sets are disjoint and add up to the tile; entries of a range contiguous in tile index; padding
after the entries.  (A schedule that dropped or doubled a quadrant would still render plausible images on most scenes.)  And the
images must not depend on it."""
import numpy as np
import pytest
import torch

from w3d_amd.synth import make_scene, make_cameras

pytestmark = pytest.mark.gpu
QMASK = {0: 0xF, 1: 0x3, 2: 0xC, 3: 0x1, 4: 0x2, 5: 0x4, 6: 0x8}


def _model(sc, dev):
    from w3d_amd.gaussian_model import GaussianModel
    m = GaussianModel(3, device=dev)
    m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
    m.active_sh_degree = 3
    return m


def check_partition(order, T):
    """order (8, cap) -> (whole tiles, tiles run as part-waves); asserts the partition property"""
    cover = np.zeros(T, np.int64)
    parts = np.zeros(T, np.int64)
    prev_hi = -1
    for x in range(8):
        e = order[x]
        valid = e != 0xFFFFFFFF
        n = int(valid.sum())
        assert valid[:n].all() and not valid[n:].any(), f"range {x}: padding inside the entries"
        tiles, part = e[:n] & 0x1FFFFFFF, e[:n] >> 29
        assert (tiles < T).all() and (part <= 6).all()
        if n:
            assert tiles.min() > prev_hi, f"range {x} overlaps the previous one"         # contiguous, ordered ranges
            prev_hi = int(tiles.max())
        for t, p in zip(tiles, part):
            q = QMASK[int(p)]
            assert cover[t] & q == 0, f"tile {t}: quadrants {q:#x} scheduled twice"
            cover[t] |= q
            parts[t] += 1
    assert (cover == 0xF).all(), f"{int((cover != 0xF).sum())} tiles not fully covered"
    return int((parts == 1).sum()), int((parts > 1).sum())


@pytest.mark.parametrize("W,H,P", [(1600, 1200, 300_000), (400, 304, 9_000), (2048, 1024, 50_000)])
def test_schedule_is_a_partition_for_any_walk_lengths(W, H, P):
    from w3d_amd.fused_step import render_raw, backward_raw, finish
    from w3d_amd.rasterizer import debug_tile_schedule
    dev = torch.device("cuda:0")
    sc = make_scene(P, seed=3)
    cam = make_cameras(36, W, H)[7].to(dev)
    m = _model(sc, dev)
    bg = torch.zeros(3, device=dev)
    T = ((W + 15) // 16) * ((H + 15) // 16)
    g = torch.Generator().manual_seed(9)
    pkg = render_raw(cam, m, bg, sync=True)                      # first render: leaves the camera's hint array behind
    ref = pkg["render"].clone()
    hint = cam.world_view_transform._w3d_tile_walk
    assert hint.numel() == T
    fills = {
        "zeros": torch.zeros(T),
        "uniform": torch.full((T,), 300.0),
        "previous": None,
        "heavy_tail": torch.distributions.Pareto(torch.tensor(50.0), torch.tensor(1.1)).sample((T,)).clamp(max=2e6),
        "one_giant": torch.cat([torch.full((T - 1,), 20.0), torch.tensor([5e6])]),
        "front_loaded": torch.cat([torch.full((T // 16,), 4000.0), torch.zeros(T - T // 16)]),
        "overflowing": torch.full((T,), 3.0e9),
        "garbage": torch.randint(0, 2 ** 32 - 1, (T,), generator=g, dtype=torch.int64).double(),
    }
    seen_split = seen_whole = 0
    for name, v in fills.items():
        if v is not None:
            hint.copy_(v.to(torch.int64).clamp(0, 2 ** 32 - 1).to(torch.uint32).view(torch.int32) if hasattr(torch, "uint32")
                       else v.to(torch.int64).to(torch.int32))
        pkg = render_raw(cam, m, bg, sync=True)
        assert finish(pkg["handle"])
        whole, split = check_partition(debug_tile_schedule(pkg["handle"]), T)
        seen_split += split
        seen_whole += whole
        assert torch.equal(pkg["render"], ref), f"[{name}] the image depends on the schedule"
        # ... and the backward's schedule, built from this forward's walk lengths
        backward_raw(m, pkg["handle"], torch.ones(3, H, W, device=dev) * 1e-3)
        check_partition(debug_tile_schedule(pkg["handle"]), T)
    # both kinds of entries were exercised (a frame with far fewer tiles than the chip has wave slots splits every tile)
    assert seen_split > 0 and (seen_whole > 0 or T < 4096)
