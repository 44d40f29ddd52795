"""tests/depth_grid_model.py (the integer restatement of w3d_binning.hip's depth_grid_kernel that the GPU tests and the probes use to
tell which in-bucket path a scene takes) against the contract the kernels rely on: the buckets PARTITION the view's key interval,
the bucket of a key is monotone in the key, and a sample that sees the distribution keeps the buckets within the LDS capacity."""
import numpy as np
import pytest

from depth_grid_model import BINS, INVALID, grid_buckets, summary


def _keys(depth, visible):
    k = np.asarray(depth, np.float32).view(np.uint32).astype(np.int64)
    k[~visible] = INVALID
    return k


CASES = {
    "uniform": lambda r, n: r.uniform(2.0, 3.0, n),
    "lognormal": lambda r, n: 0.3 + np.exp(1.2 * r.standard_normal(n)),
    "far_background": lambda r, n: np.where(r.random(n) < 0.001, r.uniform(30.0, 60.0, n), r.uniform(2.0, 3.0, n)),
    "bimodal": lambda r, n: np.where(r.random(n) < 0.5, r.uniform(2.0, 2.5, n), r.uniform(40.0, 41.0, n)),
    "thin": lambda r, n: 2.0 + 3e-4 * r.random(n),
    "two_values": lambda r, n: np.where(r.random(n) < 0.5, 2.0, 2.0000005),
}


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("n", [300, 20_000, 400_000])
def test_grid_partitions_the_interval_and_is_monotone(name, n):
    r = np.random.default_rng(len(name) * 1000 + n)
    depth = CASES[name](r, n)
    vis = r.random(n) < 0.6
    vis[:2] = True
    keys = _keys(depth, vis)
    pop, widths, nbk, b = grid_buckets(keys, return_buckets=True)
    kv = keys[vis]
    span = int(kv.max() - kv.min())
    assert pop.sum() == vis.sum() and pop.shape == (BINS,) and 1 <= nbk <= BINS
    assert (pop[nbk:] == 0).all() and (widths[nbk:] == 0).all()
    assert widths[:nbk].min() >= 1 and widths.sum() == span + 1           # a partition of [kmin, kmax]
    order = np.argsort(kv, kind="stable")
    assert (np.diff(b[order]) >= 0).all()                                 # monotone in the key
    # every key lies inside its bucket's interval
    lo = np.concatenate([[0], np.cumsum(widths)[:-1]])
    x = kv - kv.min()
    assert ((x >= lo[b]) & (x < lo[b] + widths[b])).all()


def test_far_background_keeps_buckets_in_lds_at_benchmark_size():
    """2 M Gaussians, 60 % visible, one in a thousand 30-60 units behind the slab: equal-width buckets put the slab into 14 % of
    the buckets (8-20 k keys each); the grid keeps every bucket below the 4096-key LDS capacity."""
    r = np.random.default_rng(7)
    n = 2_000_000
    depth = CASES["far_background"](r, n)
    vis = r.random(n) < 0.6
    keys = _keys(depth, vis)
    pop, widths, _ = grid_buckets(keys)
    kv = keys[vis]
    eq = np.bincount(((kv - kv.min()) * ((1 << 42) // (kv.max() - kv.min() + 1))) >> 32, minlength=BINS)
    assert eq.max() > 8192 and pop.max() <= 4096, (eq.max(), summary(pop, widths))
