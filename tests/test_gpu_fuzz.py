"""-m gpu: the round-4 differential fuzzers and repeatability probes (profiles/*_probe.py) as regression tests, a few hundred
cases each (seconds): they are what found the idle-lane race of the per-Gaussian backward and the inexact culls on needles.
Each probe runs as a child process (its own torch / library state) and prints one summary line that is asserted here."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, *args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", script), *map(str, args)], cwd=ROOT, env=e,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    return r.stdout


def test_culls_change_nothing_on_random_shapes():
    """tile_cull off vs on, deterministic backward: images AND gradients bit-identical (blobs, needles, pancakes, mixed)."""
    out = _run("fuzz_culls_probe.py", 400, 123)
    assert "cases 400 from seed 123: 0 with differences" in out, out[-3000:]


def test_shared_lists_change_nothing_on_random_shapes():
    """list_share 0 vs 1 vs 2 (tile_cull on, atomic backward): images, radii, final_T and FlashSplat contributor counts
    bit-identical on blobs, needles, pancakes and mixed shapes in random (ragged) frames; gradients of the blobs to atomic noise."""
    out = _run("fuzz_share_probe.py", 300, 555)
    assert "cases 300 from seed 555: 0 with differences" in out, out[-3000:]


def test_integer_work_matches_the_oracle_on_random_shapes():
    """radii, per-tile ranges, depth-ordered lists bit-identical to the oracle's; images for the well-conditioned shapes."""
    out = _run("fuzz_oracle_probe.py", 400, 321)
    assert "cases 400 from seed 321: 0 with differences" in out, out[-3000:]


def test_gaussian_order_does_not_matter():
    out = _run("permutation_probe.py", 3001, 24)
    assert "RESULT ok" in out, out[-3000:]


def test_forward_repeats_bit_for_bit():
    out = _run("repeat_forward_probe.py", 8)
    assert out.strip().splitlines()[-1].strip() == "TOTAL 0", out[-3000:]


@pytest.mark.parametrize("fill", ["none", "small"])
def test_one_views_backward_repeats(fill):
    """60 launches per camera of the same view's backward: no gradient element moves by more than 3e-4 of its block's maximum
    (with and without patterned torch.empty, tests/_poison.py; the bar covers the float-addition order of the atomic mode — up to
    1.8e-4 for one strongly cancelling sum since small frames run their tiles as four quadrant waves — a lost update is of order 1)."""
    out = _run("repeat_view_probe.py", fill, 60)
    assert f"mode={fill} reps=60 anomalies=0" in out, out[-3000:]


def test_backward_matches_the_oracle_on_random_scenes():
    """Every mode of the backward on 300 random well-conditioned scenes: all but a few percent within 2e-4 of every block's
    maximum, the others explained by a single pixel whose pair sits on a blend threshold, or below 1e-2 (a near-cancelling sum)."""
    out = _run("fuzz_grads_probe.py", 300, 777)
    line = [l for l in out.splitlines() if l.startswith("cases 300 from seed 777")][-1]
    n_bad = int(line.split(":")[1].split()[0])
    assert n_bad <= 9, out[-3000:]
    worst = max(float(tok) for tok in line.split("worst per block:")[1].replace(",", " ").split() if tok[0].isdigit())
    assert worst <= 1e-2, out[-3000:]


def test_densify_and_prune_cpu_and_gpu_agree_on_random_models():
    """The CPU path (pinned to the reference's own densify_and_prune by tests/golden/densify.npz) against the HIP compaction on 250
    random models, thresholds from 'nothing selected' to 'everything pruned'."""
    out = _run("fuzz_densify_probe.py", 250, 99)
    assert "cases 250 from seed 99: 0 with differences" in out, out[-3000:]


def test_fused_loss_on_random_shapes():
    """w3d_l1_ssim_fwd_bwd against the torch restatement of utils/loss_utils.py on 36 random shapes — widths and heights from 1 to
    ~400 (smaller than the 11-tap window, ragged against the 32 x 32 tiles), 1 or 3 channels, random lambda: value 2e-6, gradient
    2e-5 of its maximum."""
    import numpy as np
    import torch
    from w3d_amd.loss import photometric_loss, photometric_loss_torch
    rs = np.random.RandomState(5)
    threads = torch.get_num_threads()
    torch.set_num_threads(8)                 # (the CPU restatement's small conv2d calls crawl with one thread per core of a 256-core host)
    try:
        _loss_cases(rs, photometric_loss, photometric_loss_torch)
    finally:
        torch.set_num_threads(threads)


def _loss_cases(rs, photometric_loss, photometric_loss_torch):
    import torch
    for case in range(36):
        C = int(rs.choice([1, 3]))
        H = int(rs.choice([1, 2, 5, 11, 12, 31, 32, 33, 64, 97, int(rs.randint(1, 400))]))
        W = int(rs.choice([1, 3, 10, 11, 21, 32, 33, 63, 65, 130, int(rs.randint(1, 400))]))
        lam = float(rs.choice([0.0, 0.2, 1.0]))
        g = torch.Generator().manual_seed(case)
        gt = torch.rand(C, H, W, generator=g)
        img = (gt + 0.2 * torch.randn(C, H, W, generator=g)).clamp(0, 1)
        if case % 3 == 0:
            img.view(-1)[:: 7] = gt.view(-1)[:: 7]          # exact zeros of |x - y|: sign(0) = 0
        a = img.clone().requires_grad_(True)
        ref = photometric_loss_torch(a, gt, lam)
        ref.backward()
        b = img.cuda().requires_grad_(True)
        out = photometric_loss(b, gt.cuda(), lam)
        out.backward()
        assert abs(float(out.detach()) - float(ref.detach())) <= 2e-6 * max(1.0, abs(float(ref.detach()))), (case, C, H, W, lam)
        gerr = float((b.grad.cpu() - a.grad).abs().max() / (a.grad.abs().max() + 1e-30))
        assert gerr <= 2e-5, (case, C, H, W, lam, gerr)


def test_knn_on_random_point_clouds():
    """distCUDA2 (the device grid kNN) against the oracle's brute force on 40 random clouds: uniform, clustered, a lattice with
    exact duplicates, points on a line, a cloud with one far outlier (one huge grid cell range) — bit-identical."""
    import numpy as np
    import torch
    from simple_knn._C import distCUDA2
    from oracle.oracle import knn_dist2
    rs = np.random.RandomState(11)
    for case in range(40):
        N = int(rs.choice([1, 2, 3, 4, 5, 64, 257, 1000, 3001]))
        g = torch.Generator().manual_seed(case)
        kind = rs.choice(["uniform", "clusters", "lattice", "line", "outlier"])
        if kind == "uniform":
            pts = torch.rand(N, 3, generator=g) * float(rs.choice([1e-3, 1.0, 1e3]))
        elif kind == "clusters":
            c = torch.randn(5, 3, generator=g) * 10
            pts = c[torch.randint(0, 5, (N,), generator=g)] + 0.01 * torch.randn(N, 3, generator=g)
        elif kind == "lattice":
            pts = torch.randint(0, 4, (N, 3), generator=g).float()               # many exact duplicates
        elif kind == "line":
            pts = torch.zeros(N, 3)
            pts[:, 0] = torch.rand(N, generator=g)
        else:
            pts = torch.randn(N, 3, generator=g)
            pts[0] = torch.tensor([1e4, -1e4, 1e4])
        ref = knn_dist2(pts.numpy())
        got = distCUDA2(pts.cuda()).cpu().numpy()
        assert np.array_equal(got, ref), (case, kind, N, float(np.abs(got - ref).max()))


def test_sparse_exchange_kernels_on_random_sizes():
    """pack -> index -> rows_adam against the dense pipeline (apply per view, low-rank SH step, Adam sweep), bit for bit, on
    16 random (P, SH degree, skipped blocks) — P ragged against every granule of the kernels (64 lanes, 256-lane workgroups,
    2048-Gaussian pack groups, 4-float block alignment)."""
    import numpy as np
    from test_gpu_dist import test_rows_adam_equals_dense_pipeline_three_views as one
    rs = np.random.RandomState(3)
    skips = [(), ("opacity",), ("xyz", "f_rest"), ("f_dc", "scaling", "rotation")]
    for case in range(16):
        P = int(rs.choice([1, 2, 63, 65, 255, 257, 2047, 2049, 4095, 6001, int(rs.randint(1, 9000))]))
        one(P, int(rs.randint(4)), skips[int(rs.randint(len(skips)))])

