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
    """60 launches per camera of the same view's backward: no gradient element moves by more than 1e-4 of its block's maximum
    (with and without patterned torch.empty, tests/_poison.py)."""
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

