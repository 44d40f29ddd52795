"""-m gpu: bench.py's N > 1 code — exchange autotune (rows vs low-rank vs low-rank with the early colour gather), replica self-check, the `exchange` object of the
JSON line, stdout hygiene — run on a 1-rank RCCL group (`--force-dist`), the only world a 1-GPU box offers.  The collectives
are real RCCL calls; what N = 1 cannot show is wire time (the driver's SCALE run) and disagreement between replicas
(tests/test_gpu_two_ranks.py)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("exchange", ["auto", "dense"])
def test_bench_multi_rank_code_on_one_rank_group(exchange):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--exchange", exchange, "--points", "60000",
                        "--width", "320", "--height", "240", "--steps", "6", "--warmup", "3", "--no-extras", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1, r.stdout                   # the JSON line and nothing else (RCCL's banner goes to stderr)
    out = json.loads(lines[0])
    ex = out["exchange"]
    assert out["n_gpus"] == 1 and ex["selfcheck_ok"] is True and ex["selfcheck"]["replicas_identical_after_warmup"] is True
    if exchange == "auto":
        assert set(ex["autotune_ms_per_step"]) == {"rows", "lowrank", "lowrank_early"} and ex["mode"] in ex["autotune_ms_per_step"]
        assert ex["mode"] == min(ex["autotune_ms_per_step"], key=ex["autotune_ms_per_step"].get)
        if ex["mode"] == "rows":
            assert ex["rows"]["steps_by_form"]["rows"] > 0 and len(ex["rows"]["rows_per_view_last_step"]) == 1
    else:
        assert ex["mode"] == "dense" and "autotune_ms_per_step" not in ex
