"""CPU: bench.py's launch protocol.  `--gpus N` must really run N ranks (it starts them itself when there is no
torch.distributed environment) and report n_gpus = N — never a silent single-rank run; it must refuse a request it
cannot honour.  Uses --dry-run (gloo, stub step: nothing is measured and the line says so)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=240):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=e)


def test_gpus_2_launches_two_ranks_and_reports_them():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                     # ONE line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["config"]["views_per_step"] == 2 and "dry-run" in out["data"]


def test_single_rank_dry_run():
    r = _run(["--steps", "2", "--warmup", "1", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_more_gpus_than_visible_is_refused():
    import torch
    if torch.cuda.device_count() >= 64:
        return
    r = _run(["--gpus", "64", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and "--gpus 64" in r.stderr and not r.stdout.strip()


def test_world_size_mismatch_is_refused():
    r = _run(["--gpus", "8", "--steps", "1", "--warmup", "0", "--dry-run"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and not r.stdout.strip()
