"""-m gpu: `python bench.py --full` end to end at a reduced size — every leg under bench_legs/ (FlashSplat renders, the unmodified
loop script with and without the import redirect, trained / densified scenes, the scale model on a 1-rank RCCL group, config C1 on
the host) runs, the line stays compact and the legs' objects land in the detail file.  (The full-size run takes three minutes and
is a profiling step — profiles/collect_r06.sh — not a test.)"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_full_bench_at_reduced_size(tmp_path):
    detail = str(tmp_path / "detail.json")
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--full", "--points", "60000", "--width", "320", "--height", "240",
                        "--steps", "6", "--warmup", "3", "--trained-steps", "12", "--densify-iterations", "400", "--opaque-iterations", "0",
                        "--dropin-steps", "6", "--modules-only-steps", "4", "--detail-file", detail],
                       capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and len(lines[0]) < 4096, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["value"] > 0 and line["trained_value"] > 0 and line["densified_value"] > 0 and line["dropin_iters_per_s"] > 0
    assert line["modules_only_iters_per_s"] > 0 and line["flashsplat_views_per_s"] > 0 and line["render_mpix_per_s"] > 0
    assert line["cpu_baseline"]["value"] > 0 and line["parity"]["unattributed_outliers"] == 0 and line["parity"]["radii_differing"] == 0
    full = json.load(open(detail))
    for k in ("trained_scene", "densified_scene", "dropin", "modules_only", "scale_model", "parity_tail", "psnr"):
        assert full.get(k), k
    assert "error" not in full["densified_scene"] and "error" not in full["scale_model"], (full["densified_scene"], full["scale_model"])
    assert "prediction" in full["scale_model"]["untrained"] and "c1" in full["cpu_baseline"]
    assert "oracle_vs_oracle_other_fp32_roundings" in full["parity_tail"] and "attribution" in full["parity_tail"]
