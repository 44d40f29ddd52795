"""CPU: the ONE line bench.py prints must reach the driver whole.  Round 5's line was 38 KB and the driver's bounded stdout
tail cut off `value`, `roofline` and `config` (BENCH_r05.parsed = null).  bench.compact_line() cuts the line out of the full
record; here it is fed round 5's own full record (profiles/r05/bench_n1.json, every leg present) plus this round's `attribution`
object, and the result must (a) stay under 4 KB, (b) survive a tail cut of 8 KB of stdout, (c) carry every field the contract
and VERDICT r05 item 1 name."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _canned():
    full = json.load(open(os.path.join(ROOT, "profiles", "r05", "bench_n1.json")))
    full["parity_tail"]["attribution"] = {"flipped_pixels": 31, "wobbling_pixels": 2, "flipped_not_on_a_threshold": 0,
                                          "gaussians_blended_there": 4000, "beyond_1e4": 297, "beyond_1e4_blended_at_a_flipped_pixel": 290,
                                          "beyond_1e4_inside_own_rounding_bound": 7, "unattributed_outliers": 0}
    full["detail_file"] = "bench_detail.json"
    return full


def test_compact_line_fits_and_parses_from_a_cut_tail():
    import bench
    full = _canned()
    assert len(json.dumps(full)) > 30_000                      # the record that broke round 5
    line = bench.compact_line(full)
    assert len(line) < 4096 and "\n" not in line
    stdout = "x" * 100_000 + "\n" + line + "\n"                # whatever came before, the driver keeps a bounded tail
    out = json.loads(stdout[-8192:].splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype", "data",
              "config", "roofline", "cpu_baseline", "psnr", "parity", "detail_file"):
        assert k in out, k
    assert out["value"] == full["value"] and out["ms_per_step"] == full["ms_per_step"] and out["vs_baseline"] is None
    assert abs(out["value"] - out["n_gpus"] * 1e3 / out["ms_per_step"]) <= 2e-3 * out["value"]
    assert out["config"]["workload"].startswith("C3") and out["config"]["points"] == 2_000_000 and "model" not in out["config"]
    assert out["config"]["V"] > 0 and out["config"]["R"] > 0 and out["config"]["R_walk"] > 0
    r = out["roofline"]
    assert r["bound"] in ("hbm", "mfma", "valu") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["avg_launch_ms"] > 0 and r["algorithmic_bytes_per_launch"] > 0
    assert "traffic" in r and "kernels" not in r and "probe_stage_ms" not in r
    cb = out["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] in ("port", "reference") and cb["sample"] and "c1" not in cb
    assert abs(out["psnr"]["delta_db"]) <= out["psnr"]["bar_db"]
    p = out["parity"]
    assert p["unattributed_outliers"] == 0 and p["densify_norm"]["p99"] <= 1e-4 and "p999" in p["densify_norm"] and "outliers" in p["densify_norm"]
    assert p["statement"] == "p99 <= 1e-4; tail attributed; reference-CUDA parity unpinned"
    # the legs' scalars ride along, their objects do not
    assert out["trained_value"] == full["trained_value"] and "trained_scene" not in out and "scale_model" not in out


def test_compact_line_with_an_exchange_object_and_without_a_cpu_baseline():
    """the N > 1 line: no cpu_baseline / psnr / parity (rank 0 of a multi-GPU job does not run the oracle), an `exchange` object"""
    import bench
    full = _canned()
    for k in ("cpu_baseline", "psnr", "parity_tail"):
        full[k] = None
    full["n_gpus"] = 8
    full["exchange"] = {"all_gather_dcolor": {"ms": 0.5, "bus_GBps_per_gpu": 300.0}, "all_reduce_geometry": {"ms": 0.7, "bus_GBps_per_gpu": 250.0},
                        "all_gather_rows": {"ms": 0.2, "bus_GBps_per_gpu": 280.0}, "mode": "rows",
                        "selfcheck": {"mode_requested": "rows", "replicas_identical_after_warmup": True},
                        "autotune_ms_per_step": {"rows": 2.0, "lowrank": 2.4, "lowrank_early": 2.3},
                        "rows": {"steps_by_form": {"rows": 200, "lowrank": 0}, "rows_per_view_last_step": [110000] * 8, "row_bytes": 64,
                                 "break_even_rows": 700000},
                        "replicas_identical_after_timed_steps": True, "selfcheck_ok": True}
    line = bench.compact_line(full)
    assert len(line) < 4096
    out = json.loads(line)
    assert out["n_gpus"] == 8 and out["cpu_baseline"] is None and out["psnr"] is None and "parity" not in out
    assert out["exchange"]["selfcheck_ok"] is True and out["exchange"]["mode"] == "rows"
    # an exchange object that outgrows the limit is cut down, never the contract's fields
    full["exchange"]["rows"]["rows_per_view_last_step"] = list(range(100000, 100000 + 600))
    out = json.loads(bench.compact_line(full))
    assert out["value"] == full["value"] and out["roofline"]["frac"] > 0 and out["exchange"]["selfcheck_ok"] is True


def test_committed_lines_of_this_round_are_what_the_contract_asks():
    """The lines bench.py really printed on the GPU box this round (profiles/r06/bench_n1*.json: the default run of each
    collection) — one line each, under 4 KB, with the contract's fields, the roofline and cpu_baseline objects, and the parity
    statement with zero unattributed outliers; and the detail file written next to the newest one is the record it was cut from."""
    import glob
    import bench
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r06", "bench_n1*.json")))
    assert files
    for f in files:
        text = open(f).read().strip()
        assert len(text.splitlines()) == 1 and len(text) < 4096, f
        d = json.loads(text)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                  "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert k in d, (f, k)
        assert d["unit"] == "iters/s" and d["n_gpus"] == 1 and d["vs_baseline"] is None and d["value"] > 100.0      # north star: >= 100
        assert abs(d["value"] * d["ms_per_step"] - 1000.0) < 1.0
        r = d["roofline"]
        assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.0 < r["frac"] < 1.0
        assert d["cpu_baseline"]["kind"] in ("port", "reference") and d["cpu_baseline"]["value"] > 0
        if "parity" in d:
            assert d["parity"]["statement"] == bench.PARITY_STATEMENT
            assert d["parity"].get("unattributed_outliers", 0) == 0
    newest = json.load(open(os.path.join(ROOT, "profiles", "r06", "bench_n1.json")))
    detail = json.load(open(os.path.join(ROOT, "profiles", "r06", "bench_detail.json")))
    assert detail["value"] == newest["value"] and detail["roofline"]["frac"] == newest["roofline"]["frac"]
    assert json.loads(bench.compact_line(detail))["value"] == newest["value"]
