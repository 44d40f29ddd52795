"""-m gpu: bench.py's N > 1 code at WORLD SIZE 2 — two rank processes of the unmodified bench.py sharing the one GPU of the
test box.  RCCL refuses two ranks on one device, so a launcher (this file, `--rank`) turns the process group into gloo and
stages the collectives through host memory (tests/test_gpu_two_ranks.py's shim + the two list-form calls bench.py adds),
then runs bench.py as `__main__`.  What a 1-rank group (tests/test_gpu_bench_exchange.py) cannot show and this does: the
max-over-ranks timing, the autotune's identical choice on both ranks, per-rank cameras, the replica check between two REAL
replicas, `value` = world x steps / time, rank 1 printing nothing.  Wire time stays the driver's SCALE run."""
import json
import os
import runpy
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH_ARGS = ["--gpus", "2", "--points", "60000", "--width", "320", "--height", "240", "--steps", "6", "--warmup", "3",
              "--no-cpu-baseline"]


def _rank_main(argv):
    """One rank: gloo + host-staged collectives instead of RCCL, both ranks on cuda:0, then bench.py as it is."""
    for p in (ROOT, os.path.join(ROOT, "wheat-3dgs_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from test_gpu_two_ranks import _install_host_staged_collectives
    real_init, real_gather, real_bcast = dist.init_process_group, dist.all_gather, dist.broadcast

    def init_process_group(backend=None, rank=-1, world_size=-1, device_id=None, **kw):
        real_init("gloo", rank=rank, world_size=world_size, **{k: v for k, v in kw.items() if k == "store"})
        _install_host_staged_collectives()

    def all_gather(out_list, t, **kw):
        hs = [o.detach().cpu() for o in out_list]
        real_gather(hs, t.detach().cpu())
        for o, h in zip(out_list, hs):
            o.copy_(h)

    def broadcast(t, src, **kw):
        h = t.detach().cpu()
        real_bcast(h, src)
        t.copy_(h)
    dist.init_process_group, dist.all_gather, dist.broadcast = init_process_group, all_gather, broadcast
    sys.argv = [os.path.join(ROOT, "bench.py")] + argv
    runpy.run_path(sys.argv[0], run_name="__main__")


def _run(extra):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, WORLD_SIZE="2", RANK=str(r), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--rank"] + BENCH_ARGS + extra, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=900))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-3000:]
    return outs


@pytest.mark.gpu
@pytest.mark.parametrize("exchange", ["auto", "dense"])
def test_bench_with_two_ranks_on_one_gpu(exchange):
    (out0, _), (out1, _) = _run(["--exchange", exchange, "--no-extras"])
    assert out1.strip() == ""                               # rank 1 prints nothing
    lines = out0.splitlines()
    assert len(lines) == 1, out0
    line = json.loads(lines[0])
    ex = line["exchange"]
    assert line["n_gpus"] == 2 and line["steps"] == 6 and line["scaling"] == "weak"
    assert line["config"]["views_per_step"] == 2
    # whole-job value: both ranks' views over the max-over-ranks time
    assert abs(line["value"] - 2 * 1e3 / line["ms_per_step"]) <= 2e-3 * line["value"]
    assert ex["selfcheck"]["replicas_identical_after_warmup"] is True and ex["replicas_identical_after_timed_steps"] is True
    assert ex["selfcheck_ok"] is True
    if exchange == "auto":
        assert set(ex["autotune_ms_per_step"]) == {"rows", "lowrank", "lowrank_early"}
        assert ex["mode"] == min(ex["autotune_ms_per_step"], key=ex["autotune_ms_per_step"].get)
        if ex["mode"] == "rows":
            assert len(ex["rows"]["rows_per_view_last_step"]) == 2 and ex["rows"]["steps_by_form"]["rows"] > 0
    else:
        assert ex["mode"] == "dense"
    for k in ("all_gather_dcolor", "all_reduce_geometry"):
        assert ex[k]["ms"] > 0
    assert "scale_model" not in line                      # (the N-GPU prediction belongs to the single-GPU line)


@pytest.mark.gpu
def test_bench_full_legs_with_two_ranks(tmp_path):
    """--full at N = 2: forward-only render, FlashSplat views and the trained-scene leg run on every rank (max-over-ranks
    clocks, rank 0 alone measures the workload statistics meanwhile); the single-GPU legs (drop-in loop, densified / opaque
    scenes, scale_model) stay out of an N > 1 record.  The line stays compact; the legs' objects go to the detail file."""
    detail = str(tmp_path / "detail.json")
    (out0, _), (out1, _) = _run(["--full", "--trained-steps", "12", "--detail-file", detail])
    assert out1.strip() == ""
    lines = out0.splitlines()
    assert len(lines) == 1 and len(lines[0]) < 4096, out0
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["exchange"]["selfcheck_ok"] is True
    assert line["trained_value"] > 0 and line["render_mpix_per_s"] > 0 and line["flashsplat_views_per_s"] > 0
    assert line["roofline"]["kernel"] and line["roofline"]["frac"] > 0 and line["detail_file"] == detail
    full = json.load(open(detail))
    assert full["value"] == line["value"] and full["trained_scene"]["after_steps"] >= 12
    for k in ("dropin", "densified_scene", "opaque_scene", "scale_model", "cpu_baseline"):
        assert full.get(k) is None, k


@pytest.mark.gpu
def test_bench_default_run_has_no_legs_with_two_ranks(tmp_path):
    """the flags the driver's SCALE run uses: the headline step, the render rate, the exchange object — and nothing else"""
    detail = str(tmp_path / "detail.json")
    (out0, _), (out1, _) = _run(["--detail-file", detail])
    line = json.loads(out0.splitlines()[-1])
    assert out1.strip() == "" and line["n_gpus"] == 2 and line["exchange"]["selfcheck_ok"] is True and line["render_mpix_per_s"] > 0
    full = json.load(open(detail))
    for k in ("trained_scene", "flashsplat_views_per_s", "dropin", "densified_scene", "opaque_scene", "scale_model"):
        assert full.get(k) is None, k


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "--rank":
    _rank_main(sys.argv[2:])
