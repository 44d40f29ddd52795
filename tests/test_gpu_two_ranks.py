"""-m gpu: the view-parallel step with world_size 2 and the REAL kernels — two rank processes sharing the one GPU of the
test box.  RCCL refuses two ranks on one device, so the process group is gloo and the three collectives the step uses are
staged through host memory by a shim installed in the worker (test-side only); everything else — per-rank camera choice,
1/world pre-scaling, backward_raw_lowrank / sh_adam_lowrank / the in-place Adam sweep, chunked colour-gradient gather,
the sparse form's row packing / count + row all-gather / in-order application, statistics reduction, the dense
reduce-scatter + sharded Adam + parameter all-gather — is the product code on device tensors.  Asserted for every exchange
mode, across a densification:
  * the two replicas stay BIT-identical (parameters, both moments, statistics) although nothing re-synchronises them;
  * one step equals a single process stepping Adam on the mean of the two views' gradients.

test_rccl_ranks_* is the SAME test over RCCL itself: world 2 (and 4 / 8 where the box has them) rank processes, one GPU each,
backend "nccl", NO shim — the in-place reduce-scatter / all-gather of the dense exchange, the chunked asynchronous
colour-gradient all-gather + geometry all-reduce of the low-rank one and the row all-gather + u8 / int32 statistics
all-reduces of the sparse one run on the real links (config C5's exchange).  It
skips on a box with fewer GPUs than ranks; bench.py's `exchange.selfcheck` repeats the replica check at benchmark size."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, P = 176, 144, 5000


def _scene_and_cams(dev):
    import torch
    from w3d_amd.synth import make_scene, make_cameras
    from w3d_amd.gaussian_model import GaussianModel, OptimizationParams
    cams = [c.to(dev) for c in make_cameras(6, W, H)]
    g = torch.Generator().manual_seed(11)
    for cam in cams:
        cam.original_image = torch.rand(3, H, W, generator=g).to(dev)
    sc = make_scene(P, seed=21, scale_mean=0.03)
    m = GaussianModel(3, device=dev)
    m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
    m.active_sh_degree = 3

    class Opt(OptimizationParams):
        densify_from_iter = 2
        densification_interval = 4
        opacity_reset_interval = 6
        densify_until_iter = 100
        densify_grad_threshold = 0.00002
    opt = Opt()
    m.training_setup(opt)
    return m, opt, cams


def _install_host_staged_collectives():
    """gloo has no device collectives for everything the step uses: stage them through host memory (worker side only)."""
    import torch
    import torch.distributed as dist

    class Done:
        def wait(self):
            return True

    real = dict(ar=dist.all_reduce, ag=dist.all_gather_into_tensor, rs=dist.reduce_scatter_tensor)

    def all_reduce(t, op=dist.ReduceOp.SUM, async_op=False, **kw):
        h = t.detach().cpu()
        real["ar"](h, op=op)
        t.copy_(h)
        return Done()

    def all_gather_into_tensor(out, inp, async_op=False, **kw):
        h = out.detach().cpu()
        real["ag"](h, inp.detach().cpu().clone())
        out.copy_(h)
        return Done()

    def reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, async_op=False, **kw):
        h = out.detach().cpu().clone()
        real["rs"](h, inp.detach().cpu().clone(), op=op)
        out.copy_(h)
        return Done()
    dist.all_reduce, dist.all_gather_into_tensor, dist.reduce_scatter_tensor = all_reduce, all_gather_into_tensor, reduce_scatter_tensor


def worker(rank, world, port, mode, outdir, backend="gloo_staged"):
    for p in (ROOT, os.path.join(ROOT, "wheat-3dgs_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _poison                      # (W3D_TEST_FILL: patterned torch.empty, see tests/_poison.py; a no-op otherwise)
    _poison.install_from_env()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    if backend == "nccl":               # one GPU per rank, RCCL over xGMI, the product's collectives as they are
        dev = torch.device("cuda", rank)
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:                               # both ranks on the one GPU of the box: gloo + host-staged collectives
        dev = torch.device("cuda:0")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        _install_host_staged_collectives()
    from w3d_amd.train import Trainer
    m, opt, cams = _scene_and_cams(dev)
    # "lowrank_early": the colour-gradient all-gather is issued between the two halves of the backward
    # "<mode>_sorted": Trainer(spatial_order=True) — every replica puts its model into Morton order (at construction and at the
    # first densification), computed on each rank from its own copy of the positions: the replicas must pick the same order
    sorted_ = mode.endswith("_sorted")
    ex = mode[:-len("_sorted")] if sorted_ else mode
    tr = Trainer(m, cams, opt, torch.zeros(3, device=dev), densify=True, cameras_extent=2.0,
                 exchange="lowrank" if ex.startswith("lowrank") else ex, early_gather=ex == "lowrank_early",
                 rows_max_fraction=1.0,   # (small dense test views: take the sparse form however many rows a view has)
                 spatial_order=sorted_)
    if sorted_:
        tr.SPATIAL_ORDER_EVERY = m.spatial_order_every = 1
    assert tr.world == world and tr.rank == rank
    snaps = {}
    for it in range(1, 8):              # densifies at iteration 4, resets the opacities at iteration 6
        tr.step(it)
        if it in (1, 7):
            tr.gather_moments()
            tr.sync_stats()         # (visibility counts / radii are tracked per rank and reduced when read)
            snaps[it] = dict(flat=m.flat.detach().cpu().numpy(), m=m.optimizer.exp_avg.cpu().numpy(),
                             v=m.optimizer.exp_avg_sq.cpu().numpy(), accum=m.xyz_gradient_accum.cpu().numpy(),
                             denom=m.denom.cpu().numpy(), radii=m.max_radii2D.cpu().numpy(), P=np.array(m.num_points))
    used = np.array([tr.exchange_used.get("rows", 0), tr.exchange_used.get("lowrank", 0)])
    np.savez(os.path.join(outdir, f"rank{rank}_{mode}.npz"), used=used, **{f"{k}_{it}": v for it, s in snaps.items() for k, v in s.items()})
    dist.barrier()
    dist.destroy_process_group()


def _run_ranks(world, mode, backend, tmp_path):
    """Start `world` rank processes as fresh children (nothing of this process's GPU state is inherited: they are new
    interpreters, not forks) and return their snapshots."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([ROOT, os.path.join(ROOT, "wheat-3dgs_amd"), os.path.join(ROOT, "tests")])
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(r), str(world), str(port), mode,
                               str(tmp_path), backend],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=900)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return [np.load(os.path.join(tmp_path, f"rank{r}_{mode}.npz")) for r in range(world)]


def _check_replicas_and_mean_gradient_step(snaps, world, mode):
    import torch
    a = snaps[0]
    for r, b in enumerate(snaps[1:], 1):
        for k in a.files:                # replicas bit-identical, before and after the densification and the opacity reset
            assert np.array_equal(a[k], b[k]), f"{mode}: rank {r} differs from rank 0 in {k}"
    assert int(a["P_7"]) != P            # the schedule really densified
    if mode in ("rows", "rows_sorted"):  # ... and every iteration went through the sparse form (none was too dense)
        assert a["used"].tolist() == [7, 0], a["used"]

    # single-process reference of step 1: Adam on the MEAN of the ranks' views' gradients, statistics summed / maxed
    from w3d_amd.fused_step import backward_raw, render_raw
    from w3d_amd.fused import l1_ssim_fwd_bwd
    from w3d_amd.train import Trainer
    dev = torch.device("cuda:0")
    m, opt, cams = _scene_and_cams(dev)
    tr = Trainer(m, cams, opt, torch.zeros(3, device=dev), densify=False, spatial_order=mode.endswith("_sorted"))
    m.update_learning_rate(1)
    n = len(cams)
    total = torch.zeros_like(m.flat_grad)
    nsum = torch.zeros(P, device=dev)
    vcount = torch.zeros(P, device=dev)
    rmax = torch.zeros(P, device=dev)
    with torch.no_grad():
        for r in range(world):
            cam = cams[tr.perm[(0 * world + r) % n]]
            pkg = render_raw(cam, m, tr.bg, sync=True)
            _, dimg = l1_ssim_fwd_bwd(pkg["render"], cam.original_image, opt.lambda_dssim)
            gnorm, _ = backward_raw(m, pkg["handle"], dimg, want_norm=True)
            total += m.flat_grad
            nsum += gnorm
            vcount += (pkg["radii"] > 0).float()
            rmax = torch.max(rmax, pkg["radii"].float())
        m.flat_grad.copy_(total / world)
        m.optimizer.step(respect_none_grads=False)
    for blk, (lo, hi) in m.block_slices().items():
        ref_m = m.optimizer.exp_avg[lo:hi].cpu().numpy()
        err = np.abs(a["m_1"][lo:hi] - ref_m).max() / (np.abs(ref_m).max() + 1e-30)
        assert err <= 2e-4, f"{mode}: exp_avg of {blk} after one step: rel err {err:.2e}"
        ref_v = m.optimizer.exp_avg_sq[lo:hi].cpu().numpy()
        err = np.abs(a["v_1"][lo:hi] - ref_v).max() / (np.abs(ref_v).max() + 1e-30)
        assert err <= 4e-4, f"{mode}: exp_avg_sq of {blk} after one step: rel err {err:.2e}"
    d = np.abs(a["flat_1"] - m.flat.detach().cpu().numpy())
    assert (d > 1e-6).mean() <= 2e-3 and d.max() <= 0.11       # (a sign flip of a ~0 gradient moves a parameter by 2 lr)
    assert np.array_equal(a["denom_1"].reshape(-1), vcount.cpu().numpy())
    assert np.array_equal(a["radii_1"], rmax.cpu().numpy())
    e = np.abs(a["accum_1"].reshape(-1) - nsum.cpu().numpy()).max() / nsum.abs().max().item()
    assert e <= 2e-4, f"{mode}: summed gradient norms rel err {e:.2e}"


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["rows", "lowrank", "lowrank_early", "dense", "rows_sorted", "dense_sorted"])
def test_two_ranks_one_gpu_replicas_identical_and_equal_mean_gradient_step(mode, tmp_path):
    _check_replicas_and_mean_gradient_step(_run_ranks(2, mode, "gloo_staged", tmp_path), 2, mode)


def _gpus():
    import torch
    return torch.cuda.device_count()        # (does not initialise the GPU)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["rows", "lowrank", "lowrank_early", "dense", "rows_sorted"])
@pytest.mark.parametrize("world", [2, 4, 8])
def test_rccl_ranks_replicas_identical_and_equal_mean_gradient_step(world, mode, tmp_path):
    """Config C5's exchange on the real links: `world` ranks, one GPU each, backend nccl (= RCCL), no shims."""
    if _gpus() < world:
        pytest.skip(f"needs {world} GPUs, this box has {_gpus()}")
    _check_replicas_and_mean_gradient_step(_run_ranks(world, mode, "nccl", tmp_path), world, mode)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "--worker":
    worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6], sys.argv[7] if len(sys.argv) > 7 else "gloo_staged")
