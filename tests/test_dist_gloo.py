"""CPU, world_size 2 over gloo: the exchange step of the view-parallel loop.
Checks that (a) after reduce-scatter + sharded Adam + parameter all-gather every rank holds exactly
the parameters a single process would get from Adam on the MEAN of the per-view gradients,
(b) the densification statistics are the SUM of per-view norms / visibility and the MAX of radii —
not functions of the averaged gradient, (c) ranks render different cameras, (d) the sharded Adam
moments are made whole before densification and the replicas stay bit-identical afterwards."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tensors(item):
    import numpy as np
    return tuple(torch.from_numpy(x) if isinstance(x, np.ndarray) else x for x in item)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model(P=64):
    for p in (ROOT, os.path.join(ROOT, "wheat-3dgs_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import cpu_twins                    # (spawned workers do not run conftest.py: the torch stand-ins of the kernels, tests/cpu_twins.py)
    cpu_twins.install()
    from w3d_amd.gaussian_model import GaussianModel, OptimizationParams
    from w3d_amd.synth import make_scene
    sc = make_scene(P, seed=0, scale_mean=0.05)
    m = GaussianModel(3, device="cpu")
    m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
    opt = OptimizationParams()
    m.training_setup(opt)
    return m, opt


def _view_grad(rank, n):
    g = torch.Generator().manual_seed(100 + rank)
    return torch.randn(n, generator=g)


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m, opt = _model()
    from w3d_amd.train import Trainer
    P = m.num_points
    tr = Trainer(m, list(range(10)), opt, torch.zeros(3), densify=True)
    cams = [tr.camera_for(it) for it in (1, 2, 3)]
    ok = True
    g = torch.Generator().manual_seed(500 + rank)
    for step in range(3):
        # per-view quantities a rank would have after its own render + backward
        # (exchange()'s contract: the bucket holds the view's gradient already scaled by 1/world)
        m.flat_grad.copy_(_view_grad(rank + 10 * step, m.flat.numel()) / world)
        gnorm = torch.rand(P, generator=g) * 1e-3
        vis = torch.rand(P, generator=g) > 0.4
        radii = (torch.rand(P, generator=g) * 30).to(torch.int32) * vis
        nsum, vcount, rmax = tr.exchange(gnorm, vis, radii)
        tr.wait_stats()
        # visibility counts and radii are NOT exchanged per step: every rank tracks its own views and Trainer.sync_stats()
        # reduces them when something reads them (below, before the densification)
        ok &= vcount is None and rmax is None

        def gather(t):
            lst = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(lst, t)
            return lst
        norms, viss, rads = gather(gnorm), gather(vis.to(torch.float32)), gather(radii)
        ok &= torch.allclose(nsum, sum(n * v for n, v in zip(norms, viss)), atol=1e-9)
        want_count = sum(viss) if step == 0 else want_count + sum(viss)
        want_rmax = torch.stack(rads).max(0).values if step == 0 else torch.maximum(want_rmax, torch.stack(rads).max(0).values)
        m.xyz_gradient_accum += nsum[:, None]
        tr.optimizer_step_and_gather(zero_grad=False, skip=())
    # lock-step densification: whole statistics and moments first, then identical decisions on every rank
    ok &= float(m.denom.abs().max()) == 0.0                     # nothing folded in yet
    tr.sync_stats()
    ok &= torch.equal(m.denom.reshape(-1), want_count) and torch.equal(m.max_radii2D, want_rmax.to(m.max_radii2D.dtype))
    tr.sync_stats()                                             # (idempotent: nothing tracked since)
    ok &= torch.equal(m.denom.reshape(-1), want_count)
    tr.gather_moments()
    moments = m.optimizer.exp_avg.clone()
    torch.manual_seed(1234)
    m.densify_and_prune(2e-4, 0.005, 10.0, None)
    # (numpy payloads: a torch tensor travels as a shared-memory file that vanishes if this process exits first)
    out.put((rank, bool(ok), cams, m.num_points, m.flat.detach().numpy().copy(), moments.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_view_parallel_exchange_two_ranks():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((_tensors(q.get(timeout=180)) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, *_ in res)
    cams0, cams1 = res[0][2], res[1][2]
    assert all(a != b for a, b in zip(cams0, cams1))            # different view per rank in every step
    assert len(set(cams0 + cams1)) == 6                         # and no repetition within the cycle
    assert res[0][3] == res[1][3] > 64                          # replicas densified identically
    assert torch.equal(res[0][4], res[1][4])                    # ... and hold bit-identical parameters
    assert torch.equal(res[0][5], res[1][5])                    # ... and bit-identical (whole) Adam moments

    # single-process reference: Adam on the MEAN of the per-view gradients, three steps
    m, opt = _model()
    for step in range(3):
        m.flat_grad.copy_((_view_grad(0 + 10 * step, m.flat.numel()) + _view_grad(1 + 10 * step, m.flat.numel())) / world)
        m.optimizer.step()
    assert torch.allclose(m.optimizer.exp_avg, res[0][5], atol=1e-7)


def _lowrank_inputs(rank, step, P, nflat):
    """what a rank would hold after its own backward_raw_lowrank: dcolor (P,3) with culled rows, geometry gradients"""
    g = torch.Generator().manual_seed(1000 + 17 * rank + step)
    dcol = torch.randn(P, 3, generator=g) * 1e-2
    dcol[torch.rand(P, generator=g) < 0.3] = 0.0
    return dcol, torch.randn(nflat, generator=g) * 1e-2


def _cams(n=10):
    from types import SimpleNamespace
    g = torch.Generator().manual_seed(77)
    return [SimpleNamespace(camera_center=torch.randn(3, generator=g) * 3.0) for _ in range(n)]


def _zero_culled_geometry(m, culled):
    """A Gaussian a view does not reach has no gradient at all in that view: zero its geometry rows as well."""
    from w3d_amd.fused_step import GEO_BLOCKS
    for n in GEO_BLOCKS:
        m.grad_view(n)[culled] = 0.0


def _densified_model():
    """_model() after one densify_and_prune: every parameter tensor has been re-bound, so every .grad is None — the state the
    exchange paths run in for the rest of a training run (round-3 advisor finding: the CPU twin of sh_adam_lowrank consulted
    .grad and silently skipped f_dc / f_rest from then on)."""
    m, opt = _model()
    g = torch.Generator().manual_seed(31)
    m.xyz_gradient_accum += (torch.rand(m.num_points, 1, generator=g) * 6e-4)
    m.denom += 1.0
    torch.manual_seed(1234)
    m.densify_and_prune(2e-4, 0.005, 10.0, None)
    assert m.num_points > 64 and all(p.grad is None for p in m._p.values())
    return m, opt


def _lowrank_worker(rank, world, port, out, mode):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    densified = mode.endswith("@densified")
    mode = mode.split("@")[0]
    m, opt = _densified_model() if densified else _model()
    m.active_sh_degree = 2
    from w3d_amd.train import Trainer
    P = m.num_points
    tr = Trainer(m, _cams(), opt, torch.zeros(3), densify=True, exchange=mode.split("_")[0],
                 rows_max_fraction=0.2 if mode in ("rows_fallback", "rows_abandon") else None)
    stats = []
    for step in range(1, 4):
        dcol, geo = _lowrank_inputs(rank, step, P, m.flat.numel())
        m.flat_grad.copy_(geo / world)                       # SH blocks of the bucket are ignored by the exchange
        vis = dcol.abs().sum(1) > 0
        _zero_culled_geometry(m, ~vis)
        gnorm = torch.rand(P, generator=torch.Generator().manual_seed(7 * rank + step)) * 1e-3 * vis
        radii = (torch.rand(P, generator=torch.Generator().manual_seed(9 * rank + step)) * 30).to(torch.int32) * vis
        if mode == "rows_overflow" and step > 1:
            tr._rows_cap = 8                                     # far below the ~45 rows of a view: remainder all-gather
        if mode == "rows_abandon" and step > 1:
            tr._rows_skip, tr._rows_cap = 0, 8                   # a speculative collective on a step that turns out too dense
        ex = tr.exchange_rows if mode.startswith("rows") else tr.exchange_lowrank
        nsum, vcount, rmax = ex(dcol / world, gnorm, vis, radii)
        tr.wait_stats()
        assert vcount is None and rmax is None               # tracked per rank, reduced by sync_stats()
        stats.append((nsum.clone().numpy(),))
        tr.optimizer_step_lowrank(step, skip=({"opacity"} if step == 2 else ()))
    tr.sync_stats()
    stats.append((m.denom.numpy().copy(), m.max_radii2D.numpy().copy()))
    out.put((rank, m.flat.detach().numpy().copy(), m.optimizer.exp_avg.numpy().copy(), m.optimizer.exp_avg_sq.numpy().copy(),
             m.optimizer.step_count, stats, dict(tr.exchange_used)))
    dist.barrier()
    dist.destroy_process_group()


def _run_lowrank(mode, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_lowrank_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_rows_exchange_two_ranks_equals_lowrank_bit_for_bit():
    """The sparse form ships only the non-zero gradient rows and adds them in view order: for two ranks that is the same
    sum as the low-rank form's all-reduce (0 + a + b), so parameters, moments and statistics must agree BIT FOR BIT — also
    when a step is too dense for the sparse form and falls back (rows_max_fraction 0.2 with 70 % non-zero rows), when
    the speculative size of the row collective was too small and a second all-gather carries the remainder, and when a
    speculatively sized collective is abandoned because the step turned out too dense (its size is capped at rows_limit)."""
    import numpy as np
    ref = _run_lowrank("lowrank")
    for mode, used in (("rows", {"rows": 3, "lowrank": 0}), ("rows_fallback", {"rows": 0, "lowrank": 3}),
                       ("rows_overflow", {"rows": 3, "lowrank": 0, "rows_overflow": 2}),
                       ("rows_abandon", {"rows": 0, "lowrank": 3, "rows_abandoned": 2})):
        res = _run_lowrank(mode)
        for rank in (0, 1):
            assert res[rank][6] == used
            for k in (1, 2, 3):
                assert np.array_equal(res[rank][k], ref[0][k]), (mode, rank, k)
            for got, want in zip(res[rank][5], ref[0][5]):
                for a, b in zip(got, want):
                    assert np.array_equal(a, b), mode


def test_lowrank_exchange_two_ranks():
    """exchange_lowrank + optimizer_step_lowrank on 2 gloo ranks == one process stepping Adam on the mean over the two views
    of the DENSE gradient (SH gradient = basis(direction to that view's camera) x dcolor), and the replicas stay bit-identical
    although no parameters are ever exchanged."""
    _check_against_single_process("lowrank", _model)


def test_lowrank_and_rows_exchange_after_a_densification():
    """The same check on a model that has been densified first (every .grad None, as in every iteration of a real run after the
    first densify_and_prune), for the low-rank and the sparse form: ALL six blocks must have stepped."""
    _check_against_single_process("lowrank@densified", _densified_model)
    _check_against_single_process("rows@densified", _densified_model)


def _check_against_single_process(mode, make_model):
    world = 2
    res = [_tensors(r) for r in _run_lowrank(mode, world)]
    for k in (1, 2, 3):
        assert torch.equal(res[0][k], res[1][k])
    assert res[0][4] == 3
    # single-process reference
    from w3d_amd.sh import sh_basis
    from w3d_amd.train import Trainer
    m, opt = make_model()
    start = m.flat.detach().clone()
    m.active_sh_degree = 2
    tr = Trainer(m, _cams(), opt, torch.zeros(3), densify=True)
    tr.world = world                                        # only to enumerate the two views of each iteration
    P = m.num_points
    sl = m.block_slices()
    for step in range(1, 4):
        campos = tr.campos_of_all_ranks(step)
        total = torch.zeros_like(m.flat)
        sh = torch.zeros(P, 16, 3)
        for r in range(world):
            dcol, geo = _lowrank_inputs(r, step, P, m.flat.numel())
            m.flat_grad.copy_(geo / world)
            _zero_culled_geometry(m, ~(dcol.abs().sum(1) > 0))
            geo = m.flat_grad.clone() * world
            total += geo / world
            dirs = m._p["xyz"].detach() - campos[r][None]
            dirs = dirs / dirs.norm(dim=1, keepdim=True)
            b = sh_basis(2, dirs)
            sh[:, :9] += b[:, :, None] * (dcol / world)[:, None, :]
        m.flat_grad.copy_(total)
        m.flat_grad[sl["f_dc"][0]:sl["f_dc"][1]] = sh[:, :1].reshape(-1)
        m.flat_grad[sl["f_rest"][0]:sl["f_rest"][1]] = sh[:, 1:].reshape(-1)
        m.optimizer.step(skip=({"opacity"} if step == 2 else ()), respect_none_grads=False)
    for name, (a, b) in sl.items():                          # every block moved (none was silently skipped) ...
        assert float((m.flat[a:b] - start[a:b]).abs().max()) > 0, name
        assert float((res[0][1][a:b] - start[a:b]).abs().max()) > 0, name
    assert m.optimizer.steps == {"xyz": 3, "f_dc": 3, "f_rest": 3, "opacity": 2, "scaling": 3, "rotation": 3}
    assert torch.allclose(m.flat, res[0][1], rtol=0, atol=2e-7)
    assert torch.allclose(m.optimizer.exp_avg, res[0][2], rtol=1e-5, atol=1e-9)
    assert torch.allclose(m.optimizer.exp_avg_sq, res[0][3], rtol=1e-5, atol=1e-12)
