"""GPU, one rank: the view-parallel exchange paths of the fused step on a real RCCL process group (world_size 1 — the only
world a 1-GPU box offers; the 2-rank logic is covered on CPU by tests/test_dist_gloo.py).  Both exchanges must reproduce the
plain single-GPU step: "lowrank" (colour gradients gathered, SH gradient rebuilt by sh_adam_lowrank_kernel, geometry
all-reduced, replicated Adam), "rows" (only the non-zero gradient rows gathered, applied in view order; also with the
per-step fall-back to "lowrank" forced) and "dense" (in-place reduce-scatter, sharded Adam, in-place all-gather)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


@pytest.fixture()
def one_rank_group():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    yield
    dist.destroy_process_group()


def test_exchange_paths_equal_single_gpu_step(one_rank_group):
    from w3d_amd.synth import make_scene, make_cameras
    from w3d_amd.gaussian_model import GaussianModel, OptimizationParams
    from w3d_amd.train import Trainer
    dev = torch.device("cuda:0")
    W, H = 208, 160
    cams = [c.to(dev) for c in make_cameras(4, W, H)]
    g = torch.Generator().manual_seed(5)
    for cam in cams:
        cam.original_image = torch.rand(3, H, W, generator=g).to(dev)
    bg = torch.tensor([0.1, 0.1, 0.0], device=dev)
    sc = make_scene(6999, seed=13, scale_mean=0.02)      # odd: the SH blocks are not 16-B aligned (dword fallback paths)
    runs = {}
    for name in ("single", "lowrank", "lowrank3", "lowrank_early", "rows", "rows_fallback", "rows_overflow", "dense"):
        m = GaussianModel(3, device=dev)
        m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
        m.active_sh_degree = 2
        m.deterministic = True       # (run-to-run reproducible gradients: the sparse and the low-rank form are compared bit for bit)
        opt = OptimizationParams()
        m.training_setup(opt)
        tr = Trainer(m, cams, opt, bg, densify=False, force_exchange=(name != "single"))
        tr.fused_adam = False
        tr.exchange_mode = "lowrank" if name.startswith("lowrank") else "rows" if name.startswith("rows") else name
        tr.rows_max_fraction = 0.0 if name == "rows_fallback" else None   # 0: every step is "too dense" for the sparse form
        tr.lowrank_chunks = 3 if name == "lowrank3" else None      # colour gradients gathered in 3 row chunks
        tr.early_gather = name == "lowrank_early"                  # ... issued between the two halves of the backward
        tr.step(1)
        assert name != "rows" or tr._rows_cap is not None             # later steps size the row collective speculatively
        tr.sync_stats()                     # (visibility counts / radii: tracked per rank, reduced when read)
        one = (m.flat.clone(), m.optimizer.exp_avg.clone(), m.optimizer.exp_avg_sq.clone(), m.xyz_gradient_accum.clone(),
               m.denom.clone(), m.max_radii2D.clone())
        for it in range(2, 5):
            if name == "rows_overflow":
                tr._rows_cap = 256          # a guess far below the real row count: the remainder travels in a second all-gather
            tr.step(it)
        assert m.optimizer.step_count == 4
        if name.startswith("rows"):
            used = tr.exchange_used
            assert (used["rows"], used["lowrank"]) == ((0, 4) if name == "rows_fallback" else (4, 0))
            assert name != "rows_overflow" or used["rows_overflow"] == 3
        if name != "single":
            tr.gather_moments()
        runs[name] = (one, m.flat.clone(), m)
    ref1, ref4, mref = runs["single"]
    # the sparse form adds the same numbers (0 + g == g): bit-identical to the low-rank form on one rank
    for a, b in zip(runs["rows"][0], runs["lowrank"][0]):
        assert torch.equal(a, b)
    for name in ("rows", "rows_fallback", "rows_overflow"):
        assert torch.equal(runs[name][1], runs["lowrank"][1]), name
    for name in ("lowrank", "lowrank3", "lowrank_early", "rows", "dense"):
        one, four, m = runs[name]
        for k, tol in ((1, 2e-4), (2, 4e-4)):           # moments after one step are (1-b1) g and (1-b2) g^2
            for blk, (lo, hi) in m.block_slices().items():
                r = ref1[k][lo:hi]
                err = float((one[k][lo:hi] - r).abs().max() / (r.abs().max() + 1e-30))
                assert err <= tol, f"{name}: moment {k} of {blk}: rel err {err:.2e}"
        d1 = (one[0] - ref1[0]).abs()
        assert float((d1 > 1e-6).float().mean()) <= 2e-3 and float(d1.max()) <= 0.11, name
        assert torch.equal(one[4], ref1[4]) and torch.equal(one[5], ref1[5]), name
        assert float((one[3] - ref1[3]).abs().max() / ref1[3].abs().max()) <= 2e-4, name
        d4 = (four - ref4).abs()
        assert float((d4 > 1e-4).float().mean()) <= 2e-3 and float(d4.max()) <= 0.25, name


@pytest.mark.parametrize("mode", ["lowrank", "rows", "dense"])
def test_densifying_training_under_exchange(one_rank_group, mode):
    """The episodic host logic under the exchange paths: collectives are drained before a densification recycles the
    gradient bucket, the step is skipped / restricted in those iterations exactly as on one GPU, sharded moments (dense)
    are gathered before the resize — and the result matches the plain single-GPU run of the same schedule."""
    import numpy as np
    from w3d_amd.synth import make_scene, make_cameras
    from w3d_amd.gaussian_model import GaussianModel, OptimizationParams
    from w3d_amd.train import Trainer, render_views
    dev = torch.device("cuda:0")
    W, H = 160, 120
    cams = [c.to(dev) for c in make_cameras(6, W, H)]
    bg = torch.zeros(3, device=dev)
    target = make_scene(3000, seed=15, scale_mean=0.04)
    tm = GaussianModel(3, device=dev)
    tm.create_from_tensors(target.xyz, target.features_dc, target.features_rest, target.scaling, target.rotation, target.opacity)
    tm.active_sh_degree = 3
    for cam, img in zip(cams, render_views(tm, cams, bg)):
        cam.original_image = img.clamp(0, 1)

    class Opt(OptimizationParams):
        densify_from_iter = 10
        densification_interval = 10
        opacity_reset_interval = 25
        densify_until_iter = 45
        densify_grad_threshold = 0.00005
    results = {}
    for name in ("single", mode):
        sc = make_scene(1500, seed=16, scale_mean=0.04)
        m = GaussianModel(3, device=dev)
        m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
        m.active_sh_degree = 3
        opt = Opt()
        m.training_setup(opt)
        tr = Trainer(m, cams, opt, bg, densify=True, cameras_extent=2.0, force_exchange=(name != "single"))
        tr.exchange_mode = name
        sizes, losses = [], []
        for it in range(1, 56):
            losses.append(float(tr.step(it)))
            sizes.append(m.num_points)
        tr.gather_moments()
        assert torch.isfinite(m.flat).all() and all(np.isfinite(losses))
        assert 59 * m.num_points <= m.flat.numel() <= 59 * m.num_points + 15 and m.optimizer.exp_avg.numel() == m.flat.numel()
        results[name] = (sizes, losses)
    (s0, l0), (s1, l1) = results["single"], results[mode]
    assert len(set(s0)) > 3
    # same densification decisions while the runs have not drifted apart (first resize), same order of magnitude after
    first = next(i for i in range(1, len(s0)) if s0[i] != s0[i - 1])
    assert s0[:first + 1] == s1[:first + 1]
    assert abs(s0[-1] - s1[-1]) <= 0.05 * s0[-1]
    assert abs(np.mean(l0[-10:]) - np.mean(l1[-10:])) <= 0.02


@pytest.mark.parametrize("P,deg", [(5000, 3), (3001, 1)])
def test_sh_adam_lowrank_kernel_three_views_against_torch(P, deg):
    """sh_adam_lowrank_kernel with several views (what ranks > 1 produce) against the torch formula of the same update on a
    CPU copy of the model: gradient = sum over views of basis(direction to that view's camera) x dcolor, then Adam."""
    from w3d_amd.synth import make_scene
    from w3d_amd.gaussian_model import GaussianModel, OptimizationParams
    from w3d_amd.fused_step import sh_adam_lowrank
    sc = make_scene(P, seed=31, scale_mean=0.02)
    g = torch.Generator().manual_seed(3)
    V = 3
    dcol = torch.randn(V, P, 3, generator=g) * 1e-2
    dcol[torch.rand(V, P, generator=g) < 0.4] = 0.0                 # culled in that view
    campos = torch.randn(V, 3, generator=g) * 3.0 + torch.tensor([0.0, 0.0, 4.0])
    models = []
    for dev in ("cpu", "cuda:0"):
        m = GaussianModel(3, device=dev)
        m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
        m.active_sh_degree = deg
        m.training_setup(OptimizationParams())
        for step in range(2):
            m.optimizer.step_count += 1
            sh_adam_lowrank(m, (dcol * (step + 1)).to(dev), campos.to(dev), skip=(("f_dc",) if step == 1 else ()))
        models.append(m)
    a, b = models
    sl = a.block_slices()
    for name in ("f_dc", "f_rest"):
        lo, hi = sl[name]
        assert torch.allclose(b.flat[lo:hi].cpu(), a.flat[lo:hi], rtol=0, atol=3e-7), name
        assert torch.allclose(b.optimizer.exp_avg[lo:hi].cpu(), a.optimizer.exp_avg[lo:hi], rtol=1e-4, atol=1e-9), name
        assert torch.allclose(b.optimizer.exp_avg_sq[lo:hi].cpu(), a.optimizer.exp_avg_sq[lo:hi], rtol=2e-4, atol=1e-13), name
    for name in ("xyz", "opacity", "scaling", "rotation"):          # untouched blocks
        lo, hi = sl[name]
        assert torch.equal(b.flat[lo:hi].cpu(), a.flat[lo:hi])
    assert float(a.optimizer.exp_avg[sl["f_rest"][0]:sl["f_rest"][1]].abs().max()) > 0


@pytest.mark.parametrize("P,frac", [(10007, 0.07), (4096, 1.0), (513, 0.0)])
def test_gradient_row_kernels_against_torch(P, frac):
    """w3d_pack_gradient_rows / w3d_apply_gradient_rows (the sparse exchange's two kernels): the packed rows are exactly the
    non-zero rows (each once, any order, -0.0 counts as zero), and applying three "views" in order reproduces the dense
    sums bit for bit — against the CPU branch of the same functions and plain torch.  P = 10007: odd row counts, padded blocks."""
    from w3d_amd.fused_step import GEO_BLOCKS, ROW_FLOATS, apply_gradient_rows, pack_gradient_rows
    from w3d_amd.gaussian_model import GaussianModel, OptimizationParams
    from w3d_amd.synth import make_scene
    dev = torch.device("cuda:0")
    sc = make_scene(P, seed=3)
    models = {}
    for d in ("cpu", dev):
        m = GaussianModel(3, device=d)
        m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
        m.training_setup(OptimizationParams())
        models[str(d)] = m
    mc, mg = models["cpu"], models[str(dev)]
    g = torch.Generator().manual_seed(P)
    views = []
    for v in range(3):
        keep = torch.rand(P, generator=g) < frac
        flat = torch.randn(mc.flat.numel(), generator=g)
        dcol = torch.randn(P, 3, generator=g) * keep[:, None]
        gnorm = torch.rand(P, generator=g) * keep
        mc.flat_grad.copy_(flat)
        for n in GEO_BLOCKS:
            mc.grad_view(n).mul_(keep.view(P, *([1] * (mc.grad_view(n).dim() - 1))))
        if frac > 0 and v == 0:
            # a row whose only non-zero value is the norm, one whose only non-zero is a rotation component, a row of -0.0
            z = (~keep).nonzero()[:3, 0] if (~keep).sum() >= 3 else keep.nonzero()[:3, 0]
            for n in GEO_BLOCKS:
                mc.grad_view(n)[z] = 0.0
            dcol[z] = 0.0
            gnorm[z] = 0.0
            gnorm[z[0]] = 0.5
            mc.grad_view("rotation")[z[1], 3] = 1e-30
            mc.grad_view("opacity")[z[2]] = -0.0
        mg.flat_grad.copy_(mc.flat_grad.to(dev))
        rows_c, cnt_c = pack_gradient_rows(mc, dcol, gnorm, norm_scale=2.0)
        rows_g, cnt_g = pack_gradient_rows(mg, dcol.to(dev), gnorm.to(dev), norm_scale=2.0)
        n = int(cnt_c[0])
        assert int(cnt_g[0]) == n
        rc, rg = rows_c[:n], rows_g[:n].cpu()
        order = rg[:, 0].contiguous().view(torch.int32).argsort()
        assert torch.equal(rg[order].view(torch.int32), rc.view(torch.int32))      # same rows, bit for bit (CPU branch is index-ordered)
        views.append((rows_g[:max(n, 1)].contiguous(), cnt_g, dict(dcol=dcol, gnorm=gnorm * 2.0, geo=mc.flat_grad.clone())))
        # capacity smaller than the row count: counted, not written past the end
        small = torch.full((max(n // 2, 1), ROW_FLOATS), 7.0, device=dev)
        cnt = torch.zeros(1, dtype=torch.int32, device=dev)
        from w3d_amd._lib import check, lib, ptr, stream_ptr
        from w3d_amd.fused_step import _geo_grads
        import ctypes
        check(lib.w3d_pack_gradient_rows(P, ptr(dcol.to(dev)), ctypes.byref(_geo_grads(mg)), None, 1.0, ptr(small),
                                         small.shape[0] - (1 if n > 1 else 0), ptr(cnt), stream_ptr(dev)))
        torch.cuda.synchronize()
        if n > 1:
            assert bool((small[-1] == 7.0).all())
    # apply the three views in order on both devices and against dense torch sums in the same order
    nmax = max(r.shape[0] for r, _, _ in views)
    for m, d in ((mc, "cpu"), (mg, dev)):
        m.flat_grad.zero_()
    d_all = {k: torch.zeros(3, P, 3, device=k) for k in ("cpu", dev)}
    nsum = {k: torch.zeros(P, device=k) for k in ("cpu", dev)}
    want_geo = torch.zeros_like(mc.flat_grad)
    want_n = torch.zeros(P)
    sl = mc.block_slices()
    a, b = sl["xyz"][0], sl["rotation"][1]
    for v, (rows, cnt, ref) in enumerate(views):
        pad = torch.zeros(nmax, ROW_FLOATS, device=dev)
        pad[:rows.shape[0]] = rows
        apply_gradient_rows(mg, pad, cnt, nmax, d_all[dev][v], nsum[dev])
        apply_gradient_rows(mc, pad.cpu(), cnt.cpu(), nmax, d_all["cpu"][v], nsum["cpu"])
        want_geo[a:b] += ref["geo"][a:b]
        want_n += ref["gnorm"]
        assert torch.equal(d_all[dev][v].cpu(), ref["dcol"] + 0.0)
    for got in (mg.flat_grad.cpu(), mc.flat_grad):
        for n in GEO_BLOCKS:                             # (block by block: the <= 3 padding floats in front of a block are nobody's)
            lo, hi = sl[n]
            assert torch.equal(got[lo:hi], want_geo[lo:hi] + 0.0), n
        assert float(got[:a].abs().max() if a else 0) == 0 and float(got[b:].abs().max()) == 0      # SH blocks untouched
    assert torch.equal(nsum[dev].cpu(), want_n) and torch.equal(nsum["cpu"], want_n)


@pytest.mark.parametrize("P,deg,skip", [(10007, 3, ()), (4096, 1, ("opacity",)), (700, 2, ("f_dc", "xyz"))])
def test_rows_adam_equals_dense_pipeline_three_views(P, deg, skip):
    """w3d_index_gradient_rows + w3d_rows_norm_sum + w3d_rows_adam (the optimizer step straight from the gathered rows) against
    the dense pipeline — w3d_apply_gradient_rows per view, w3d_sh_adam_lowrank, the Adam sweep over the geometry blocks — on
    three views with overlapping row sets: parameters, both moments and the norm sums must agree BIT FOR BIT, after two
    consecutive steps (second step: moments non-zero, one view empty)."""
    from w3d_amd.fused_step import (GEO_BLOCKS, ROW_FLOATS, SH_BLOCKS, GatheredRows, apply_gradient_rows, pack_gradient_rows,
                                    rows_adam, sh_adam_lowrank)
    from w3d_amd.gaussian_model import GaussianModel, OptimizationParams
    from w3d_amd.synth import make_scene
    dev = torch.device("cuda:0")
    sc = make_scene(P, seed=5)
    ms = []
    for _ in range(3):                       # [0] packs the views' rows, [1] rows_adam, [2] dense pipeline
        m = GaussianModel(3, device=dev)
        m.create_from_tensors(sc.xyz, sc.features_dc, sc.features_rest, sc.scaling, sc.rotation, sc.opacity)
        m.active_sh_degree = deg
        m.training_setup(OptimizationParams())
        ms.append(m)
    packer, ma, mb = ms
    gen = torch.Generator().manual_seed(P + deg)
    campos = (torch.randn(3, 3, generator=gen) * 3.0).to(dev)
    for step in (1, 2):
        rows_v, cnt_v = [], []
        for v in range(3):
            keep = (torch.rand(P, generator=gen) < (0.0 if (step == 2 and v == 1) else 0.3)).to(dev)
            packer.flat_grad.copy_(torch.randn(packer.flat.numel(), generator=gen).to(dev) * 1e-2)
            for n in GEO_BLOCKS:
                gv = packer.grad_view(n)
                gv.mul_(keep.view(P, *([1] * (gv.dim() - 1))))
            dcol = torch.randn(P, 3, generator=gen).to(dev) * 1e-2 * keep[:, None]
            dcol[::7] = 0.0                                                   # rows whose colour gradient is fully clamped
            gnorm = torch.rand(P, generator=gen).to(dev) * keep
            r, c = pack_gradient_rows(packer, dcol, gnorm)
            rows_v.append(r)
            cnt_v.append(c)
        counts = torch.cat(cnt_v)
        cap = max(int(counts.max()), 1) + 5                                   # (slack rows beyond the counts hold garbage)
        rows_all = torch.full((3, cap, ROW_FLOATS), float("nan"), device=dev)
        for v in range(3):
            n = int(counts[v])
            rows_all[v, :n] = rows_v[v][:n]
        # A: indexed rows, one optimizer kernel
        ga = GatheredRows(ma, rows_all, counts)
        nsum_a = ga.norm_sum()
        ma.optimizer.advance(GEO_BLOCKS + SH_BLOCKS, skip)
        rows_adam(ma, ga, campos, skip)
        # B: dense arrays, low-rank SH step, Adam sweep over the geometry blocks
        d_all = torch.zeros(3, P, 3, device=dev)
        nsum_b = torch.zeros(P, device=dev)
        mb.flat_grad.zero_()
        for v in range(3):
            apply_gradient_rows(mb, rows_all[v], counts[v:v + 1], cap, d_all[v], nsum_b)
        mb.optimizer.advance(GEO_BLOCKS + SH_BLOCKS, skip)
        sh_adam_lowrank(mb, d_all, campos, skip=skip)
        mb.optimizer.step(only=GEO_BLOCKS, skip=skip, advance=False, respect_none_grads=False)
        assert torch.equal(nsum_a, nsum_b)
        assert ma.optimizer.steps == mb.optimizer.steps
        for name, x, y in (("flat", ma.flat, mb.flat), ("exp_avg", ma.optimizer.exp_avg, mb.optimizer.exp_avg),
                           ("exp_avg_sq", ma.optimizer.exp_avg_sq, mb.optimizer.exp_avg_sq)):
            assert torch.equal(x.view(torch.int32), y.view(torch.int32)), (step, name)
        assert torch.isfinite(ma.flat).all()
    for n in skip:                                                            # a skipped block was left alone
        lo, hi = ma.block_slices()[n]
        assert float(ma.optimizer.exp_avg[lo:hi].abs().max()) == 0.0
