"""The training step around the rasterizer — what "train iters/sec" measures.

One step = the loop body of reference train_vanilla_3dgs.py:55-115: LR schedule, SH degree
ramp, pick a camera, render(), 0.8*L1 + 0.2*(1-SSIM), backward, densification statistics,
(densify / prune / opacity reset on their schedule), Adam step, zero_grad.

View-parallel data parallelism (the reference itself is single-GPU): one process
per GPU, every rank holds a full replica of the Gaussians and renders a DIFFERENT camera of the
same step; then
  * the densification statistics are exchanged BEFORE gradients are averaged — they are sums of
    per-view gradient NORMS, visibility counts and a max of radii (scene/gaussian_model.py:461-463,
    train_vanilla_3dgs.py:102-103), not functions of the averaged gradient;
  * the gradients are averaged over the views — fused step: 14 floats per Gaussian and view cross the links (colour
    gradient + 11 geometry gradients; the SH gradient is rebuilt on every rank), as packed rows of only the Gaussians the
    view gave a gradient to (exchange_rows, default) or densely (exchange_lowrank), and the optimizer is replicated;
    autograd step / exchange="dense": the 59 x P fp32 bucket is reduce-scattered, Adam sharded, parameters all-gathered;
  * densify/prune then run identically on every rank (same statistics, same RNG seed for the
    split samples) so the replicas stay in lock-step without a parameter broadcast.
"""
import torch
import torch.distributed as dist

from .gaussian_renderer import render
from .loss import photometric_loss


class PipelineParams:
    """Defaults of reference arguments/__init__.py:64-69."""
    convert_SHs_python = False
    compute_cov3D_python = False
    debug = False


def dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def nccl_inplace_shard(full, lo, hi, rank, world):
    """NCCL / RCCL document exactly one aliasing of send and receive buffers as legal: reduce-scatter with
    recvbuff == sendbuff + rank * recvcount, all-gather with sendbuff == recvbuff + rank * sendcount.  Returns the shard view
    of `full` if [lo, hi) is that slot (the collective then runs in place, no staging copy of the bucket), else None — the
    caller stages through a temporary.  A checked invariant instead of an assumption about how the buffers were sliced."""
    n = hi - lo
    if n <= 0 or full.numel() != world * n or lo != rank * n or not full.is_contiguous():
        return None
    shard = full[lo:hi]
    if shard.data_ptr() != full.data_ptr() + rank * n * full.element_size():
        return None
    return shard


class _LazyVisible:
    """`radii > 0`, evaluated only if somebody multiplies by it (the too-dense fallback of the sparse exchange)."""

    def __init__(self, radii):
        self._radii, self._v = radii, None

    def _get(self):
        if self._v is None:
            self._v = self._radii > 0
        return self._v

    def __rmul__(self, other):
        return other * self._get()

    def __mul__(self, other):
        return self._get() * other


class Trainer:
    ROWS_RETRY = 16          # exchange="rows": steps in the low-rank form after one that was too dense, before rows are counted again
    ROWS_WINDOW = 48         # ... and the speculative size of the row collective follows the largest count of this many steps
    # (w3d_view.list_share of the fused step is chosen from the walked fraction the trainer measures: rasterizer.adapt_list_share)
    SHARE_PROBE_EVERY = 64

    def __init__(self, model, cameras, opt, background, pipe=None, cameras_extent=1.0, seed=0,
                 densify=True, loss_fn=photometric_loss, fused=None, force_exchange=False, fused_adam=True,
                 exchange="rows", early_gather=False, lowrank_chunks=None, rows_max_fraction=None, white_background=None,
                 spatial_order=True):
        """spatial_order (default on): keep the Gaussians stored in Morton order of their positions (GaussianModel.sort_spatially:
        culled Gaussians then come in runs and whole waves of the per-Gaussian forward skip their SH rows; +2-6 % per step).  The
        model is sorted HERE — its buffers and nn.Parameters are rebound, as a densification rebinds them: take references to them
        after constructing the Trainer — at the first and then every SPATIAL_ORDER_EVERY-th densification, and once more when
        densification ends.  The row order is then no longer the point cloud's / the reference's (survivors, clones, children);
        `initial_perm` is the permutation applied here.  False keeps the rows where they are.
        white_background: the dataset flag of reference train_vanilla_3dgs.py:44,109 — with it the opacities are ALSO reset once
        at iteration == opt.densify_from_iter (besides every opacity_reset_interval).  The reference derives the background colour
        AND this extra reset from the one flag; None (default) does the same from the other end: a `background` of all ones is a
        white-background dataset.
        fused_adam: single GPU — the optimizer update is applied by the backward kernel itself
        (fused_step.backward_raw_adam), except in the iterations that densify / reset opacity (there the reference
        skips the replaced parameters' update).
        exchange: view-parallel exchange of the fused step.  "lowrank" ships dL/dRGB (3 floats per Gaussian and view)
        plus the 11 geometry gradients and replicates the optimizer; "rows" ships the same 14 floats but only for the
        Gaussians that received a gradient in the view (exchange_rows; falls back to "lowrank" in a step where that would
        be more bytes); "dense" reduce-scatters the 59-float bucket, shards Adam and all-gathers the parameters (also
        what the autograd step uses).
        rows_max_fraction: "rows" is used while the largest per-view row count stays below this fraction of P
        (None: the break-even of the two byte counts for this world size, rows_limit()).
        early_gather (lowrank only): issue the colour-gradient all-gather between the blend backward and the
        per-Gaussian backward, so that it travels while that kernel runs (costs a 25-us extraction kernel).
        lowrank_chunks: row chunks of the colour-gradient all-gather (None: 4 above 256 k Gaussians)."""
        self.model, self.cameras, self.opt = model, cameras, opt
        self.bg = background
        self.pipe = pipe or PipelineParams()
        self.extent = cameras_extent
        self.densify = densify
        if white_background is None:
            white_background = bool(torch.all(torch.as_tensor(background).detach().float() == 1.0))
        self.white_background = bool(white_background)
        self.loss_fn = loss_fn
        # fused=None: use the fused raw-parameter step whenever the model lives on the GPU, the
        # default pipeline flags are in force and the loss is the standard photometric one
        if fused is None:
            fused = bool(model.flat.is_cuda and loss_fn is photometric_loss and
                         not self.pipe.convert_SHs_python and not self.pipe.compute_cov3D_python)
        self.fused = fused
        self.rank, self.world = dist_info()
        if self.world > 1 and model.flat_store.numel() % self.world != 0:
            # the flat buffers are padded to a multiple of 256 elements (GaussianModel._bind): the dense exchange shards
            # them evenly only for world sizes that divide 256
            raise ValueError(f"world size {self.world} does not divide the padded parameter buffer "
                             f"({model.flat_store.numel()} elements, a multiple of 256): use 1, 2, 4, 8, ... ranks")
        # force_exchange: run the multi-rank exchange path (collectives, sharded optimizer) even in a
        # 1-rank process group — lets the RCCL code path be exercised on a single GPU
        self.force_exchange = bool(force_exchange)
        if exchange not in ("rows", "lowrank", "dense"):
            raise ValueError("exchange must be 'rows', 'lowrank' or 'dense'")
        self.rows_max_fraction = rows_max_fraction
        self.rows_speculate = True            # size the row collective from the previous step (exchange_rows)
        self._rows_cap = None
        self._rows_skip = 0
        self._rows = None                     # fused_step.GatheredRows of the current iteration (GPU path of exchange_rows)
        self._rows_bufs = None                # persistent per-step buffers of exchange_rows (index arrays, pinned counts)
        self._rows_recent = []                # largest per-view row counts of the last ROWS_WINDOW steps (sizes the speculation)
        self.exchange_used = {"rows": 0, "lowrank": 0}        # steps per form actually taken (rows mode decides per step)
        self.fused_adam = bool(fused_adam)
        if exchange == "rows" and self.world > 32:
            exchange = "lowrank"              # (the per-Gaussian view mask of the sparse form is one 32-bit word)
        self.exchange_mode = exchange
        self.early_gather = bool(early_gather)
        self._d_chunks, self._geo_work = [], None
        self.lowrank_chunks = lowrank_chunks
        g = torch.Generator(device="cpu").manual_seed(seed)
        self.perm = torch.randperm(len(cameras), generator=g).tolist()
        self.last = {}
        self.spatial_order = bool(spatial_order)
        self.initial_perm = None
        if self.spatial_order and model.num_points:
            if model.spatial_order_every <= 0:
                model.spatial_order_every = self.SPATIAL_ORDER_EVERY
            self.initial_perm = model.sort_spatially()
        # list_share: None on the model = chosen from the measured walk fraction (adapt_list_share, kept on the model as
        # _list_share_chosen); a number = the caller's
        self.share_rho = None

    # densifications between two full re-sorts (the new Gaussians of the rounds between stay appended).  A re-sort of 2 M
    # Gaussians is 2 ms (1 ms for the permutation, 1 ms for the compaction pass); every second densification it costs 0.01 ms per
    # iteration of the reference schedule and leaves at most one round's clones and children out of place
    SPATIAL_ORDER_EVERY = 2

    def _order_before_step(self, iteration):
        """spatial_order: the last densification is over — the Gaussians appended since the last re-sort go into place (no
        gradient is pending at the start of a step; moments and statistics move with their rows)."""
        if self.spatial_order and self.densify and iteration == self.opt.densify_until_iter and self.model._densify_calls:
            self.sync_stats()
            self.gather_moments()
            self.model.sort_spatially()

    def adapt_list_share(self, handle):
        """Every SHARE_PROBE_EVERY fused steps (and on the first three): the walked fraction of the view just rendered decides the
        list_share of the following steps (rasterizer.adapt_list_share; kept on the model)."""
        from .rasterizer import adapt_list_share
        combine = None
        if self.world > 1:
            # every rank probes in the same iterations (same call count): one scalar all-reduce per probe, the mean decides for all
            def combine(rho):
                t = torch.tensor([rho], dtype=torch.float64, device=self.model.flat.device)
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                return float(t) / self.world
        rho = adapt_list_share(self.model, handle, every=self.SHARE_PROBE_EVERY, combine=combine)
        if rho is not None:
            self.share_rho = rho

    def camera_for(self, iteration):
        """rank r of N renders camera perm[(it*N + r) mod n] — N distinct views per step."""
        return self.cameras[self.perm[((iteration - 1) * self.world + self.rank) % len(self.cameras)]]

    # ------------------------------------------------------------------------------------------
    def shard_range(self):
        """[lo, hi) of the padded flat buffers this rank owns (optimizer state sharding)."""
        n = self.model.flat_store.numel() // self.world
        return self.rank * n, (self.rank + 1) * n

    def exchange(self, grad2d_norm, visible, radii, tracking=True):
        """The one exchange step of the view-parallel loop.  Returns the reduced
        (sum of per-view norms, visibility count, max radii) — None each once densification is over.

        Contract: the flat gradient bucket holds THIS view's gradient already scaled by 1/world (the step
        scales dL/dimage, a 3xHxW pass, instead of dividing the 59*P bucket afterwards; exact for power-of-two
        worlds), and `grad2d_norm` is the norm of the unscaled 2-D gradient.

        Gradients: reduce-scatter (SUM) of the flat 59*P bucket — every rank receives the averaged gradient of
        ITS 1/N slice only, steps Adam on that slice, and the updated parameters are all-gathered
        (optimizer_step_and_gather).  Same bytes on the wire as an all-reduce, but the optimizer sweep
        (the largest HBM stream of the step) shrinks N-fold per GPU.  On RCCL both collectives run in place
        (recv = send + rank*count), so no staging copy of the 472 MB bucket exists; the reduce-scatter is
        issued first and asynchronously, the small statistics all-reduces queue behind it and overlap Adam."""
        m = self.model
        if self.world > 1 or self.force_exchange:
            lo, hi = self.shard_range()
            shard = nccl_inplace_shard(m.flat_grad_store, lo, hi, self.rank, self.world) if dist.get_backend() == "nccl" else None
            if shard is not None:
                self._rs_work = dist.reduce_scatter_tensor(shard, m.flat_grad_store, op=dist.ReduceOp.SUM, async_op=True)
            else:
                shard = torch.empty(hi - lo, dtype=m.flat_grad_store.dtype, device=m.flat_grad_store.device)
                dist.reduce_scatter_tensor(shard, m.flat_grad_store, op=dist.ReduceOp.SUM)
                m.flat_grad_store[lo:hi].copy_(shard)
                self._rs_work = None
            if not tracking:
                return None, None, None
            self.track_local(visible, radii)
            nsum = grad2d_norm * visible
            self._stat_work = (dist.all_reduce(nsum, op=dist.ReduceOp.SUM, async_op=True),)
            return nsum, None, None
        return grad2d_norm * visible, visible.to(grad2d_norm.dtype), radii

    def track_local(self, visible, radii):
        """Visibility counts and radii of THIS rank's views since the last sync_stats().  `denom` (a sum of 0 / 1 over views) and
        `max_radii2D` (a maximum) are only read when a densification or a checkpoint consumes them, and both reductions are
        exact in any order — integer-valued sums, maxima — so the ranks accumulate their own views locally and reduce ONCE,
        right before the consumer (sync_stats), instead of with two all-reduces of P-sized arrays in every step (rounds 1-4).
        The sum of the gradient norms still travels every step (in the rows, or as one all-reduce): a float sum is not."""
        P = int(radii.shape[0])
        ver = getattr(self.model, "_rows_version", 0)
        if getattr(self, "_vis_local", None) is not None and (self._vis_local_version != ver or self._vis_local.shape[0] != P):
            # (model.reorder() / sort_spatially() / prune_points() called directly between two sync points used to leave
            #  these counters on the old rows without any error)
            raise RuntimeError("the model's rows were moved (reorder / sort_spatially / prune / densify called directly) while this rank "
                               "held visibility counts of its own views: call Trainer.sync_stats() on every rank first")
        if getattr(self, "_vis_local", None) is None or self._vis_local.device != radii.device:
            self._vis_local = torch.zeros(P, dtype=torch.int32, device=radii.device)
            self._rmax_local = torch.zeros(P, dtype=torch.int32, device=radii.device)
            self._vis_local_version = ver
        if radii.is_cuda and radii.dtype == torch.int32 and radii.is_contiguous():
            from ._lib import check, ptr, stream_ptr
            from .fused_step import lib             # (the module that declares the entry point's argument types)
            with torch.cuda.device(radii.device):             # one launch instead of five torch kernels (`visible` IS radii > 0)
                check(lib.w3d_track_visibility(P, ptr(radii), ptr(self._vis_local), ptr(self._rmax_local), stream_ptr(radii.device)))
            return
        from ._host_twins import twin
        twin("track_visibility", "Trainer.track_local")(radii > 0 if visible is None or isinstance(visible, _LazyVisible) else visible, radii,
                                                        self._vis_local, self._rmax_local)

    def sync_stats(self):
        """Fold the locally tracked visibility counts and radii of all ranks into model.denom / model.max_radii2D (track_local).
        Called before anything reads them: densify_and_prune, capture().  A collective pair — every rank must call it."""
        v = getattr(self, "_vis_local", None)
        if v is None:
            return
        m = self.model
        if v.shape[0] == m.num_points:
            dist.all_reduce(v, op=dist.ReduceOp.SUM)
            dist.all_reduce(self._rmax_local, op=dist.ReduceOp.MAX)
            m.denom += v[:, None].to(m.denom.dtype)
            m.max_radii2D = torch.max(m.max_radii2D, self._rmax_local.to(m.max_radii2D.dtype))
        self._vis_local = self._rmax_local = None

    def campos_of_all_ranks(self, iteration):
        """(world, 3) camera centres of the views all ranks render in this iteration (known locally: same camera list,
        same permutation).  On the GPU a slice of a table kept on the device (the camera centres in permutation order, twice in a
        row so that a step's `world` consecutive entries never wrap): no per-step stack + host-to-device copy in front of the
        optimizer kernel."""
        n = len(self.cameras)
        dev = self.model.flat.device
        if dev.type == "cuda" and self.world <= n:
            tab = getattr(self, "_campos_table", None)
            if tab is None or tab.device != dev or tab.shape[0] != 2 * n:
                c = torch.stack([self.cameras[i].camera_center.detach().reshape(3).float() for i in self.perm]).to(dev)
                tab = self._campos_table = torch.cat([c, c]).contiguous()
            s0 = ((iteration - 1) * self.world) % n
            return tab[s0:s0 + self.world]
        cams = [self.cameras[self.perm[((iteration - 1) * self.world + r) % n]] for r in range(self.world)]
        return torch.stack([c.camera_center.detach().reshape(3).float() for c in cams])

    def exchange_lowrank(self, dcolor, grad2d_norm, visible, radii, tracking=True):
        """Low-rank exchange of the view-parallel step.  Contract as exchange(): dcolor and the geometry blocks of the
        gradient bucket hold this view's values scaled by 1/world.  Issues, asynchronously and in this order: the
        all-gather of the (P,3) colour gradients (the SH update waits for it), the all-reduce (SUM) of the geometry
        blocks of the bucket (xyz | opacity | scaling | rotation are one contiguous span) and the statistics.
        14 floats per Gaussian and view cross the links instead of 59, and nothing is gathered afterwards because every
        rank applies the identical update (optimizer_step_lowrank).  Returns the reduced statistics."""
        m = self.model
        P = m.num_points
        if dcolor is not None:           # (None: gather_colors() was called early, between the two halves of the backward)
            self.gather_colors(dcolor)
        sl = m.block_slices()
        a, b = sl["xyz"][0], sl["rotation"][1]                # xyz | opacity | scaling | rotation: one contiguous span
        # (one contiguous span: 11 P gradient floats + at most 3 zero padding floats in front of each block, flat_layout)
        assert 11 * P <= b - a <= 11 * P + 9 and sl["xyz"][0] < sl["opacity"][0] < sl["scaling"][0] < sl["rotation"][0]
        self._geo_work = [dist.all_reduce(m.flat_grad[a:b], op=dist.ReduceOp.SUM, async_op=True)]
        if not tracking:
            return None, None, None
        self.track_local(visible, radii)              # (visibility counts and radii: reduced when consumed, sync_stats)
        nsum = grad2d_norm * visible
        self._stat_work = (dist.all_reduce(nsum, op=dist.ReduceOp.SUM, async_op=True),)
        return nsum, None, None

    def rows_limit(self, P):
        """Largest per-view row count for which the sparse form moves fewer bytes than the low-rank one.  Per rank and step
        the low-rank exchange receives (N-1) * (12 + 88/N) * P bytes (all-gather of (P,3) + ring all-reduce of 11 floats),
        the sparse one (N-1) * 64 * rows: break-even rows / P = (12 + 88/N) / 64, taken with a 10 % margin for the count
        exchange and the host wait that sizes the collective."""
        frac = self.rows_max_fraction
        if frac is None:
            frac = min(1.0, 0.9 * (12.0 + 88.0 / max(self.world, 1)) / 64.0)
        return int(frac * P)

    def exchange_rows(self, dcolor, grad2d_norm, visible, radii, tracking=True, packed=None):
        """Sparse exchange of the view-parallel step.  Contract as exchange_lowrank().  A view gives a gradient only to the
        Gaussians its pixels blended — every other row of dcolor and of the geometry gradients is exactly zero — so the
        ranks all-gather their NON-ZERO rows (64 B: index, ||dL/dmean2D||, dL/dRGB, 11 geometry gradients; packed by
        w3d_pack_gradient_rows).  On the GPU nothing dense is rebuilt from them: w3d_index_gradient_rows leaves a per-Gaussian
        view mask + row positions (fused_step.GatheredRows) and the replicated optimizer step reads the rows through it,
        every Gaussian's views IN VIEW ORDER (optimizer_step_lowrank -> w3d_rows_adam) — identical additions in identical
        order on every rank keep the replicas bit-identical, as in the low-rank form.  (CPU tensors, the host-logic tests:
        the same sums through dense arrays, w3d_apply_gradient_rows' torch twin, and the low-rank step.)
        Sizing the collective.  The row counts are all-gathered too, and the kernels read them on the device.  The
        first step (and every step after one that was too dense) waits for them on the host, sizes the collective exactly
        and decides — identically on every rank — whether the views are sparse enough (rows_limit) or go through
        exchange_lowrank (as do the ROWS_RETRY steps after such a one, without counting rows).  Later steps are SPECULATIVE:
        the collective is sized 1.15 x the largest count of the last ROWS_WINDOW steps (never beyond rows_limit, the size at
        which the sparse form stops paying) and enqueued together with the indexing kernel before the host looks at the counts
        (which arrive in pinned memory in the meantime); a view that produced more rows than that gets a second all-gather for
        the remainder, a step that turned out too dense altogether abandons the rows (`rows_abandoned`) for the low-rank form.  The host wait then falls where the GPU still has the
        collective to run, instead of leaving it idle.
        Statistics: the norms travel in the rows; visibility counts and radii are tracked per rank and reduced when a
        densification or a checkpoint reads them (track_local / sync_stats) — an ordinary step of this form issues TWO
        collectives: the counts and the rows."""
        from .fused_step import ROW_FLOATS, GatheredRows, apply_gradient_rows, pack_gradient_rows
        m = self.model
        P = m.num_points
        if visible is None:
            visible = _LazyVisible(radii)
        dev = radii.device
        if packed is not None:
            # (rows, count) straight from the per-Gaussian backward (fused_step.backward_raw_rows): dcolor / grad2d_norm are None;
            # a step that turns out too dense for this form rebuilds the dense arrays from its own rows (too_dense below)
            assert self._rows_skip == 0 and dcolor is None
        if self._rows_skip > 0:
            # a recent step was too dense for this form: the next ROWS_RETRY steps take the low-rank form without counting
            # their rows first (same state on every rank: it derives from the gathered counts)
            self._rows_skip -= 1
            self.exchange_used["lowrank"] += 1
            return self.exchange_lowrank(dcolor, grad2d_norm, visible, radii, tracking=tracking)
        rows, count = packed if packed is not None else pack_gradient_rows(m, dcolor, grad2d_norm if tracking else None)
        counts = torch.empty(self.world, dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(counts, count)
        pinned = ev = None
        bufs = self._rows_bufs
        if counts.is_cuda and (bufs is None or bufs["P"] != P or bufs["dev"] != dev):
            # index arrays and the pinned landing place of the counts live as long as P does (resized after a densification)
            bufs = self._rows_bufs = {"P": P, "dev": dev, "pinned": torch.empty(self.world, dtype=torch.int32, pin_memory=True),
                                      "viewmask": torch.empty(max(P, 1), dtype=torch.int32, device=dev),
                                      "slots": torch.empty(self.world * max(P, 1), dtype=torch.int32, device=dev)}
        index_bufs = None if bufs is None or not counts.is_cuda else (bufs["viewmask"], bufs["slots"])
        if counts.is_cuda:                       # the counts start their way to the host now; whoever needs them waits on ev
            pinned = bufs["pinned"]
            pinned.copy_(counts, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()

        def host_counts():
            if ev is None:
                return counts.tolist()
            ev.synchronize()
            return pinned.tolist()
        # everything that does not depend on the counts is enqueued BEFORE the host reads them
        gpu = radii.is_cuda
        vcount = rmax = nsum = None
        self._stat_work = ()
        if tracking:
            self.track_local(visible, radii)          # no collective: reduced when consumed (sync_stats); the norms travel in the rows
        sl = m.block_slices()
        a, b = sl["xyz"][0], sl["rotation"][1]                    # xyz | opacity | scaling | rotation: one contiguous span
        assert 11 * P <= b - a <= 11 * P + 9

        def gather(first, n):
            """rows [first, first + n) of every view: one all-gather -> (world, n, 16)"""
            part = torch.empty(self.world, n, ROW_FLOATS, dtype=torch.float32, device=dev)
            dist.all_gather_into_tensor(part.view(-1), rows[first:first + n].reshape(-1))
            return part

        def too_dense(hc):
            # too dense for the sparse form: this step's gradients travel as in exchange_lowrank; the norms, which
            # would have travelled in the rows, as one more all-reduce.  Decided from the gathered counts: identically on every rank
            self.last_row_counts = hc
            self.exchange_used["lowrank"] += 1
            self._rows_skip = self.ROWS_RETRY
            self._rows_cap = None
            self._rows_recent = []
            ns = None
            dense_color = dcolor
            if packed is not None:
                # the dense arrays the low-rank form exchanges were never written: rebuild them from this view's own rows
                dense_color = torch.zeros(P, 3, dtype=torch.float32, device=dev)
                m.flat_grad[a:b].zero_()
                own_norm = torch.zeros(P, dtype=torch.float32, device=dev) if tracking else None
                apply_gradient_rows(m, rows, count, rows.shape[0], dense_color, own_norm)
            if tracking:
                ns = own_norm if packed is not None else grad2d_norm * visible
                self._stat_work = self._stat_work + (dist.all_reduce(ns, op=dist.ReduceOp.SUM, async_op=True),)
            self.exchange_lowrank(dense_color, None, None, None, tracking=False)
            return ns, vcount, rmax

        cap = self._rows_cap
        limit = self.rows_limit(P)
        gathered = None
        if cap is not None:
            # (never speculate beyond the size at which the sparse form stops paying: a step that turned dense — after a
            #  densification or an SH degree bump — then costs one wasted collective of at most `limit` rows, not of P)
            cap = max(1, min(cap, max(P, 1), max(limit, 1)))
            parts = [gather(0, cap)]
            if gpu:
                gathered = GatheredRows(m, parts[0], counts, index_bufs)      # (indexing kernel enqueued behind the collective)
            hc = host_counts()                                    # the GPU is busy with the collective meanwhile
            nmax = max(hc)
            if nmax > limit:
                self.exchange_used["rows_abandoned"] = self.exchange_used.get("rows_abandoned", 0) + 1
                return too_dense(hc)
            if nmax > cap:                                        # a view outgrew the guess: the remainder follows
                parts.append(gather(cap, nmax - cap))
                self.exchange_used["rows_overflow"] = self.exchange_used.get("rows_overflow", 0) + 1
                if gpu:
                    gathered = GatheredRows(m, torch.cat(parts, 1).contiguous(), counts, index_bufs)
        else:
            hc = host_counts()                                    # host wait: backward + a world-int collective
            nmax = max(hc)
            if nmax > limit:
                return too_dense(hc)
            parts = [gather(0, max(nmax, 1))]
            if gpu:
                gathered = GatheredRows(m, parts[0], counts, index_bufs)
        self.last_row_counts = hc
        self.exchange_used["rows"] += 1
        # next step's guess — or back to the exact, host-sized form when the views have become too dense for this one
        # (the views differ: one camera's row count says little about the next one's — sized from the previous STEP alone, 27 of
        #  100 benchmark steps outgrew the guess and paid a second collective; the largest count of a window of steps, + 15 %)
        self._rows_recent = (self._rows_recent + [nmax])[-self.ROWS_WINDOW:]
        self._rows_cap = None if not self.rows_speculate else \
            min(max(P, 1), (int(1.15 * max(self._rows_recent)) + 1024) // 1024 * 1024)
        if gpu:
            # the optimizer kernel reads the rows through the per-Gaussian index (optimizer_step_lowrank -> w3d_rows_adam):
            # no dense per-view array is zero-filled, scattered into or read
            if tracking:
                # (the norms go straight into the running statistic: _post_backward's `xyz_gradient_accum += nsum` without the array)
                gathered.norm_accumulate(m.xyz_gradient_accum)
                self._norms_accumulated = True
            self._rows, self._d_chunks, self._geo_work = gathered, [], []
            return None, vcount, rmax
        # host-logic path (CPU tensors, tests/test_dist_gloo.py): the same sums through dense arrays and the low-rank step
        if tracking:
            nsum = torch.zeros(P, dtype=torch.float32, device=dev)
        d_all = torch.zeros(self.world, P, 3, dtype=torch.float32, device=dev)
        m.flat_grad[a:b].zero_()
        done = 0
        for part in parts:
            left = counts if done == 0 else (counts - done).clamp_(min=0)
            for v in range(self.world):
                apply_gradient_rows(m, part[v], left[v:v + 1], part.shape[1], d_all[v], nsum)
            done += part.shape[1]
        self._d_chunks, self._geo_work = [[(0, P), d_all, None]], []
        return nsum, vcount, rmax

    def gather_colors(self, dcolor):
        """Asynchronous all-gather of the (P,3) colour gradients, in a few row chunks so that the SH update of the first
        chunk starts while the later ones (and then the geometry all-reduce) are still on the wire."""
        P = self.model.num_points
        nchunk = self.lowrank_chunks or (4 if (dcolor.is_cuda and P >= (1 << 18)) else 1)
        step = ((P + nchunk - 1) // nchunk + 255) // 256 * 256
        dcolor = dcolor.contiguous()
        self._d_chunks = []
        for r0 in range(0, P, step):
            r1 = min(P, r0 + step)
            d_all = torch.empty(self.world, r1 - r0, 3, dtype=torch.float32, device=dcolor.device)
            work = dist.all_gather_into_tensor(d_all.view(-1), dcolor[r0:r1].view(-1), async_op=True)
            self._d_chunks.append([(r0, r1), d_all, work])

    def optimizer_step_lowrank(self, iteration, skip):
        """Replicated optimizer step after exchange_lowrank: SH blocks from the gathered colour gradients (needs the
        pre-update xyz, so it runs first, while the geometry all-reduces are still in flight), then the geometry blocks
        from the reduced bucket.  Identical inputs and a fixed view order keep the replicas bit-identical."""
        from .fused_step import GEO_BLOCKS, SH_BLOCKS, rows_adam, sh_adam_lowrank
        m = self.model
        m.optimizer.advance(GEO_BLOCKS + SH_BLOCKS, skip)
        campos = self.campos_of_all_ranks(iteration).to(m.flat.device)
        if self._rows is not None:           # sparse exchange on the GPU: all six blocks in one kernel, straight from the rows
            rows_adam(m, self._rows, campos, skip)
            m._bucket_claimed = False
            self._rows = None
            return
        if not self._d_chunks:
            raise RuntimeError("optimizer_step_lowrank: no exchanged gradients to step from (exchange_lowrank / exchange_rows first)")
        whole = len(self._d_chunks) == 1
        for rows, d_all, work in self._d_chunks:
            if work is not None:
                work.wait()
            sh_adam_lowrank(m, d_all, campos, skip=skip, rows=None if whole else rows)
        for w in self._geo_work or ():
            w.wait()
        m.optimizer.step(only=GEO_BLOCKS, skip=skip, advance=False, respect_none_grads=False)
        self._d_chunks, self._geo_work = [], None

    def _drain_lowrank(self):
        """Make the current stream wait for the gradient collectives (before anything may recycle their buffers)."""
        for c in self._d_chunks:
            if c[2] is not None:
                c[2].wait()
                c[2] = None
        for w in self._geo_work or ():
            w.wait()
        self._geo_work = None
        # (self._rows — the sparse form's gathered rows — needs no wait: its collectives were synchronous on this stream; an
        #  opacity-reset iteration still steps from it after _post_backward)

    def wait_stats(self):
        """Make the current stream wait for the statistics all-reduces of exchange()."""
        for w in getattr(self, "_stat_work", None) or ():
            w.wait()
        self._stat_work = None

    def optimizer_step_and_gather(self, zero_grad, skip):
        """Adam on this rank's shard, then all-gather of the updated parameters (single GPU: plain step)."""
        m = self.model
        if self.world == 1 and not self.force_exchange:
            m.optimizer.step(zero_grad=zero_grad, skip=skip, respect_none_grads=False)
            return
        lo, hi = self.shard_range()
        if getattr(self, "_rs_work", None) is not None:
            self._rs_work.wait()
            self._rs_work = None
        m.optimizer.step(zero_grad=zero_grad, skip=skip, elem_range=(lo, hi), respect_none_grads=False)
        shard = nccl_inplace_shard(m.flat_store, lo, hi, self.rank, self.world) if dist.get_backend() == "nccl" else None
        if shard is not None:
            dist.all_gather_into_tensor(m.flat_store, shard)
        else:
            dist.all_gather_into_tensor(m.flat_store, m.flat_store[lo:hi].clone())
        self._moments_sharded = True

    def gather_moments(self):
        """Before anything reads or rebuilds the optimizer state (densify, checkpoint): make the
        sharded Adam moments whole again on every rank."""
        if (self.world == 1 and not self.force_exchange) or not getattr(self, "_moments_sharded", False):
            return
        m = self.model
        lo, hi = self.shard_range()
        n = m.flat.numel()
        for buf in (m.optimizer.exp_avg, m.optimizer.exp_avg_sq):
            full = torch.zeros(m.flat_store.numel(), dtype=buf.dtype, device=buf.device)
            full[:n].copy_(buf)
            dist.all_gather_into_tensor(full, full[lo:hi].clone())
            buf.copy_(full[:n])
        self._moments_sharded = False

    def capture(self):
        """GaussianModel.capture() of a consistent replica: in the dense exchange every rank only steps the Adam moments
        of its shard, so they are gathered first."""
        self.gather_moments()
        self.sync_stats()
        assert not getattr(self, "_moments_sharded", False)
        return self.model.capture()

    def _structure_change_due(self, iteration):
        """Does _post_backward densify / prune / reset opacity in this iteration?"""
        opt = self.opt
        if not (iteration < opt.densify_until_iter and self.densify):
            return False
        return (iteration > opt.densify_from_iter and iteration % opt.densification_interval == 0) or \
            self._opacity_reset_due(iteration)

    def _opacity_reset_due(self, iteration):
        """reference train_vanilla_3dgs.py:109"""
        opt = self.opt
        return iteration % opt.opacity_reset_interval == 0 or (self.white_background and iteration == opt.densify_from_iter)

    def background_for(self, iteration):
        """reference train_vanilla_3dgs.py:71: a fresh uniform colour per iteration when opt.random_background is set"""
        if getattr(self.opt, "random_background", False):
            return torch.rand(3, device=self.bg.device)
        return self.bg

    def _post_backward(self, iteration, nsum, vcount, rmax, stats_done):
        """Densification bookkeeping + optimizer step shared by both step flavours."""
        m, opt = self.model, self.opt
        # the reference replaces nn.Parameters on densify / opacity reset, so their .grad is None
        # and torch's Adam skips them in that iteration's step; `skip` reproduces that.
        skip = set()
        if iteration < opt.densify_until_iter:
            if not stats_done and getattr(self, "_norms_accumulated", False):
                self._norms_accumulated = False           # (exchange_rows on the GPU added the rows' norms itself)
            elif not stats_done:
                m.xyz_gradient_accum += nsum[:, None]
                if vcount is not None:              # (several ranks: tracked locally, reduced by sync_stats below when read)
                    m.max_radii2D = torch.max(m.max_radii2D, rmax.to(m.max_radii2D.dtype))
                    m.denom += vcount[:, None]
            if self.densify:
                if iteration > opt.densify_from_iter and iteration % opt.densification_interval == 0:
                    self.sync_stats()
                    self.gather_moments()
                    size_threshold = 20 if iteration > opt.opacity_reset_interval else None
                    torch.manual_seed(1234 + iteration)     # identical split samples on every rank
                    m.densify_and_prune(opt.densify_grad_threshold, 0.005, self.extent, size_threshold)
                    skip = {"xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"}
                if self._opacity_reset_due(iteration):
                    self.gather_moments()
                    m.reset_opacity()
                    skip.add("opacity")
        return skip

    def step_fused(self, iteration):
        """The same step with everything between the parameter buffers and the image fused into
        the HIP kernels (fused_step.py): raw-parameter forward, fused loss value+gradient, raw-parameter backward.
        One GPU: the backward applies Adam and the densification statistics itself (backward_raw_adam); iterations that
        densify / reset opacity write the gradient bucket and step separately.  Several ranks: low-rank or dense
        exchange (exchange_lowrank / exchange).  No autograd graph is built."""
        from .fused import l1_ssim_fwd_bwd
        from .fused_step import (backward_blend_dcolor, backward_raw, backward_raw_adam, backward_raw_lowrank, backward_raw_rows,
                                 finish, render_raw)
        m, opt = self.model, self.opt
        m.update_learning_rate(iteration)
        if iteration % 1000 == 0:
            m.oneupSHdegree()
        cam = self.camera_for(iteration)
        bg = self.background_for(iteration)
        with torch.no_grad():
            tracking = iteration < opt.densify_until_iter
            single = self.world == 1 and not self.force_exchange
            lowrank = (not single) and self.exchange_mode in ("lowrank", "rows") and m.max_sh_degree == 3
            rows_form = lowrank and self.exchange_mode == "rows"
            dcol = packed = None
            use_adam = (single and self.fused_adam and iteration < opt.iterations and
                        m.max_sh_degree == 3 and not self._structure_change_due(iteration))
            attempts = 0
            while True:
                attempts += 1
                if attempts > 4:      # one repeat sizes the list buffer exactly; more means the counters are corrupt
                    raise RuntimeError("fused step: the forward keeps reporting an overflowing list buffer")
                pkg = render_raw(cam, m, bg, sync=False, color_only=True)      # (the loss reads `render` only)
                loss, dimg = l1_ssim_fwd_bwd(pkg["render"], cam.original_image, opt.lambda_dssim)
                if not single and self.world > 1:
                    dimg.mul_(1.0 / self.world)      # the bucket then holds grad/world: reduce-scatter(SUM) = mean
                if use_adam:
                    gnorm = backward_raw_adam(m, pkg["handle"], dimg, want_norm=False, update_stats=tracking)
                elif lowrank and self.early_gather and not rows_form:
                    # blend backward + dL/dRGB; make sure this forward is final BEFORE a collective is issued (a rank that
                    # repeats its view must not issue it twice); then gather, then the per-Gaussian backward
                    early = backward_blend_dcolor(m, pkg["handle"], dimg)
                    if not finish(pkg["handle"]):
                        continue
                    self.gather_colors(early)
                    gnorm, _ = backward_raw_lowrank(m, pkg["handle"], None, want_norm=True)
                    dcol = None
                elif rows_form and self._rows_skip == 0:
                    # sparse form: the per-Gaussian backward emits the non-zero rows itself — no dense 14-float arrays, no pack pass
                    packed = backward_raw_rows(m, pkg["handle"], dimg, norm_scale=float(self.world) if tracking else 0.0)
                    gnorm = dcol = None
                elif lowrank:
                    gnorm, dcol = backward_raw_lowrank(m, pkg["handle"], dimg, want_norm=True)
                else:
                    # (statistics of this flavour are applied on the host after finish(): the forward was speculative)
                    gnorm, _ = backward_raw(m, pkg["handle"], dimg, update_stats=False, want_norm=True)
                if finish(pkg["handle"]):        # the only host wait of the step, with the backward already queued
                    break                        # (on overflow the fused-Adam kernel updated nothing: repeat the view)
            self.adapt_list_share(pkg["handle"])
            if single and tracking and not use_adam:       # (the fused-Adam kernel updated the statistics itself)
                vis = pkg["radii"] > 0
                m.xyz_gradient_accum += gnorm[:, None]          # gnorm is 0 on culled Gaussians
                m.denom += vis[:, None]
                m.max_radii2D = torch.max(m.max_radii2D, pkg["radii"].to(m.max_radii2D.dtype))
            if not single:
                # (the sparse form on the GPU needs no visibility mask: the rows carry the norms and track_visibility reads the radii)
                vis = None if packed is not None else pkg["radii"] > 0
                if self.world > 1 and gnorm is not None:
                    gnorm = gnorm * float(self.world)           # statistics use the unscaled per-view norm
                stepped_early = False
                if packed is not None or lowrank:
                    if packed is not None:
                        nsum, vcount, rmax = self.exchange_rows(None, None, vis, pkg["radii"], tracking=tracking, packed=packed)
                    else:
                        nsum, vcount, rmax = (self.exchange_rows if rows_form else self.exchange_lowrank)(
                            dcol, gnorm, vis, pkg["radii"], tracking=tracking)
                    # (both forms: a too-dense rows step falls back to exchange_lowrank and leaves an asynchronous all-reduce on
                    #  the gradient bucket, which a densification recycles — drain first)
                    if self._structure_change_due(iteration):
                        self._drain_lowrank()        # densification recycles the gradient bucket
                    elif iteration < opt.iterations:
                        # ordinary iteration: step BEFORE waiting for the statistics — they are the last collectives in
                        # the queue, and waiting for them first would serialise the SH update behind the geometry
                        # all-reduce it is meant to overlap
                        self.optimizer_step_lowrank(iteration, ())
                        stepped_early = True
                else:
                    nsum, vcount, rmax = self.exchange(gnorm, vis, pkg["radii"], tracking=tracking)
                self.wait_stats()
            else:
                nsum = vcount = rmax = None
            skip = self._post_backward(iteration, nsum, vcount, rmax, stats_done=single)
            if use_adam:
                m.optimizer.note_fused_step()
            elif lowrank:
                if stepped_early:
                    assert not skip
                elif iteration < opt.iterations and len(skip) < 6:
                    self.optimizer_step_lowrank(iteration, skip)
                else:
                    # (the reference's step skips every replaced parameter in a densification iteration)
                    self._drain_lowrank()
                    self._rows, self._d_chunks = None, []
            elif iteration < opt.iterations:
                # the next backward overwrites the whole bucket: no zeroing pass needed
                self.optimizer_step_and_gather(zero_grad=bool(skip), skip=skip)
        self.last = dict(loss=loss, image=pkg["render"], radii=pkg["radii"])
        return loss

    def step(self, iteration):
        self._order_before_step(iteration)
        if self.fused:
            return self.step_fused(iteration)
        m, opt = self.model, self.opt
        m.update_learning_rate(iteration)
        if iteration % 1000 == 0:
            m.oneupSHdegree()
        cam = self.camera_for(iteration)
        pkg = render(cam, m, self.pipe, self.background_for(iteration))
        image = pkg["render"]
        loss = self.loss_fn(image, cam.original_image, opt.lambda_dssim)
        (loss / self.world if self.world > 1 else loss).backward()
        with torch.no_grad():
            vis, radii = pkg["visibility_filter"], pkg["radii"]
            gnorm = pkg["viewspace_points"].grad[:, :2].norm(dim=-1) * float(self.world)
            tracking = iteration < opt.densify_until_iter
            nsum, vcount, rmax = self.exchange(gnorm, vis, radii, tracking=tracking)
            self.wait_stats()
            skip = self._post_backward(iteration, nsum, vcount, rmax, stats_done=False)
            if iteration < opt.iterations:
                self.optimizer_step_and_gather(zero_grad=True, skip=skip)
        self.last = dict(loss=loss.detach(), num_rendered=None, image=image.detach())
        return loss.detach()


RENDER_STREAMS = 3      # frames in flight of render_views (independent frames overlap each other's latency-bound stages).
                        # 2 M Gaussians at 1600x1200: 1 stream 0.74 ms per frame, 2-4 streams 0.62-0.66 (profiles/render_host_probe.py;
                        # the host needs 0.19 ms to enqueue a frame, so the loop is GPU-bound).  Three, not two: HIP streams share
                        # a handful of hardware queues, and WHICH two pool streams a process gets decides whether they overlap at
                        # all — the same loop ran at 2550 Mpix/s on one pair and 3050 on another within one process, and at
                        # 2940-3100 with any three (profiles/r03/render_stream_pairs.json).


def render_views(model, cameras, background, pipe=None):
    """Forward-only rendering of a list of views (what reference render.py:24-35 times)."""
    pipe = pipe or PipelineParams()
    out = []
    raw = (hasattr(model, "flat") and model.flat.is_cuda and not pipe.convert_SHs_python
           and not pipe.compute_cov3D_python)
    with torch.no_grad():
        if not raw:
            return [render(cam, model, pipe, background)["render"] for cam in cameras]
        # pre-activation parameters straight into the kernels (no exp/sigmoid/normalize/cat launches), no host round
        # trip per frame: the list buffer is sized speculatively (fused_step.ListCapacity) and the counters of frame i
        # are only inspected after frame i+1 has been enqueued; a frame whose buffer was too small is rendered again
        from .fused_step import finish, render_raw
        pending = []
        # consecutive frames go to alternating streams: sort and binning are latency-bound kernels that leave most of the
        # chip idle, so two independent frames in flight overlap them (frames do not depend on each other)
        main = torch.cuda.current_stream(model.flat.device)
        if RENDER_STREAMS > 1:
            streams = getattr(model, "_render_streams", None)       # (kept on the model: creating a HIP stream costs ~0.2 ms)
            if streams is None or len(streams) != RENDER_STREAMS or streams[0].device != model.flat.device:
                streams = model._render_streams = [torch.cuda.Stream(device=model.flat.device) for _ in range(RENDER_STREAMS)]
        else:
            streams = [main]
        for st in streams:
            st.wait_stream(main)

        def settle(i, pkg):
            if not finish(pkg["handle"]):
                with torch.cuda.stream(streams[i % len(streams)]):
                    out[i] = render_raw(cameras[i], model, background, color_only=True)["render"]      # synchronous: exact buffer size
        for i, cam in enumerate(cameras):
            with torch.cuda.stream(streams[i % len(streams)]):
                pkg = render_raw(cam, model, background, sync=False, color_only=True)      # (only `render` is returned)
            out.append(pkg["render"])
            pending.append((i, pkg))
            if len(pending) > len(streams):
                settle(*pending.pop(0))
        for item in pending:
            settle(*item)
        for st in streams:
            main.wait_stream(st)
        if len(streams) > 1:
            for img in out:
                img.record_stream(main)
    return out
