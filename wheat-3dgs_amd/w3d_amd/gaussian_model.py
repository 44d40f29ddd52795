"""Parameter store feeding the rasterizer — the caller-side contract of SURVEY.md §8 rows A9-A11.

Same public surface as reference scene/gaussian_model.py (GaussianModel): `get_xyz`,
`get_scaling` (exp), `get_rotation` (normalize), `get_opacity` (sigmoid), `get_features`
(cat(dc, rest)), `get_covariance`, `active_sh_degree`/`oneupSHdegree`, `training_setup`,
`update_learning_rate`, `add_densification_stats`, `densify_and_prune`, `reset_opacity`,
`capture`/`restore`, `create_from_pcd`, `reset_label`.  The storage is MI355X-first rather than a
translation: all 59 trainable floats per Gaussian live in ONE flat fp32 buffer laid out as
six contiguous blocks [xyz | f_dc | f_rest | opacity | scaling | rotation], and the gradients in
a second flat buffer of the same layout, so the view-parallel step all-reduces one bucket
(236 B x P) over xGMI without packing, and the optimizer can sweep one array.  The six
nn.Parameters are views into the flat buffer.
"""
import math
import os

import numpy as np
import torch
from torch import nn

# Order of the blocks inside the flat buffers: the four geometry blocks first (11 contiguous floats per Gaussian: the
# view-parallel exchange all-reduces exactly that span of the gradient bucket in one collective), then the SH blocks.
BLOCKS = (("xyz", (3,)), ("opacity", (1,)), ("scaling", (3,)), ("rotation", (4,)), ("f_dc", (1, 3)), ("f_rest", (15, 3)))
FLOATS_PER_GAUSSIAN = sum(int(np.prod(s)) for _, s in BLOCKS)  # 59
BLOCK_ALIGN = 4        # every block of the flat buffers starts on a multiple of 4 floats (16 B)


def flat_layout(P):
    """name -> (start, stop) element offsets of each parameter block inside the flat buffers, and the total length.
    Blocks lie back to back, except that every block STARTS on a 16-byte boundary (round 4): the kernels stream the blocks
    with 16-B vector accesses and fall back to dword paths when a block's base is misaligned — with back-to-back blocks that
    was the case whenever P was not a multiple of 4, i.e. after three densifications out of four (the fused backward + Adam
    kernel took 0.72 instead of 0.55 ms at 1.9 M Gaussians).  At most 3 floats of padding per block; they stay zero.
    The same rule is applied by w3d_densify_compact (include/w3d.h)."""
    off, out = 0, {}
    for name, shape in BLOCKS:
        off = (off + BLOCK_ALIGN - 1) // BLOCK_ALIGN * BLOCK_ALIGN
        n = P * int(np.prod(shape))
        out[name] = (off, off + n)
        off += n
    return out, off


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


def get_expon_lr_func(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """Log-linear LR decay with optional warm-up; same values as reference
    utils/general_utils.py:29-62 (pinned by tests/golden/lr.npz)."""
    def helper(step):
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0
        delay = 1.0
        if lr_delay_steps > 0:
            delay = lr_delay_mult + (1 - lr_delay_mult) * math.sin(0.5 * math.pi * min(max(step / lr_delay_steps, 0), 1))
        t = min(max(step / max_steps, 0), 1)
        return delay * math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)
    return helper


def quat_to_rotmat(q):
    """(N,4) quaternion (normalised here, as reference utils/general_utils.py:78-99) -> (N,3,3)."""
    # the norm as an explicit left-to-right sum of squares: torch's vectorised .norm() rounds differently in the last
    # bit, and the split children's positions are compared bit for bit with the reference's (tests/golden/densify.npz)
    n = torch.sqrt(q[:, 0] * q[:, 0] + q[:, 1] * q[:, 1] + q[:, 2] * q[:, 2] + q[:, 3] * q[:, 3])
    q = q / n[:, None]
    r, x, y, z = q.unbind(1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)


class OptimizationParams:
    """Defaults of reference arguments/__init__.py:71-90."""
    iterations = 15_000
    position_lr_init = 0.00016
    position_lr_final = 0.0000016
    position_lr_delay_mult = 0.01
    position_lr_max_steps = 30_000
    feature_lr = 0.0025
    opacity_lr = 0.05
    scaling_lr = 0.005
    rotation_lr = 0.001
    percent_dense = 0.01
    lambda_dssim = 0.2
    densification_interval = 100
    opacity_reset_interval = 3000
    densify_from_iter = 500
    densify_until_iter = 11_000
    densify_grad_threshold = 0.0002
    random_background = False


class GaussianModel:
    def __init__(self, sh_degree: int = 3, device="cuda"):
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        self.device = torch.device(device)
        self.flat = torch.empty(0, device=self.device)
        self.flat_grad = torch.empty(0, device=self.device)
        self._p = {}
        self.max_radii2D = torch.empty(0, device=self.device)
        self.xyz_gradient_accum = torch.empty(0, device=self.device)
        self.denom = torch.empty(0, device=self.device)
        self._which_object = torch.empty(0, device=self.device)
        self.optimizer = None
        # optional behaviours of the raw-parameter rasterizer path (fused_step.render_raw reads them; rasterizer.py header)
        self.tile_cull = True           # exact footprint culling of tile instances (identical outputs, shorter lists)
        self.deterministic = False      # deterministic blend backward (bit-identical gradients run to run)
        self.list_share = None          # 0 / 1 / 2: w3d_view.list_share (speed only); None: what a Trainer measured to be best for
        self._list_share_chosen = None  # this model (train.Trainer.adapt_list_share), else rasterizer.LIST_SHARE_DEFAULT
        self._bucket_claimed = False    # a backward node of this pass already writes the flat gradient bucket directly
        # storage order (sort_spatially): > 0 keeps the Gaussians in Morton order of their positions — sorted at the first and
        # then every `spatial_order_every`-th densify_and_prune and when a checkpoint is restored.  A Gaussian set has no
        # order of its own; the reference's (survivors, clones, children) is what 0 (the default) keeps.  W3D_SPATIAL_ORDER
        # sets the default for models a script creates itself (the import redirect).
        self.spatial_order_every = max(0, int(os.environ.get("W3D_SPATIAL_ORDER", "0") or 0))
        self._densify_calls = 0
        self.percent_dense = 0.0
        self.spatial_lr_scale = 1.0
        self._train_args = None

    # ------------------------------------------------------------------ storage
    def _bind(self, blocks: dict):
        """(Re)build the flat buffers from a dict name -> tensor (P, *shape)."""
        P = blocks["xyz"].shape[0]
        layout, n = flat_layout(P)
        # storage padded to a multiple of 256 elements: every world size that divides 256 (1, 2, 4, 8, ...) shards it
        # evenly for the reduce-scatter / all-gather exchange (train.py checks) without repacking
        n_pad = (n + 255) // 256 * 256
        self.flat_store = torch.zeros(n_pad, dtype=torch.float32, device=self.device)
        self.flat_grad_store = torch.zeros(n_pad, dtype=torch.float32, device=self.device)
        self._full, self._spare = {}, {}
        self.flat = self.flat_store[:n]
        self.flat_grad = self.flat_grad_store[:n]
        self._p = {}
        for name, shape in BLOCKS:
            a, b = layout[name]
            view = self.flat[a:b].view(P, *shape)
            view.copy_(blocks[name].reshape(P, *shape))
            p = nn.Parameter(view, requires_grad=True)
            p.grad = self.flat_grad[a:b].view(P, *shape)
            self._p[name] = p

    def grad_view(self, name):
        """The block of the flat gradient bucket that belongs to parameter `name`, shaped like it (whether or not the
        parameter's .grad currently points at it)."""
        a, b = self.block_slices()[name]
        return self.flat_grad[a:b].view(self._p[name].shape)

    def bind_grad_views(self):
        """(Re)attach every parameter's .grad to its block of the flat bucket (after zero_grad(set_to_none=True))."""
        for name, p in self._p.items():
            p.grad = self.grad_view(name)

    def block_slices(self):
        """name -> (start, stop) element offsets of each block inside the flat buffers (flat_layout)."""
        return flat_layout(self.num_points)[0]

    @property
    def num_points(self):
        return 0 if not self._p else int(self._p["xyz"].shape[0])

    # names used by the reference's code
    _xyz = property(lambda s: s._p["xyz"])
    _features_dc = property(lambda s: s._p["f_dc"])
    _features_rest = property(lambda s: s._p["f_rest"])
    _opacity = property(lambda s: s._p["opacity"])
    _scaling = property(lambda s: s._p["scaling"])
    _rotation = property(lambda s: s._p["rotation"])

    # ------------------------------------------------------------------ activations (gaussian_model.py:33-41,101-132)
    @property
    def get_xyz(self):
        return self._p["xyz"]

    @property
    def get_scaling(self):
        return torch.exp(self._p["scaling"])

    @property
    def get_rotation(self):
        return torch.nn.functional.normalize(self._p["rotation"])

    @property
    def get_opacity(self):
        return torch.sigmoid(self._p["opacity"])

    @property
    def get_features(self):
        return torch.cat((self._p["f_dc"], self._p["f_rest"]), dim=1)

    @property
    def get_which_object(self):
        return self._which_object

    def get_covariance(self, scaling_modifier=1):
        R = quat_to_rotmat(self._p["rotation"])
        L = R * (scaling_modifier * self.get_scaling)[:, None, :]
        S = L @ L.transpose(1, 2)
        return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1)

    def oneupSHdegree(self):
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    # ------------------------------------------------------------------ construction
    def create_from_points(self, points, colors, spatial_lr_scale=1.0):
        """Initialisation of reference create_from_pcd (gaussian_model.py:138-165): isotropic
        log-scale from the 3-NN mean squared distance, identity rotation, opacity 0.1."""
        from .rasterizer import dist2_knn3
        self.spatial_lr_scale = spatial_lr_scale
        pts = torch.as_tensor(points, dtype=torch.float32, device=self.device)
        col = torch.as_tensor(colors, dtype=torch.float32, device=self.device)
        P = pts.shape[0]
        dist2 = torch.clamp_min(dist2_knn3(pts), 0.0000001)
        rots = torch.zeros(P, 4, device=self.device)
        rots[:, 0] = 1
        self._bind(dict(xyz=pts, f_dc=((col - 0.5) / 0.28209479177387814)[:, None, :],
                        f_rest=torch.zeros(P, 15, 3, device=self.device),
                        opacity=inverse_sigmoid(0.1 * torch.ones(P, 1, device=self.device)),
                        scaling=torch.log(torch.sqrt(dist2))[:, None].repeat(1, 3), rotation=rots))
        self._reset_stats()
        self._which_object = torch.zeros(P, 1, dtype=torch.int, device=self.device)

    def create_from_pcd(self, pcd, spatial_lr_scale: float):
        """Reference signature (scene/gaussian_model.py:138): `pcd` carries .points and .colors (BasicPointCloud)."""
        self.create_from_points(np.asarray(pcd.points), np.asarray(pcd.colors), spatial_lr_scale)

    def create_from_tensors(self, xyz, features_dc, features_rest, scaling, rotation, opacity, spatial_lr_scale=1.0):
        """Directly from pre-activation tensors (synthetic scenes, checkpoints)."""
        self.spatial_lr_scale = spatial_lr_scale
        dev = self.device
        self._bind(dict(xyz=xyz.to(dev), f_dc=features_dc.to(dev), f_rest=features_rest.to(dev),
                        opacity=opacity.to(dev), scaling=scaling.to(dev), rotation=rotation.to(dev)))
        self._reset_stats()
        self._which_object = torch.zeros(self.num_points, 1, dtype=torch.int, device=dev)

    def _reset_stats(self):
        P = self.num_points
        self.xyz_gradient_accum = torch.zeros(P, 1, device=self.device)
        self.denom = torch.zeros(P, 1, device=self.device)
        self.max_radii2D = torch.zeros(P, device=self.device)

    # ------------------------------------------------------------------ optimisation (gaussian_model.py:167-194)
    def _group_lrs(self, a):
        return {"xyz": a.position_lr_init * self.spatial_lr_scale, "f_dc": a.feature_lr, "f_rest": a.feature_lr / 20.0,
                "opacity": a.opacity_lr, "scaling": a.scaling_lr, "rotation": a.rotation_lr}

    def training_setup(self, training_args, moments=None):
        self._train_args = training_args
        self.percent_dense = training_args.percent_dense
        if moments is None:
            self._reset_stats()
        lrs = self._group_lrs(training_args)
        from .optim import FlatAdam
        self.optimizer = FlatAdam(self, lrs, eps=1e-15, moments=moments)
        self.xyz_scheduler_args = get_expon_lr_func(
            lr_init=training_args.position_lr_init * self.spatial_lr_scale,
            lr_final=training_args.position_lr_final * self.spatial_lr_scale,
            lr_delay_mult=training_args.position_lr_delay_mult, max_steps=training_args.position_lr_max_steps)

    def update_learning_rate(self, iteration):
        lr = self.xyz_scheduler_args(iteration)
        self.optimizer.set_lr("xyz", lr)
        return lr

    # ------------------------------------------------------------------ densification (gaussian_model.py:399-463)
    def add_densification_stats(self, viewspace_point_tensor, update_filter):
        """reference :461-463.  Same values, written as masked arithmetic over all P rows: boolean-mask indexing costs a
        host sync per statement (the size of the selection), `x + 0` leaves the other rows bit-identical."""
        g = viewspace_point_tensor.grad if isinstance(viewspace_point_tensor, torch.Tensor) and \
            viewspace_point_tensor.grad is not None else viewspace_point_tensor
        if update_filter.dtype != torch.bool:        # an index list: the reference's statement as it is
            self.xyz_gradient_accum[update_filter] += torch.norm(g[update_filter, :2], dim=-1, keepdim=True)
            self.denom[update_filter] += 1
            return
        P = self.num_points
        if g.dim() != 2 or g.shape[0] != P or update_filter.numel() != P:
            raise RuntimeError(f"add_densification_stats: gradient {tuple(g.shape)} / filter {tuple(update_filter.shape)} "
                               f"do not match the model's {P} points")
        if g.is_cuda and g.dtype == torch.float32 and g.is_contiguous() and g.shape[1] == 3 and \
                update_filter.is_cuda and update_filter.device == g.device and self.xyz_gradient_accum.device == g.device and \
                self.xyz_gradient_accum.shape[0] == P and self.denom.shape[0] == P and \
                self.xyz_gradient_accum.is_contiguous() and self.denom.is_contiguous():
            from .fused import add_densification_stats
            add_densification_stats(g, update_filter.reshape(-1).contiguous(), self.xyz_gradient_accum, self.denom)
            return
        if g.is_cuda and update_filter.is_cuda:
            # (a strided / non-fp32 gradient on the device: the reference's own statement, as masked arithmetic)
            norm = torch.norm(g[:, :2].float(), dim=-1, keepdim=True)
            self.xyz_gradient_accum += torch.where(update_filter.reshape(-1, 1), norm, torch.zeros_like(norm))
            self.denom += update_filter.reshape(-1, 1).to(self.denom.dtype)
            return
        from ._host_twins import twin
        twin("add_densification_stats", "GaussianModel.add_densification_stats")(self, g, update_filter)

    def _bind_store(self, flat_store, flat_grad_store, P):
        """Adopt already-filled flat buffers (layout of _bind) without copying.  The new nn.Parameters start with
        .grad None, as the reference's replaced parameters do (scene/gaussian_model.py:340-397 builds new nn.Parameters
        in _prune_optimizer / cat_tensors_to_optimizer): the `optimizer.step()` that follows a densification in the
        reference's loop (train_vanilla_3dgs.py:105-115) therefore skips them, and so does FlatAdam.step()."""
        layout, n = flat_layout(P)
        self.flat_store, self.flat_grad_store = flat_store, flat_grad_store
        self.flat, self.flat_grad = flat_store[:n], flat_grad_store[:n]
        self._p = {}
        self._bucket_claimed = False
        for name, shape in BLOCKS:
            a, b = layout[name]
            self._p[name] = nn.Parameter(self.flat[a:b].view(P, *shape), requires_grad=True)

    def _compact(self, src, n_keep, n_child0=None, child_xyz=None, child_scaling=None, reset_stats=True):
        """The single resize path.  New row r = old row src[r]; rows >= n_keep are new points
        (zero Adam moments, reference cat_tensors_to_optimizer :356-374), rows >= n_child0 are split children whose
        xyz / scaling come from child_xyz / child_scaling.  Parameters and both moments move in ONE pass of
        csrc/w3d_densify.hip (a model on the CPU is refused: w3d_amd/_host_twins.py)."""
        P_old, P_new = self.num_points, int(src.numel())
        # (the rows move: whoever keeps per-row data of its own between two calls — Trainer.track_local's per-rank visibility counters —
        #  compares this counter to notice)
        self._rows_version = getattr(self, "_rows_version", 0) + 1
        n_child0 = P_new if n_child0 is None else int(n_child0)
        layout_old, layout_new = flat_layout(P_old)[0], flat_layout(P_new)[0]
        n = flat_layout(P_new)[1]
        n_pad = (n + 255) // 256 * 256
        opt = self.optimizer
        # Buffers come from a two-generation pool with 25 % head room: a fresh multi-GB hipMalloc costs 100+ ms,
        # the compaction itself ~5 ms at 5 M Gaussians, and P changes at every densification.
        spare, self._spare = getattr(self, "_spare", {}), {}
        cur = getattr(self, "_full", {})

        def take(key):
            buf = spare.pop(key, None)
            if buf is None or buf.numel() < n_pad or buf.device != self.flat.device:
                buf = torch.empty((int(n_pad * 1.25) + 255) // 256 * 256, dtype=torch.float32, device=self.flat.device)
            return buf
        full = {"store": take("store")}
        new_store = full["store"][:n_pad]
        new_store[n:].zero_()

        def zero_padding(buf):                           # (the <= 3 padding floats in front of a block stay zero)
            prev_end = 0
            for name, _ in BLOCKS:
                a, b = layout_new[name]
                if a > prev_end:
                    buf[prev_end:a].zero_()
                prev_end = b
        zero_padding(new_store)
        g_full = cur.get("grad")
        full["grad"] = g_full if g_full is not None and g_full.numel() >= n_pad else take("grad")
        new_grad = full["grad"][:n_pad]
        new_grad.zero_()
        m_new = v_new = None
        if opt is not None:
            full["m"], full["v"] = take("m"), take("v")
            m_new, v_new = full["m"][:n], full["v"][:n]
            zero_padding(m_new)
            zero_padding(v_new)
        dims = [int(np.prod(shape)) for _, shape in BLOCKS]
        names = [name for name, _ in BLOCKS]
        if self.flat.is_cuda:
            from .fused import densify_compact
            densify_compact(dims, names.index("xyz"), names.index("scaling"), P_old, src, n_keep, n_child0, self.flat,
                            None if opt is None else opt.exp_avg, None if opt is None else opt.exp_avg_sq,
                            new_store, m_new, v_new, child_xyz, child_scaling)
        else:
            from ._host_twins import twin
            twin("densify_compact", "GaussianModel._compact")(self, names, dims, layout_old, layout_new, P_old, P_new, src, n_keep, n_child0,
                                                              new_store, m_new, v_new, child_xyz, child_scaling)
        src64 = src.to(torch.int64)
        self._which_object = self._which_object.index_select(0, src64)
        stats = None if reset_stats else (self.xyz_gradient_accum.index_select(0, src64), self.denom.index_select(0, src64),
                                          self.max_radii2D.index_select(0, src64))
        steps = dict(opt.steps) if opt is not None else 0
        # the buffers just vacated serve the next compaction
        self._spare = {"store": cur.get("store", self.flat_store)}
        if opt is not None:
            self._spare["m"], self._spare["v"] = cur.get("m", opt.exp_avg), cur.get("v", opt.exp_avg_sq)
        self._bind_store(new_store, new_grad, P_new)
        self._full = full
        if opt is not None:
            self.training_setup(self._train_args, moments=(m_new, v_new, steps))
        if reset_stats:
            self._reset_stats()      # reference zeroes the statistics after every growth (densification_postfix :395-397)
        else:
            self.xyz_gradient_accum, self.denom, self.max_radii2D = stats

    def prune_points(self, mask, during_training=True):
        """reference :328-352 — drops the rows where mask is True, keeps the statistics of the survivors.
        `during_training=False` (run_3d_seg.py:334,346: one object's Gaussians cut out of a deepcopy for its PLY) leaves the
        optimizer alone there; here parameters and moments share one compaction pass, so both flavours take it."""
        src = (~mask).nonzero().squeeze(1)
        self._compact(src, n_keep=src.numel(), reset_stats=False)

    # ------------------------------------------------------------------ storage order
    def spatial_permutation(self):
        """Row order that puts the Gaussians in Morton (Z-order) order of their positions: 21 bits per axis over the box of the
        0.1 % .. 99.9 % quantiles of a strided subsample (outliers clamp to its faces), stable in the current index — a function of the positions
        alone, so replicas holding identical positions get the identical order."""
        xyz = torch.nan_to_num(self._p["xyz"].detach().float(), nan=0.0, posinf=0.0, neginf=0.0)
        P = xyz.shape[0]
        if P < 2:
            return torch.arange(P, device=xyz.device)
        # (the quantiles of every (P / 65536)-th row: a sort of <= 2^17 rows — torch.kthvalue over all P takes 12 ms per call)
        sub = torch.sort(xyz[::max(1, P >> 16)], dim=0).values
        n = sub.shape[0]
        lo, hi = sub[int(0.001 * n)], sub[min(n - 1, n - 1 - int(0.001 * n))]
        q = ((xyz - lo) / (hi - lo).clamp_min(1e-12) * 2097151.0).clamp_(0.0, 2097151.0).to(torch.int64)

        def spread(v):                                   # 21 bits -> every third bit of 63
            v = (v | (v << 32)) & 0x1F00000000FFFF
            v = (v | (v << 16)) & 0x1F0000FF0000FF
            v = (v | (v << 8)) & 0x100F00F00F00F00F
            v = (v | (v << 4)) & 0x10C30C30C30C30C3
            return (v | (v << 2)) & 0x1249249249249249
        code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
        return torch.argsort(code, stable=True)

    def reorder(self, perm):
        """New row r = old row perm[r] of EVERY per-Gaussian array: parameters, both Adam moments, labels, densification
        statistics (one pass of the compaction kernel; step counters, learning rates and statistics kept)."""
        P = self.num_points
        perm = perm.to(self.device)
        if perm.numel() != P or (P and not bool(torch.equal(torch.sort(perm).values, torch.arange(P, device=self.device)))):
            raise ValueError("reorder: not a permutation of the rows")
        if P:
            self._compact(perm, n_keep=P, reset_stats=False)

    def sort_spatially(self):
        """Store the Gaussians in Morton order of their positions and return the permutation applied (new row r = old row
        perm[r]).  Why: which Gaussians a camera sees is a property of WHERE they are, so in this order the culled ones come
        in long runs — whole waves of the per-Gaussian forward skip their 180-B SH rows (preprocess_fwd 0.158 -> 0.113 ms
        at 2 M Gaussians).  Nothing the rasterizer computes depends on the order except the
        tie-break of exactly equal depths."""
        perm = self.spatial_permutation()
        if perm.numel():
            self._compact(perm, n_keep=perm.numel(), reset_stats=False)        # (an argsort: a permutation by construction)
        return perm

    def __deepcopy__(self, memo):
        """run_3d_seg.py:327 deep-copies the model per identified object.  The generic deepcopy would clone every
        nn.Parameter on its own and break the invariant that they are VIEWS of the flat buffers; this one copies the flat
        buffers and rebinds.  Per-call caches kept on the model (scratch buffers, capacity hints, streams) are not copied."""
        new = GaussianModel(self.max_sh_degree, device=self.device)
        memo[id(self)] = new
        for k in ("active_sh_degree", "tile_cull", "deterministic", "list_share", "_list_share_chosen", "spatial_order_every", "_densify_calls", "percent_dense", "spatial_lr_scale", "_train_args"):
            setattr(new, k, getattr(self, k))
        if self.num_points:
            new._bind({n: p.detach() for n, p in self._p.items()})
            new.flat_grad.copy_(self.flat_grad)
            for n, p in self._p.items():
                if p.grad is None:
                    new._p[n].grad = None
        new._which_object = self._which_object.clone()
        new.max_radii2D, new.xyz_gradient_accum, new.denom = (self.max_radii2D.clone(), self.xyz_gradient_accum.clone(),
                                                              self.denom.clone())
        if self.optimizer is not None:
            opt = self.optimizer
            new.training_setup(self._train_args, moments=(opt.exp_avg.clone(), opt.exp_avg_sq.clone(), dict(opt.steps)))
            new.optimizer.lrs.update(opt.lrs)
        return new

    def _split_children(self, sel_idx, N):
        """xyz and (raw) scaling of the N children of every selected Gaussian, reference :407-414 (same torch.normal
        call on the same stds, so the same random stream is consumed)."""
        stds = self.get_scaling.detach()[sel_idx].repeat(N, 1)
        samples = torch.normal(mean=torch.zeros_like(stds), std=stds)
        rots = quat_to_rotmat(self._p["rotation"].detach()[sel_idx]).repeat(N, 1, 1)
        child_xyz = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + self._p["xyz"].detach()[sel_idx].repeat(N, 1)
        return child_xyz, torch.log(stds / (0.8 * N))

    def _selection(self, grads, grad_threshold, scene_extent):
        hot = torch.norm(grads, dim=-1) >= grad_threshold
        smax = self.get_scaling.detach().max(dim=1).values
        return hot & (smax <= self.percent_dense * scene_extent), hot & (smax > self.percent_dense * scene_extent), smax

    def densify_and_clone(self, grads, grad_threshold, scene_extent):
        """reference :425-439"""
        P = self.num_points
        clone, _, _ = self._selection(grads, grad_threshold, scene_extent)
        src = torch.cat([torch.arange(P, device=self.device), clone.nonzero().squeeze(1)])
        self._compact(src, n_keep=P)

    def densify_and_split(self, grads, grad_threshold, scene_extent, N=2):
        """reference :399-423 (grads shorter than P are zero-padded, as there)"""
        P = self.num_points
        padded = torch.zeros(P, 1, device=self.device)
        padded[:grads.shape[0]] = grads.reshape(-1, 1)
        _, split, _ = self._selection(padded, grad_threshold, scene_extent)
        sel_idx = split.nonzero().squeeze(1)
        child_xyz, child_scaling = self._split_children(sel_idx, N)
        keep = (~split).nonzero().squeeze(1)
        self._compact(torch.cat([keep, sel_idx.repeat(N)]), n_keep=keep.numel(), n_child0=keep.numel(),
                      child_xyz=child_xyz, child_scaling=child_scaling)

    def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size, N=2):
        """reference :441-455, with clone -> split -> prune folded into ONE compaction of the buffers.  The surviving rows
        and their order are those of the reference's three steps: originals that are neither split nor pruned, then
        the clones, then the children (all first samples, then all second samples).  Quirk kept: densification_postfix
        zeroes max_radii2D before the screen-size test :449, so that test can never fire; only opacity and the
        world-size test prune."""
        grads = self.xyz_gradient_accum / self.denom
        grads[grads.isnan()] = 0.0
        clone, split, smax = self._selection(grads, max_grad, extent)
        low = (self.get_opacity.detach() < min_opacity).squeeze(-1)
        prune_parent = (low | (smax > 0.1 * extent)) if max_screen_size else low
        sel_idx = split.nonzero().squeeze(1)
        child_xyz, child_scaling = self._split_children(sel_idx, N)
        child_prune = low[sel_idx].repeat(N)
        if max_screen_size:
            child_prune = child_prune | (torch.exp(child_scaling).max(dim=1).values > 0.1 * extent)
        idx_keep = (~split & ~prune_parent).nonzero().squeeze(1)
        idx_clone = (clone & ~prune_parent).nonzero().squeeze(1)
        ck = ~child_prune
        src = torch.cat([idx_keep, idx_clone, sel_idx.repeat(N)[ck]])
        self._compact(src, n_keep=idx_keep.numel(), n_child0=idx_keep.numel() + idx_clone.numel(),
                      child_xyz=child_xyz[ck], child_scaling=child_scaling[ck])
        # (spatial_order_every: the clones and children were appended; every so many rounds the whole set goes back into
        #  Morton order — one more pass of the same kernel, amortised over `every` x densification_interval iterations)
        self._densify_calls += 1
        if self.spatial_order_every > 0 and (self._densify_calls - 1) % self.spatial_order_every == 0:
            self.sort_spatially()

    def reset_opacity(self):
        new = inverse_sigmoid(torch.min(self.get_opacity, torch.ones_like(self.get_opacity) * 0.01))
        with torch.no_grad():
            self._p["opacity"].copy_(new)
        # reference :305-318 replace_tensor_to_optimizer: a NEW nn.Parameter with zeroed moments — its .grad is None, so the
        # optimizer.step() of the same iteration does not touch the opacities (and the stale pre-reset gradient is dropped)
        self._p["opacity"].grad = None
        a, b = self.block_slices()["opacity"]
        self.flat_grad[a:b].zero_()
        self.optimizer.zero_moments("opacity")

    # ------------------------------------------------------------------ PLY snapshots (gaussian_model.py:196-293)
    def _ply_attributes(self):
        names = ["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)]
        names += [f"f_rest_{i}" for i in range(self._features_rest.shape[1] * 3)]
        names += ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)] + ["which_object"]
        return names

    def save_ply(self, path):
        """Binary little-endian PLY with the reference's attribute order and layout: all float32;
        f_dc / f_rest stored CHANNEL-major (transpose(1,2).flatten) as scene/gaussian_model.py:217-218 does."""
        import os
        os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)
        P = self.num_points
        cols = [self._xyz.detach(), torch.zeros(P, 3, device=self.device),
                self._features_dc.detach().transpose(1, 2).flatten(start_dim=1),
                self._features_rest.detach().transpose(1, 2).flatten(start_dim=1),
                self._opacity.detach(), self._scaling.detach(), self._rotation.detach(),
                self._which_object.to(torch.float32).reshape(P, 1)]
        data = torch.cat([c.reshape(P, -1).float() for c in cols], dim=1).contiguous().cpu().numpy().astype("<f4")
        names = self._ply_attributes()
        assert data.shape[1] == len(names)
        header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % P
        header += "".join("property float %s\n" % n for n in names) + "end_header\n"
        with open(path, "wb") as f:
            f.write(header.encode("ascii"))
            f.write(data.tobytes())

    def load_ply(self, path):
        """Inverse of save_ply (reference load_ply, gaussian_model.py:239-293): attribute lookup by NAME, so
        files written by the reference load as well."""
        with open(path, "rb") as f:
            raw = f.read()
        end = raw.index(b"end_header\n") + len(b"end_header\n")
        lines = raw[:end].decode("ascii").split("\n")
        assert "binary_little_endian" in lines[1], "only binary little-endian PLY is supported"
        P = int([ln for ln in lines if ln.startswith("element vertex")][0].split()[-1])
        props = [ln.split()[-1] for ln in lines if ln.startswith("property")]
        assert all(ln.split()[1] == "float" for ln in lines if ln.startswith("property")), "expected float32 properties"
        data = np.frombuffer(raw, dtype="<f4", count=P * len(props), offset=end).reshape(P, len(props))
        col = {n: i for i, n in enumerate(props)}

        def take(prefix):
            names = sorted((n for n in props if n.startswith(prefix)), key=lambda n: int(n.split("_")[-1]))
            return torch.tensor(np.stack([data[:, col[n]] for n in names], axis=1))
        xyz = torch.tensor(np.stack([data[:, col[k]] for k in ("x", "y", "z")], axis=1))
        n_rest = len([n for n in props if n.startswith("f_rest_")])
        assert n_rest == 3 * ((self.max_sh_degree + 1) ** 2 - 1)
        f_dc = take("f_dc_").reshape(P, 3, 1).transpose(1, 2).contiguous()
        f_rest = take("f_rest_").reshape(P, 3, n_rest // 3).transpose(1, 2).contiguous()
        self._bind(dict(xyz=xyz, f_dc=f_dc, f_rest=f_rest, opacity=torch.tensor(data[:, col["opacity"]].copy())[:, None],
                        scaling=take("scale_"), rotation=take("rot_")))
        wo = data[:, col["which_object"]] if "which_object" in col else np.zeros(P, np.float32)
        self._which_object = torch.tensor(wo.copy()).to(torch.int)[:, None].to(self.device)
        self._reset_stats()
        self.active_sh_degree = self.max_sh_degree

    # ------------------------------------------------------------------ labels (gaussian_model.py:465-506)
    def reset_label(self, obj_used_mask, set_which_object_to=None, overlap_threshold=0.8):
        """Assign `set_which_object_to` to the Gaussians of obj_used_mask unless they largely (> overlap_threshold) belong
        to earlier objects already; then the dominant earlier object decides: if the new set covers less than 60 % of
        itself inside that object it becomes a new object anyway (returns None), otherwise it is merged into the old one
        (returns the old id).  Decision rules and return values of reference reset_label; the counts are computed on the
        device (one bincount instead of unique + .cpu() copies of two P-sized masks)."""
        mask = obj_used_mask.reshape(-1).to(torch.bool)
        wo = self._which_object.reshape(-1)
        sel = wo[mask]
        n_sel = int(mask.sum())
        nonzero_count = int(torch.count_nonzero(sel))
        if nonzero_count > 0:
            overlap_ratio = nonzero_count / n_sel
            if overlap_ratio > overlap_threshold:
                counts = torch.bincount(sel.to(torch.int64).clamp_min(0))
                counts[0] = 0
                # torch.unique returns ascending values and argmax the first maximum: smallest id among equal counts
                which_overlap_object = int(torch.argmax(counts))
                old = wo == which_overlap_object
                n_new = int(mask.sum())
                intersect_ratio = float((mask & old).sum()) / n_new if n_new > 0 else 0.0
                if intersect_ratio < 0.6:
                    self._which_object[mask] = set_which_object_to
                    return None
                self._which_object[mask] = which_overlap_object
                return which_overlap_object
            self._which_object[mask] = set_which_object_to
            return None
        if set_which_object_to is not None:
            self._which_object[mask] = set_which_object_to
        return None

    # ------------------------------------------------------------------ checkpoint (gaussian_model.py:63-99)
    def capture(self):
        """The reference's 13-tuple, optimizer state in torch.optim.Adam's state_dict layout (FlatAdam.state_dict), so a
        checkpoint written here restores in the reference and vice versa."""
        return (self.active_sh_degree, self._xyz.detach().clone(), self._features_dc.detach().clone(),
                self._features_rest.detach().clone(), self._scaling.detach().clone(), self._rotation.detach().clone(),
                self._opacity.detach().clone(), self._which_object, self.max_radii2D, self.xyz_gradient_accum,
                self.denom, self.optimizer.state_dict() if self.optimizer is not None else None, self.spatial_lr_scale)

    def restore(self, model_args, training_args):
        (self.active_sh_degree, xyz, f_dc, f_rest, scaling, rotation, opacity, which_object, max_radii2D, accum, denom,
         opt_dict, self.spatial_lr_scale) = model_args
        det = lambda t: t.detach() if isinstance(t, torch.Tensor) else t  # noqa: E731  (reference tuples hold nn.Parameters)
        self._bind(dict(xyz=det(xyz), f_dc=det(f_dc), f_rest=det(f_rest), opacity=det(opacity), scaling=det(scaling),
                        rotation=det(rotation)))
        self._which_object = which_object.to(self.device)
        self.training_setup(training_args)
        self.max_radii2D, self.xyz_gradient_accum, self.denom = (max_radii2D.to(self.device), accum.to(self.device),
                                                                 denom.to(self.device))
        if opt_dict is not None:
            self.optimizer.load_state_dict(opt_dict)
        if self.spatial_order_every > 0:
            self.sort_spatially()
