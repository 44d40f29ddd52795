"""Parameter store feeding the rasterizer — the caller-side contract of SURVEY.md §8 rows A9-A11.

Same public surface as reference scene/gaussian_model.py (GaussianModel): `get_xyz`,
`get_scaling` (exp), `get_rotation` (normalize), `get_opacity` (sigmoid), `get_features`
(cat(dc, rest)), `get_covariance`, `active_sh_degree`/`oneupSHdegree`, `training_setup`,
`update_learning_rate`, `add_densification_stats`, `densify_and_prune`, `reset_opacity`,
`capture`/`restore`, `create_from_points`.  The storage is MI355X-first rather than a
translation: all 59 trainable floats per Gaussian live in ONE flat fp32 buffer laid out as
six contiguous blocks [xyz | f_dc | f_rest | opacity | scaling | rotation], and the gradients in
a second flat buffer of the same layout, so the view-parallel step all-reduces one bucket
(236 B x P) over xGMI without packing, and the optimizer can sweep one array.  The six
nn.Parameters are views into the flat buffer.
"""
import math

import numpy as np
import torch
from torch import nn

BLOCKS = (("xyz", (3,)), ("f_dc", (1, 3)), ("f_rest", (15, 3)), ("opacity", (1,)), ("scaling", (3,)),
          ("rotation", (4,)))
FLOATS_PER_GAUSSIAN = sum(int(np.prod(s)) for _, s in BLOCKS)  # 59


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
    q = q / q.norm(dim=1, keepdim=True)
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
        self.percent_dense = 0.0
        self.spatial_lr_scale = 1.0
        self._train_args = None

    # ------------------------------------------------------------------ storage
    def _bind(self, blocks: dict):
        """(Re)build the flat buffers from a dict name -> tensor (P, *shape)."""
        P = blocks["xyz"].shape[0]
        n = P * FLOATS_PER_GAUSSIAN
        # storage padded to a multiple of 256 elements: any world size up to 256 can shard it evenly for the
        # reduce-scatter / all-gather exchange (train.py) without repacking
        n_pad = (n + 255) // 256 * 256
        self.flat_store = torch.zeros(n_pad, dtype=torch.float32, device=self.device)
        self.flat_grad_store = torch.zeros(n_pad, dtype=torch.float32, device=self.device)
        self.flat = self.flat_store[:n]
        self.flat_grad = self.flat_grad_store[:n]
        self._p = {}
        off = 0
        for name, shape in BLOCKS:
            n = P * int(np.prod(shape))
            view = self.flat[off:off + n].view(P, *shape)
            view.copy_(blocks[name].reshape(P, *shape))
            p = nn.Parameter(view, requires_grad=True)
            p.grad = self.flat_grad[off:off + n].view(P, *shape)
            self._p[name] = p
            off += n

    def block_slices(self):
        """name -> (start, stop) element offsets of each block inside the flat buffers."""
        P, off, out = self.num_points, 0, {}
        for name, shape in BLOCKS:
            n = P * int(np.prod(shape))
            out[name] = (off, off + n)
            off += n
        return out

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
        g = viewspace_point_tensor.grad if isinstance(viewspace_point_tensor, torch.Tensor) and \
            viewspace_point_tensor.grad is not None else viewspace_point_tensor
        self.xyz_gradient_accum[update_filter] += torch.norm(g[update_filter, :2], dim=-1, keepdim=True)
        self.denom[update_filter] += 1

    def _rebuild(self, keep, extra=None):
        """Single resize path: keep rows where `keep` is True, then append `extra` rows.
        Parameters, both Adam moments and the bookkeeping vectors move together."""
        cur = {n: self._p[n].detach() for n, _ in BLOCKS}
        mom = self.optimizer.moments() if self.optimizer is not None else None
        new, new_m, new_v = {}, {}, {}
        for n, _ in BLOCKS:
            parts = [cur[n][keep]]
            if extra is not None:
                parts.append(extra[n])
            new[n] = torch.cat(parts, 0)
            if mom is not None:
                m, v = mom[n]
                pad = [] if extra is None else [torch.zeros_like(extra[n])]
                new_m[n] = torch.cat([m[keep]] + pad, 0)
                new_v[n] = torch.cat([v[keep]] + pad, 0)
        n_extra = 0 if extra is None else extra["xyz"].shape[0]
        wo = self._which_object[keep]
        if extra is not None:
            wo = torch.cat([wo, extra["which_object"]], 0)
        stats = (self.xyz_gradient_accum[keep], self.denom[keep], self.max_radii2D[keep])
        steps = self.optimizer.step_count if self.optimizer is not None else 0
        self._bind(new)
        self._which_object = wo
        if self.optimizer is not None:
            self.training_setup(self._train_args, moments=(new_m, new_v, steps))
        if n_extra:
            self._reset_stats()      # reference zeroes the statistics after every growth (densification_postfix)
        else:
            self.xyz_gradient_accum, self.denom, self.max_radii2D = stats

    def prune_points(self, mask):
        self._rebuild(~mask)

    def _select(self, sel, repeat=1):
        d = {n: self._p[n].detach()[sel].repeat(repeat, *([1] * (self._p[n].dim() - 1))) for n, _ in BLOCKS}
        d["which_object"] = self._which_object[sel].repeat(repeat, 1)
        return d

    def densify_and_clone(self, grads, grad_threshold, scene_extent):
        sel = (torch.norm(grads, dim=-1) >= grad_threshold) & \
              (self.get_scaling.max(dim=1).values <= self.percent_dense * scene_extent)
        self._rebuild(torch.ones(self.num_points, dtype=torch.bool, device=self.device), self._select(sel))

    def densify_and_split(self, grads, grad_threshold, scene_extent, N=2):
        P = self.num_points
        padded = torch.zeros(P, device=self.device)
        padded[:grads.shape[0]] = grads.squeeze()
        sel = (padded >= grad_threshold) & (self.get_scaling.max(dim=1).values > self.percent_dense * scene_extent)
        new = self._select(sel, N)
        stds = self.get_scaling.detach()[sel].repeat(N, 1)
        samples = torch.normal(mean=torch.zeros_like(stds), std=stds)
        rots = quat_to_rotmat(self._p["rotation"].detach()[sel]).repeat(N, 1, 1)
        new["xyz"] = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + new["xyz"]
        new["scaling"] = torch.log(stds / (0.8 * N))
        keep = torch.ones(P, dtype=torch.bool, device=self.device)
        self._rebuild(keep, new)
        drop = torch.cat((sel, torch.zeros(N * int(sel.sum()), dtype=torch.bool, device=self.device)))
        self.prune_points(drop)

    def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size):
        grads = self.xyz_gradient_accum / self.denom
        grads[grads.isnan()] = 0.0
        self.densify_and_clone(grads, max_grad, extent)
        self.densify_and_split(grads, max_grad, extent)
        prune = (self.get_opacity < min_opacity).squeeze()
        if max_screen_size:
            prune = prune | (self.max_radii2D > max_screen_size) | (self.get_scaling.max(dim=1).values > 0.1 * extent)
        self.prune_points(prune)

    def reset_opacity(self):
        new = inverse_sigmoid(torch.min(self.get_opacity, torch.ones_like(self.get_opacity) * 0.01))
        with torch.no_grad():
            self._p["opacity"].copy_(new)
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

    # ------------------------------------------------------------------ checkpoint (gaussian_model.py:63-99)
    def capture(self):
        return (self.active_sh_degree, self._xyz.detach().clone(), self._features_dc.detach().clone(),
                self._features_rest.detach().clone(), self._scaling.detach().clone(), self._rotation.detach().clone(),
                self._opacity.detach().clone(), self._which_object, self.max_radii2D, self.xyz_gradient_accum,
                self.denom, self.optimizer.state_dict() if self.optimizer is not None else None, self.spatial_lr_scale)

    def restore(self, model_args, training_args):
        (self.active_sh_degree, xyz, f_dc, f_rest, scaling, rotation, opacity, which_object, max_radii2D, accum, denom,
         opt_dict, self.spatial_lr_scale) = model_args
        self._bind(dict(xyz=xyz, f_dc=f_dc, f_rest=f_rest, opacity=opacity, scaling=scaling, rotation=rotation))
        self._which_object = which_object
        self.training_setup(training_args)
        self.max_radii2D, self.xyz_gradient_accum, self.denom = max_radii2D, accum, denom
        if opt_dict is not None:
            self.optimizer.load_state_dict(opt_dict)
