"""Stand-in for reference scene/gaussian_model.py: the TRAINING-TIME surface of its GaussianModel in this repo's wording —
six nn.Parameters activated with torch ops on every access (:33-41,101-121), torch.optim.Adam over six named groups with
eps 1e-15 (:167-186), the exponential xyz schedule (:188-194), add_densification_stats (:461-463) and the checkpoint
13-tuple (:63-99).  No densification, PLY or label code: the measured iterations do not reach them."""
import math

import torch
from torch import nn
from simple_knn._C import distCUDA2  # noqa: F401  (the import the reference's module has, :20)


def _xyz_schedule(lr_init, lr_final, delay_mult, max_steps):
    def at(step):
        t = min(max(step / max_steps, 0.0), 1.0)
        return math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)
    return at


class GaussianModel:
    def __init__(self, sh_degree: int):
        self.active_sh_degree, self.max_sh_degree = 0, sh_degree
        self.optimizer = None
        self.spatial_lr_scale = 0.0

    get_xyz = property(lambda s: s._xyz)
    get_scaling = property(lambda s: torch.exp(s._scaling))
    get_rotation = property(lambda s: torch.nn.functional.normalize(s._rotation))
    get_opacity = property(lambda s: torch.sigmoid(s._opacity))
    get_features = property(lambda s: torch.cat((s._features_dc, s._features_rest), dim=1))
    get_which_object = property(lambda s: s._which_object)

    def oneupSHdegree(self):
        self.active_sh_degree = min(self.active_sh_degree + 1, self.max_sh_degree)

    def training_setup(self, a):
        P = self._xyz.shape[0]
        self.percent_dense = a.percent_dense
        self.xyz_gradient_accum = torch.zeros(P, 1, device="cuda")
        self.denom = torch.zeros(P, 1, device="cuda")
        s = self.spatial_lr_scale
        self.optimizer = torch.optim.Adam([
            {"params": [self._xyz], "lr": a.position_lr_init * s, "name": "xyz"},
            {"params": [self._features_dc], "lr": a.feature_lr, "name": "f_dc"},
            {"params": [self._features_rest], "lr": a.feature_lr / 20.0, "name": "f_rest"},
            {"params": [self._opacity], "lr": a.opacity_lr, "name": "opacity"},
            {"params": [self._scaling], "lr": a.scaling_lr, "name": "scaling"},
            {"params": [self._rotation], "lr": a.rotation_lr, "name": "rotation"}], lr=0.0, eps=1e-15)
        self._xyz_lr = _xyz_schedule(a.position_lr_init * s, a.position_lr_final * s, a.position_lr_delay_mult,
                                     a.position_lr_max_steps)

    def update_learning_rate(self, iteration):
        for g in self.optimizer.param_groups:
            if g["name"] == "xyz":
                g["lr"] = self._xyz_lr(iteration)
                return g["lr"]

    def add_densification_stats(self, viewspace_point_tensor, update_filter):
        self.xyz_gradient_accum[update_filter] += torch.norm(viewspace_point_tensor.grad[update_filter, :2], dim=-1, keepdim=True)
        self.denom[update_filter] += 1

    def capture(self):
        return (self.active_sh_degree, self._xyz, self._features_dc, self._features_rest, self._scaling, self._rotation,
                self._opacity, self._which_object, self.max_radii2D, self.xyz_gradient_accum, self.denom,
                self.optimizer.state_dict(), self.spatial_lr_scale)

    def restore(self, model_args, training_args):
        (self.active_sh_degree, xyz, f_dc, f_rest, scaling, rotation, opacity, self._which_object, self.max_radii2D, accum, denom,
         opt_dict, self.spatial_lr_scale) = model_args
        par = lambda t: nn.Parameter(t.detach().clone().cuda().contiguous().requires_grad_(True))  # noqa: E731
        self._xyz, self._features_dc, self._features_rest = par(xyz), par(f_dc), par(f_rest)
        self._scaling, self._rotation, self._opacity = par(scaling), par(rotation), par(opacity)
        self.training_setup(training_args)
        self.xyz_gradient_accum, self.denom = accum, denom
        if opt_dict is not None:
            self.optimizer.load_state_dict(opt_dict)
