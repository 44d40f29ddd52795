"""Generates tests/golden/*.npz|json by IMPORTING the reference's own Python pieces.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden.py
The reference's rasterizer itself (CUDA submodules) is absent, so the vectors here pin what the
reference CAN compute on a CPU: SH evaluation, covariance construction, camera matrices, the loss,
PSNR, the LR schedule, FlashSplat's label assignment, and the exact argument marshalling of
gaussian_renderer.render()/flashsplat_render() (captured with stub rasterizer modules).
Only data (inputs + expected outputs) is written; no reference source is copied.
"""
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

# ---- make `device="cuda"` allocations land on the CPU while reference code runs
_orig = {n: getattr(torch, n) for n in ("zeros", "ones", "zeros_like", "tensor", "empty", "rand")}


def _cpuify(fn):
    def w(*a, **k):
        if "device" in k and str(k["device"]).startswith("cuda"):
            k["device"] = "cpu"
        return fn(*a, **k)
    return w


for _n, _f in _orig.items():
    setattr(torch, _n, _cpuify(_f))
torch.Tensor.cuda = lambda self, *a, **k: self

# ---- stub the modules the reference imports but this image lacks
captured = {}


class _Settings(dict):
    def __init__(self, **kw):
        super().__init__(**kw)
        self.__dict__.update(kw)


def _make_rasterizer(tag, nout):
    class _R:
        def __init__(self, raster_settings=None):
            self.s = raster_settings

        def __call__(self, **kw):
            rec = {"settings": {k: (list(v.shape) if torch.is_tensor(v) else v) for k, v in self.s.items()},
                   "kwargs": {k: (None if v is None else {"shape": list(v.shape), "dtype": str(v.dtype),
                                                          "requires_grad": bool(v.requires_grad),
                                                          "contiguous": bool(v.is_contiguous())})
                              for k, v in kw.items()}}
            captured.setdefault(tag, []).append(rec)
            P = kw["means3D"].shape[0]
            H, W = self.s["image_height"], self.s["image_width"]
            outs = [torch.zeros(3, H, W), torch.zeros(P, dtype=torch.int32), torch.zeros(1, H, W), torch.zeros(1, H, W)]
            if nout == 8:
                outs += [torch.zeros(H, W), torch.zeros(self.s["num_obj"] + 1, P), torch.zeros(P, 2), torch.zeros(P)]
            return tuple(outs)
    return _R


for name, nout in (("diff_gaussian_rasterization", 4), ("flashsplat_rasterization", 8)):
    m = types.ModuleType(name)
    m.GaussianRasterizationSettings = _Settings
    m.GaussianRasterizer = _make_rasterizer(name, nout)
    sys.modules[name] = m
knn = types.ModuleType("simple_knn")
knn_c = types.ModuleType("simple_knn._C")
knn_c.distCUDA2 = lambda pts: torch.ones(pts.shape[0])
sys.modules["simple_knn"] = knn
sys.modules["simple_knn._C"] = knn_c
ply = types.ModuleType("plyfile")
ply.PlyData = ply.PlyElement = object
sys.modules["plyfile"] = ply


class _Anything(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _Anything(self.__name__ + "." + k)

    def __call__(self, *a, **k):
        return None


for _name in ("ffmpeg", "torchvision", "torchvision.utils", "torchvision.transforms", "torchvision.transforms.functional",
              "shapely", "shapely.geometry", "wandb", "cv2", "open3d"):
    if _name not in sys.modules:
        try:
            __import__(_name)
        except Exception:
            sys.modules[_name] = _Anything(_name)
sys.path.insert(0, REF)

from utils.sh_utils import eval_sh  # noqa: E402
from utils.general_utils import build_rotation, get_expon_lr_func  # noqa: E402
from utils.graphics_utils import getWorld2View2, getProjectionMatrix, focal2fov  # noqa: E402
from utils.loss_utils import l1_loss, ssim  # noqa: E402
from utils.image_utils import psnr  # noqa: E402
from scene.gaussian_model import GaussianModel  # noqa: E402
from scene.cameras import MiniCam  # noqa: E402
import gaussian_renderer  # noqa: E402

g = torch.Generator().manual_seed(1234)

# (1) SH evaluation, degrees 0..3, combined with the +0.5 / clamp of render()'s python branch
N = 64
shs = torch.randn(N, 16, 3, generator=g)
shs[:, 1:] *= 0.3
xyz = torch.randn(N, 3, generator=g)
campos = torch.tensor([0.3, -0.2, 2.5])
d = xyz - campos
d = d / d.norm(dim=1, keepdim=True)
sh_out = {}
for deg in range(4):
    sh_out[f"rgb_deg{deg}"] = torch.clamp_min(eval_sh(deg, shs.transpose(1, 2), d) + 0.5, 0.0).numpy()
np.savez(os.path.join(OUT, "sh_eval.npz"), shs=shs.numpy(), xyz=xyz.numpy(), campos=campos.numpy(), **sh_out)

# (2) covariance from scaling / rotation through GaussianModel.get_covariance
gm = GaussianModel(3)
gm._scaling = torch.randn(N, 3, generator=g) * 0.5 - 3.0
gm._rotation = torch.randn(N, 4, generator=g)
cov = {}
for mod in (1.0, 0.7):
    cov[f"cov_mod{mod}"] = gm.get_covariance(mod).numpy()
np.savez(os.path.join(OUT, "cov3d.npz"), log_scaling=gm._scaling.numpy(), rotation_raw=gm._rotation.numpy(),
         scaling=gm.get_scaling.numpy(), rotation=gm.get_rotation.numpy(),
         R=build_rotation(gm._rotation).numpy(), **cov)

# (3) camera matrices as scene/cameras.py builds them
cams = {}
rng = np.random.RandomState(7)
for i in range(3):
    A = rng.randn(3, 3)
    Q, _ = np.linalg.qr(A)
    if np.linalg.det(Q) < 0:
        Q[:, 0] *= -1
    T = rng.randn(3) * 0.5 + np.array([0, 0, 2.5])
    fovx, fovy = focal2fov(1.2 * 64, 64), focal2fov(1.2 * 64, 48)
    wvt = torch.tensor(getWorld2View2(Q, T, np.array([0.0, 0.0, 0.0]), 1.0).astype(np.float32)).transpose(0, 1)
    proj = getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)
    full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
    center = wvt.inverse()[3, :3]
    cams[f"R{i}"], cams[f"T{i}"] = Q, T
    cams[f"fov{i}"] = np.array([fovx, fovy])
    cams[f"view{i}"], cams[f"proj{i}"], cams[f"full{i}"], cams[f"center{i}"] = wvt.numpy(), proj.numpy(), full.numpy(), center.numpy()
np.savez(os.path.join(OUT, "camera.npz"), **cams)

# (4) loss pieces and PSNR
img1 = torch.rand(3, 40, 56, generator=g)
img2 = (img1 + 0.1 * torch.randn(3, 40, 56, generator=g)).clamp(0, 1)
np.savez(os.path.join(OUT, "loss.npz"), img1=img1.numpy(), img2=img2.numpy(), l1=l1_loss(img1, img2).numpy(),
         ssim=ssim(img1, img2).numpy(), psnr=psnr(img1, img2).numpy())

# (5) LR schedule (arguments/__init__.py:76-79 defaults, spatial_lr_scale 1)
f = get_expon_lr_func(lr_init=0.00016, lr_final=0.0000016, lr_delay_mult=0.01, max_steps=30000)
steps = np.array([0, 1, 100, 1000, 15000, 30000, 40000])
np.savez(os.path.join(OUT, "lr.npz"), steps=steps, lr=np.array([f(int(s)) for s in steps], dtype=np.float64))

# (6) FlashSplat label assignment (run_3d_seg.py:54-72)
sys.modules.setdefault("wandb", types.ModuleType("wandb"))
import importlib.util  # noqa: E402
src = open(os.path.join(REF, "run_3d_seg.py")).read()
start = src.index("def multi_instance_opt")
end = src.index("def opt_label_w_seg")
ns = {"torch": torch, "F": torch.nn.functional, "tqdm": (lambda x, **k: x)}
exec(compile(src[start:end], "run_3d_seg_excerpt", "exec"), ns)   # executed, not stored
counts2 = torch.rand(2, 50, generator=g)
countsK = torch.rand(6, 50, generator=g)
np.savez(os.path.join(OUT, "multi_instance_opt.npz"), counts2=counts2.numpy(),
         labels2=ns["multi_instance_opt"](counts2, 0.0).numpy(), countsK=countsK.numpy(),
         labelsK=ns["multi_instance_opt"](countsK, 0.0).numpy(),
         labelsK_g=ns["multi_instance_opt"](countsK, 0.2).numpy())

# (7) argument marshalling of render() / flashsplat_render()
P = 10
gm = GaussianModel(3)
gm._xyz = torch.randn(P, 3, generator=g).requires_grad_()
gm._features_dc = torch.randn(P, 1, 3, generator=g).requires_grad_()
gm._features_rest = torch.randn(P, 15, 3, generator=g).requires_grad_()
gm._scaling = torch.randn(P, 3, generator=g).requires_grad_()
gm._rotation = torch.randn(P, 4, generator=g).requires_grad_()
gm._opacity = torch.randn(P, 1, generator=g).requires_grad_()
gm.active_sh_degree = 2
cam = MiniCam(64, 48, cams["fov0"][1], cams["fov0"][0], 0.01, 100.0, torch.tensor(cams["view0"]), torch.tensor(cams["full0"]))
bg = torch.zeros(3)


class Pipe:
    convert_SHs_python = False
    compute_cov3D_python = False
    debug = False


out = gaussian_renderer.render(cam, gm, Pipe(), bg)
keys_render = sorted(out.keys())
Pipe.convert_SHs_python = True
Pipe.compute_cov3D_python = True
gaussian_renderer.render(cam, gm, Pipe(), bg)
Pipe.convert_SHs_python = False
Pipe.compute_cov3D_python = False
used = torch.zeros(P, dtype=torch.bool)
used[::2] = True
out2 = gaussian_renderer.flashsplat_render(cam, gm, Pipe(), bg, gt_mask=torch.zeros(48, 64), obj_num=1)
gaussian_renderer.flashsplat_render(cam, gm, Pipe(), bg, used_mask=used)
json.dump({"captured": captured, "render_keys": keys_render, "flashsplat_keys": sorted(out2.keys())},
          open(os.path.join(OUT, "render_marshalling.json"), "w"), indent=1, default=str)
print("golden fixtures written to", OUT)

# (8) densify_and_prune on the reference's own GaussianModel (scene/gaussian_model.py:399-455) with its torch.optim.Adam
#     state surgery (:318-374): pre-state -> post-state, CPU random stream (torch.manual_seed) for the split samples.
from types import SimpleNamespace  # noqa: E402
torch.cuda.empty_cache = lambda: None


def _densify_case(seed, P, max_screen_size, tag, out):
    gg = torch.Generator().manual_seed(seed)
    gm = GaussianModel(3)
    gm.spatial_lr_scale = 1.0
    gm._xyz = torch.nn.Parameter(torch.randn(P, 3, generator=gg))
    gm._features_dc = torch.nn.Parameter(torch.randn(P, 1, 3, generator=gg))
    gm._features_rest = torch.nn.Parameter(torch.randn(P, 15, 3, generator=gg) * 0.1)
    gm._scaling = torch.nn.Parameter(torch.log(torch.rand(P, 3, generator=gg) * 0.06 + 1e-3))   # around percent_dense * extent = 0.02
    gm._scaling.data[::17] = torch.log(torch.tensor(0.5))                                       # some beyond 0.1 * extent
    gm._rotation = torch.nn.Parameter(torch.randn(P, 4, generator=gg))
    gm._opacity = torch.nn.Parameter(torch.randn(P, 1, generator=gg) * 3.0 - 2.0)               # some below min_opacity
    gm._which_object = torch.arange(P, dtype=torch.int32).reshape(P, 1)
    gm.max_radii2D = torch.rand(P, generator=gg) * 40
    args = SimpleNamespace(percent_dense=0.01, position_lr_init=0.00016, position_lr_final=0.0000016,
                           position_lr_delay_mult=0.01, position_lr_max_steps=30000, feature_lr=0.0025, opacity_lr=0.05,
                           scaling_lr=0.005, rotation_lr=0.001)
    gm.training_setup(args)
    for _ in range(2):
        for grp in gm.optimizer.param_groups:
            p = grp["params"][0]
            p.grad = torch.randn(p.shape, generator=gg) * 1e-2
        gm.optimizer.step()
    gm.xyz_gradient_accum = torch.rand(P, 1, generator=gg) * 6e-4
    gm.denom = (torch.rand(P, 1, generator=gg) > 0.1).float()          # zeros give 0/0 -> NaN -> 0 (:443)

    def snap(prefix):
        for grp in gm.optimizer.param_groups:
            p = grp["params"][0]
            st = gm.optimizer.state[p]
            out[f"{tag}_{prefix}_{grp['name']}"] = p.detach().numpy().copy()
            out[f"{tag}_{prefix}_m_{grp['name']}"] = st["exp_avg"].numpy().copy()
            out[f"{tag}_{prefix}_v_{grp['name']}"] = st["exp_avg_sq"].numpy().copy()
        out[f"{tag}_{prefix}_accum"] = gm.xyz_gradient_accum.numpy().copy()
        out[f"{tag}_{prefix}_denom"] = gm.denom.numpy().copy()
        out[f"{tag}_{prefix}_max_radii2D"] = gm.max_radii2D.numpy().copy()
        out[f"{tag}_{prefix}_which_object"] = gm._which_object.numpy().copy()
    snap("pre")
    torch.manual_seed(1000 + seed)
    gm.densify_and_prune(0.0002, 0.005, 2.0, max_screen_size)
    snap("post")
    out[f"{tag}_args"] = np.array([0.0002, 0.005, 2.0, -1.0 if max_screen_size is None else float(max_screen_size), 1000 + seed])


dens = {}
_densify_case(3, 300, 20, "a", dens)
_densify_case(4, 257, None, "b", dens)
np.savez_compressed(os.path.join(OUT, "densify.npz"), **dens)
print("densify fixture:", {k: v.shape for k, v in dens.items() if k.endswith("_xyz")})
