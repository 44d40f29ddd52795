"""Seeded synthetic wheat-plot scenes and cameras.

No wheat-plot data ships with the reference, so every config of BASELINE.json runs on this
generator: a plot-shaped slab of Gaussians seen by 36 overhead cameras (3 rows x 12).
The camera matrices follow the reference's conventions exactly:
  * world->view from (R, T) with the values of utils/graphics_utils.py:38-49 (getWorld2View2),
  * projection with the values of utils/graphics_utils.py:51-71 (getProjectionMatrix), znear 0.01 / zfar 100
    (scene/cameras.py:50-51),
  * both handed to the rasterizer TRANSPOSED, full_proj = view^T-form @ proj^T-form, camera
    centre = inverse(view^T-form)[3, :3]   (scene/cameras.py:56-59).
Everything is generated on the CPU with a fixed torch.Generator and then moved to the device.
"""
import math
from dataclasses import dataclass

import numpy as np
import torch


def world_to_view(R, t):
    """4x4 world->view matrix of a camera stored the reference's way (R = camera-to-world rotation, t = world-to-camera
    translation: scene/dataset_readers.py, utils/graphics_utils.py:38-49 with the default translate = 0, scale = 1, for
    which the centre re-normalisation there is the identity).  Pinned by tests/golden/camera.npz."""
    m = np.eye(4)
    m[:3, :3] = np.asarray(R, np.float64).T
    m[:3, 3] = np.asarray(t, np.float64)
    return m


def perspective(znear, zfar, fovX, fovY):
    """The reference's projection (utils/graphics_utils.py:51-71): a symmetric frustum, z mapped to [0, 1], w = z."""
    tx, ty = math.tan(0.5 * fovX), math.tan(0.5 * fovY)
    return torch.tensor([[1.0 / tx, 0.0, 0.0, 0.0],
                         [0.0, 1.0 / ty, 0.0, 0.0],
                         [0.0, 0.0, zfar / (zfar - znear), -(zfar * znear) / (zfar - znear)],
                         [0.0, 0.0, 1.0, 0.0]], dtype=torch.float32)


def focal2fov(focal, pixels):
    return 2 * math.atan(pixels / (2 * focal))


class SynthCamera:
    """Duck-types the fields of reference scene/cameras.py Camera / MiniCam that render() reads."""

    def __init__(self, uid, R, T, FoVx, FoVy, width, height, device="cpu"):
        self.uid = uid
        self.R, self.T = R, T
        self.FoVx, self.FoVy = FoVx, FoVy
        self.image_width, self.image_height = int(width), int(height)
        self.znear, self.zfar = 0.01, 100.0
        wvt = torch.tensor(world_to_view(R, T).astype(np.float32)).transpose(0, 1)
        proj = perspective(self.znear, self.zfar, FoVx, FoVy).transpose(0, 1)
        self.world_view_transform = wvt.to(device)
        self.projection_matrix = proj.to(device)
        self.full_proj_transform = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0).to(device)
        self.camera_center = wvt.inverse()[3, :3].to(device)
        self.original_image = None

    def to(self, device):
        for k in ("world_view_transform", "projection_matrix", "full_proj_transform", "camera_center"):
            setattr(self, k, getattr(self, k).to(device))
        if self.original_image is not None:
            self.original_image = self.original_image.to(device)
        return self


def look_at_camera(uid, eye, target, width, height, focal_px, device="cpu"):
    """COLMAP-style camera (x right, y down, z forward) at `eye` looking at `target`."""
    eye = np.asarray(eye, np.float64)
    fwd = np.asarray(target, np.float64) - eye
    fwd /= np.linalg.norm(fwd)
    up_hint = np.array([0.0, 1.0, 0.0]) if abs(fwd[1]) < 0.95 else np.array([1.0, 0.0, 0.0])
    right = np.cross(fwd, up_hint)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    R_c2w = np.stack([right, down, fwd], axis=1)     # columns = camera axes in world
    R_w2c = R_c2w.T
    T = -R_w2c @ eye
    # reference stores R = transposed W2C rotation (== C2W), T = W2C translation (dataset_readers)
    return SynthCamera(uid, R_c2w, T, focal2fov(focal_px, width), focal2fov(focal_px, height), width, height, device)


def make_cameras(n=36, width=1600, height=1200, device="cpu", focal_mult=1.2):
    """3 rows x 12 cameras on arcs 2.0-2.5 units above the slab, looking at its centre."""
    cams = []
    rows = 3
    per_row = max(1, n // rows)
    target = np.array([0.0, 0.0, 0.3])
    for i in range(n):
        r, k = divmod(i, per_row)
        r = min(r, rows - 1)
        height_z = 2.0 + 0.25 * r
        ang = (k / per_row) * 2.0 * math.pi + 0.13 * r
        rad = 0.45 + 0.1 * r
        eye = np.array([rad * math.cos(ang), 0.6 * rad * math.sin(ang), 0.3 + height_z])
        cams.append(look_at_camera(i, eye, target, width, height, focal_mult * width, device))
    return cams


@dataclass
class SynthScene:
    xyz: torch.Tensor            # (P,3)
    features_dc: torch.Tensor    # (P,1,3)
    features_rest: torch.Tensor  # (P,15,3)
    scaling: torch.Tensor        # (P,3) log-scale (pre-activation)
    rotation: torch.Tensor       # (P,4) un-normalised quaternion (pre-activation)
    opacity: torch.Tensor        # (P,1) logit (pre-activation)

    def to(self, device):
        return SynthScene(*(getattr(self, f).to(device) for f in
                            ("xyz", "features_dc", "features_rest", "scaling", "rotation", "opacity")))

    def take(self, rows):
        """The scene with its Gaussians in another order (row r = old row rows[r])."""
        rows = rows.to(self.xyz.device)
        return SynthScene(*(getattr(self, f)[rows].contiguous() for f in
                            ("xyz", "features_dc", "features_rest", "scaling", "rotation", "opacity")))

    @property
    def P(self):
        return self.xyz.shape[0]


def make_scene(P, seed=0, scale_mean=0.006, scale_sigma=0.6):
    """Plot-shaped slab: xyz ~ U([-1.5,1.5]x[-0.75,0.75]x[0,0.6]),
    log-scale ~ N(log scale_mean, scale_sigma^2) per axis, quat ~ N(0,1)^4, opacity logit ~ N(0,2^2),
    SH DC ~ N(0,1), higher bands ~ N(0,0.15^2)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    u = torch.rand(P, 3, generator=g)
    xyz = torch.stack([u[:, 0] * 3.0 - 1.5, u[:, 1] * 1.5 - 0.75, u[:, 2] * 0.6], 1)
    scaling = math.log(scale_mean) + scale_sigma * torch.randn(P, 3, generator=g)
    rotation = torch.randn(P, 4, generator=g)
    opacity = 2.0 * torch.randn(P, 1, generator=g)
    f_dc = torch.randn(P, 1, 3, generator=g)
    f_rest = 0.15 * torch.randn(P, 15, 3, generator=g)
    return SynthScene(xyz.contiguous(), f_dc, f_rest, scaling, rotation, opacity)


def small_test_scene(P=200, W=64, H=48, seed=0, scale=0.05, n_cams=4):
    """A tiny scene for parity tests: fat Gaussians so a 64x48 image is well covered, one of
    them behind the camera (culled), one far outside the frustum (FoV clamp / empty rect)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    sc = make_scene(P, seed=seed, scale_mean=scale, scale_sigma=0.5)
    sc.xyz[:, 0] *= 0.35
    sc.xyz[:, 1] *= 0.5
    if P > 4:
        sc.xyz[1] = torch.tensor([0.0, 0.0, 5.0])       # behind / above the cameras -> near-culled
        sc.xyz[2] = torch.tensor([40.0, 0.0, 0.3])      # far off-axis -> FoV clamp, empty rect
        sc.features_dc[3] = torch.tensor([[-9.0, 0.2, 0.1]])   # forces an SH clamp on one channel
        sc.opacity[4] = 12.0                                    # alpha saturates at 0.99
        sc.scaling[4] = math.log(scale * 3)
    cams = make_cameras(n_cams, W, H, focal_mult=1.2)
    _ = g
    return sc, cams


def make_opaque_scene(seed=0, ground=400_000, heads=4000, per_head=30, per_stem=12):
    """A wheat plot made of OPAQUE surfaces — what a trained 3DGS model of a photographed scene converges to, and the regime
    make_scene's random translucent slab is not: a ground sheet of flat, nearly opaque discs with a low-frequency texture,
    and `heads` ellipsoidal ears of `per_head` opaque Gaussians on thin stems.  A pixel's ray meets a handful of them before
    it saturates (contributors per pixel in the tens at most), whatever the number of Gaussians behind.
    Pre-activation blocks as make_scene; seeded."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g)       # noqa: E731
    n = lambda *s: torch.randn(*s, generator=g)      # noqa: E731
    # ---- ground: jittered discs over the plot (slightly larger than the slab of make_scene so the image borders are covered)
    gx = r(ground) * 3.4 - 1.7
    gy = r(ground) * 1.9 - 0.95
    gz = 0.01 * n(ground)
    g_xyz = torch.stack([gx, gy, gz], 1)
    s0 = math.log(1.62 * math.sqrt(3.4 * 1.9 / ground))      # disc sigma = 1.6 x the mean spacing: the sheet is closed (0.0065 at 400 k)
    g_scale = torch.stack([s0 + 0.25 * n(ground), s0 + 0.25 * n(ground), s0 + math.log(0.15) + 0.2 * n(ground)], 1)
    g_rot = torch.cat([torch.ones(ground, 1), 0.05 * n(ground, 3)], 1)
    g_op = 3.5 + 0.5 * n(ground, 1)
    tex = 0.5 + 0.25 * torch.sin(7.0 * gx + 1.3) * torch.cos(9.0 * gy) + 0.15 * torch.sin(23.0 * gx * gy + 0.7)
    soil = torch.stack([0.45 * tex + 0.15, 0.32 * tex + 0.12, 0.18 * tex + 0.08], 1) + 0.05 * n(ground, 3)
    # ---- ears on stems
    hx, hy = r(heads) * 3.0 - 1.5, r(heads) * 1.5 - 0.75
    hz = 0.25 + 0.3 * r(heads)
    tilt = 0.05 * n(heads, 2)
    idx = torch.arange(heads).repeat_interleave(per_head)
    off = n(heads * per_head, 3) * torch.tensor([0.012, 0.012, 0.03])
    e_xyz = torch.stack([hx[idx], hy[idx], hz[idx]], 1) + off
    e_scale = math.log(0.006) + 0.3 * n(heads * per_head, 3)
    e_rot = n(heads * per_head, 4)
    e_op = 3.0 + 0.7 * n(heads * per_head, 1)
    hue = 0.75 + 0.2 * r(heads)
    ear = torch.stack([hue, 0.85 * hue, 0.25 + 0.1 * r(heads)], 1)[idx] + 0.06 * n(heads * per_head, 3)
    sidx = torch.arange(heads).repeat_interleave(per_stem)
    t = r(heads * per_stem)
    s_xyz = torch.stack([hx[sidx] + tilt[sidx, 0] * t, hy[sidx] + tilt[sidx, 1] * t, hz[sidx] * t], 1) + 0.001 * n(heads * per_stem, 3)
    s_scale = torch.stack([math.log(0.0025) + 0.2 * n(heads * per_stem), math.log(0.0025) + 0.2 * n(heads * per_stem),
                           math.log(0.02) + 0.2 * n(heads * per_stem)], 1)
    s_rot = torch.cat([torch.ones(heads * per_stem, 1), 0.03 * n(heads * per_stem, 3)], 1)
    s_op = 2.5 + 0.5 * n(heads * per_stem, 1)
    stem = torch.tensor([0.35, 0.55, 0.2]) + 0.05 * n(heads * per_stem, 3)
    xyz = torch.cat([g_xyz, e_xyz, s_xyz])
    rgb = torch.cat([soil, ear, stem]).clamp(0.02, 0.98)
    P = xyz.shape[0]
    f_dc = ((rgb - 0.5) / 0.28209479177387814)[:, None, :]
    f_rest = 0.03 * n(P, 15, 3)
    return SynthScene(xyz.contiguous(), f_dc.contiguous(), f_rest, torch.cat([g_scale, e_scale, s_scale]).contiguous(),
                      torch.cat([g_rot, e_rot, s_rot]).contiguous(), torch.cat([g_op, e_op, s_op]).contiguous())
