"""Python face of the CPU oracle.  TEST INFRASTRUCTURE ONLY (see oracle/w3d_oracle.c header).

Two independent restatements of the rasterizer behind Wheat-3DGS's
``gaussian_renderer.render()`` (reference gaussian_renderer/__init__.py:22-106) live here:

* ``COracle`` — ctypes binding of ``w3d_oracle.c`` (fp32, explicit forward AND explicit backward,
  i.e. the formulas a kernel implements).  Used as the checker of the HIP path and as the
  ``cpu_baseline`` of bench.py (kind "port").
* ``torch_render`` — a differentiable pure-PyTorch restatement whose backward comes from
  ``torch.autograd``; it exists to check the explicit backward formulas of the C oracle
  (and runs in float64 for finite-difference quality).  Small scenes only.

PARITY UNPINNED: the reference's rasterizer sources are absent (un-vendored submodules), so
neither restatement can be compared with the reference CUDA code; they are pinned against the
importable reference pieces through tests/golden/ and against each other.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libw3d_oracle.so")


def build(force=False):
    """Compile w3d_oracle.c with gcc (seconds)."""
    src = os.path.join(_HERE, "w3d_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


class _View(ctypes.Structure):
    _fields_ = [("H", ctypes.c_int), ("W", ctypes.c_int),
                ("tanfovx", ctypes.c_float), ("tanfovy", ctypes.c_float),
                ("scale_modifier", ctypes.c_float),
                ("sh_degree", ctypes.c_int), ("sh_coeffs", ctypes.c_int),
                ("bg", ctypes.c_float * 3), ("view", ctypes.c_float * 16),
                ("proj", ctypes.c_float * 16), ("campos", ctypes.c_float * 3)]


def _f32(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _ptr(a, ty=ctypes.c_float):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ty))


class COracle:
    """One forward (+ optional backward) through the C restatement."""

    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            build()
            lib = ctypes.CDLL(_LIB_PATH)
            lib.w3do_forward.restype = ctypes.c_void_p
            lib.w3do_num_rendered.restype = ctypes.c_long
            lib.w3do_num_rendered.argtypes = [ctypes.c_void_p]
            lib.w3do_free.argtypes = [ctypes.c_void_p]
            cls._lib = lib
        return cls._lib

    def __init__(self, H, W, tanfovx, tanfovy, bg, viewmatrix, projmatrix, campos, sh_degree=3,
                 scale_modifier=1.0, sh_coeffs=16, nthreads=1):
        v = _View()
        v.H, v.W = int(H), int(W)
        v.tanfovx, v.tanfovy = float(tanfovx), float(tanfovy)
        v.scale_modifier = float(scale_modifier)
        v.sh_degree, v.sh_coeffs = int(sh_degree), int(sh_coeffs)
        v.bg[:] = [float(x) for x in np.asarray(bg).reshape(-1)]
        v.view[:] = [float(x) for x in np.asarray(viewmatrix, dtype=np.float32).reshape(-1)]
        v.proj[:] = [float(x) for x in np.asarray(projmatrix, dtype=np.float32).reshape(-1)]
        v.campos[:] = [float(x) for x in np.asarray(campos).reshape(-1)]
        self.v = v
        self.H, self.W = int(H), int(W)
        self.nthreads = int(nthreads)
        self.h = None
        self.inputs = None

    def __del__(self):
        self.free()

    def free(self):
        if self.h is not None:
            self.lib().w3do_free(ctypes.c_void_p(self.h))
            self.h = None

    def forward(self, means3D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, gt_mask=None, num_obj=0):
        self.free()
        lib = self.lib()
        means3D = _f32(means3D)
        P = means3D.shape[0]
        self.P = P
        opacities = _f32(opacities).reshape(-1)
        shs, colors_precomp = _f32(shs), _f32(colors_precomp)
        scales, rotations, cov3D_precomp = _f32(scales), _f32(rotations), _f32(cov3D_precomp)
        if (shs is None) == (colors_precomp is None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
           ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        if shs is not None:
            assert shs.shape[1] == self.v.sh_coeffs and shs.shape[2] == 3
        H, W = self.H, self.W
        color = np.zeros((3, H, W), np.float32)
        depth = np.zeros((1, H, W), np.float32)
        alpha = np.zeros((1, H, W), np.float32)
        radii = np.zeros((P,), np.int32)
        flash = gt_mask is not None or num_obj > 0
        used_count = contrib_num = proj_xy = gs_depth = None
        gt = _f32(gt_mask)
        if flash:
            used_count = np.zeros((num_obj + 1, P), np.float32)
            contrib_num = np.zeros((H, W), np.int32)
            proj_xy = np.zeros((P, 2), np.float32)
            gs_depth = np.zeros((P,), np.float32)
        self.h = lib.w3do_forward(ctypes.byref(self.v), ctypes.c_int(P), _ptr(means3D), _ptr(shs),
                                  _ptr(colors_precomp), _ptr(opacities), _ptr(scales), _ptr(rotations),
                                  _ptr(cov3D_precomp), _ptr(color), _ptr(depth), _ptr(alpha),
                                  _ptr(radii, ctypes.c_int), ctypes.c_int(self.nthreads),
                                  _ptr(gt), ctypes.c_int(num_obj), _ptr(used_count),
                                  _ptr(contrib_num, ctypes.c_int), _ptr(proj_xy), _ptr(gs_depth))
        self.inputs = dict(means3D=means3D, shs=shs, colors_precomp=colors_precomp, opacities=opacities,
                           scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)
        out = dict(color=color, depth=depth, alpha=alpha, radii=radii)
        if flash:
            out.update(used_count=used_count, contrib_num=contrib_num, proj_xy=proj_xy, gs_depth=gs_depth)
        return out

    @classmethod
    def set_exp_mode(cls, mode):
        """0: expf(power) (the restated algorithm).  1: exp2f(power * log2 e) — a rounding-sensitivity probe for the tests:
        the spread between the two runs is the uncertainty any fp32 implementation has against this oracle."""
        cls.lib().w3do_set_exp_mode(ctypes.c_int(int(mode)))

    def fragile_pixels(self, eps):
        """(H,W) bool: pixels whose walk meets a (pixel, Gaussian) pair within `eps` (relative) of one of the blend's
        thresholds — alpha vs 1/255, transmittance vs 1e-4, power vs 0 (w3d_oracle.c: w3do_fragile_pixels)."""
        out = np.zeros((self.H, self.W), np.uint8)
        lib = self.lib()
        lib.w3do_fragile_pixels.argtypes = [ctypes.c_void_p, ctypes.c_float, ctypes.POINTER(ctypes.c_ubyte)]
        lib.w3do_fragile_pixels(ctypes.c_void_p(self.h), ctypes.c_float(eps), _ptr(out, ctypes.c_ubyte))
        return out.astype(bool)

    def contributors_of(self, pixel_mask):
        """(P,) bool: Gaussians blended at any pixel of `pixel_mask` (H,W) (w3do_mark_contributors)."""
        pm = np.ascontiguousarray(np.asarray(pixel_mask).reshape(self.H, self.W).astype(np.uint8))
        flags = np.zeros(self.P, np.uint8)
        lib = self.lib()
        lib.w3do_mark_contributors.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ubyte), ctypes.POINTER(ctypes.c_ubyte)]
        lib.w3do_mark_contributors(ctypes.c_void_p(self.h), _ptr(pm, ctypes.c_ubyte), _ptr(flags, ctypes.c_ubyte))
        return flags.astype(bool)

    def num_rendered(self):
        return int(self.lib().w3do_num_rendered(ctypes.c_void_p(self.h)))

    def binning(self):
        gx, gy = (self.W + 15) // 16, (self.H + 15) // 16
        ranges = np.zeros((gx * gy, 2), np.uint32)
        pl = np.zeros((max(self.num_rendered(), 1),), np.uint32)
        self.lib().w3do_get_binning(ctypes.c_void_p(self.h), _ptr(ranges, ctypes.c_uint32), _ptr(pl, ctypes.c_uint32))
        return ranges, pl[: self.num_rendered()]

    def geom(self):
        P = self.P
        d = dict(depth=np.zeros(P, np.float32), xy=np.zeros((P, 2), np.float32),
                 conic_opacity=np.zeros((P, 4), np.float32), rgb=np.zeros((P, 3), np.float32),
                 cov3D=np.zeros((P, 6), np.float32), rect=np.zeros((P, 4), np.int32),
                 clamped=np.zeros((P, 3), np.uint8))
        self.lib().w3do_get_geom(ctypes.c_void_p(self.h), _ptr(d["depth"]), _ptr(d["xy"]), _ptr(d["conic_opacity"]),
                                 _ptr(d["rgb"]), _ptr(d["cov3D"]), _ptr(d["rect"], ctypes.c_int),
                                 _ptr(d["clamped"], ctypes.c_ubyte))
        return d

    def pixel_state(self):
        ft = np.zeros((self.H, self.W), np.float32)
        nc = np.zeros((self.H, self.W), np.uint32)
        self.lib().w3do_get_pixel_state(ctypes.c_void_p(self.h), _ptr(ft), _ptr(nc, ctypes.c_uint32))
        return ft, nc

    def backward(self, dL_dcolor, dL_ddepth=None, dL_dalpha=None, abs_sums=False):
        """abs_sums=True: also g["means2D_abs"] (P,2) float64 = sum over the summands of |term| of dL/dmean2D.x / .y — the
        conditioning of the densification statistic (w3do_set_abs_sums)."""
        assert self.h is not None, "forward first"
        i = self.inputs
        P = self.P
        absbuf = np.zeros((P, 2), np.float64) if abs_sums else None
        self.lib().w3do_set_abs_sums.argtypes = [ctypes.POINTER(ctypes.c_double)]
        self.lib().w3do_set_abs_sums(_ptr(absbuf, ctypes.c_double))
        M = self.v.sh_coeffs
        dL_dcolor = _f32(dL_dcolor).reshape(3, self.H, self.W)
        dL_ddepth = None if dL_ddepth is None else _f32(dL_ddepth).reshape(self.H, self.W)
        dL_dalpha = None if dL_dalpha is None else _f32(dL_dalpha).reshape(self.H, self.W)
        g = dict(means3D=np.zeros((P, 3), np.float32), means2D=np.zeros((P, 3), np.float32),
                 opacities=np.zeros((P, 1), np.float32))
        g["colors_precomp"] = np.zeros((P, 3), np.float32) if i["colors_precomp"] is not None else None
        g["shs"] = np.zeros((P, M, 3), np.float32) if i["shs"] is not None else None
        g["scales"] = np.zeros((P, 3), np.float32) if i["scales"] is not None else None
        g["rotations"] = np.zeros((P, 4), np.float32) if i["rotations"] is not None else None
        g["cov3D"] = np.zeros((P, 6), np.float32)
        self.lib().w3do_backward(ctypes.c_void_p(self.h), ctypes.byref(self.v), _ptr(i["means3D"]), _ptr(i["shs"]),
                                 _ptr(i["colors_precomp"]), _ptr(i["opacities"]), _ptr(i["scales"]),
                                 _ptr(i["rotations"]), _ptr(i["cov3D_precomp"]), _ptr(dL_dcolor), _ptr(dL_ddepth),
                                 _ptr(dL_dalpha), _ptr(g["means3D"]), _ptr(g["means2D"]), _ptr(g["colors_precomp"]),
                                 _ptr(g["shs"]), _ptr(g["opacities"]), _ptr(g["scales"]), _ptr(g["rotations"]),
                                 _ptr(g["cov3D"]), ctypes.c_int(self.nthreads))
        self.lib().w3do_set_abs_sums(None)
        if abs_sums:
            g["means2D_abs"] = absbuf
        if i["cov3D_precomp"] is None:
            g["cov3D_precomp"] = None
        else:
            g["cov3D_precomp"] = g["cov3D"]
        return g


def knn_dist2(points, nthreads=1):
    """Mean squared distance to the 3 nearest other points (reference scene/gaussian_model.py:148)."""
    pts = _f32(points)
    out = np.zeros((pts.shape[0],), np.float32)
    COracle.lib().w3do_knn_dist2(ctypes.c_int(pts.shape[0]), _ptr(pts), _ptr(out), ctypes.c_int(nthreads))
    return out


# --------------------------------------------------------------------------------------------
# Differentiable PyTorch restatement (autograd gives the backward).  Small scenes only.
# --------------------------------------------------------------------------------------------
_C0 = 0.28209479177387814
_C1 = 0.4886025119029199
_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
       1.445305721320277, -0.5900435899266435]


def torch_sh_to_rgb(deg, shs, means3D, campos):
    """SH (P,M,3) -> RGB (P,3), +0.5 and clamp at 0 — the python branch of the reference
    (gaussian_renderer/__init__.py:77-82 with utils/sh_utils.py:57-112)."""
    import torch
    d = means3D - campos[None, :]
    d = d / d.norm(dim=1, keepdim=True)
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    r = _C0 * shs[:, 0]
    if deg > 0:
        r = r - _C1 * y * shs[:, 1] + _C1 * z * shs[:, 2] - _C1 * x * shs[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        r = (r + _C2[0] * xy * shs[:, 4] + _C2[1] * yz * shs[:, 5] + _C2[2] * (2 * zz - xx - yy) * shs[:, 6]
             + _C2[3] * xz * shs[:, 7] + _C2[4] * (xx - yy) * shs[:, 8])
    if deg > 2:
        r = (r + _C3[0] * y * (3 * xx - yy) * shs[:, 9] + _C3[1] * xy * z * shs[:, 10]
             + _C3[2] * y * (4 * zz - xx - yy) * shs[:, 11] + _C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * shs[:, 12]
             + _C3[4] * x * (4 * zz - xx - yy) * shs[:, 13] + _C3[5] * z * (xx - yy) * shs[:, 14]
             + _C3[6] * x * (xx - 3 * yy) * shs[:, 15])
    return torch.clamp_min(r + 0.5, 0.0)


def torch_cov3d(scales, rotations, mod=1.0):
    """(P,6) covariance from scale and (unit) quaternion; reference scene/gaussian_model.py:27-31."""
    import torch
    r, x, y, z = rotations[:, 0], rotations[:, 1], rotations[:, 2], rotations[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)
    L = R * (mod * scales)[:, None, :]
    S = L @ L.transpose(1, 2)
    return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], dim=1)


def torch_render(H, W, tanfovx, tanfovy, bg, viewmatrix, projmatrix, campos, means3D, opacities, means2D=None,
                 shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, sh_degree=3,
                 scale_modifier=1.0):
    """Differentiable restatement of SURVEY.md Appendix A.1-A.3.  Returns (color, radii, depth, alpha).

    ``means2D`` (P,3), if given, is added (scaled to pixels) to the projected centre so that its
    autograd gradient is the screen-space gradient with the reference's W/2, H/2 convention.
    Deviations from plain autograd that mirror the published explicit backward: the 0.99 alpha
    cap passes gradient straight through; the FoV clamp contributes no gradient.
    """
    import torch
    dt = means3D.dtype
    P = means3D.shape[0]
    V = viewmatrix.to(dt)
    M = projmatrix.to(dt)
    bg = bg.to(dt)
    ones = torch.ones(P, 1, dtype=dt)
    p_view = torch.cat([means3D, ones], 1) @ V          # row-vector convention: transposed matrices
    p_hom = torch.cat([means3D, ones], 1) @ M
    depth = p_view[:, 2]
    pw = 1.0 / (p_hom[:, 3] + 1e-7)
    p_proj = p_hom[:, :3] * pw[:, None]
    if cov3D_precomp is not None:
        cov3D = cov3D_precomp
    else:
        cov3D = torch_cov3d(scales, rotations, scale_modifier)
    fx, fy = W / (2 * tanfovx), H / (2 * tanfovy)
    tz = p_view[:, 2]
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    txtz, tytz = p_view[:, 0] / tz, p_view[:, 1] / tz
    clx = (txtz < -limx) | (txtz > limx)
    cly = (tytz < -limy) | (tytz > limy)
    tx = torch.where(clx, (txtz.clamp(-limx, limx) * tz).detach(), p_view[:, 0])
    ty = torch.where(cly, (tytz.clamp(-limy, limy) * tz).detach(), p_view[:, 1])
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -(fx * tx) / (tz * tz), zero, fy / tz, -(fy * ty) / (tz * tz)], 1).reshape(P, 2, 3)
    Wm = V[:3, :3].t()   # maths rotation: W(r,c) = V[c, r] in torch (V is the transposed matrix)
    T = J @ Wm[None]
    S = torch.stack([cov3D[:, 0], cov3D[:, 1], cov3D[:, 2], cov3D[:, 1], cov3D[:, 3], cov3D[:, 4],
                     cov3D[:, 2], cov3D[:, 4], cov3D[:, 5]], 1).reshape(P, 3, 3)
    cov2 = T @ S @ T.transpose(1, 2)
    a, b, c = cov2[:, 0, 0] + 0.3, cov2[:, 0, 1], cov2[:, 1, 1] + 0.3
    det = a * c - b * b
    conic = torch.stack([c / det, -b / det, a / det], 1)
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam)).detach()
    px = ((p_proj[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((p_proj[:, 1] + 1.0) * H - 1.0) * 0.5
    if means2D is not None:
        px = px + means2D[:, 0] * (0.5 * W)
        py = py + means2D[:, 1] * (0.5 * H)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    pxd, pyd = px.detach(), py.detach()
    minx = ((pxd - radius) / 16).trunc().clamp(0, gx).long()
    maxx = ((pxd + radius + 15) / 16).trunc().clamp(0, gx).long()
    miny = ((pyd - radius) / 16).trunc().clamp(0, gy).long()
    maxy = ((pyd + radius + 15) / 16).trunc().clamp(0, gy).long()
    visible = (depth.detach() > 0.2) & (det.detach() != 0) & ((maxx - minx) * (maxy - miny) > 0)
    radii = torch.where(visible, radius, torch.zeros_like(radius)).to(torch.int32)
    if colors_precomp is not None:
        rgb = colors_precomp
    else:
        rgb = torch_sh_to_rgb(sh_degree, shs, means3D, campos.to(dt))
    # depth order, ties by index (stable)
    key = depth.detach().to(torch.float32).contiguous().view(torch.int32).to(torch.int64)
    order = torch.argsort(key, stable=True)
    order = order[visible[order]]
    color = torch.zeros(3, H, W, dtype=dt)
    odepth = torch.zeros(1, H, W, dtype=dt)
    oalpha = torch.zeros(1, H, W, dtype=dt)
    opac = opacities.reshape(-1)
    for tyi in range(gy):
        for txi in range(gx):
            sel = order[(minx[order] <= txi) & (txi < maxx[order]) & (miny[order] <= tyi) & (tyi < maxy[order])]
            y0, y1 = tyi * 16, min(tyi * 16 + 16, H)
            x0, x1 = txi * 16, min(txi * 16 + 16, W)
            ys, xs = torch.meshgrid(torch.arange(y0, y1, dtype=dt), torch.arange(x0, x1, dtype=dt), indexing="ij")
            ys, xs = ys.reshape(-1, 1), xs.reshape(-1, 1)
            n = sel.numel()
            if n == 0:
                color[:, y0:y1, x0:x1] = bg[:, None, None]
                continue
            dx = px[sel][None, :] - xs
            dy = py[sel][None, :] - ys
            q = conic[sel]
            power = -0.5 * (q[:, 0][None] * dx * dx + q[:, 2][None] * dy * dy) - q[:, 1][None] * dx * dy
            araw = opac[sel][None] * torch.exp(power)
            alpha = araw + (torch.clamp_max(araw, 0.99) - araw).detach()     # straight-through cap
            accepted = (power.detach() <= 0) & (alpha.detach() >= 1.0 / 255.0)
            aeff = torch.where(accepted, alpha, torch.zeros_like(alpha))
            Tafter = torch.cumprod(1 - aeff, dim=1)
            Tbefore = torch.cat([torch.ones_like(Tafter[:, :1]), Tafter[:, :-1]], 1)
            stop = accepted & (Tafter.detach() < 1e-4)
            stopped = torch.cumsum(stop.to(torch.int32), 1) > 0
            applied = accepted & ~stopped
            w = torch.where(applied, aeff * Tbefore, torch.zeros_like(aeff))
            Tfinal = torch.where(applied, 1 - aeff, torch.ones_like(aeff)).prod(dim=1)
            C = w @ rgb[sel] + Tfinal[:, None] * bg[None, :]
            D = w @ depth[sel]
            A = w.sum(1)
            hh, ww = y1 - y0, x1 - x0
            color[:, y0:y1, x0:x1] = C.t().reshape(3, hh, ww)
            odepth[0, y0:y1, x0:x1] = D.reshape(hh, ww)
            oalpha[0, y0:y1, x0:x1] = A.reshape(hh, ww)
    return color, radii, odepth, oalpha
