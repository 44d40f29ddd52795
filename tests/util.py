"""Shared helpers of the test-suite (inputs for one view of a synthetic scene)."""
import math

import numpy as np
import torch

from w3d_amd.synth import small_test_scene, make_scene, make_cameras


def view_inputs(sc, cam, sh_degree=3, precomp_color=False, precomp_cov=False, scale_modifier=1.0, device=None):
    """Activated rasterizer inputs exactly as reference render() marshals them
    (gaussian_renderer/__init__.py:57-84).  device: where the activations of scene/gaussian_model.py:33-41 are evaluated — the
    reference evaluates them with torch ON THE GPU and hands the results to the rasterizer, so the tests of the raw-parameter
    kernels (which must reproduce those bits) feed the oracle torch's device results; None: the host's."""
    from oracle.oracle import torch_cov3d, torch_sh_to_rgb
    means = sc.xyz.float().contiguous()
    act = (lambda f, t: f(t.float().to(device)).cpu()) if device is not None else (lambda f, t: f(t))  # noqa: E731
    opac = act(torch.sigmoid, sc.opacity).float().contiguous()
    scales = act(torch.exp, sc.scaling).float().contiguous()
    rots = act(torch.nn.functional.normalize, sc.rotation).float().contiguous()
    shs = torch.cat([sc.features_dc, sc.features_rest], 1).float().contiguous()
    d = dict(means3D=means, opacities=opac, shs=shs, colors_precomp=None, scales=scales, rotations=rots,
             cov3D_precomp=None)
    if precomp_color:
        d["colors_precomp"] = torch_sh_to_rgb(sh_degree, shs, means, cam.camera_center.float()).contiguous()
        d["shs"] = None
    if precomp_cov:
        d["cov3D_precomp"] = torch_cov3d(scales, rots, scale_modifier).contiguous()
        d["scales"] = d["rotations"] = None
    return d


def cam_settings(cam, bg, sh_degree=3, scale_modifier=1.0):
    return dict(H=cam.image_height, W=cam.image_width, tanfovx=math.tan(cam.FoVx * 0.5),
                tanfovy=math.tan(cam.FoVy * 0.5), bg=bg, viewmatrix=cam.world_view_transform,
                projmatrix=cam.full_proj_transform, campos=cam.camera_center, sh_degree=sh_degree,
                scale_modifier=scale_modifier)


def make_oracle(cam, bg, sh_degree=3, scale_modifier=1.0, nthreads=1):
    from oracle.oracle import COracle
    s = cam_settings(cam, bg, sh_degree, scale_modifier)
    return COracle(s["H"], s["W"], s["tanfovx"], s["tanfovy"], np.asarray(bg, np.float32),
                   cam.world_view_transform.cpu().numpy(), cam.full_proj_transform.cpu().numpy(),
                   cam.camera_center.cpu().numpy(), sh_degree=sh_degree, scale_modifier=scale_modifier,
                   nthreads=nthreads)


def np_inputs(d):
    return {k: (None if v is None else v.detach().cpu().numpy()) for k, v in d.items()}


def psnr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    mse = ((a - b) ** 2).mean()
    return 20 * math.log10(1.0 / math.sqrt(mse)) if mse > 0 else float("inf")


def rel_err(x, ref):
    x, ref = np.asarray(x, np.float64), np.asarray(ref, np.float64)
    return float(np.abs(x - ref).max() / (np.abs(ref).max() + 1e-30))


def needle_scene(seed, P=20000):
    """A stress scene for the footprint culls: needles — one axis 0.05 ... 8 scene units, the two others 1e-5 ... 2e-3, so the
    projected conics have condition numbers up to 1e7 — many of them centred outside the frame.  For such a Gaussian the blend's
    fp32 exponent carries an absolute rounding error of 0.01 ... 1 hundreds of pixels from the centre."""
    import math
    import torch
    from w3d_amd.synth import make_scene
    sc = make_scene(P, seed=100 + seed, scale_mean=0.03)
    g = torch.Generator().manual_seed(seed)
    sc.scaling[:, 0] = torch.empty(P).uniform_(math.log(0.05), math.log(8.0), generator=g)
    sc.scaling[:, 1:] = torch.empty(P, 2).uniform_(math.log(1e-5), math.log(2e-3), generator=g)
    sc.xyz[:, 0] *= 3.0
    sc.xyz[:, 1] *= 3.0
    sc.opacity[:] = torch.empty(P, 1).normal_(0.0, 2.0, generator=g)
    return sc


# ------------------------------------------------------------------ the import redirect on the stand-in checkout
STANDIN_TOPS = ("scene", "gaussian_renderer", "utils", "train_loop")


class standin_checkout:
    """Context manager: `tests/standin_checkout/` (the reference's module names and import lines, tests/standin_checkout/README.md)
    first on sys.path, its modules freshly imported — under w3d_amd.dropin's redirect (hook=True: all of it; a tuple: those
    modules only) or as they are (hook=False: only the rasterizer packages are this repo's).  Yields the `train_loop` module;
    on exit the redirect is removed and the stand-in's modules are dropped from sys.modules."""

    def __init__(self, hook):
        self.hook = hook

    def __enter__(self):
        import importlib
        import os
        import sys
        from w3d_amd import dropin
        self.dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "standin_checkout")
        dropin.uninstall(purge=STANDIN_TOPS)
        sys.path.insert(0, self.dir)
        if self.hook:
            dropin.install(only=None if self.hook is True else tuple(self.hook))
        mod = importlib.import_module("train_loop")
        assert os.path.dirname(os.path.abspath(mod.__file__)) == self.dir, mod.__file__
        return mod

    def __exit__(self, *exc):
        import sys
        from w3d_amd import dropin
        dropin.uninstall(purge=STANDIN_TOPS)
        if self.dir in sys.path:
            sys.path.remove(self.dir)
        return False


def checkpoint_tuple(sc, sh_degree=3, spatial_lr_scale=1.0, device="cuda"):
    """A scene's pre-activation blocks as the 13-tuple of reference GaussianModel.capture() (scene/gaussian_model.py:63-79)
    with no optimizer state: what `--start_checkpoint` hands to restore() (train_vanilla_3dgs.py:38-40)."""
    P = sc.xyz.shape[0]
    t = lambda x: x.to(device).float().contiguous()  # noqa: E731
    return (sh_degree, t(sc.xyz), t(sc.features_dc), t(sc.features_rest), t(sc.scaling), t(sc.rotation), t(sc.opacity),
            torch.zeros(P, 1, dtype=torch.int, device=device), torch.zeros(P, device=device), torch.zeros(P, 1, device=device),
            torch.zeros(P, 1, device=device), None, spatial_lr_scale)


# ------------------------------------------------------------------ full-size gradient statistics (tests/test_gpu_fullsize.py, bench.py's parity_tail)

def raw_grads_from_oracle(gref, sc):
    """Chain the oracle's gradients w.r.t. the ACTIVATED inputs through the activations of reference
    scene/gaussian_model.py:33-41 (exp, sigmoid, normalize, cat(dc, rest)) — float64 torch formulas."""
    t = lambda a: torch.from_numpy(np.asarray(a)).double()  # noqa: E731
    out = {"xyz": t(gref["means3D"])}
    shs = t(gref["shs"])
    out["f_dc"], out["f_rest"] = shs[:, :1], shs[:, 1:]
    o = torch.sigmoid(sc.opacity.double())
    out["opacity"] = t(gref["opacities"]).reshape(-1, 1) * o * (1 - o)
    out["scaling"] = t(gref["scales"]) * torch.exp(sc.scaling.double())
    r = sc.rotation.double()
    n = r.norm(dim=1, keepdim=True)
    q = r / n
    gq = t(gref["rotations"])
    out["rotation"] = (gq - q * (q * gq).sum(1, keepdim=True)) / n
    return {k: v.numpy() for k, v in out.items()}


def per_gaussian_error(got, ref):
    """max_d |got - ref| / max_d |ref| per Gaussian, over the Gaussians whose reference gradient is not zero (hidden
    Gaussians behind saturated pixels have none; a tiny floor keeps denormal-sized gradients from dominating)."""
    got, ref = np.asarray(got, np.float64).reshape(len(ref), -1), np.asarray(ref, np.float64).reshape(len(ref), -1)
    mag = np.abs(ref).max(1)
    nz = mag > 0
    floor = 1e-4 * np.median(mag[nz]) if nz.any() else 1.0
    return np.abs(got - ref).max(1) / (mag + floor), nz, np.abs(got - ref).max(1) / (1e-4 * mag + 1e-3 * (np.median(mag[nz]) if nz.any() else 1.0))


def flip_pixels(out, ref):
    """Pixels whose contributor set evidently differs between the two implementations: a colour / alpha difference far
    above fp32 noise (5e-5; rounding noise of the blend is ~1e-6).  (n_contrib cannot be compared: it is a position in the
    tile's list, and the footprint-culled lists of the product path are shorter than the oracle's.)"""
    d = np.abs(out["color"] - ref["color"]).max(0) > 5e-5
    d |= np.abs(out["alpha"][0] - ref["alpha"][0]) > 5e-5
    return int(d.sum())


def gradient_stats(got, want, vis, bulk=1e-4):
    """Per block: percentiles of the per-Gaussian relative error, number of Gaussians beyond `bulk`."""
    stats = {}
    for k, ref in want.items():
        g = np.asarray(got[k]).reshape(ref.shape)
        e, nz, mixed = per_gaussian_error(g, ref)
        e = e[nz & vis]
        stats[k] = dict(n=int(e.size), p50=float(np.percentile(e, 50)), p99=float(np.percentile(e, 99)),
                        p999=float(np.percentile(e, 99.9)), max=float(e.max()), outliers=int((e > bulk).sum()),
                        worst_mixed=float(mixed[vis].max()),
                        culled_zero=bool(np.all(g.reshape(len(ref), -1)[~vis] == 0)))
    return stats


def densify_norm_error(n_own, n_ref, vis):
    nz = vis & (n_ref > 0)
    e = np.abs(n_own - n_ref)[nz] / (n_ref[nz] + 1e-4 * np.median(n_ref[nz]))
    return dict(n=int(e.size), p50=float(np.percentile(e, 50)), p99=float(np.percentile(e, 99)), p999=float(np.percentile(e, 99.9)),
                max=float(e.max()), outliers=int((e > 1e-4).sum()))
