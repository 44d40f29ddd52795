"""The fused training-step path: the rasterizer called on the
PRE-ACTIVATION parameters of GaussianModel, gradients written straight into the flat gradient
bucket, densification statistics updated in the same backward kernel.  No autograd graph, no
activation / cat / split / accumulate kernels: PyTorch only owns the memory and the stream.

Same arithmetic as render() + loss.backward() of the drop-in path (reference
gaussian_renderer/__init__.py:22-106 followed by autograd through scene/gaussian_model.py:101-121);
tests/test_gpu_fused.py checks the two paths against each other.
"""
import ctypes
import math

import torch

from ._lib import W3DView, check, lib, ptr, stream_ptr
from .rasterizer import GaussianRasterizationSettings, _View

_vp, _i32 = ctypes.c_void_p, ctypes.c_int32


class W3DRawParams(ctypes.Structure):
    _fields_ = [(n, _vp) for n in ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")]


class W3DRawGrads(ctypes.Structure):
    _fields_ = [(n, _vp) for n in ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")]


class W3DDensifyStats(ctypes.Structure):
    _fields_ = [(n, _vp) for n in ("dL_dmeans2D", "grad2d_norm", "radii", "xyz_gradient_accum", "denom", "max_radii2D")]


lib.w3d_forward_stage1_raw.argtypes = [ctypes.POINTER(W3DView), _i32, ctypes.POINTER(W3DRawParams), _vp, _vp, _vp, _vp, _vp]
lib.w3d_forward_stage1_raw.restype = ctypes.c_int
lib.w3d_backward_raw.argtypes = [ctypes.POINTER(W3DView), _i32, ctypes.POINTER(W3DRawParams), _vp, _vp, _vp, _vp, _vp,
                                 ctypes.POINTER(W3DRawGrads), ctypes.POINTER(W3DDensifyStats), _vp, _vp]
lib.w3d_backward_raw.restype = ctypes.c_int


def _raw_params(model):
    p = W3DRawParams()
    p.xyz, p.f_dc, p.f_rest = model._xyz.data_ptr(), model._features_dc.data_ptr(), model._features_rest.data_ptr()
    p.opacity, p.scaling, p.rotation = model._opacity.data_ptr(), model._scaling.data_ptr(), model._rotation.data_ptr()
    return p


def render_raw(cam, model, bg_color, scaling_modifier=1.0):
    """Forward on the raw parameters.  Returns the dict of render() (minus viewspace_points) plus a
    `handle` for backward_raw()."""
    dev = model.flat.device
    if not model.flat.is_cuda:
        raise RuntimeError("the fused step needs the model on the GPU; there is no CPU path")
    P = model.num_points
    H, W = int(cam.image_height), int(cam.image_width)
    s = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), bg_color,
                                      scaling_modifier, cam.world_view_transform, cam.full_proj_transform,
                                      model.active_sh_degree, cam.camera_center, False, False)
    view = _View(s, (model.max_sh_degree + 1) ** 2, dev)
    prm = _raw_params(model)
    with torch.cuda.device(dev):
        stream = stream_ptr(dev)
        sb, tb = ctypes.c_uint64(), ctypes.c_uint64()
        check(lib.w3d_forward_sizes(P, H, W, ctypes.byref(sb), ctypes.byref(tb)))
        state = torch.empty(sb.value, dtype=torch.uint8, device=dev)
        scratch = torch.empty(tb.value, dtype=torch.uint8, device=dev)
        radii = torch.empty(P, dtype=torch.int32, device=dev)
        counts = (ctypes.c_uint32 * 2)()
        check(lib.w3d_forward_stage1_raw(ctypes.byref(view.c), P, ctypes.byref(prm), ptr(radii), ptr(state),
                                         ptr(scratch), ctypes.cast(counts, _vp), stream))
        R = int(counts[1])
        plist = torch.empty(max(R, 1), dtype=torch.int32, device=dev)
        color = torch.empty(3, H, W, dtype=torch.float32, device=dev)
        depth = torch.empty(1, H, W, dtype=torch.float32, device=dev)
        alpha = torch.empty(1, H, W, dtype=torch.float32, device=dev)
        check(lib.w3d_forward_stage2(ctypes.byref(view.c), P, ptr(state), ptr(scratch), ptr(plist), ctypes.c_uint64(R),
                                     ptr(color), ptr(depth), ptr(alpha), None, 0, None, None, None, None, stream))
    handle = dict(view=view, P=P, state=state, point_list=plist, radii=radii, num_rendered=R, num_visible=int(counts[0]))
    return {"render": color, "radii": radii, "depth": depth, "alpha": alpha, "handle": handle}


def backward_raw(model, handle, dL_dimage, dL_ddepth=None, dL_dalpha=None, update_stats=False, want_norm=False,
                 want_means2D=False):
    """Backward into model.flat_grad (OVERWRITTEN).  update_stats: add_densification_stats and the
    max_radii2D update happen inside the kernel (single-GPU step).  want_norm: also return the
    per-Gaussian ||dL/dmean2D|| (the view-parallel exchange needs it before reduction)."""
    dev = model.flat.device
    P, view = handle["P"], handle["view"]
    if P != model.num_points:
        raise RuntimeError("model was resized between forward and backward")
    prm = _raw_params(model)
    g = W3DRawGrads()
    for n, attr in (("xyz", "xyz"), ("f_dc", "f_dc"), ("f_rest", "f_rest"), ("opacity", "opacity"),
                    ("scaling", "scaling"), ("rotation", "rotation")):
        setattr(g, n, model._p[attr].grad.data_ptr())
    st = W3DDensifyStats()
    gnorm = torch.empty(P, dtype=torch.float32, device=dev) if want_norm else None
    m2d = torch.empty(P, 3, dtype=torch.float32, device=dev) if want_means2D else None
    st.dL_dmeans2D = None if m2d is None else m2d.data_ptr()
    st.grad2d_norm = None if gnorm is None else gnorm.data_ptr()
    st.radii = handle["radii"].data_ptr()
    if update_stats:
        st.xyz_gradient_accum, st.denom = model.xyz_gradient_accum.data_ptr(), model.denom.data_ptr()
        st.max_radii2D = model.max_radii2D.data_ptr()
    with torch.cuda.device(dev):
        sb = ctypes.c_uint64()
        check(lib.w3d_backward_sizes(P, ctypes.byref(sb)))
        scratch = torch.empty(sb.value, dtype=torch.uint8, device=dev)
        check(lib.w3d_backward_raw(ctypes.byref(view.c), P, ctypes.byref(prm), ptr(handle["state"]),
                                   ptr(handle["point_list"]), ptr(dL_dimage.contiguous()), ptr(dL_ddepth), ptr(dL_dalpha),
                                   ctypes.byref(g), ctypes.byref(st), ptr(scratch), stream_ptr(dev)))
    return gnorm, m2d
