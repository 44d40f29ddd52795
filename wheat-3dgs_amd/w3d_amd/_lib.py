"""ctypes binding of libw3d_hip.so (the C-ABI of include/w3d.h).

The product path has NO fallback: if the HIP library is missing or fails to load, importing
this module raises.  torch is imported first on purpose — the library's DT_NEEDED
``libamdhip64.so.7`` then resolves to the HIP runtime torch has already loaded, so streams and
device pointers are shared between torch and our kernels (one runtime per process).
"""
import ctypes
import os

import torch  # noqa: F401  (must precede the CDLL below, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("W3D_HIP_LIB", os.path.join(_HERE, "..", "lib", "libw3d_hip.so"))

c_f32p = ctypes.c_void_p
c_u32p = ctypes.c_void_p


class W3DView(ctypes.Structure):
    """Mirror of ``w3d_view`` (include/w3d.h)."""
    _fields_ = [("struct_size", ctypes.c_uint32),
                ("image_height", ctypes.c_int32), ("image_width", ctypes.c_int32),
                ("tanfovx", ctypes.c_float), ("tanfovy", ctypes.c_float),
                ("scale_modifier", ctypes.c_float),
                ("sh_degree", ctypes.c_int32), ("sh_coeffs", ctypes.c_int32),
                ("prefiltered", ctypes.c_int32), ("debug", ctypes.c_int32),
                ("bg", ctypes.c_void_p), ("viewmatrix", ctypes.c_void_p),
                ("projmatrix", ctypes.c_void_p), ("campos", ctypes.c_void_p),
                ("tile_cull", ctypes.c_int32), ("deterministic", ctypes.c_int32),
                ("det_list_capacity", ctypes.c_uint64), ("tile_walk_hint", ctypes.c_void_p),
                ("records_kept_clean", ctypes.c_int32), ("list_share", ctypes.c_int32)]

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.struct_size = ctypes.sizeof(W3DView)      # checked by every entry point (include/w3d.h)


ABI_MAJOR = 3       # W3D_ABI_VERSION // 100 of the include/w3d.h these ctypes mirrors were written against


def _load():
    path = os.path.abspath(LIB_PATH)
    if not os.path.exists(path):
        raise ImportError(
            f"libw3d_hip.so not found at {path}. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or wheat-3dgs_amd/csrc/build.sh — there is no CPU or PyTorch fallback for the rasterizer.")
    lib = ctypes.CDLL(path)
    vp, i32, u64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_uint64
    lib.w3d_version.restype = ctypes.c_int
    ver = int(lib.w3d_version())
    if ver // 100 != ABI_MAJOR:
        raise ImportError(f"{path} reports ABI version {ver}, this binding mirrors ABI {ABI_MAJOR}xx of include/w3d.h: a stale "
                          "library (or a stale binding) — rebuild with wheat-3dgs_amd/csrc/build.sh")
    lib.w3d_last_error.restype = ctypes.c_char_p
    lib.w3d_forward_sizes.argtypes = [i32, i32, i32, ctypes.POINTER(u64), ctypes.POINTER(u64)]
    lib.w3d_forward_stage1.argtypes = [ctypes.POINTER(W3DView), i32] + [vp] * 7 + [vp, vp, vp, vp, vp]
    lib.w3d_forward_stage2.argtypes = [ctypes.POINTER(W3DView), i32, vp, vp, vp, u64, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp]
    lib.w3d_backward_sizes.argtypes = [i32, ctypes.POINTER(u64)]
    lib.w3d_backward_det_sizes.argtypes = [i32, u64, ctypes.POINTER(u64)]
    lib.w3d_backward.argtypes = [ctypes.POINTER(W3DView), i32] + [vp] * 7 + [vp, vp] + [vp] * 3 + [vp] * 8 + [vp, vp]
    lib.w3d_knn_dist2.argtypes = [i32, vp, vp, vp]
    lib.w3d_knn_sizes.argtypes = [i32, ctypes.POINTER(u64)]
    lib.w3d_knn_dist2_grid.argtypes = [i32, vp, vp, vp, vp]
    lib.w3d_debug_tile_ranges.argtypes = [i32, i32, i32, vp, vp, vp]
    lib.w3d_debug_pixel_state.argtypes = [i32, i32, i32, vp, vp, vp, vp]
    for name in ("w3d_forward_sizes", "w3d_forward_stage1", "w3d_forward_stage2", "w3d_backward_sizes", "w3d_backward_det_sizes",
                 "w3d_backward", "w3d_knn_dist2", "w3d_knn_sizes", "w3d_knn_dist2_grid", "w3d_debug_tile_ranges",
                 "w3d_debug_pixel_state"):
        getattr(lib, name).restype = ctypes.c_int
    return lib


lib = _load()


class W3DError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise W3DError(f"libw3d_hip error {rc}: {lib.w3d_last_error().decode(errors='replace')}")


def ptr(t):
    """Device (or NULL) pointer of a tensor as an int for ctypes."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def loaded_hip_runtimes():
    """Paths of every libamdhip64 mapped into this process (must be exactly one)."""
    out = set()
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    out.add(line.split()[-1])
    except OSError:
        pass
    return sorted(out)
