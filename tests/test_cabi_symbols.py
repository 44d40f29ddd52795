"""CPU: libw3d_hip.so loads (no GPU needed) and exports every function include/w3d.h declares;
argument validation paths that never touch the device behave as documented."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "w3d.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(w3d_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_boundary():
    names = declared_functions()
    for must in ("w3d_forward_stage1", "w3d_forward_stage2", "w3d_backward", "w3d_knn_dist2", "w3d_l1_ssim_fwd_bwd",
                 "w3d_adam_step", "w3d_forward_sizes", "w3d_backward_sizes", "w3d_last_error", "w3d_version"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from w3d_amd import _lib
    for name in declared_functions():
        assert hasattr(_lib.lib, name), f"{name} declared in include/w3d.h but not exported"
    assert _lib.lib.w3d_version() // 100 == _lib.ABI_MAJOR
    assert len(_lib.loaded_hip_runtimes()) == 1          # shares torch's HIP runtime


def test_host_side_validation_without_a_device():
    from w3d_amd import _lib
    lib = _lib.lib
    st, sc = ctypes.c_uint64(), ctypes.c_uint64()
    assert lib.w3d_forward_sizes(2_000_000, 1200, 1600, ctypes.byref(st), ctypes.byref(sc)) == 0
    assert st.value > 2_000_000 * 48 and sc.value > 2_000_000 * 16
    assert lib.w3d_forward_sizes(-1, 10, 10, ctypes.byref(st), ctypes.byref(sc)) != 0
    assert b"bad sizes" in lib.w3d_last_error()
    assert lib.w3d_forward_sizes(0, 16, 16, ctypes.byref(st), ctypes.byref(sc)) == 0
    # NULL view is rejected before anything is launched
    assert lib.w3d_forward_stage1(None, 0, None, None, None, None, None, None, None, None, None, None, None, None) == 1
    assert b"view is NULL" in lib.w3d_last_error()
    bs = ctypes.c_uint64()
    assert lib.w3d_backward_sizes(1000, ctypes.byref(bs)) == 0 and bs.value == 1000 * 64
    # a view built against another header (shorter struct: the two fields round 3 appended are missing) is refused before
    # any of its fields is trusted
    v = _lib.W3DView()
    assert v.struct_size == ctypes.sizeof(_lib.W3DView)
    v.struct_size -= 16
    v.image_height = v.image_width = 16
    assert lib.w3d_forward_stage1(ctypes.byref(v), 0, None, None, None, None, None, None, None, None, None, None, None, None) == 1
    assert b"w3d_view size" in lib.w3d_last_error()


def test_product_path_has_no_cpu_fallback():
    import torch
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    s = GaussianRasterizationSettings(16, 16, 0.5, 0.5, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 0,
                                      torch.zeros(3), False, False)
    with pytest.raises(RuntimeError, match="no CPU path"):
        GaussianRasterizer(s)(means3D=torch.zeros(2, 3), means2D=torch.zeros(2, 3), shs=torch.zeros(2, 16, 3),
                              colors_precomp=None, opacities=torch.zeros(2, 1), scales=torch.ones(2, 3),
                              rotations=torch.zeros(2, 4), cov3D_precomp=None)
    # nothing in the product package imports the oracle
    pkg = os.path.join(ROOT, "wheat-3dgs_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".sh")):
                assert "oracle" not in open(os.path.join(dp, f)).read().replace("oracle's", "").replace("CPU oracle", ""), f


def test_host_side_validation_of_the_training_step_entry_points():
    """Argument errors of the newer entry points are reported before anything touches the device."""
    from w3d_amd import _lib
    lib = _lib.lib
    i32, u64, vp, f = ctypes.c_int32, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_float
    lib.w3d_knn_dist2_grid.argtypes = [i32, vp, vp, vp, vp]
    sb = u64()
    assert lib.w3d_knn_sizes(2_000_000, ctypes.byref(sb)) == 0 and sb.value > 2_000_000 * 20
    assert lib.w3d_knn_sizes(-1, ctypes.byref(sb)) != 0
    assert lib.w3d_knn_dist2_grid(0, None, None, None, None) == 0            # nothing to do
    assert lib.w3d_knn_dist2_grid(10, None, None, None, None) != 0 and b"knn" in lib.w3d_last_error()
    lib.w3d_densify_compact.argtypes = [i32, ctypes.POINTER(i32), i32, i32, u64, u64, u64, u64] + [vp] * 10
    dims = (i32 * 6)(3, 1, 3, 4, 3, 45)
    assert lib.w3d_densify_compact(6, dims, 0, 2, 10, 0, 0, 0, *([None] * 10)) == 0          # P_new = 0: nothing to do
    assert lib.w3d_densify_compact(9, dims, 0, 2, 10, 5, 5, 5, *([None] * 10)) != 0          # > 8 blocks
    assert b"blocks" in lib.w3d_last_error()
    assert lib.w3d_densify_compact(6, dims, 0, 2, 10, 5, 6, 5, *([None] * 10)) != 0          # NULL buffers / bad counts
    lib.w3d_sh_adam_lowrank.argtypes = [i32, i32, i32] + [vp] * 9 + [f, f, i32, i32] + [f] * 5 + [vp]
    args = [None] * 9 + [0.1, 0.1, 0, 0, 0.9, 0.999, 1e-15, 0.1, 0.001, None]
    assert lib.w3d_sh_adam_lowrank(0, 2, 3, *args) == 0                                      # P = 0: nothing to do
    assert lib.w3d_sh_adam_lowrank(10, 2, 4, *args) != 0 and b"sizes" in lib.w3d_last_error()  # degree 4 unsupported
    assert lib.w3d_sh_adam_lowrank(10, 2, 3, *args) != 0 and b"NULL" in lib.w3d_last_error()
    lib.w3d_adam_step.argtypes = [u64, vp, vp, vp, vp, f, f, f, f, f, f, i32, vp]
    assert lib.w3d_adam_step(0, None, None, None, None, 0.1, 0.9, 0.999, 1e-15, 0.1, 0.001, 0, None) == 0
    assert lib.w3d_adam_step(8, None, None, None, None, 0.1, 0.9, 0.999, 1e-15, 0.1, 0.001, 0, None) != 0


def test_ctypes_structs_match_the_c_header(tmp_path):
    """The Python mirrors of the header's structs (W3DView, the raw-parameter blocks, the fused-Adam block) have the size and
    field offsets a C compiler gives include/w3d.h — a field added on one side only shifts every pointer behind it."""
    import subprocess
    from w3d_amd import _lib, fused_step
    src = tmp_path / "layout.c"
    src.write_text('''#include <stdio.h>
#include <stddef.h>
#include "w3d.h"
int main(void) {
    printf("view %zu %zu %zu %zu %zu %zu %zu %zu %zu %d\\n", sizeof(w3d_view), offsetof(w3d_view, bg), offsetof(w3d_view, tile_cull),
           offsetof(w3d_view, deterministic), offsetof(w3d_view, det_list_capacity), offsetof(w3d_view, tile_walk_hint),
           offsetof(w3d_view, records_kept_clean), offsetof(w3d_view, list_share), offsetof(w3d_view, struct_size), W3D_ABI_VERSION);
    printf("raw %zu %zu\\n", sizeof(w3d_raw_params), sizeof(w3d_raw_grads));
    printf("stats %zu\\n", sizeof(w3d_densify_stats));
    printf("adam %zu %zu %zu %zu\\n", sizeof(w3d_adam_fused), offsetof(w3d_adam_fused, lr), offsetof(w3d_adam_fused, beta1),
           offsetof(w3d_adam_fused, bias_correction2));
    return 0;
}
''')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = dict((ln.split()[0], [int(x) for x in ln.split()[1:]]) for ln in subprocess.check_output([str(exe)], text=True).splitlines())
    V = _lib.W3DView
    assert out["view"] == [ctypes.sizeof(V), V.bg.offset, V.tile_cull.offset, V.deterministic.offset, V.det_list_capacity.offset,
                           V.tile_walk_hint.offset, V.records_kept_clean.offset, V.list_share.offset, V.struct_size.offset,
                           _lib.lib.w3d_version()]
    assert V().struct_size == ctypes.sizeof(V) and out["view"][-1] // 100 == _lib.ABI_MAJOR
    assert out["raw"] == [ctypes.sizeof(fused_step.W3DRawParams), ctypes.sizeof(fused_step.W3DRawGrads)]
    assert out["stats"] == [ctypes.sizeof(fused_step.W3DDensifyStats)]
    A = fused_step.W3DAdamFused
    assert out["adam"] == [ctypes.sizeof(A), A.lr.offset, A.beta1.offset, A.bias_correction2.offset]
