"""The C-ABI without PyTorch: tests/c_client/w3d_c_client.c (plain C11, gcc, buffers from hipMalloc, a stream from
hipStreamCreate) runs forward + deterministic backward of one view; the same inputs through the Python binding must give the
same bits.  Shows that include/w3d.h is a C header and that libw3d_hip.so needs nothing from torch."""
import os
import struct
import subprocess

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLIENT = os.path.join(ROOT, "tests", "c_client", "_bin", "w3d_c_client")


def test_plain_c_host_gets_the_same_bits_as_the_python_binding(tmp_path):
    src = os.path.join(ROOT, "tests", "c_client", "w3d_c_client.c")
    hdr = os.path.join(ROOT, "include", "w3d.h")
    # (normally built by __graft_entry__.build(); gcc is part of the image.  A binary older than the header it was compiled
    #  against has the wrong struct layout: rebuild)
    if not os.path.exists(CLIENT) or os.path.getmtime(CLIENT) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        import sys
        sys.path.insert(0, ROOT)
        import __graft_entry__
        __graft_entry__.build_c_client()
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from w3d_amd import rasterizer
    from w3d_amd.synth import small_test_scene
    from util import view_inputs
    P, W, H = 3000, 208, 160
    sc, cams = small_test_scene(P=P, W=W, H=H, seed=4, scale=0.03)
    import math
    cam = cams[1]
    tfx, tfy = float(np.float32(math.tan(cam.FoVx * 0.5))), float(np.float32(math.tan(cam.FoVy * 0.5)))
    d = view_inputs(sc, cam)
    bg = np.array([0.1, 0.2, 0.3], np.float32)
    dL = np.random.RandomState(2).randn(3, H, W).astype(np.float32)
    f32 = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float32))  # noqa: E731
    M = d["shs"].shape[1]
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        f.write(struct.pack("<5i3f", P, H, W, 3, M, tfx, tfy, 1.0))
        for a in (bg, cam.world_view_transform, cam.full_proj_transform, cam.camera_center, d["means3D"], d["shs"],
                  d["opacities"], d["scales"], d["rotations"], dL):
            a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a
            f.write(f32(a).tobytes())
    r = subprocess.run([CLIENT, str(fin), str(fout)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = np.fromfile(fout, dtype=np.uint8)
    off = 0

    def take(n, dt):
        nonlocal off
        a = raw[off:off + n * 4].view(dt)
        off += n * 4
        return a
    counts = take(2, np.uint32)
    c = {"radii": take(P, np.int32), "color": take(3 * H * W, np.float32), "depth": take(H * W, np.float32),
         "alpha": take(H * W, np.float32), "means3D": take(P * 3, np.float32), "means2D": take(P * 3, np.float32),
         "shs": take(P * M * 3, np.float32), "opacities": take(P, np.float32), "scales": take(P * 3, np.float32),
         "rotations": take(P * 4, np.float32)}
    assert off == raw.size

    dev = torch.device("cuda:0")
    t = {k: torch.as_tensor(f32(d[k]), device=dev).requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    means2D = torch.zeros(P, 3, device=dev, requires_grad=True)
    tod = lambda a: torch.as_tensor(f32(a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a), device=dev)  # noqa: E731
    s = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=tfx, tanfovy=tfy,
                                      bg=tod(bg), scale_modifier=1.0, viewmatrix=tod(cam.world_view_transform),
                                      projmatrix=tod(cam.full_proj_transform), sh_degree=3, campos=tod(cam.camera_center),
                                      prefiltered=False, debug=False, deterministic=True)
    color, radii, depth, alpha = GaussianRasterizer(raster_settings=s)(
        means3D=t["means3D"], means2D=means2D, shs=t["shs"], colors_precomp=None, opacities=t["opacities"],
        scales=t["scales"], rotations=t["rotations"], cov3D_precomp=None)
    (color * torch.as_tensor(dL, device=dev)).sum().backward()
    assert int(counts[0]) == int((radii > 0).sum()) and int(counts[1]) > 0
    eq = lambda a, b: np.array_equal(a.reshape(-1), b.detach().cpu().numpy().reshape(-1))  # noqa: E731
    assert eq(c["radii"], radii)
    assert eq(c["color"], color) and eq(c["depth"], depth) and eq(c["alpha"], alpha)
    assert eq(c["means2D"], means2D.grad)
    for k in ("means3D", "shs", "opacities", "scales", "rotations"):
        assert eq(c[k], t[k].grad), k
    assert float(np.abs(c["shs"]).max()) > 0
