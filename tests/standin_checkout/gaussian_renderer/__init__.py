"""Stand-in for reference gaussian_renderer/__init__.py: render() marshals ACTIVATED tensors into the rasterizer package it
imports by name (:14, :22-106) — with `wheat-3dgs_amd/` on the path that package is this repo's."""
import math

import torch
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
from scene.gaussian_model import GaussianModel  # noqa: F401  (:15; render.py:22 imports the class from here)


def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None):
    assert override_color is None and not pipe.compute_cov3D_python and not pipe.convert_SHs_python
    means2D = torch.zeros_like(pc.get_xyz, requires_grad=True) + 0
    means2D.retain_grad()
    cam = viewpoint_camera
    rasterizer = GaussianRasterizer(raster_settings=GaussianRasterizationSettings(
        image_height=int(cam.image_height), image_width=int(cam.image_width), tanfovx=math.tan(cam.FoVx * 0.5),
        tanfovy=math.tan(cam.FoVy * 0.5), bg=bg_color, scale_modifier=scaling_modifier, viewmatrix=cam.world_view_transform,
        projmatrix=cam.full_proj_transform, sh_degree=pc.active_sh_degree, campos=cam.camera_center, prefiltered=False,
        debug=False))
    image, radii, depth, alpha = rasterizer(means3D=pc.get_xyz, means2D=means2D, shs=pc.get_features, colors_precomp=None,
                                            opacities=pc.get_opacity, scales=pc.get_scaling, rotations=pc.get_rotation,
                                            cov3D_precomp=None)
    return {"render": image, "viewspace_points": means2D, "visibility_filter": radii > 0, "radii": radii, "depth": depth,
            "alpha": alpha}


def flashsplat_render(*a, **k):
    raise NotImplementedError("stand-in checkout: only the training loop's render() is restated")
