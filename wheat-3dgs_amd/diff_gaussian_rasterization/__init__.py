"""Drop-in for the `diff_gaussian_rasterization` package imported at reference
gaussian_renderer/__init__.py:14 and gaussian_renderer/render_helper.py:3 (the depth/alpha
fork pinned in README.md:21).  Backed by hand-written gfx950 kernels (wheat-3dgs_amd/csrc)."""
from w3d_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians  # noqa: F401

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians"]
