"""Drop-in for the `flashsplat_rasterization` package imported at reference
gaussian_renderer/__init__.py:18-19 (14-field settings, 8 outputs, forward-only)."""
from w3d_amd.rasterizer import FlashSplatRasterizationSettings as GaussianRasterizationSettings  # noqa: F401
from w3d_amd.rasterizer import FlashSplatRasterizer as GaussianRasterizer  # noqa: F401

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer"]
