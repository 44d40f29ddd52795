"""`from simple_knn._C import distCUDA2` (reference scene/gaussian_model.py:20,148)."""
from w3d_amd.rasterizer import dist2_knn3 as distCUDA2  # noqa: F401

__all__ = ["distCUDA2"]
