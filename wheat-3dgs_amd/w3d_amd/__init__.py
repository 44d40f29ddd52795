"""wheat-3dgs_amd — MI355X-native Gaussian rasterizer behind Wheat-3DGS's render() boundary.

Put this directory (`wheat-3dgs_amd/`) on PYTHONPATH: it exposes the three import names the
reference binds (`diff_gaussian_rasterization`, `flashsplat_rasterization`, `simple_knn._C`)
plus this package, the host-side core.  Importing any rasterizer entry point loads
lib/libw3d_hip.so and fails loudly when it is absent — there is no CPU fallback.
"""
__version__ = "0.1.0"
