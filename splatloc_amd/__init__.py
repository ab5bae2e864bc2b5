"""splatloc_amd — MI355X-native (gfx950) differentiable Gaussian tile rasterizer for SplatLoc.

Host-side mirror of the `diff_gauss` / `simple_knn` extension API that SplatLoc's Python
calls (gaussian_splatting/gaussian_renderer/__init__.py:4-7,117-126;
gaussian_splatting/scene/gaussian_model.py:18,206).  All compute goes through the C ABI in
include/splatraster.h (hand-written HIP, loaded with ctypes); there is no CPU fallback.
"""
from .rasterizer import (  # noqa: F401
    GaussianRasterizationSettings,
    GaussianRasterizer,
    rasterize_gaussians,
    rasterize_window,
    _RasterizeGaussians,
    _RasterizeWindow,
)
from .knn import distCUDA2  # noqa: F401

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "rasterize_window", "distCUDA2"]
