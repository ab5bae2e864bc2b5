"""Drop-in for the `diff_gauss` package SplatLoc imports
(gaussian_splatting/gaussian_renderer/__init__.py:4-7) — MI355X-native implementation."""
from splatloc_amd.rasterizer import (  # noqa: F401
    GaussianRasterizationSettings,
    GaussianRasterizer,
    rasterize_gaussians,
    rasterize_window,
    _RasterizeGaussians,
)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "rasterize_window"]
