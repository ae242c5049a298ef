"""Drop-in `diff_gaussian_rasterization` package (what renderer.py:13-16 of D3GA imports), served by d3ga_amd.

Put /root/repo/compat and /root/repo on PYTHONPATH ahead of any CUDA build of the original package."""
from d3ga_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer,  # noqa: F401
                                 rasterize_gaussians)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians"]
