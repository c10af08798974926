"""MI355X-native differentiable Gaussian rasterizer: drop-in for the reference's
``diff_gaussian_rasterization`` operator and ``gaussian_renderer.render`` host.

    from mvs_gaussian_splatting_amd import GaussianRasterizationSettings, GaussianRasterizer, render
"""
from .rasterizer import (GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians,  # noqa: F401
                         rasterize_gaussians_fused)
from .renderer import render  # noqa: F401
from .losses import l1_loss, l1_dssim_loss, add_densification_stats  # noqa: F401

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "render", "l1_loss", "l1_dssim_loss",
           "add_densification_stats"]
