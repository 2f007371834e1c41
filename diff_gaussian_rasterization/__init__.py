"""Drop-in for the third-party extension the reference imports at gaussian_renderer/__init__.py:15
(`from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer`;
upstream: ashawkey/diff-gaussian-rasterization, reference README.md:23).  Re-exports the MI355X-native
implementation, so gaussian_renderer.render(), scene/gaussian_model.py and train.py run unchanged."""
from scgaussian_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer,  # noqa: F401
                                       rasterize_gaussians)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians"]
