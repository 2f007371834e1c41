"""scgaussian_amd — MI355X-native differentiable Gaussian rasterizer (hand-written HIP for gfx950)
behind the reference's GaussianRasterizer / GaussianRasterizationSettings API
(gaussian_renderer/__init__.py:15,38-53,100-108)."""
__version__ = "0.1.0"
