"""Drop-in shim: makes the reference's import line
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
(gaussian_splatting/gaussian_renderer/__init__.py:14) resolve to the gfx950 rasterizer when this repository root is
on sys.path.  See INTEGRATION.md."""
from gaussian_gan_decoder_amd.rasterizer import (  # noqa: F401
    GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians, _RasterizeGaussians, mark_visible,
    rasterize_gaussians_native, rasterize_gaussians_backward_native)
