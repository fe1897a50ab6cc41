"""gaussian_gan_decoder_amd -- MI355X-native (gfx950) Gaussian-splatting decode/render hot path.

Only what the path needs: the rasterizer (C ABI in include/ggd_raster.h, HIP kernels in
csrc/), its PyTorch-facing mirror of the reference's `diff_gaussian_rasterization` API, the render()/render_simple()
callers, the camera / Gaussian-container prologue and the synthetic scenes used by tests and bench.
Importing this package does not load the native library; the first rasterizer call does (and fails loudly if
libggd_raster.so has not been built).
"""
__version__ = "0.1.0"
