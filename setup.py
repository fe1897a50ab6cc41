"""pip entry point.  The reference installs its rasterizer with

    pip install submodules/diff-gaussian-rasterization          (/root/reference/environment.yml:35)

Pointing that line at this repository instead (`pip install --no-build-isolation /path/to/this/repo`, or `-e` for a
development install) installs `gaussian_gan_decoder_amd` AND the top-level shim package `diff_gaussian_rasterization`,
so the reference's import line (gaussian_splatting/gaussian_renderer/__init__.py:14) resolves to the gfx950 library.
The build step compiles the HIP kernels in-tree with hipcc (gaussian_gan_decoder_amd/build.py; hipcc cross-compiles
gfx950 without a GPU) before the package files are collected; GGD_SKIP_NATIVE_BUILD=1 skips it (the library is then
built on first use by `python -m gaussian_gan_decoder_amd.build`)."""
import os
import sys

from setuptools import setup
from setuptools.command.build_py import build_py
from setuptools.command.develop import develop

ROOT = os.path.dirname(os.path.abspath(__file__))


def _build_native():
    if os.environ.get("GGD_SKIP_NATIVE_BUILD") == "1":
        return
    sys.path.insert(0, ROOT)
    try:
        from gaussian_gan_decoder_amd import build as hip_build
        print("building", hip_build.LIB_PATH, "with hipcc (gfx950)")
        hip_build.build()
    finally:
        sys.path.remove(ROOT)


class BuildPyWithHip(build_py):
    def run(self):
        _build_native()
        super().run()


class DevelopWithHip(develop):
    def run(self):
        _build_native()
        super().run()


setup(
    name="ggd-mi355x",
    version="0.3.0",
    description="MI355X-native (gfx950) Gaussian-splatting decode/render path: drop-in for diff_gaussian_rasterization",
    python_requires=">=3.9",
    packages=["gaussian_gan_decoder_amd", "diff_gaussian_rasterization"],
    package_data={"gaussian_gan_decoder_amd": ["libggd_raster.so", "csrc/*"]},
    cmdclass={"build_py": BuildPyWithHip, "develop": DevelopWithHip},
)
