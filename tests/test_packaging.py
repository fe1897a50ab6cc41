"""The reference installs its rasterizer with `pip install submodules/diff-gaussian-rasterization`
(/root/reference/environment.yml:35).  Pointing that line at this repository must install both packages -- the library and
the top-level shim -- with the compiled gfx950 library inside, and make the reference's import line
(gaussian_splatting/gaussian_renderer/__init__.py:14) resolve from any working directory.  Installed into a throw-away
--target directory (this image has no venv module); no index access is needed (--no-deps --no-build-isolation)."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pip_install_resolves_the_reference_import_line(tmp_path):
    target = tmp_path / "site"
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    litter = [p for p in (os.path.join(ROOT, "build"), os.path.join(ROOT, "ggd_mi355x.egg-info")) if not os.path.exists(p)]
    try:
        _install_and_import(tmp_path, target, env)
    finally:
        for p in litter:            # what pip's in-tree build leaves behind in the source directory
            shutil.rmtree(p, ignore_errors=True)


def _install_and_import(tmp_path, target, env):
    r = subprocess.run([sys.executable, "-m", "pip", "install", "--no-deps", "--no-build-isolation", "--no-index", "-q",
                        "--target", str(target), ROOT], capture_output=True, text=True, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert (target / "gaussian_gan_decoder_amd" / "libggd_raster.so").exists(), "the compiled library was not packaged"
    assert (target / "diff_gaussian_rasterization" / "__init__.py").exists()
    code = ("import os, gaussian_gan_decoder_amd as g;"
            "from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer;"   # the reference's line
            "from gaussian_gan_decoder_amd import _capi;"
            "lib = _capi.load();"
            "assert all(hasattr(lib, s) for s in _capi.EXPORTS);"
            "print(os.path.dirname(g.__file__)); print(GaussianRasterizationSettings._fields[:3])")
    env["PYTHONPATH"] = str(target)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert os.path.samefile(r.stdout.splitlines()[0], str(target / "gaussian_gan_decoder_amd"))
    assert "image_height" in r.stdout
