"""Build the gfx950 rasterizer library (libggd_raster.so) in-tree with hipcc.

    python -m gaussian_gan_decoder_amd.build [--force] [--save-temps]

hipcc cross-compiles for gfx950 without a GPU.  Flags that are part of the numerical contract:
  -ffp-contract=off                              no implicit FMA contraction (integer anchors bit-exact vs oracle)
  -fhip-fp32-correctly-rounded-divide-sqrt       IEEE division / sqrt
  -munsafe-fp-atomics                            float atomicAdd -> global_atomic_add_f32 (no CAS loop)
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
CSRC = os.path.join(_PKG, "csrc")
LIB_PATH = os.path.join(_PKG, "libggd_raster.so")
SOURCES = ["ggd_capi.hip", "ggd_preprocess.hip", "ggd_binning.hip", "ggd_rowbin.hip", "ggd_blend.hip", "ggd_triplane.hip", "ggd_mlp.hip", "ggd_imgloss.hip",
           "ggd_preprocess_bwd.hip", "ggd_surface.hip"]
HEADERS = ["ggd_common.h", "ggd_math.h", "ggd_mlp_bwd.inc", "ggd_mlp_wgrad.inc", "ggd_mlp_pack.inc", "ggd_mlp_hl.inc", "ggd_mlp_gelu.inc", "ggd_scan.inc", "ggd_rowbin_wide.inc", "ggd_msd_finish.inc"]
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 to build the gfx950 rasterizer)")


def flags() -> list[str]:
    return ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
            "-fhip-fp32-correctly-rounded-divide-sqrt", "-munsafe-fp-atomics", "-fno-gpu-rdc",
            "-Wall", "-Wno-unused-function", "-I" + os.path.join(_ROOT, "include"), "-I" + CSRC] + \
        os.environ.get("GGD_EXTRA_HIPCC_FLAGS", "").split()


OBJ_DIR = os.path.join(_PKG, "build")


def _deps() -> list[str]:
    return [os.path.join(CSRC, f) for f in HEADERS] + [os.path.join(_ROOT, "include", "ggd_raster.h"),
                                                       os.path.abspath(__file__)]


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(d) > t for d in [os.path.join(CSRC, f) for f in SOURCES] + _deps())


def _compile_one(src: str, force: bool, save_temps: bool, verbose: bool) -> tuple[str, bool]:
    """One translation unit -> build/<name>.o (skipped when newer than the source and every header)."""
    obj = os.path.join(OBJ_DIR, os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    if not force and os.path.exists(obj):
        t = os.path.getmtime(obj)
        if all(os.path.getmtime(d) <= t for d in [path] + _deps()):
            return obj, False
    cmd = [_hipcc()] + flags() + ["-c", path, "-o", obj]
    cwd = CSRC
    if save_temps:
        cwd = os.path.join(_ROOT, "gpurun_out", "temps")
        os.makedirs(cwd, exist_ok=True)
        cmd.insert(1, "-save-temps")
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, cwd=cwd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n" + res.stdout + res.stderr)
    if verbose and res.stderr.strip():
        print(res.stderr)
    return obj, True


def build(force: bool = False, save_temps: bool = False, verbose: bool = False) -> str:
    """Per-file objects compiled in parallel (only the stale ones), then one link step."""
    if not force and not is_stale():
        return LIB_PATH
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJ_DIR, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(lambda f: _compile_one(f, force, save_temps, verbose), SOURCES))
    cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-fno-gpu-rdc"] + [o for o, _ in objs] + ["-o", LIB_PATH]
    res = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + res.stdout + res.stderr)
    return LIB_PATH


if __name__ == "__main__":
    p = build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv, verbose=True)
    print("built", p)
