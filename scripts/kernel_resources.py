"""Register / LDS / spill figures of every gfx950 kernel in the built library, read from the code objects embedded in
libggd_raster.so (no GPU needed):  python scripts/kernel_resources.py [path/to/lib.so] [substring ...]

The `.hip_fatbin` section of the shared library holds one clang offload bundle per translation unit; each is unbundled with
clang-offload-bundler and its AMDGPU metadata note read with llvm-readelf.  Used by tests/test_capi_and_host.py (the kernels
that name fixed physical registers must own them) and for the occupancy column of DESIGN.md."""
from __future__ import annotations

import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def demangle(names):
    try:
        out = subprocess.run([shutil.which("llvm-cxxfilt") or shutil.which("c++filt") or os.path.join(LLVM, "llvm-cxxfilt")], input="\n".join(names), capture_output=True, text=True, check=True).stdout
        return out.splitlines()
    except (OSError, subprocess.CalledProcessError):
        return list(names)


def kernel_resources(lib_path: str) -> dict:
    """{demangled kernel name: {vgpr, agpr, sgpr, lds, scratch, vgpr_spill, sgpr_spill, wg, kernarg}}"""
    res = {}
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib_path, fat], check=True)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        for i, s in enumerate(starts):
            part = os.path.join(td, f"b{i}.bin")
            open(part, "wb").write(blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            co = os.path.join(td, f"k{i}.co")
            r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={part}", f"--output={co}"],
                               capture_output=True, text=True)
            if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
            cur = None
            keys = {"vgpr_count": "vgpr", "sgpr_count": "sgpr", "vgpr_spill_count": "vgpr_spill", "sgpr_spill_count": "sgpr_spill",
                    "group_segment_fixed_size": "lds", "private_segment_fixed_size": "scratch", "max_flat_workgroup_size": "wg",
                    "kernarg_segment_size": "kernarg", "agpr_count": "agpr"}

            def commit(c):
                if c and "_sym" in c:
                    res[c["_sym"]] = {a: b for a, b in c.items() if not a.startswith("_")}
            for line in notes.splitlines():
                if re.match(r"\s*- \.agpr_count:", line):    # the first (alphabetical) key of a kernel record
                    commit(cur)
                    cur = {}
                m = re.match(r"\s*-?\s*\.(\w+):\s+(\S.*)$", line)
                if not m or cur is None:
                    continue
                k, v = m.group(1), m.group(2).strip()
                if k in keys:
                    cur[keys[k]] = int(v)
                elif k == "symbol":
                    cur["_sym"] = v.replace(".kd", "")
            commit(cur)
    names = list(res)
    return dict(zip(demangle(names), (res[n] for n in names)))


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void\s+", "", name)
    return re.sub(r"\(.*$", "", name)


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else os.path.join(root, "gaussian_gan_decoder_amd", "libggd_raster.so")
    pats = [a for a in sys.argv[1:] if not a.endswith(".so")]
    tab = kernel_resources(lib)
    print(f"{'kernel':70s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'lds B':>7s} {'scratch':>7s} {'v.spill':>7s} {'kernarg':>7s} {'waves/SIMD':>10s}")
    for name in sorted(tab, key=short):
        if pats and not any(p in name for p in pats):
            continue
        r = tab[name]
        regs = max(1, r.get("vgpr", 0) + r.get("agpr", 0))
        waves = min(8, 512 // ((regs + 7) // 8 * 8))
        print(f"{short(name)[:70]:70s} {r.get('vgpr', 0):5d} {r.get('agpr', 0):5d} {r.get('sgpr', 0):5d} {r.get('lds', 0):7d} "
              f"{r.get('scratch', 0):7d} {r.get('vgpr_spill', 0):7d} {r.get('kernarg', 0):7d} {waves:10d}")
