#!/bin/bash
# hl_variants.sh <suffix>...: kernel times of the reference-precision decoder kernels for the library and for the builds
# gaussian_gan_decoder_amd/libggd_raster<suffix>.so (scripts/build_variant.sh) -- timing experiments
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "" "$@"; do
  if [ -n "$v" ]; then export GGD_LIB_PATH=$R/gaussian_gan_decoder_amd/libggd_raster$v.so; else unset GGD_LIB_PATH; fi
  rm -rf /tmp/hv; rocprofv3 --kernel-trace --stats -d /tmp/hv -o p --output-format csv -- python $R/scripts/hl_only.py 3 > /tmp/hv.log 2>&1
  echo "== variant '$v'"
  python - <<'PY'
import csv
for r in csv.DictReader(open("/tmp/hv/p_kernel_stats.csv")):
    n = r["Name"]
    if "hl_kernel" in n:
        tag = "backward" if "backward" in n else ("wgrad" if "wgrad" in n else ("pack" if "pack" in n else ("forward+z" if "ILb1E" in n or "<true>" in n else "forward")))
        print("   %-12s avg_us=%9.1f" % (tag, float(r["AverageNs"]) / 1e3))
PY
done
