#!/bin/bash
# variant_traces.sh <lib.so|""> ...: the binning kernels of one traced frame (cube and shell) for each library build
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; export TMPDIR=/tmp; cd "$ROOT"
for lib in "$@"; do
  for w in 1M_1024_cube 1M_1024_shell; do
    rm -rf gpurun_out/trv; GGD_LIB_PATH=$lib rocprofv3 --kernel-trace -d gpurun_out/trv -o p --output-format csv -- python scripts/fwd_only.py $w 16 > /dev/null 2>&1
    echo "== lib=$lib $w"; python scripts/frame_trace.py $(find gpurun_out/trv -name 'p_kernel_trace.csv') | grep "rb_\|kernels:" | cut -c1-100
  done
done
