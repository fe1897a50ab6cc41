#!/bin/bash
# variant_kernel.sh <kernel-name regex> <lib.so> ...: per library build, the named kernel's duration in one traced cube frame
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; export TMPDIR=/tmp; cd "$ROOT"
PAT=$1; shift
for lib in "$@"; do
  for w in ${WORKLOADS:-1M_1024_cube}; do
    rm -rf gpurun_out/trv; GGD_LIB_PATH=$lib rocprofv3 --kernel-trace -d gpurun_out/trv -o p --output-format csv -- python scripts/fwd_only.py $w 16 > /dev/null 2>&1
    echo "== lib=$(basename $lib) $w: $(python scripts/frame_trace.py $(find gpurun_out/trv -name 'p_kernel_trace.csv') | grep -E "$PAT|kernels:" | cut -c1-100 | tr '\n' '|')"
  done
done
