#!/bin/bash
# frame_traces.sh <outdir> [workloads...]: rocprofv3 kernel trace of 16 forward frames per workload -> <outdir>/frame_trace_<workload>.txt
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; export TMPDIR=/tmp
cd "$ROOT"
OUT=$1; shift
mkdir -p $OUT
for w in ${@:-1M_1024_cube 1M_1024_shell}; do
  rm -rf gpurun_out/tr_$w; rocprofv3 --kernel-trace -d gpurun_out/tr_$w -o p --output-format csv -- python scripts/fwd_only.py $w 16 > /dev/null 2>&1
  python scripts/frame_trace.py $(find gpurun_out/tr_$w -name 'p_kernel_trace.csv') > $OUT/frame_trace_$w.txt; cat $OUT/frame_trace_$w.txt
done
