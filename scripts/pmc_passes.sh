#!/bin/bash
# usage: scripts/pmc_passes.sh <outdir under gpurun_out> <script.py> <script args...>   (run on the GPU box via gpurun)
# Three SQ passes (8 counters each) + FETCH_SIZE + WRITE_SIZE + a kernel trace, each in its own rocprofv3 run.
set -u
OUT=$1; shift
SCRIPT=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$OUT
run() { local tag=$1; shift; rocprofv3 "$@" -d $R/gpurun_out/$OUT/$tag -o p --output-format csv -- python $R/$SCRIPT $ARGS > $R/gpurun_out/$OUT/$tag.log 2>&1; }
ARGS="$*"
run sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
run sq2 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY
run sq3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH
run mfma --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run trace --kernel-trace --stats
find $R/gpurun_out/$OUT -name "*.csv" | head -20
