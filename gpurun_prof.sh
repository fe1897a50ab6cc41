cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ROOT=$PWD
OUT=/tmp/prof
mkdir -p $OUT gpurun_out/prof
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r01 -- python $ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --backward > $ROOT/gpurun_out/prof/trace_bench.json 2> $ROOT/gpurun_out/prof/trace.err
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY -d $OUT/pmc_sq -o r01 -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --backward > /dev/null 2> $ROOT/gpurun_out/prof/pmc_sq.err
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o r01 -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --backward > /dev/null 2> $ROOT/gpurun_out/prof/pmc_fetch.err
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o r01 -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --backward > /dev/null 2> $ROOT/gpurun_out/prof/pmc_write.err
cd $ROOT
find $OUT -type f | xargs ls -la | head -40
cp $OUT/trace/*kernel_stats.csv gpurun_out/prof/ 2>/dev/null
for d in pmc_sq pmc_fetch pmc_write; do f=$(ls $OUT/$d/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f gpurun_out/prof/${d}_counter_collection.csv; done
ls -la gpurun_out/prof
