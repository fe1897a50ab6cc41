cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r01b
mkdir -p $O
cd /tmp
# 1) kernel trace + stats of the default headline command (forward only section, plus backward extra)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $R/bench.py --steps 100 --warmup 10 --backward --no-train --no-decode --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/kt.err
find /tmp/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
# 2) train step kernel stats (fused decoder)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -o kt -- python $R/bench.py --steps 5 --warmup 2 --no-decode --no-cpu-baseline --train-iters 4 > $O/bench_train_under_rocprof.json 2> $O/kt2.err
find /tmp/kt2 -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_train.csv \;
# 3) PMC passes (separate runs; counters only)
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_SALU" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $pass --output-format csv -d /tmp/pmc_$tag -o pmc -- python $R/bench.py --steps 3 --warmup 1 --backward --no-train --no-decode --no-cpu-baseline > /dev/null 2> $O/pmc_$tag.err
  find /tmp/pmc_$tag -name "*counter_collection.csv" -exec cp {} /tmp/pmc_$tag.csv \;
done
python $R/profiles/summarize_pmc.py /tmp/pmc_SQ_WAVES.csv /tmp/pmc_FETCH_SIZE.csv /tmp/pmc_WRITE_SIZE.csv "rocprofv3 PMC summary, round 1 final: bench.py --steps 3 --backward, 1M Gaussians / 1024x1024 'cube' (R = 4.15M)" > $O/pmc_summary.txt 2> $O/pmc_sum.err
ls -la $O; tail -3 $O/*.err | cut -c1-200
