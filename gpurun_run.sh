cd $GRAFT_REPO_ROOT
timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -3 gpurun_out/bench.err; cat gpurun_out/bench.json
