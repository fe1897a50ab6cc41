cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -5
python __graft_entry__.py --smoke 2>&1 | tail -2
python bench.py --backward > gpurun_out/bench1.json 2> gpurun_out/bench1.err; tail -c 3000 gpurun_out/bench1.json; tail -5 gpurun_out/bench1.err
