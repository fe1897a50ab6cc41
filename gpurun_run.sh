cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q > gpurun_out/t.log 2>&1; tail -15 gpurun_out/t.log | cut -c1-200
