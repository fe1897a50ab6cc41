cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
GGD_EXP_MODE=0 python -m pytest tests/test_raster_backward_gpu.py -m gpu -q -s > gpurun_out/bwd0.log 2>&1
GGD_EXP_MODE=1 python -m pytest tests/test_raster_backward_gpu.py -m gpu -q -s > gpurun_out/bwd1.log 2>&1
tail -3 gpurun_out/bwd0.log gpurun_out/bwd1.log
