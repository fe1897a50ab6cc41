cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_decoder_gpu.py -m gpu -q -x 2>&1 | tail -25 | cut -c1-220
