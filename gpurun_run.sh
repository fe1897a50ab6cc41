cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O2 scripts/probes/tr_b16_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
timeout 300 python -m pytest tests/test_decoder_gpu.py -m gpu -q -x 2>&1 | tail -25 | cut -c1-220
timeout 400 python scripts/profile_train.py --fused > gpurun_out/train_fused.txt 2>&1
grep -v "^---" gpurun_out/train_fused.txt | cut -c1-64,150-260 | tail -30
