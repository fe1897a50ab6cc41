cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 | cut -c1-220
timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; python -c "
import json; d=json.load(open('gpurun_out/bench.json')); print(d['value'], d['stage_ms']); print(d['decode_render']['frames_per_s'], d['train']['ms_per_iter'], d['train_fused_decoder']['ms_per_iter'])"
