cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -3
timeout 300 python scripts/quick_timing.py 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['P'], d['S'], d['kind'], 'blend', d['stages']['blend'], 'bwd', d['stages']['blend_bwd'], 'fwd_ms', round(d['fwd_ms'],3))"
timeout 200 python scripts/ab_blend.py 1 2>&1 | grep -v amdgpu | cut -c1-220
