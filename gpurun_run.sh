cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_raster_gpu.py -m gpu -q -x 2>&1 | tail -4 | cut -c1-220
timeout 300 python bench.py --no-train --no-decode --no-cpu-baseline --steps 100 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms'])"
