cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --no-train --no-decode --no-cpu-baseline --steps 100 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms'])"
timeout 300 python bench.py --workload 100k_512_cube --no-train --no-decode --no-cpu-baseline --steps 100 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms'])"
