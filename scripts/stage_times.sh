python bench.py --no-cpu-baseline --no-train --no-decode --no-sweep 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read()); print(round(b['value']), b['stage_ms'])"
