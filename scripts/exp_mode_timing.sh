for m in 2 1 0; do
  echo -n "exp mode $m: "
  GGD_EXP_MODE=$m python bench.py --no-cpu-baseline --no-train --no-decode --no-sweep --steps 400 2>/dev/null | python -c 'import json,sys; b=json.loads(sys.stdin.read()); print(round(b["value"]), b["stage_ms"]["blend"])'
done
