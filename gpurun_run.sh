cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --no-cpu-baseline --no-decode --steps 20 --warmup 5 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['train'])"
timeout 600 python bench.py --no-cpu-baseline --no-decode --steps 20 --warmup 5 --train-amp | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['train'])"
