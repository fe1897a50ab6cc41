cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -3
for w in 1M_1024_cube 1M_1024_shell 100k_512_cube; do for b in 0 1; do echo "$w BINNING=$b"; GGD_BINNING=$b timeout 200 python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline --no-train | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'fps', d['ms_per_step'], d['stage_ms'])"; done; done
