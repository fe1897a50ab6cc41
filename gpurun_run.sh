cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_raster_gpu.py -m gpu -q -x 2>&1 | tail -4 | cut -c1-220
cd /tmp
GGD_BINNING=3 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $R/bench.py --steps 50 --warmup 5 --no-train --no-decode --no-cpu-baseline > /tmp/b.json 2> /tmp/kt.err
python -c "import json; d=json.load(open('/tmp/b.json')); print(d['value'], d['stage_ms'])"
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    n=r['Name'].replace('(anonymous namespace)::','').split('(')[0][:60]
    print(f"{n:62s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.2f}")
PY
