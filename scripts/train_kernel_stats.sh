#!/bin/bash
# Per-kernel GPU time of one fused-decoder train step (rocprofv3 kernel trace of scripts/profile_train.py --fused).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pt; rocprofv3 --kernel-trace --stats -d /tmp/pt -o p --output-format csv -- python $R/scripts/profile_train.py --fused "$@" > /tmp/pt.log 2>&1
python - <<'PY'
import csv
rows = list(csv.DictReader(open("/tmp/pt/p_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("GPU ms/step", round(tot / 6 / 1e6, 2), "launches/step", round(sum(int(r["Calls"]) for r in rows) / 6, 1))
for r in rows[:40]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print("%-75s calls/step=%6.1f avg_us=%8.1f ms/step=%6.2f" % (n[:75], int(r["Calls"]) / 6, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 6e6))
PY
grep "ms/iter" /tmp/pt.log
