#!/bin/bash
# Average duration of the three decoder training kernels in one fused train step (rocprofv3 kernel trace of scripts/profile_train.py).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pt; rocprofv3 --kernel-trace --stats -d /tmp/pt -o p --output-format csv -- python $R/scripts/profile_train.py --fused > /tmp/pt.log 2>&1
python - <<'PY'
import csv
for r in csv.DictReader(open('/tmp/pt/p_kernel_stats.csv')):
    if 'decoder_' in r['Name'] or 'tpb_acc' in r['Name']: print(r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
grep "ms/iter" /tmp/pt.log
