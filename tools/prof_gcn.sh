#!/bin/bash
# On the GPU box: per-kernel table of the GCN epoch at the S1M shape (rocprofv3 --kernel-trace --stats over tools/probe_gcn.py)
# usage: bash tools/prof_gcn.sh <tag>  -> gpurun_out/<tag>_gcn_epoch.csv (per epoch: calls, microseconds)
TAG=${1:-gcn}
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_gcn
EPOCHS=${EPOCHS:-20} timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_gcn -o p -- python3 $GRAFT_REPO_ROOT/tools/probe_gcn_epoch.py 2>/dev/null | grep "epoch" | tee $OUT/${TAG}_gcn_epoch.txt
python3 - "$OUT/${TAG}_gcn_epoch.csv" <<'PY'
import csv, os, sys
ep = int(os.environ.get("EPOCHS", 20)) + 6       # timed epochs + the six warm-up ones probe_gcn_epoch.py runs
rows = list(csv.DictReader(open('/tmp/prof_gcn/p_kernel_stats.csv')))
with open(sys.argv[1], 'w') as f:
    f.write('kernel,calls_per_epoch,us_per_call,us_per_epoch\n')
    tot = 0.0
    for r in rows[:28]:
        per = float(r['TotalDurationNs']) / 1e3 / ep
        tot += per
        f.write(f"\"{r['Name'][:90]}\",{int(r['Calls']) / ep:.2f},{float(r['AverageNs']) / 1e3:.1f},{per:.1f}\n")
    f.write(f'"(sum of the 28 largest)",,,{tot:.1f}\n')
print(open(sys.argv[1]).read())
PY
