#!/bin/bash
# On the GPU box: kernel timeline of one SDRF iteration at the bench shape (device draw), from k_imp_insert to the next one.
# usage: bash tools/timeline_step.sh <tag> -> gpurun_out/<tag>_step_timeline.txt
TAG=${1:-step}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl_step
K=12 INC=${INC:-0} timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_step -o p -- python3 $GRAFT_REPO_ROOT/tools/probe_step.py > /dev/null 2>&1
python3 - > $OUT/${TAG}_step_timeline.txt <<'PY'
import csv
rows = list(csv.DictReader(open('/tmp/tl_step/p_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_imp_insert' in r['Kernel_Name']]
# the first measured run is device-draw: take its 10th iteration (8 warm-up + a few)
a, b = idx[14], idx[15]
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
    print(f"{s:9.1f} {e:9.1f} {e - s:8.1f}  {r['Kernel_Name'].split('(')[0][:80]}")
PY
cat $OUT/${TAG}_step_timeline.txt
