#!/bin/bash
# On the GPU box: per-kernel times (rocprofv3 --kernel-trace --stats) of the curvature pass on the bench graph.
# usage: bash tools/prof_pass.sh [tag]   (env DCR_PASS, N, M, REPS pass through); summary also written to gpurun_out/prof_<tag>.txt
TAG=${1:-pass}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_pass
REPS=${REPS:-10} timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pass -o p -- python3 $GRAFT_REPO_ROOT/tools/probe_pass.py 2>/dev/null | grep "pass ms" | tee $OUT/prof_$TAG.txt || exit 1
python3 - >> $OUT/prof_$TAG.txt <<'PY'
import csv
for r in list(csv.DictReader(open('/tmp/prof_pass/p_kernel_stats.csv')))[:14]:
    print(f"   {float(r['AverageNs'])/1e3:10.1f} us x{r['Calls']:>4}  {r['Name'][:110]}")
PY
cat $OUT/prof_$TAG.txt
