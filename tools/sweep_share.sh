#!/bin/bash
# On the GPU box: pass ms for a list of DCR_H2_SHARE settings ("l,m,s2,s1,s0" percentages), interleaved over <rounds> rounds.
# usage: bash tools/sweep_share.sh <tag> <rounds> <share...>     (env N, REPS)
tag=$1; rounds=$2; shift 2
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/share_$tag.txt
: > $OUT
for r in $(seq 1 $rounds); do
  for sh in "$@"; do
    ms=$(cd $R && DCR_H2_SHARE=$sh REPS=${REPS:-40} timeout -k 10 300 python3 tools/probe_pass.py 2>&1 | grep "pass ms" | awk '{print $3}')
    echo "$sh $ms" | tee -a $OUT
  done
done
python3 - $OUT <<'PY' | tee -a $OUT
import sys, collections, statistics
d = collections.defaultdict(list)
for l in open(sys.argv[1]):
    p = l.split()
    if len(p) == 2:
        try: d[p[0]].append(float(p[1]))
        except ValueError: pass
for k, v in sorted(d.items(), key=lambda kv: statistics.median(kv[1])):
    print(f'median {k:22s} {statistics.median(v):8.4f} ms   min {min(v):8.4f}   runs {len(v)}')
PY
