#!/bin/bash
# On the GPU box: pass ms (tools/probe_pass.py) with an environment switch off / on, interleaved rounds
#   bash tools/ab_env.sh <VAR> <rounds> [N]      -> prints one line per run and the medians
var=$1; rounds=$2; export N=${3:-100000}
R=$GRAFT_REPO_ROOT
for r in $(seq 1 $rounds); do
  for v in 0 1; do
    ms=$(cd $R && env $var=$v REPS=${REPS:-40} timeout -k 10 300 python3 tools/probe_pass.py 2>&1 | grep "pass ms" | awk '{print $3}')
    echo "$var=$v $ms"
  done
done | tee /tmp/ab_env.txt
python3 - <<'PY'
import collections, statistics
d = collections.defaultdict(list)
for l in open('/tmp/ab_env.txt'):
    p = l.split()
    if len(p) == 2:
        try: d[p[0]].append(float(p[1]))
        except ValueError: pass
for k, v in sorted(d.items()):
    print(f'median {k:24s} {statistics.median(v):8.4f} ms   min {min(v):8.4f}   runs {len(v)}')
PY
