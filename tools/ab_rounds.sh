#!/bin/bash
# On the GPU box: pass ms of several builds of the library, INTERLEAVED (a box drifts by more than the differences looked for):
#   bash tools/ab_rounds.sh <tag> <rounds> <name...>     name = default | a csrc/variants/libdcr_hip_<name>.so (tools/build_variant.sh)
# env: N (nodes, default 100000), REPS (passes per run, default 40), SERIAL=1 -> per-kernel times of each build, class kernels one
# after the other (tools/prof_pass.sh).  -> gpurun_out/ab_rounds_<tag>.txt (one line per run, then the medians)
tag=$1; rounds=$2; shift 2
R=$GRAFT_REPO_ROOT; C=$R/discrete-curvature-rewiring_amd/csrc; OUT=$R/gpurun_out/ab_rounds_$tag.txt
: > $OUT
for r in $(seq 1 $rounds); do
  for n in "$@"; do
    if [ "$n" = default ]; then unset DCR_LIB; else export DCR_LIB=$C/variants/libdcr_hip_$n.so; fi
    ms=$(cd $R && REPS=${REPS:-40} timeout -k 10 300 python3 tools/probe_pass.py 2>&1 | grep "pass ms" | awk '{print $3}')
    echo "$n $ms" | tee -a $OUT
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
for k, v in d.items():
    print(f'median {k:12s} {statistics.median(v):8.4f} ms   min {min(v):8.4f}   runs {len(v)}')
PY
if [ -n "$SERIAL" ]; then
  for n in "$@"; do
    if [ "$n" = default ]; then unset DCR_LIB; else export DCR_LIB=$C/variants/libdcr_hip_$n.so; fi
    echo "=== serial $n" | tee -a $OUT
    DCR_SERIAL_BINS=1 REPS=10 bash $R/tools/prof_pass.sh abr_${tag}_$n | tee -a $OUT
  done
fi
