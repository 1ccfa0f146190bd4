#!/bin/bash
# round 5, second GPU call: stream layouts of the two-hop pass (DCR_H2_LAYOUT) x hardware queues, interleaved rounds; timelines;
# the tightened GCN gradient tests.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
run() {  # name, N, reps, env...
  local name=$1 n=$2 reps=$3; shift 3
  ms=$(env "$@" N=$n REPS=$reps timeout -k 10 300 python3 tools/probe_pass.py 2>&1 | grep "pass ms" | awk '{print $3}')
  echo "$n $name $ms"
}
for r in 1 2 3; do
  run base-q4 100000 40 GPU_MAX_HW_QUEUES=4
  run base-q8 100000 40 GPU_MAX_HW_QUEUES=8
  run l01111-q4 100000 40 GPU_MAX_HW_QUEUES=4 DCR_H2_LAYOUT=0,1,1,1,1
  run l01221-q4 100000 40 GPU_MAX_HW_QUEUES=4 DCR_H2_LAYOUT=0,1,2,2,1
  run l01222-q4 100000 40 GPU_MAX_HW_QUEUES=4 DCR_H2_LAYOUT=0,1,2,2,2
  run l01122-q4 100000 40 GPU_MAX_HW_QUEUES=4 DCR_H2_LAYOUT=0,1,1,2,2
  run l01233-q4 100000 40 GPU_MAX_HW_QUEUES=4 DCR_H2_LAYOUT=0,1,2,3,3
  run l00111-q4 100000 40 GPU_MAX_HW_QUEUES=4 DCR_H2_LAYOUT=0,0,1,1,1
done | tee $OUT/r05_layouts.txt
for l in base 0,1,1,1,1 0,1,2,2,1 0,1,2,2,2; do
  if [ $l = base ]; then run base 1000000 10 GPU_MAX_HW_QUEUES=4; else run l$l 1000000 10 GPU_MAX_HW_QUEUES=4 DCR_H2_LAYOUT=$l; fi
done | tee -a $OUT/r05_layouts.txt
python3 - <<'PY' | tee -a $GRAFT_REPO_ROOT/gpurun_out/r05_layouts.txt
import collections, statistics, os
d = collections.defaultdict(list)
for l in open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r05_layouts.txt'):
    p = l.split()
    if len(p) == 3:
        try: d[(p[0], p[1])].append(float(p[2]))
        except ValueError: pass
for k, v in sorted(d.items()):
    print(f'median {k[0]:>8s} {k[1]:14s} {statistics.median(v):8.4f} ms   min {min(v):8.4f}   runs {len(v)}')
PY
DCR_H2_LAYOUT=0,1,1,1,1 bash tools/timeline_pass.sh r05_l01111
DCR_H2_LAYOUT=0,1,2,2,1 bash tools/timeline_pass.sh r05_l01221
timeout -k 10 900 python3 -m pytest tests/test_gcn.py -x -q -m gpu -k "first_layer or activation_fused" 2>&1 | tail -15 | tee $OUT/r05_gcn_tests.txt
timeout -k 10 600 python3 -m pytest tests/test_gcn_configs_gpu.py -x -q -m gpu -k "s1m_training_step" 2>&1 | tail -15 | tee -a $OUT/r05_gcn_tests.txt
