#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout -k 10 900 python3 -m pytest tests/test_h2_engine_gpu.py tests/test_checkers_gpu.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -4
for r in 1 2 3; do
  for v in 0 1; do
    ms=$(DCR_H2_TRI_SETS=$v REPS=40 timeout -k 10 300 python3 tools/probe_pass.py 2>&1 | grep "pass ms" | awk '{print $3}')
    echo "S100k tri_sets=$v $ms"
  done
done | tee $OUT/r05_tri_sets.txt
for v in 0 1; do
  ms=$(DCR_H2_TRI_SETS=$v N=1000000 REPS=10 timeout -k 10 300 python3 tools/probe_pass.py 2>&1 | grep "pass ms" | awk '{print $3}')
  echo "S1M tri_sets=$v $ms"
done | tee -a $OUT/r05_tri_sets.txt
bash tools/timeline_pass.sh r05_tri
