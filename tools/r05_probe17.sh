#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout -k 10 900 python3 -m pytest tests/test_h2_engine_gpu.py tests/test_checkers_gpu.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
for r in 1 2 3; do
  for v in 0 1; do
    ms=$(DCR_H2_CLIST=$v REPS=40 timeout -k 10 300 python3 tools/probe_pass.py 2>&1 | grep "pass ms" | awk '{print $3}')
    echo "S100k clist=$v $ms"
  done
done | tee $OUT/r05_clist.txt
for v in 0 1; do
  ms=$(DCR_H2_CLIST=$v N=1000000 REPS=10 timeout -k 10 300 python3 tools/probe_pass.py 2>&1 | grep "pass ms" | awk '{print $3}')
  echo "S1M clist=$v $ms"
done | tee -a $OUT/r05_clist.txt
DCR_LIB=$R/discrete-curvature-rewiring_amd/csrc/variants/libdcr_hip_ut.so REPS=1 timeout -k 10 300 python3 tools/probe_pass.py 2>&1 | grep -A6 "class L" | tail -7 | cut -c1-230
DCR_SERIAL_BINS=1 REPS=10 bash tools/prof_pass.sh r05_serial2 2>&1 | head -8 | cut -c1-120
