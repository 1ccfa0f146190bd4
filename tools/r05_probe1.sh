#!/bin/bash
# round 5, first GPU call: (1) does the number of hardware queues (GPU_MAX_HW_QUEUES, default 4) hold back the fifth class
# kernel?  interleaved rounds of tools/probe_pass.py; (2) section stamps of the class kernels (H2_PROF variant);
# (3) pass timeline with 8 queues.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
for r in 1 2 3; do
  for q in 4 8; do
    ms=$(GPU_MAX_HW_QUEUES=$q REPS=40 timeout -k 10 300 python3 tools/probe_pass.py 2>&1 | grep "pass ms" | awk '{print $3}')
    echo "S100k hwq=$q $ms"
  done
done | tee $OUT/r05_hwq.txt
for q in 4 8; do
  ms=$(N=1000000 GPU_MAX_HW_QUEUES=$q REPS=10 timeout -k 10 300 python3 tools/probe_pass.py 2>&1 | grep "pass ms" | awk '{print $3}')
  echo "S1M hwq=$q $ms"
done | tee -a $OUT/r05_hwq.txt
DCR_LIB=$R/discrete-curvature-rewiring_amd/csrc/variants/libdcr_hip_prof.so REPS=5 timeout -k 10 300 python3 tools/probe_pass.py > $OUT/r05_h2_prof.txt 2>&1
tail -20 $OUT/r05_h2_prof.txt
GPU_MAX_HW_QUEUES=8 bash tools/timeline_pass.sh r05_hwq8
