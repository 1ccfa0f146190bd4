#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
DCR_LIB=$R/discrete-curvature-rewiring_amd/csrc/variants/libdcr_hip_ut.so REPS=1 timeout -k 10 300 python3 tools/probe_pass.py 2>&1 | tail -n 26 > $OUT/r05_unit_times.txt
cat $OUT/r05_unit_times.txt | cut -c1-260
timeout -k 10 900 python3 -m pytest tests/test_experiment_gpu.py -x -q -m gpu 2>&1 | tail -5
for r in 1 2; do
  EPOCHS=40 timeout -k 10 300 python3 tools/probe_gcn_epoch.py 2>&1 | grep "epoch ms"
  DCR_FUSED_HEAD=0 EPOCHS=40 timeout -k 10 300 python3 tools/probe_gcn_epoch.py 2>&1 | grep "epoch ms"
done | tee $OUT/r05_gcn_ab.txt
bash tools/trace_gcn_epoch.sh > $OUT/r05_gcn_epoch_trace.txt 2>&1
cat $OUT/r05_gcn_epoch_trace.txt | cut -c1-130
N=2120 M=2 F=3703 H=64 C=6 bash tools/trace_gcn_epoch.sh > $OUT/r05_gcn_citeseer_trace.txt 2>&1
cat $OUT/r05_gcn_citeseer_trace.txt | cut -c1-130
N=2120 M=2 F=3703 H=64 C=6 DCR_FIRST_FUSED=0 bash tools/trace_gcn_epoch.sh 2>&1 | cut -c1-130
