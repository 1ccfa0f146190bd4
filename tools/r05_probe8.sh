#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
timeout -k 10 900 python3 -m pytest tests/test_h2_engine_gpu.py tests/test_checkers_gpu.py -x -q -m gpu 2>&1 | tail -5
timeout -k 10 600 python3 -m pytest tests/test_gcn.py -x -q -m gpu -k "first_layer" 2>&1 | tail -3
DCR_LIB=$R/discrete-curvature-rewiring_amd/csrc/variants/libdcr_hip_ut.so REPS=1 timeout -k 10 300 python3 tools/probe_pass.py 2>&1 | tail -n 26 > $OUT/r05_unit_times.txt
cat $OUT/r05_unit_times.txt | cut -c1-260
for r in 1 2 3; do
  REPS=40 timeout -k 10 300 python3 tools/probe_pass.py 2>&1 | grep "pass ms"
done
N=1000000 REPS=10 timeout -k 10 300 python3 tools/probe_pass.py 2>&1 | grep "pass ms"
N=2120 M=2 F=3703 H=64 C=6 bash tools/trace_gcn_epoch.sh > $OUT/r05_gcn_citeseer_trace.txt 2>&1
cat $OUT/r05_gcn_citeseer_trace.txt | cut -c1-130
N=2485 M=2 F=1433 H=128 C=7 bash tools/trace_gcn_epoch.sh 2>&1 | cut -c1-130
N=2120 M=2 F=3703 H=64 C=6 EPOCHS=200 timeout -k 10 300 python3 tools/probe_gcn_epoch.py 2>&1 | grep "epoch ms"
N=2120 M=2 F=3703 H=64 C=6 EPOCHS=200 DCR_FIRST_FUSED=0 timeout -k 10 300 python3 tools/probe_gcn_epoch.py 2>&1 | grep "epoch ms"
