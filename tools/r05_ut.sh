#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
DCR_LIB=$R/discrete-curvature-rewiring_amd/csrc/variants/libdcr_hip_ut.so REPS=1 timeout -k 10 300 python3 tools/probe_pass.py 2>&1 | tail -n 26 > $OUT/r05_unit_times.txt
cat $OUT/r05_unit_times.txt | cut -c1-300
