#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
DCR_LIB=$R/discrete-curvature-rewiring_amd/csrc/variants/libdcr_hip_tricheck.so timeout -k 10 300 python3 -m pytest tests/test_h2_engine_gpu.py -x -q -m gpu -k reference_fixtures 2>&1 | grep -E "Error|assert|invariant|passed|failed" | head -8
timeout -k 10 600 python3 -m pytest tests/test_h2_engine_gpu.py tests/test_checkers_gpu.py -x -q -m gpu 2>&1 | tail -3
for r in 1 2 3; do REPS=40 timeout -k 10 300 python3 tools/probe_pass.py 2>&1 | grep "pass ms"; done
bash tools/timeline_pass.sh r05_bal | tail -16
