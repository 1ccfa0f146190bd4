#!/bin/bash
# round 5: the long versions of the GPU-box checkers (the edge-by-edge incremental pass over thousands of iterations)
cd $GRAFT_REPO_ROOT
ITERS=3000 timeout -k 10 900 python3 tests/check_soak.py 2>&1 | grep -v amdgpu | tee gpurun_out/r05_soak.txt
echo "== with removals (bound -1.19, tau 180)" | tee -a gpurun_out/r05_soak.txt
ITERS=1500 BOUND=-1.19 TAU=180 timeout -k 10 900 python3 tests/check_soak.py 2>&1 | grep -v amdgpu | tee -a gpurun_out/r05_soak.txt
timeout -k 10 600 python3 tests/check_hub_sdrf.py 2>&1 | grep -v amdgpu | tail -4 | tee -a gpurun_out/r05_soak.txt
