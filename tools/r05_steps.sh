#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_scale_golden_gpu.py tests/test_checkers_gpu.py -x -q -m gpu > gpurun_out/r05_inc_tests.txt 2>&1
echo "rc=$?" >> gpurun_out/r05_inc_tests.txt; tail -3 gpurun_out/r05_inc_tests.txt
timeout -k 10 300 python3 tools/probe_inc_adversarial.py 2>&1 | grep -v amdgpu
N=1000000 timeout -k 10 500 python3 tools/probe_inc_adversarial.py 2>&1 | grep -v amdgpu
K=300 timeout -k 10 200 python3 tools/probe_inc.py 2>&1 | grep incremental
