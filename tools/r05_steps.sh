#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_scale_golden_gpu.py tests/test_wrappers_gpu.py tests/test_checkers_gpu.py -x -q -m gpu > gpurun_out/r05_sdrf_tests.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r05_sdrf_tests.txt
tail -4 gpurun_out/r05_sdrf_tests.txt
K=100 timeout -k 10 200 python3 tools/probe_step.py 2>&1 | grep -v amdgpu.ids
bash tools/timeline_step.sh r05c > /dev/null 2>&1
head -12 gpurun_out/r05c_step_timeline.txt
