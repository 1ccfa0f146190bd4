#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_scale_golden_gpu.py tests/test_h2_engine_gpu.py tests/test_engine_roles_gpu.py tests/test_wrappers_gpu.py tests/test_checkers_gpu.py -x -q -m gpu > gpurun_out/r05_inc_tests.txt 2>&1
echo "rc=$?" >> gpurun_out/r05_inc_tests.txt; tail -5 gpurun_out/r05_inc_tests.txt
for r in 1 2; do
 echo "fine edges: $(K=300 timeout -k 10 200 python3 tools/probe_inc.py 2>&1 | grep incremental)"
 echo "class kernels: $(DCR_NC_FINE=0 K=300 timeout -k 10 200 python3 tools/probe_inc.py 2>&1 | grep incremental)"
done
INC=1 bash tools/timeline_step.sh r05g_inc > /dev/null 2>&1
sed -n 7,30p gpurun_out/r05g_inc_step_timeline.txt
