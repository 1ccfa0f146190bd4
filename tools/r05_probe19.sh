#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout -k 10 1000 python3 -m pytest tests/test_gpu_parity.py tests/test_scale_golden_gpu.py tests/test_experiment_gpu.py tests/test_wrappers_gpu.py -x -q -m gpu 2>&1 | tail -3
INC=0 bash tools/timeline_step.sh r05b > /dev/null 2>&1; head -9 $OUT/r05b_step_timeline.txt | cut -c1-100
timeout -k 10 600 python3 bench.py --steps 100 --warmup 5 --no-gcn --no-cpu-baseline --no-s1m 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('it/s', d['value'], 'ms', d['ms_per_step'], 'pass', d['bfc_pass_ms'], 'outside', d['outside_pass_ms'], 'inc', d['incremental_mode']['ms_per_step'])"
