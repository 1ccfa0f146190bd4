#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout -k 10 900 python3 -m pytest tests/test_experiment_gpu.py tests/test_gcn.py -x -q -m gpu 2>&1 | tail -3
for r in 1 2; do EPOCHS=40 timeout -k 10 300 python3 tools/probe_gcn_epoch.py 2>&1 | grep "epoch ms"; done
bash tools/trace_gcn_epoch.sh 2>&1 | cut -c1-110
