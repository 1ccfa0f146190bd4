#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout -k 10 600 python3 -m pytest tests/test_experiment_gpu.py -x -q -m gpu -k "deeper or head" 2>&1 | tail -4
timeout -k 10 900 python3 tools/probe_spmm_order.py 2>&1 | grep -v Warn | tee $OUT/r05_spmm_order.txt
