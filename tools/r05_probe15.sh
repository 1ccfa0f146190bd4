#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout -k 10 600 python3 -X faulthandler -m pytest tests/test_experiment_gpu.py -x -q -m gpu -k "deeper" > $OUT/r05_deeper.log 2>&1
grep -n "File\|Error\|error\|Fatal\|fault" $OUT/r05_deeper.log | head -40
