#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee $OUT/r05_gputests.log
