#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 300 python3 -m pytest tests/test_h2_engine_gpu.py -x -q -m gpu -k reference_fixtures 2>&1 | grep -E "Error|assert|invariant|passed|failed" | head -20
DCR_H2_TRI_SETS=0 timeout -k 10 300 python3 -m pytest tests/test_h2_engine_gpu.py -x -q -m gpu -k reference_fixtures 2>&1 | grep -E "Error|assert|invariant|passed|failed" | head -20
