#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_h2_engine_gpu.py -x -q -m gpu 2>&1 | grep -E "Error|assert|invariant|passed|failed" | head -8
