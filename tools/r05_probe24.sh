#!/bin/bash
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/discrete-curvature-rewiring_amd/csrc/variants/libdcr_hip_tricheck.so
DCR_LIB=$V timeout -k 10 300 python3 -m pytest tests/test_h2_engine_gpu.py -x -q -m gpu -k reference_fixtures 2>&1 | grep -E "Error|passed|failed" | head -5
