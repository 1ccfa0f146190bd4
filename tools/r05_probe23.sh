#!/bin/bash
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/discrete-curvature-rewiring_amd/csrc/variants/libdcr_hip_bal.so
for env in "X=1" "DCR_SERIAL_BINS=1" "DCR_H2_LAYOUT=five" "DCR_H2_LAYOUT=0,0,0,0,0"; do
  echo "== $env"
  env $env DCR_LIB=$V timeout -k 10 300 python3 -m pytest tests/test_h2_engine_gpu.py -x -q -m gpu -k reference_fixtures 2>&1 | grep -E "AssertionError: \(|passed|failed" | head -3
done
