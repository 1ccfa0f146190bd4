#!/bin/bash
# On the GPU box: kernel tests of the one-kernel first layer, then tools/probe_first_bwd.py per build (default + variants named)
R=$GRAFT_REPO_ROOT; C=$R/discrete-curvature-rewiring_amd/csrc
for v in default "$@" default "$@"; do
  echo "== $v"
  if [ $v = default ]; then unset DCR_LIB; else export DCR_LIB=$C/variants/libdcr_hip_$v.so; fi
  cd $R
  [ -z "$NOTEST" ] && timeout -k 10 300 python -m pytest tests/test_gcn.py -x -q -m gpu -k "first_layer or one_kernel" 2>&1 | tail -1
  timeout -k 10 200 python tools/probe_first_bwd.py 2>&1 | grep "one kernel\|prof" | tail -3
done
