#!/bin/bash
# On the GPU box: kernel tests of the one-kernel first layer, then tools/probe_first_layer.py per build (default + variants named)
#   bash tools/first_layer_ab.sh <name...>
R=$GRAFT_REPO_ROOT; C=$R/discrete-curvature-rewiring_amd/csrc
cd $R && timeout -k 10 600 python -m pytest tests/test_gcn.py -x -q -m gpu -k "first_layer or one_kernel" > gpurun_out/first_tests.log 2>&1; tail -3 gpurun_out/first_tests.log
for v in default "$@" default; do
  echo "== $v"
  if [ $v = default ]; then unset DCR_LIB; else export DCR_LIB=$C/variants/libdcr_hip_$v.so; fi
  timeout -k 10 200 python tools/probe_first_layer.py 2>&1 | grep -v amdgpu.ids
done
