#!/bin/bash
# On the GPU box: the two-hop pass with the default library and with every csrc/variants/libdcr_hip_*.so (tools/build_variant.sh),
# each loaded through DCR_LIB: parity (tests/test_h2_engine_gpu.py unless AB_NOTEST=1), pass ms on S100k (and S1M with AB_S1M=1),
# and per-kernel times with the class kernels one after the other (DCR_SERIAL_BINS=1; AB_SERIAL=1).
# usage: bash tools/ab_h2.sh <tag> [variant names...]   -> gpurun_out/ab_h2_<tag>.txt
tag=$1; shift
R=$GRAFT_REPO_ROOT; C=$R/discrete-curvature-rewiring_amd/csrc; OUT=$R/gpurun_out/ab_h2_$tag.txt
: > $OUT
one() {  # name, lib ('' = default)
  echo "=== $1" | tee -a $OUT
  if [ -n "$2" ]; then export DCR_LIB=$2; else unset DCR_LIB; fi
  if [ -z "$AB_NOTEST" ]; then
    (cd $R && timeout -k 10 300 python3 -m pytest tests/test_h2_engine_gpu.py -x -q 2>&1 | tail -2) | tee -a $OUT
  fi
  for i in 1 2; do (cd $R && REPS=${REPS:-30} timeout -k 10 200 python3 tools/probe_pass.py) | tee -a $OUT; done
  if [ -n "$AB_S1M" ]; then (cd $R && N=1000000 REPS=4 timeout -k 10 300 python3 tools/probe_pass.py) | tee -a $OUT; fi
  if [ -n "$AB_SERIAL" ]; then DCR_SERIAL_BINS=1 bash $R/tools/prof_pass.sh ab_${tag}_$1 | tee -a $OUT; fi
}
if [ $# -eq 0 ]; then
  one default ""
  for v in $C/variants/libdcr_hip_*.so; do [ -f "$v" ] || continue; n=$(basename $v .so); one ${n#libdcr_hip_} $v; done
else
  for n in "$@"; do if [ "$n" = default ]; then one default ""; else one $n $C/variants/libdcr_hip_$n.so; fi; done
fi
