#!/bin/bash
# On the GPU box: time the pass (S100k, and S1M with AB_S1M=1; the incremental SDRF iteration with AB_INC=1) with the
# default library and with every csrc/variants/libdcr_hip_*.so (built by tools/build_variant.sh).  Variants are loaded
# through DCR_LIB: the default library is never overwritten.
C=discrete-curvature-rewiring_amd/csrc
run() {
  REPS=${REPS:-20} timeout -k 10 200 python3 tools/probe_pass.py || exit 1
  if [ -n "$AB_S1M" ]; then N=1000000 REPS=3 timeout -k 10 200 python3 tools/probe_pass.py || exit 1; fi
  if [ -n "$AB_INC" ]; then timeout -k 10 200 python3 tools/probe_iter_inc.py || exit 1; fi
}
echo base; run
for v in $C/variants/libdcr_hip_*.so; do
  [ -f "$v" ] || continue
  echo $v; DCR_LIB=$PWD/$v run
done
