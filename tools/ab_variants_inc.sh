#!/bin/bash
# like tools/ab_variants.sh, for the full pass and the incremental SDRF iteration
C=discrete-curvature-rewiring_amd/csrc
cp $C/libdcr_hip.so /tmp/libdcr_base.so
run() { REPS=20 timeout -k 10 200 python3 tools/probe_pass.py || exit 1; timeout -k 10 200 python3 tools/probe_iter_inc.py || exit 1; }
echo base; run
for v in $C/variants/libdcr_hip_*.so; do
  cp $v $C/libdcr_hip.so; echo $v; run
done
cp /tmp/libdcr_base.so $C/libdcr_hip.so
