#!/bin/bash
# On the GPU box: time the pass (S100k, and S1M with AB_S1M=1) with the default library and with every
# csrc/variants/libdcr_hip_*.so (built by tools/build_variant.sh)
C=discrete-curvature-rewiring_amd/csrc
cp $C/libdcr_hip.so /tmp/libdcr_base.so
run() { REPS=${REPS:-20} timeout -k 10 200 python3 tools/probe_pass.py || exit 1; if [ -n "$AB_S1M" ]; then N=1000000 REPS=3 timeout -k 10 200 python3 tools/probe_pass.py || exit 1; fi; }
echo base; run
for v in $C/variants/libdcr_hip_*.so; do
  cp $v $C/libdcr_hip.so; echo $v; run
done
cp /tmp/libdcr_base.so $C/libdcr_hip.so
