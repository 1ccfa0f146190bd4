#!/bin/bash
# On the GPU box: time the pass with the default library and with every csrc/variants/libdcr_hip_*.so
C=discrete-curvature-rewiring_amd/csrc
cp $C/libdcr_hip.so /tmp/libdcr_base.so
echo base; REPS=20 timeout -k 10 200 python3 tools/probe_pass.py || exit 1
for v in $C/variants/libdcr_hip_*.so; do
  cp $v $C/libdcr_hip.so; echo $v; REPS=20 timeout -k 10 200 python3 tools/probe_pass.py || exit 1
done
cp /tmp/libdcr_base.so $C/libdcr_hip.so
