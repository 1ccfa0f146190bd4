#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
V=$R/discrete-curvature-rewiring_amd/csrc/variants
echo "== citeseer 3703 default"; N=2120 F=3703 H=64 C=6 REPS=200 timeout -k 10 200 python3 tools/probe_first_layer.py 2>&1 | grep -v Warn
echo "== citeseer 3703 nofence"; DCR_LIB=$V/libdcr_hip_nofence.so N=2120 F=3703 H=64 C=6 REPS=200 timeout -k 10 200 python3 tools/probe_first_layer.py 2>&1 | grep "one kernel"
echo "== 3712 (vector staging)"; N=2120 F=3712 H=64 C=6 REPS=200 timeout -k 10 200 python3 tools/probe_first_layer.py 2>&1 | grep "one kernel\|library"
echo "== 512 -> 128 (wide, few chunks) n=2120"; N=2120 F=512 H=128 C=7 REPS=200 timeout -k 10 200 python3 tools/probe_first_layer.py 2>&1 | grep "one kernel\|library"
echo "== cora 1433 -> 128"; N=2485 F=1433 H=128 C=7 REPS=200 timeout -k 10 200 python3 tools/probe_first_layer.py 2>&1 | grep "one kernel\|library\|act_linear"
