#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 600 python3 -m pytest tests/test_gcn.py -x -q -m gpu -k "first_layer" 2>&1 | tail -3
timeout -k 10 600 python3 -m pytest tests/test_gcn_configs_gpu.py -x -q -m gpu -k "reference_dataset_shapes" 2>&1 | tail -3
echo "== citeseer 3703"; N=2120 F=3703 H=64 C=6 REPS=200 timeout -k 10 200 python3 tools/probe_first_layer.py 2>&1 | grep "one kernel\|library\|act_linear"
echo "== cora 1433 -> 128"; N=2485 F=1433 H=128 C=7 REPS=200 timeout -k 10 200 python3 tools/probe_first_layer.py 2>&1 | grep "one kernel\|library\|act_linear"
N=2120 M=2 F=3703 H=64 C=6 EPOCHS=200 timeout -k 10 300 python3 tools/probe_gcn_epoch.py 2>&1 | grep "epoch ms"
N=2120 M=2 F=3703 H=64 C=6 EPOCHS=200 DCR_FIRST_FUSED=0 timeout -k 10 300 python3 tools/probe_gcn_epoch.py 2>&1 | grep "epoch ms"
N=2485 M=2 F=1433 H=128 C=7 EPOCHS=200 timeout -k 10 300 python3 tools/probe_gcn_epoch.py 2>&1 | grep "epoch ms"
N=2485 M=2 F=1433 H=128 C=7 EPOCHS=200 DCR_FIRST_FUSED=0 timeout -k 10 300 python3 tools/probe_gcn_epoch.py 2>&1 | grep "epoch ms"
