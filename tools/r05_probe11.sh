#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 900 python3 -m pytest tests/test_gcn_configs_gpu.py -x -q -m gpu -k "reference_dataset_shapes or citeseer" 2>&1 | tail -5
timeout -k 10 900 python3 -m pytest tests/test_gcn.py tests/test_experiment_gpu.py -x -q -m gpu 2>&1 | tail -5
N=2120 M=2 F=3703 H=64 C=6 EPOCHS=200 SPARSEX=1 timeout -k 10 300 python3 tools/probe_gcn_epoch.py 2>&1 | grep "epoch ms"
N=2485 M=2 F=1433 H=128 C=7 EPOCHS=200 SPARSEX=1 timeout -k 10 300 python3 tools/probe_gcn_epoch.py 2>&1 | grep "epoch ms"
N=2120 M=2 F=3703 H=64 C=6 SPARSEX=1 bash tools/trace_gcn_epoch.sh 2>&1 | cut -c1-130
