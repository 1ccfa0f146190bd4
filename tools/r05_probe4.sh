#!/bin/bash
# round 5, fourth GPU call: K-chunked first-layer kernels + the fused head (tests), unit timings of the block classes
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gcn.py -x -q -m gpu -k "first_layer or one_kernel" 2>&1 | tail -25 | tee $OUT/r05_first_tests.txt
timeout -k 10 900 python3 -m pytest tests/test_gcn_configs_gpu.py -x -q -m gpu -k "reference_dataset_shapes or citeseer" 2>&1 | tail -25 | tee -a $OUT/r05_first_tests.txt
timeout -k 10 900 python3 -m pytest tests/test_experiment_gpu.py -x -q -m gpu 2>&1 | tail -25 | tee $OUT/r05_experiment_tests.txt
DCR_LIB=$R/discrete-curvature-rewiring_amd/csrc/variants/libdcr_hip_ut.so REPS=2 timeout -k 10 300 python3 tools/probe_pass.py > $OUT/r05_unit_times.txt 2>&1
tail -n 40 $OUT/r05_unit_times.txt
