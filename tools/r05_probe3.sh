#!/bin/bash
# round 5, third GPU call: the K-chunked first-layer kernels (tests), unit timings of the block classes, default layout check
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gcn.py -x -q -m gpu -k "first_layer or one_kernel" 2>&1 | tail -25 | tee $OUT/r05_first_tests.txt
timeout -k 10 900 python3 -m pytest tests/test_gcn_configs_gpu.py -x -q -m gpu -k "reference_dataset_shapes or citeseer" 2>&1 | tail -25 | tee -a $OUT/r05_first_tests.txt
DCR_LIB=$R/discrete-curvature-rewiring_amd/csrc/variants/libdcr_hip_ut.so REPS=2 timeout -k 10 300 python3 tools/probe_pass.py > $OUT/r05_unit_times.txt 2>&1
tail -n 40 $OUT/r05_unit_times.txt
for r in 1 2; do
  for l in five default; do
    if [ $l = five ]; then ms=$(DCR_H2_LAYOUT=five REPS=40 timeout -k 10 300 python3 tools/probe_pass.py 2>&1 | grep "pass ms" | awk '{print $3}');
    else ms=$(REPS=40 timeout -k 10 300 python3 tools/probe_pass.py 2>&1 | grep "pass ms" | awk '{print $3}'); fi
    echo "S100k layout=$l $ms"
  done
done | tee $OUT/r05_layout_default.txt
