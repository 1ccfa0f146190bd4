set -x
cd $GRAFT_REPO_ROOT
bash tools/pmc_traffic.sh r05 > gpurun_out/r05_pmc_traffic.log 2>&1 && cp gpurun_out/r05_pmc_traffic.json profiles/r05_pmc_traffic.json
bash tools/pmc_bound.sh r05 > gpurun_out/r05_pmc_bound.log 2>&1 && cp gpurun_out/r05_pmc_bound.json profiles/r05_pmc_bound.json
bash tools/kernel_times.sh r05 > gpurun_out/r05_kernel_times.log 2>&1 && cp gpurun_out/r05_kernel_times.json profiles/r05_kernel_times.json
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench.json.log 2> gpurun_out/r05_bench.err
python bench.py --steps 100 --warmup 5 --no-gcn --no-cpu-baseline > gpurun_out/r05_bench_100steps.json.log 2>/dev/null
bash tools/timeline_pass.sh r05 > /dev/null 2>&1
N=1000000 REPS=3 bash tools/timeline_pass.sh r05_s1m > /dev/null 2>&1
INC=0 bash tools/timeline_step.sh r05 > /dev/null 2>&1
INC=1 bash tools/timeline_step.sh r05_incremental > /dev/null 2>&1
DCR_SERIAL_BINS=1 REPS=10 bash tools/prof_pass.sh r05_serial > /dev/null 2>&1
REPS=10 bash tools/prof_pass.sh r05_concurrent > /dev/null 2>&1
bash tools/prof_gcn.sh r05 > /dev/null 2>&1
bash tools/trace_gcn_epoch.sh > gpurun_out/r05_gcn_epoch_trace.txt 2>&1
N=2120 M=2 F=3703 H=64 C=6 SPARSEX=1 bash tools/trace_gcn_epoch.sh > gpurun_out/r05_gcn_citeseer_trace.txt 2>&1
bash tools/clock_first_layer.sh > gpurun_out/r05_first_layer_clock.txt 2>&1
PROBE=probe_first_bwd.py bash tools/clock_first_layer.sh >> gpurun_out/r05_first_layer_clock.txt 2>&1
bash tools/pmc_first_layer.sh r05_first_layer_fwd > /dev/null 2>&1
PROBE=probe_first_bwd.py bash tools/pmc_first_layer.sh r05_first_layer_bwd > /dev/null 2>&1
python -m pytest tests/ -q -m gpu > gpurun_out/r05_gputests.log 2>&1; tail -2 gpurun_out/r05_gputests.log
tail -c 400 gpurun_out/r05_bench.json.log
