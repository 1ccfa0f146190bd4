set -x
cd $GRAFT_REPO_ROOT
bash tools/pmc_traffic.sh r04 > gpurun_out/r04_pmc_traffic.log 2>&1 && cp gpurun_out/r04_pmc_traffic.json profiles/r04_pmc_traffic.json
bash tools/pmc_bound.sh r04 > gpurun_out/r04_pmc_bound.log 2>&1 && cp gpurun_out/r04_pmc_bound.json profiles/r04_pmc_bound.json
bash tools/kernel_times.sh r04 > gpurun_out/r04_kernel_times.log 2>&1 && cp gpurun_out/r04_kernel_times.json profiles/r04_kernel_times.json
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench.json.log 2> gpurun_out/r04_bench.err
python bench.py --steps 100 --warmup 5 --no-gcn --no-cpu-baseline > gpurun_out/r04_bench_100steps.json.log 2>/dev/null
bash tools/timeline_pass.sh r04 > /dev/null 2>&1
N=1000000 REPS=3 bash tools/timeline_pass.sh r04_s1m > /dev/null 2>&1
INC=0 bash tools/timeline_step.sh r04 > /dev/null 2>&1
INC=1 bash tools/timeline_step.sh r04_incremental > /dev/null 2>&1
DCR_SERIAL_BINS=1 REPS=10 bash tools/prof_pass.sh r04_serial > /dev/null 2>&1
REPS=10 bash tools/prof_pass.sh r04_concurrent > /dev/null 2>&1
bash tools/prof_gcn.sh r04 > /dev/null 2>&1
bash tools/clock_first_layer.sh > gpurun_out/r04_first_layer_clock.txt 2>&1
PROBE=probe_first_bwd.py bash tools/clock_first_layer.sh >> gpurun_out/r04_first_layer_clock.txt 2>&1
bash tools/pmc_first_layer.sh r04_first_layer_fwd > /dev/null 2>&1
PROBE=probe_first_bwd.py bash tools/pmc_first_layer.sh r04_first_layer_bwd > /dev/null 2>&1
python -m pytest tests/ -q -m gpu > gpurun_out/r04_gputests.log 2>&1; tail -2 gpurun_out/r04_gputests.log
tail -c 400 gpurun_out/r04_bench.json.log
