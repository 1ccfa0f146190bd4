#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
DCR_LIB=$R/discrete-curvature-rewiring_amd/csrc/variants/libdcr_hip_ut.so REPS=2 timeout -k 10 300 python3 tools/probe_pass.py > $OUT/r05_unit_times.txt 2>&1
tail -n 40 $OUT/r05_unit_times.txt
timeout -k 10 900 python3 bench.py --steps 20 --warmup 5 > $OUT/r05_bench_a.json.log 2> $OUT/r05_bench_a.err
tail -c 600 $OUT/r05_bench_a.err
python3 - <<'PY'
import json, os
d = json.loads(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r05_bench_a.json.log').read().strip().splitlines()[-1])
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'pass', d['bfc_pass_ms'], 'outside', d['outside_pass_ms'])
print('inc', d.get('incremental_mode', {}).get('ms_per_step'), 's1m', d.get('s1m_pass', {}).get('bfc_pass_ms'))
g = d.get('gcn', {})
print('gcn', {k: g.get(k) for k in ('ms_per_epoch', 'value', 'ms_per_epoch_all_rows', 'epoch_floor_ms', 'adam', 'error')})
print('citeseer', d.get('gcn_citeseer_shape'))
print('cpu', d.get('cpu_baseline', {}).get('value'), d.get('cpu_baseline', {}).get('spread'))
PY
