#!/bin/bash
# Average duration of every kernel of the bench's SDRF leg (rocprofv3 --kernel-trace --stats over bench.py without its side
# legs): bench.py quotes the arg-ext and improvement kernels' rooflines from this record while the hash of the SDRF kernel
# sources (tools/kernel_hash.py) still matches.  On the GPU box, from the repo root:
#   bash tools/kernel_times.sh <tag>   -> gpurun_out/<tag>_kernel_times.json, gpurun_out/<tag>_bench_kernel_stats.csv
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_$tag
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$tag -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 5 --no-gcn --no-cpu-baseline --no-config2 --no-incremental --no-s1m > $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench_profiled.json.log 2> /dev/null
cp /tmp/kt_$tag/p_kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench_kernel_stats.csv
python3 - "$tag" <<'PY'
import csv, json, os, sys
tag = sys.argv[1]
root = os.environ['GRAFT_REPO_ROOT']
sys.path.insert(0, os.path.join(root, 'tools'))
from kernel_hash import pass_sources_hash
k = {}
for r in csv.DictReader(open(f'/tmp/kt_{tag}/p_kernel_stats.csv')):
    name = r['Name'].split('(')[0].replace('void ', '').replace('dcr::', '').split('<')[0]
    tot = k.setdefault(name, [0.0, 0])
    tot[0] += float(r['TotalDurationNs']); tot[1] += int(r['Calls'])
line = [l for l in open(f'{root}/gpurun_out/{tag}_bench_profiled.json.log') if l.startswith('{')]
slots = 0
rec = {'_about': 'rocprofv3 --kernel-trace --stats over bench.py --steps 100 (SDRF leg only) on MI355X: average microseconds per launch',
       'kernels_us': {n: t / c / 1e3 for n, (t, c) in sorted(k.items())}, 'calls': {n: c for n, (t, c) in sorted(k.items())},
       'sdrf_sources_hash': pass_sources_hash(('dcr_sdrf.hip', 'dcr_internal.h')), 'pass_sources_hash': pass_sources_hash(),
       'bench_line': json.loads(line[-1]) if line else None}
# adjacency slots of the bench graph as dcr_graph_create lays it out: every row its degree + max(8, degree / 4) of slack
import numpy as np
sys.path[:0] = [root, os.path.join(root, 'discrete-curvature-rewiring_amd')]
from dcr import synthetic
ei, n = synthetic.powerlaw_graph(100000, 10, seed=12345)
deg = np.bincount(ei[0], minlength=n)
rec['adjacency_slots'] = int((deg + np.maximum(8, deg // 4)).sum())
json.dump(rec, open(f'{root}/gpurun_out/{tag}_kernel_times.json', 'w'), indent=1)
print(json.dumps(rec['kernels_us']))
PY
