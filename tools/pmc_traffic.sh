#!/bin/bash
# HBM-side traffic of the curvature pass: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE runs
# (MI355X_MICROARCH.md: they do not fit one pass), over tools/probe_pass.py (curvature passes only).
# Usage on the GPU box, from the repo root:  bash tools/pmc_traffic.sh <tag>   -> gpurun_out/<tag>_pmc_*.{csv,json}
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp
REPS=9 timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmcf_$tag -o p -- python3 $GRAFT_REPO_ROOT/tools/probe_pass.py > $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc_fetch.log 2>&1
REPS=9 timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmcw_$tag -o p -- python3 $GRAFT_REPO_ROOT/tools/probe_pass.py > $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc_write.log 2>&1
python3 - "$tag" <<'PY'
import csv, json, sys, os, collections
tag = sys.argv[1]
root = os.environ['GRAFT_REPO_ROOT']
def load(path, counter):
    acc = collections.defaultdict(float); n = collections.defaultdict(set)
    for row in csv.DictReader(open(path)):
        if row['Counter_Name'] != counter: continue
        k = row['Kernel_Name'].split('(')[0]
        if 'k_nc_' not in k: continue
        acc[k] += float(row['Counter_Value']); n[k].add(row['Dispatch_Id'])
    return acc, n
fa, fn = load(f'/tmp/pmcf_{tag}/p_counter_collection.csv', 'FETCH_SIZE')
wa, wn = load(f'/tmp/pmcw_{tag}/p_counter_collection.csv', 'WRITE_SIZE')
passes = max(len(v) for v in fn.values())
with open(f'{root}/gpurun_out/{tag}_pmc_fetch_write_summary.csv', 'w') as f:
    f.write('kernel,launches,FETCH_SIZE_KiB_per_pass,WRITE_SIZE_KiB_per_pass\n')
    for k in sorted(fa):
        f.write(f'"{k}",{len(fn[k])},{fa[k] / passes:.1f},{wa.get(k, 0.0) / passes:.1f}\n')
fetch = sum(fa.values()) / passes; write = sum(wa.values()) / passes
rec = {'_about': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/probe_pass.py: curvature passes only) on MI355X; per curvature pass = k_nc_clear + k_nc_plan<0,1> + the class kernels; raw counter units are KiB',
       'config': 'S100k N=100000 m=10', 'passes_profiled': passes, 'fetch_size_kib_per_pass': fetch, 'write_size_kib_per_pass': write,
       'gfx950_fetch_correction': 2.0, 'traffic_bytes_per_pass': (2.0 * fetch + write) * 1024.0,
       'note': 'FETCH_SIZE on gfx950 reports half the bytes of wide (16 B/lane) coalesced reads (MI355X_MICROARCH.md HBM section): doubled. Counts fabric-side requests, Infinity-Cache hits included.'}
json.dump(rec, open(f'{root}/gpurun_out/{tag}_pmc_traffic.json', 'w'), indent=1)
print(json.dumps(rec))
PY
