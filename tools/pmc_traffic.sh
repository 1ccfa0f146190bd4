#!/bin/bash
# HBM-side traffic and vector-instruction count of the curvature pass: rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE and
# --pmc SQ_INSTS_VALU in SEPARATE runs (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass), over
# tools/probe_pass.py (curvature passes only).  The record carries the hash of the kernel sources it was measured on
# (tools/kernel_hash.py); bench.py quotes it only while that hash still matches.
# Usage on the GPU box, from the repo root:  bash tools/pmc_traffic.sh <tag>   -> gpurun_out/<tag>_pmc_*.{csv,json}
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
  REPS=9 timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_${c}_$tag -o p -- python3 $GRAFT_REPO_ROOT/tools/probe_pass.py > $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc_$c.log 2>&1
done
python3 - "$tag" <<'PY'
import csv, json, sys, os, collections
tag = sys.argv[1]
root = os.environ['GRAFT_REPO_ROOT']
sys.path.insert(0, os.path.join(root, 'tools'))
from kernel_hash import pass_sources_hash
PASS = ('k_nc_', 'k_h2_', 'k_edge_pass', 'k_classify', 'k_clear_counts', 'fillBuffer')
def load(counter):
    acc = collections.defaultdict(float); n = collections.defaultdict(set)
    for row in csv.DictReader(open(f'/tmp/pmc_{counter}_{tag}/p_counter_collection.csv')):
        if row['Counter_Name'] != counter: continue
        k = row['Kernel_Name'].split('(')[0]
        if not any(p in k for p in PASS): continue
        acc[k] += float(row['Counter_Value']); n[k].add(row['Dispatch_Id'])
    return acc, n
fa, fn = load('FETCH_SIZE'); wa, wn = load('WRITE_SIZE'); va, vn = load('SQ_INSTS_VALU')
# passes profiled: launches of the kernel every pass starts with exactly once (some kernels run twice per pass)
once = [k for k in fn if k.endswith('k_h2_clear') or k.endswith('k_nc_clear') or k.endswith('k_clear_counts')]
passes = max(len(fn[k]) for k in once) if once else min(len(v) for v in fn.values())
with open(f'{root}/gpurun_out/{tag}_pmc_fetch_write_summary.csv', 'w') as f:
    f.write('kernel,launches,FETCH_SIZE_KiB_per_pass,WRITE_SIZE_KiB_per_pass,SQ_INSTS_VALU_per_pass\n')
    for k in sorted(fa):
        f.write(f'"{k}",{len(fn[k])},{fa[k] / passes:.1f},{wa.get(k, 0.0) / passes:.1f},{va.get(k, 0.0) / passes:.1f}\n')
fetch = sum(fa.values()) / passes; write = sum(wa.values()) / passes; valu = sum(va.values()) / passes
rec = {'_about': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU (separate runs, tools/probe_pass.py: curvature passes only) on MI355X; all kernels of one curvature pass; raw FETCH/WRITE units are KiB',
       'config': 'S100k N=100000 m=10', 'passes_profiled': passes, 'fetch_size_kib_per_pass': fetch, 'write_size_kib_per_pass': write,
       'gfx950_fetch_correction': 2.0, 'traffic_bytes_per_pass': (2.0 * fetch + write) * 1024.0, 'sq_insts_valu_per_pass': valu,
       'pass_sources_hash': pass_sources_hash(),
       'note': 'FETCH_SIZE on gfx950 reports half the bytes of wide (16 B/lane) coalesced reads (MI355X_MICROARCH.md HBM section): doubled. Counts fabric-side requests, Infinity-Cache hits included.'}
json.dump(rec, open(f'{root}/gpurun_out/{tag}_pmc_traffic.json', 'w'), indent=1)
print(json.dumps(rec))
PY
