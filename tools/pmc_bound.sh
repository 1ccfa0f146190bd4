#!/bin/bash
# What binds the curvature pass: SQ wait / issue / LDS counters per kernel (rocprofv3 --pmc, three counter sets in SEPARATE runs,
# each over tools/probe_pass.py = curvature passes only, the program directly after `--`), summarised with the hash of the
# kernel sources (tools/kernel_hash.py) so that bench.py quotes them only while they describe the kernels it runs.
# usage on the GPU box, from the repo root:  bash tools/pmc_bound.sh <tag>   -> gpurun_out/<tag>_pmc_bound.json (+ the raw per-kernel tables)
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
  n=$1; shift
  rm -rf /tmp/pmcb_${n}_$tag
  REPS=5 timeout -k 10 400 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmcb_${n}_$tag -o p -- python3 $GRAFT_REPO_ROOT/tools/probe_pass.py > $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc_bound_$n.log 2>&1
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pmcb_${n}_$tag/p_counter_collection.csv k_h2 > $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc_sq_$n.txt
}
run waits SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run insts SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_WAVES
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_WAVES SQ_INSTS_LDS
python3 - "$tag" <<'PY'
import csv, json, os, sys, collections
tag = sys.argv[1]
root = os.environ['GRAFT_REPO_ROOT']
sys.path.insert(0, os.path.join(root, 'tools'))
from kernel_hash import pass_sources_hash
def load(name):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for row in csv.DictReader(open(f'/tmp/pmcb_{name}_{tag}/p_counter_collection.csv')):
        k = row['Kernel_Name'].split('(')[0].replace('void ', '').replace('dcr::', '')
        if 'k_h2' not in k: continue
        acc[k][row['Counter_Name']] += float(row['Counter_Value']); disp[k].add(row['Dispatch_Id'])
    return {k: {c: v / len(disp[k]) for c, v in acc[k].items()} for k in acc}, {k: len(v) for k, v in disp.items()}
w, nd = load('waits'); i, _ = load('insts'); l, _ = load('lds')
passes = max(nd.get(k, 0) for k in nd if k.endswith('k_h2_clear')) if any(k.endswith('k_h2_clear') for k in nd) else 1
per = {}
tot_valu = tot_quad = 0.0
for k in sorted(w):
    wc = w[k].get('SQ_WAVE_CYCLES', 0.0)
    if wc <= 0: continue
    launches_per_pass = nd[k] / passes
    rec = {'launches_per_pass': launches_per_pass,
           'wait_any_over_wave_cycles': w[k].get('SQ_WAIT_ANY', 0.0) / wc,
           'active_any_over_wave_cycles': w[k].get('SQ_ACTIVE_INST_ANY', 0.0) / wc,
           'active_valu_over_wave_cycles': w[k].get('SQ_ACTIVE_INST_VALU', 0.0) / wc,
           'wait_inst_lds_over_wave_cycles': w[k].get('SQ_WAIT_INST_LDS', 0.0) / wc}
    if k in i:
        rec['insts_valu_per_launch'] = i[k].get('SQ_INSTS_VALU', 0.0)
        rec['insts_salu_per_launch'] = i[k].get('SQ_INSTS_SALU', 0.0)
        rec['insts_lds_per_launch'] = i[k].get('SQ_INSTS_LDS', 0.0)
        tot_valu += i[k].get('SQ_INSTS_VALU', 0.0) * launches_per_pass
        tot_quad += i[k].get('SQ_ACTIVE_INST_VALU', 0.0) * launches_per_pass
    if k in l and l[k].get('SQ_LDS_IDX_ACTIVE', 0.0) > 0:
        rec['lds_bank_conflict_over_lds_active'] = l[k].get('SQ_LDS_BANK_CONFLICT', 0.0) / l[k]['SQ_LDS_IDX_ACTIVE']
    per[k] = rec
out = {'_about': 'rocprofv3 --pmc, SQ block, three counter sets in separate runs over tools/probe_pass.py (S100k curvature passes) on MI355X; '
                 'per launch averages; SQ_WAVE_CYCLES / SQ_WAIT_ANY / SQ_ACTIVE_INST_* count in the same (4-cycle) units, so the ratios are '
                 'fractions of a wave\'s lifetime: parked on s_waitcnt, issuing anything, issuing vector ALU',
       'config': 'S100k N=100000 m=10', 'passes_profiled': passes, 'per_kernel': per,
       'pass': {'sq_insts_valu_per_pass': tot_valu, 'sq_active_inst_valu_quads_per_pass': tot_quad},
       'pass_sources_hash': pass_sources_hash()}
json.dump(out, open(f'{root}/gpurun_out/{tag}_pmc_bound.json', 'w'), indent=1)
print(json.dumps({k: {kk: round(vv, 3) for kk, vv in v.items() if 'over' in kk} for k, v in per.items()}, indent=0))
PY
