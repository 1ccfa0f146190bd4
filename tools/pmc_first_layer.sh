#!/bin/bash
# SQ counters of the one-kernel first layer (three instantiations: pair / train / eval), rocprofv3 --pmc in separate runs
# usage on the GPU box: bash tools/pmc_first_layer.sh <tag> -> gpurun_out/<tag>_pmc_first.txt
tag=$1
cd /tmp && export TMPDIR=/tmp
run() {
  n=$1; shift
  rm -rf /tmp/pmcf_$n
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmcf_$n -o p -- python3 $GRAFT_REPO_ROOT/tools/${PROBE:-probe_first_layer.py} > /dev/null 2>&1
  python3 - /tmp/pmcf_$n/p_counter_collection.csv <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for row in csv.DictReader(open(sys.argv[1])):
    k = row['Kernel_Name']
    if 'k_first_layer' not in k and 'k_act_linear' not in k and 'k_atb' not in k and 'Cijk' not in k: continue
    k = k.split('(')[0].replace('void dcr::', '')[:40]
    acc[k][row['Counter_Name']] += float(row['Counter_Value']); disp[k].add(row['Dispatch_Id'])
for k in sorted(acc):
    print(k, {c: round(v / len(disp[k])) for c, v in sorted(acc[k].items())})
PY
}
{
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run b SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES
run c GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC
} > $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc_first.txt 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc_first.txt
