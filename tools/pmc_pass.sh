#!/bin/bash
# SQ counters of the curvature pass kernels (two sets, separate runs): tools/pmc_pass.sh <tag> [kernel-name filter]
# (run on the GPU box from the repo root; writes gpurun_out/<tag>_pmc_sq_{insts,waits}.txt)
F=${2:-k_h2}
cd /tmp && export TMPDIR=/tmp
REPS=5 timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD --output-format csv -d /tmp/pmca_$1 -o p -- python3 $GRAFT_REPO_ROOT/tools/probe_pass.py > $GRAFT_REPO_ROOT/gpurun_out/$1_pmc_a.log 2>&1 || exit 1
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pmca_$1/p_counter_collection.csv $F > $GRAFT_REPO_ROOT/gpurun_out/$1_pmc_sq_insts.txt
REPS=5 timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d /tmp/pmcb_$1 -o p -- python3 $GRAFT_REPO_ROOT/tools/probe_pass.py > $GRAFT_REPO_ROOT/gpurun_out/$1_pmc_b.log 2>&1 || exit 1
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pmcb_$1/p_counter_collection.csv $F > $GRAFT_REPO_ROOT/gpurun_out/$1_pmc_sq_waits.txt
REPS=5 timeout -k 10 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_WAVES SQ_INSTS_LDS --output-format csv -d /tmp/pmcc_$1 -o p -- python3 $GRAFT_REPO_ROOT/tools/probe_pass.py > $GRAFT_REPO_ROOT/gpurun_out/$1_pmc_c.log 2>&1 || exit 1
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pmcc_$1/p_counter_collection.csv $F > $GRAFT_REPO_ROOT/gpurun_out/$1_pmc_lds.txt
