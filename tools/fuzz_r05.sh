#!/bin/bash
# round 5: the randomised parity sweeps on the final sources with other seeds than the test suite's
# (progress goes straight to files under gpurun_out/: a pipe into tail looks like a hung run to the box's watchdog)
cd $GRAFT_REPO_ROOT
for s in ${SEEDS:-2 3}; do
  SEED=$s HUB_PROB=${HUB_PROB:-0.06} SECONDS_BUDGET=${BUDGET:-360} timeout -k 10 700 python3 tests/fuzz_parity.py > gpurun_out/r05_fuzz_seed$s.txt 2>&1
  tail -1 gpurun_out/r05_fuzz_seed$s.txt
done
SEED=${SEEDS##* } SECONDS_BUDGET=${BUDGET_ENGINES:-200} timeout -k 10 700 python3 tests/fuzz_engines.py > gpurun_out/r05_fuzz_engines_more.txt 2>&1
tail -1 gpurun_out/r05_fuzz_engines_more.txt
