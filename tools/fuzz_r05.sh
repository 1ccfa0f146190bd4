#!/bin/bash
# round 5: the randomised parity sweep (tests/fuzz_parity.py) on the final sources — default routes, then the incremental pass's
# edge list taken from the flagged nodes' rows (large graphs' route) forced on every graph
cd $GRAFT_REPO_ROOT
SECONDS_BUDGET=420 timeout -k 10 700 python3 tests/fuzz_parity.py 2>&1 | tail -1 | tee gpurun_out/r05_fuzz.txt
DCR_NC_FINE_SWEEP=0 SECONDS_BUDGET=200 timeout -k 10 500 python3 tests/fuzz_parity.py 2>&1 | tail -1 | tee -a gpurun_out/r05_fuzz.txt
DCR_NC_FINE_FULL=1000000000 SECONDS_BUDGET=150 timeout -k 10 500 python3 tests/fuzz_parity.py 2>&1 | tail -1 | tee -a gpurun_out/r05_fuzz.txt
