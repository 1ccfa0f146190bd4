#!/bin/bash
# round 5: the randomised parity sweep (tests/fuzz_parity.py) on the final sources with other seeds than the test suite's
# (progress goes straight to a file under gpurun_out/: a pipe into tail looks like a hung run to the box's watchdog)
cd $GRAFT_REPO_ROOT
SEED=2 SECONDS_BUDGET=360 timeout -k 10 600 python3 tests/fuzz_parity.py > gpurun_out/r05_fuzz_seed2.txt 2>&1
tail -1 gpurun_out/r05_fuzz_seed2.txt
SEED=3 HUB_PROB=0.12 SECONDS_BUDGET=300 timeout -k 10 600 python3 tests/fuzz_parity.py > gpurun_out/r05_fuzz_seed3.txt 2>&1
tail -1 gpurun_out/r05_fuzz_seed3.txt
