#!/bin/bash
cd $GRAFT_REPO_ROOT
SECONDS_BUDGET=150 timeout -k 10 500 python3 tests/fuzz_parity.py 2>&1 | tail -5
