#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== edge by edge where its estimate wins (default)"; timeout -k 10 200 python3 tools/probe_small_sdrf.py 2>&1 | grep -v amdgpu
echo "== class kernels (DCR_NC_FINE=0)"; DCR_NC_FINE=0 timeout -k 10 200 python3 tools/probe_small_sdrf.py 2>&1 | grep -v amdgpu
