#!/bin/bash
cd $GRAFT_REPO_ROOT
V=discrete-curvature-rewiring_amd/csrc/variants
DCR_LIB=$V/libdcr_hip_dstats.so K=3 timeout -k 10 200 python3 tools/probe_inc.py 2>&1 | grep -v amdgpu | tail -16
