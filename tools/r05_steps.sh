#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 1 2; do
 echo "full, verdict first (default): $(K=100 timeout -k 10 200 python3 tools/probe_step.py 2>&1 | grep 'device_draw=1' | head -1)"
 echo "full, no round trip: $(DCR_DRAW_SYNC=0 K=100 timeout -k 10 200 python3 tools/probe_step.py 2>&1 | grep 'device_draw=1' | head -1)"
done
