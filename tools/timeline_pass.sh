#!/bin/bash
# On the GPU box: kernel timeline of one curvature pass on the bench graph (rocprofv3 --kernel-trace over tools/probe_pass.py,
# read by tools/trace_pass.py).  usage: bash tools/timeline_pass.sh <tag>  -> gpurun_out/<tag>_pass_timeline.txt
TAG=${1:-pass}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl_pass
REPS=${REPS:-10} timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_pass -o p -- python3 $GRAFT_REPO_ROOT/tools/probe_pass.py 2>/dev/null | grep "pass ms" > $OUT/${TAG}_pass_timeline.txt || exit 1
python3 $GRAFT_REPO_ROOT/tools/trace_pass.py /tmp/tl_pass/p_kernel_trace.csv 3 >> $OUT/${TAG}_pass_timeline.txt
cat $OUT/${TAG}_pass_timeline.txt
