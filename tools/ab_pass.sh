#!/bin/bash
# On the GPU box: per-kernel times of the curvature pass for the default library and every csrc/variants/libdcr_hip_*.so,
# each loaded through DCR_LIB (the default library is never overwritten); usage: bash tools/ab_pass.sh  (env DCR_PASS, N, M, REPS)
C=$GRAFT_REPO_ROOT/discrete-curvature-rewiring_amd/csrc
echo base; bash $GRAFT_REPO_ROOT/tools/prof_pass.sh ab_base
for v in $C/variants/libdcr_hip_*.so; do
  [ -f "$v" ] || continue
  n=$(basename $v .so); n=${n#libdcr_hip_}
  echo $n; DCR_LIB=$v bash $GRAFT_REPO_ROOT/tools/prof_pass.sh ab_$n
done
