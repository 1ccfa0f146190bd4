#!/bin/bash
# On the GPU box: per-kernel times (rocprofv3 --kernel-trace --stats) of the two-hop curvature pass (DCR_PASS=h2) for the
# default library and every csrc/variants/libdcr_hip_*.so; usage: bash tools/ab_h2.sh  (env N, M, REPS pass through)
C=$GRAFT_REPO_ROOT/discrete-curvature-rewiring_amd/csrc
cp $C/libdcr_hip.so /tmp/libdcr_base.so
cd /tmp && export TMPDIR=/tmp
export DCR_PASS=${DCR_PASS:-h2}
run() {
  rm -rf /tmp/prof_ab
  REPS=${REPS:-10} timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ab -o p -- python3 $GRAFT_REPO_ROOT/tools/probe_pass.py 2>/dev/null | grep "pass ms" || exit 1
  python3 - <<'PY'
import csv
for r in list(csv.DictReader(open('/tmp/prof_ab/p_kernel_stats.csv')))[:6]:
    print(f"   {float(r['AverageNs'])/1e3:10.1f} us x{r['Calls']:>4}  {r['Name'][:70]}")
PY
}
echo base; run
for v in $C/variants/libdcr_hip_*.so; do
  [ -f "$v" ] || continue
  cp $v $C/libdcr_hip.so; echo $(basename $v); run
done
cp /tmp/libdcr_base.so $C/libdcr_hip.so
