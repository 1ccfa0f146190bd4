#!/bin/bash
# Build libdcr_hip.so for gfx950 (cross-compiles without a GPU).  No fast-math, no FMA contraction:
# the float64 closing expression must round exactly like the reference's Python arithmetic.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -I../../include -Wall -Wno-unused-function"
OBJS=""
PIDS=""
for f in dcr_graph dcr_bfc dcr_bfc_nc dcr_bfc_h2 dcr_bfc_dense dcr_bfc_giant dcr_sdrf dcr_gcn dcr_gcn_first dcr_gemm; do
  [ -f $f.hip ] || continue
  if [ ! -f $f.o ] || [ $f.hip -nt $f.o ] || [ dcr_internal.h -nt $f.o ] || [ dcr_bfc_common.h -nt $f.o ] || [ dcr_philox.h -nt $f.o ] || [ ../../include/dcr.h -nt $f.o ]; then
    rm -f $f.o                      # a failed compile must not leave the previous object for the link
    $HIPCC $FLAGS -c $f.hip -o $f.o &
    PIDS="$PIDS $!"
  fi
  OBJS="$OBJS $f.o"
done
# host-only helper (exact vectorised cumsum of the SDRF draw): plain C++, no fast-math, no contraction
if [ ! -f dcr_host_draw.o ] || [ dcr_host_draw.cpp -nt dcr_host_draw.o ] || [ ../../include/dcr.h -nt dcr_host_draw.o ]; then
  rm -f dcr_host_draw.o
  g++ -O2 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -I../../include -Wall -c dcr_host_draw.cpp -o dcr_host_draw.o &
  PIDS="$PIDS $!"
fi
OBJS="$OBJS dcr_host_draw.o"
for p in $PIDS; do wait $p || { echo "compile failed" >&2; exit 1; }; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libdcr_hip.so $OBJS
echo "built $(pwd)/libdcr_hip.so"
