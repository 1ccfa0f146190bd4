#!/bin/bash
# tools/build_variant.sh <name> <extra hipcc flags...>: builds csrc/variants/libdcr_hip_<name>.so with -D switches for
# on-GPU A/B runs (copy it over csrc/libdcr_hip.so on the GPU box between probe runs).
set -e
cd "$(dirname "$0")/../discrete-curvature-rewiring_amd/csrc"
name=$1; shift
mkdir -p variants/$name
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -I../../include -Wall -Wno-unused-function"
for f in dcr_graph dcr_bfc dcr_bfc_nc dcr_bfc_h2 dcr_bfc_dense dcr_bfc_giant dcr_sdrf dcr_gcn dcr_gcn_first dcr_gemm; do
  /opt/rocm/bin/hipcc $FLAGS "$@" -c $f.hip -o variants/$name/$f.o &
done
g++ -O2 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -I../../include -Wall -c dcr_host_draw.cpp -o variants/$name/dcr_host_draw.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libdcr_hip_$name.so variants/$name/*.o
rm -rf variants/$name
echo built variants/libdcr_hip_$name.so
