#!/bin/bash
# A/B build of libpn2hip: mlp.hip (and chain_coop.hip, which shares the tile body) compiled with extra -D flags, everything else from the regular
# object files.   tools/build_variant.sh <name> [-DPN2_...]...   ->  build_diag/libpn2hip_<name>.so   (PN2_LIB selects it)
set -e
name=$1; shift
cd "$(dirname "$0")/../extracting-tree-morphology-from-point-clouds_amd"
mkdir -p build_diag
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math"
/opt/rocm/bin/hipcc $F "$@" -c csrc/mlp.hip -o build_diag/mlp_$name.o &
/opt/rocm/bin/hipcc $F "$@" -c csrc/chain_coop.hip -o build_diag/chain_coop_$name.o &
wait
objs=$(ls build/*.o | grep -v "/mlp.o" | grep -v "/chain_coop.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_diag/libpn2hip_$name.so build_diag/mlp_$name.o build_diag/chain_coop_$name.o $objs
echo build_diag/libpn2hip_$name.so
