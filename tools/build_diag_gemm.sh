#!/bin/bash
# Diagnostic build of libpn2hip: mlp.hip with -DPN2_GEMM_DIAG (per-phase cycle stamps inside the one-tile-per-workgroup GEMM
# kernels), everything else from the regular object files.  Used by tools/diag_gemm.py on the GPU box.
set -e
cd "$(dirname "$0")/../extracting-tree-morphology-from-point-clouds_amd"
python build.py > /dev/null
mkdir -p build_diag
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DPN2_GEMM_DIAG -c csrc/mlp.hip -o build_diag/mlp.o
objs=$(ls build/*.o | grep -v "/mlp.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_diag/libpn2hip_gemm_diag.so build_diag/mlp.o $objs
echo build_diag/libpn2hip_gemm_diag.so
