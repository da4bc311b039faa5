#!/bin/bash
# Local helper: rebuild the library and the -DPN2_FPS_DIAG variant, then A/B + phase stamps on the GPU box.
set -e
cd "$(dirname "$0")/.."
(cd extracting-tree-morphology-from-point-clouds_amd && python build.py 2>&1 | grep -E "error|warning" -A3 | head -30) || true
bash tools/build_diag.sh > /dev/null
/usr/local/graft/bin/gpurun --timeout 600 -- 'timeout -k 10 200 python tools/ab_fps_sort.py 2>&1 | tail -6 && timeout -k 10 200 python tools/diag_fps.py 2>&1 | tail -5' 2>&1 | tail -12
