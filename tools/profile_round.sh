# GPU box: bench lines + rocprofv3 kernel stats + PMC traffic for profiles/ (run through gpurun from the repo root):
#   gpurun --timeout 1100 -- 'bash tools/profile_round.sh r02'
# then copy gpurun_out/round/* into profiles/ with the round prefix (tools/collect_profiles.py).
set -e
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/round
mkdir -p $O
cd $R
if [ -z "$PN2_SKIP_BENCH" ]; then
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench (default flags: monolithic depth 4, f32) done"
timeout -k 10 300 python bench.py --mode rasterized > $O/bench_rasterized.json 2>> $O/bench.err
echo "bench rasterized done"
timeout -k 10 300 python bench.py --dtype bf16 --no-cpu-baseline > $O/bench_bf16.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --dtype bf16 --graph --no-cpu-baseline > $O/bench_bf16_graph.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --graph --no-cpu-baseline > $O/bench_graph.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --mode rasterized --dtype bf16 --no-cpu-baseline > $O/bench_rasterized_bf16.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --trees 8 --points 65536 --no-cpu-baseline > $O/bench_cfg3_8x65536.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --depth 5 --no-cpu-baseline > $O/bench_depth5.json 2>> $O/bench.err
PN2_NO_COOP=1 timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_no_coop.json 2>> $O/bench.err
PN2_COOP_MAX_ROWS=10000 timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_coop_all_levels.json 2>> $O/bench.err
PN2_NO_TILE32=1 PN2_NO_COOP=1 timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_no_tile32.json 2>> $O/bench.err
PN2_BF16_STORAGE=0 timeout -k 10 300 python bench.py --dtype bf16 --no-cpu-baseline > $O/bench_bf16_fp32_rows.json 2>> $O/bench.err
PN2_NO_HOIST=1 timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_no_hoist.json 2>> $O/bench.err
PN2_NO_HOIST=1 timeout -k 10 300 python bench.py --mode rasterized --no-cpu-baseline > $O/bench_rasterized_no_hoist.json 2>> $O/bench.err
PN2_NO_HOIST_GROUP=1 timeout -k 10 300 python bench.py --mode rasterized --no-cpu-baseline > $O/bench_rasterized_no_group_hoist.json 2>> $O/bench.err
PN2_NO_HOIST=1 timeout -k 10 300 python bench.py --dtype bf16 --no-cpu-baseline > $O/bench_bf16_no_hoist.json 2>> $O/bench.err
PN2_NO_PAIR_DGRAD=1 timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_no_pair_dgrad.json 2>> $O/bench.err
PN2_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_gloo2_selfspawn.json 2>> $O/bench.err
echo "bench variants done"
fi
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof /tmp/prof_r /tmp/pmc_fetch /tmp/pmc_write
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o $TAG -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof.err
cp $(find /tmp/prof -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
echo "kernel trace (monolithic) done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r -o $TAG -- python3 $R/bench.py --mode rasterized --no-cpu-baseline > $O/bench_rasterized_under_rocprof.json 2>> $O/rocprof.err
cp $(find /tmp/prof_r -name "*kernel_stats.csv" | head -1) $O/bench_rasterized_kernel_stats.csv
echo "kernel trace (rasterized) done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_fetch -o f -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>> $O/rocprof.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_write -o w -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>> $O/rocprof.err
python3 $R/tools/pmc_traffic.py $(find /tmp/pmc_fetch -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_write -name "*counter_collection.csv" | head -1) $O/pmc_traffic.json > $O/pmc_traffic.txt
echo "pmc done"
cd $R
timeout -k 10 600 python tools/bench_features.py > $O/bench_features.json 2>> $O/bench.err
timeout -k 10 300 python tools/bench_datapath.py > $O/bench_datapath.json 2>> $O/bench.err
timeout -k 10 300 python tools/bench_serialization.py > $O/bench_serialization.json 2>> $O/bench.err
timeout -k 10 300 python tools/bench_ptv3_attention.py > $O/bench_ptv3_attention.json 2>> $O/bench.err
timeout -k 10 300 python tools/bench_ptv3_cpe.py --widths 32,64,128,256 > $O/bench_ptv3_cpe.json 2>> $O/bench.err
timeout -k 10 600 python tools/bench_ptv3_model.py > $O/bench_ptv3_model.json 2>> $O/bench.err
timeout -k 10 600 python tools/bench_ptv3_model.py --precision bf16 >> $O/bench_ptv3_model.json 2>> $O/bench.err
timeout -k 10 600 python tools/bench_ptv3_model.py --train > $O/bench_ptv3_train.json 2>> $O/bench.err
timeout -k 10 300 python tools/bench_ptv3_wgrad.py 2>/dev/null | grep -v amdgpu.ids > $O/bench_ptv3_wgrad.txt || true
timeout -k 10 300 python tools/bench_linear_rows.py 2>/dev/null | grep -v amdgpu.ids > $O/bench_linear_rows.txt || true
timeout -k 10 300 python tools/bench_chain.py --reps 50 > $O/bench_chain.txt 2>> $O/bench.err
if [ -f extracting-tree-morphology-from-point-clouds_amd/build_diag/libpn2hip_gemm_diag.so ]; then   # tools/build_diag_gemm.sh
  timeout -k 10 200 python tools/diag_gemm.py 2>/dev/null | grep -v amdgpu.ids > $O/diag_gemm.txt || true
fi
echo "secondary benches done"
ls -la $O
