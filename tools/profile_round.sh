# GPU box: bench lines + rocprofv3 kernel stats + PMC traffic for profiles/ (run through gpurun from the repo root)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/round
mkdir -p $O
cd $R
if [ -z "$PN2_SKIP_BENCH" ]; then
timeout -k 10 300 python bench.py > $O/bench_full.json 2> $O/bench.err
echo "bench full done"
timeout -k 10 300 python bench.py --trees 8 --points 65536 --no-cpu-baseline > $O/bench_cfg3_8x65536.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --depth 5 --no-cpu-baseline > $O/bench_depth5.json 2>> $O/bench.err
echo "bench variants done"
fi
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof /tmp/pmc_fetch /tmp/pmc_write
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r01 -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof.err
cp $(find /tmp/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
echo "kernel trace done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_fetch -o f -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>> $O/rocprof.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_write -o w -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>> $O/rocprof.err
python3 $R/tools/pmc_traffic.py $(find /tmp/pmc_fetch -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_write -name "*counter_collection.csv" | head -1) $O/pmc_traffic.json > $O/pmc_traffic.txt
echo "pmc done"
cd $R
timeout -k 10 300 python tools/bench_rasterized.py > $O/bench_rasterized.json 2>> $O/bench.err
timeout -k 10 600 python tools/bench_features.py > $O/bench_features.json 2>> $O/bench.err
echo "secondary benches done"
ls -la $O
