# GPU box: which non-library kernels does a step still launch? (rocprofv3 kernel stats of bench.py)
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/glue
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/glue -o g -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/bench_glue.json 2> /dev/null
cp $(find /tmp/glue -name "*kernel_stats.csv" | head -1) $R/gpurun_out/glue_stats.csv
