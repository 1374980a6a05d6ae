# Kernel trace (no counters) of one bench workload: bash scripts/trace_only.sh TAG WORKLOAD -> gpurun_out/TAG/WORKLOAD_kernel_stats.csv + bench_WORKLOAD_traced.json
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$1; mkdir -p $O
W=$2
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/bench.py --workload $W --no-cpu-baseline --sustained-seconds 0 --steps 4 --warmup 1 > $O/bench_${W}_traced.json 2> $O/err.txt) || true
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${W}_kernel_stats.csv
rm -rf $O/trace
