# end-of-round measurement pass (run on the GPU box from the repo root): kernel trace of config 3, the PMC passes of the AMaZE kernel,
# bench lines for every workload.  Everything lands in gpurun_out/final/.
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final; mkdir -p $O
python bench.py > $O/bench_c3.json 2> $O/bench_c3.err
python bench.py --workload amaze --no-cpu-baseline --sustained-seconds 0 > $O/bench_amaze.json 2>> $O/bench_c3.err
python bench.py --workload rcd --no-cpu-baseline --sustained-seconds 0 > $O/bench_rcd.json 2>> $O/bench_c3.err
python bench.py --workload c4 --no-cpu-baseline --sustained-seconds 0 > $O/bench_c4.json 2>> $O/bench_c3.err
python bench.py --workload c5 --no-cpu-baseline --sustained-seconds 0 --steps 5 > $O/bench_c5.json 2>> $O/bench_c3.err
python bench.py --lanes 3 --no-cpu-baseline --sustained-seconds 0 > $O/bench_c3_lanes3.json 2>> $O/bench_c3.err
python bench.py --tone neutral --no-cpu-baseline --sustained-seconds 0 > $O/bench_c3_neutral.json 2>> $O/bench_c3.err
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c3 -- python $R/bench.py --no-cpu-baseline --sustained-seconds 0 --steps 5 > /dev/null 2>&1) || true
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c4 -- python $R/bench.py --workload c4 --no-cpu-baseline --sustained-seconds 0 --steps 5 > /dev/null 2>&1) || true
find $O/trace_c4 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/c4_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES"; do
  n=$(echo $c | cut -d' ' -f1)
  (cd /tmp && timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$n -- python $R/bench.py --workload amaze --no-cpu-baseline --sustained-seconds 0 --steps 3 --warmup 1 > /dev/null 2>&1) || true
done
python scripts/pmc_summary.py amaze_kernel $O/amaze_final_pmc_summary.json $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_INSTS_VALU
find $O/trace_c3 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/c3_kernel_stats.csv
ls -la $O
