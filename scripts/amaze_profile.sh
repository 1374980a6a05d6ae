# AMaZE profile pass (run on the GPU box from the repo root): kernel trace + the PMC passes of the demosaic-only workload.
# usage: bash scripts/amaze_profile.sh TAG [pmc]     -> gpurun_out/TAG/
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$1; mkdir -p $O
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/bench.py --workload amaze --no-cpu-baseline --sustained-seconds 0 --steps 5 --warmup 2 > $O/bench_traced.json 2> $O/err.txt) || true
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/amaze_kernel_stats.csv
python scripts/kstat.py $O/trace amaze border
if [ "$2" = "pmc" ]; then
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_LDS"; do
  n=$(echo $c | cut -d' ' -f1)
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$n -- python $R/bench.py --workload amaze --no-cpu-baseline --sustained-seconds 0 --steps 3 --warmup 1 > /dev/null 2>&1) || true
done
python scripts/pmc_summary.py amaze_stream_kernel $O/amaze_stream_pmc_summary.json $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_INSTS_VALU $O/pmc_SQ_INSTS_LDS $O/pmc_SQ_WAIT_INST_ANY
python scripts/pmc_summary.py amaze_kernel $O/amaze_arena_pmc_summary.json $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_INSTS_VALU
cat $O/amaze_stream_pmc_summary.json
fi
rm -rf $O/trace/*/*.db 2>/dev/null
