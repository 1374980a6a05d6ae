# Kernel trace + PMC passes of one bench workload (run on the GPU box from the repo root).
# usage: bash scripts/kernel_profile.sh TAG WORKLOAD KERNEL_SUBSTRING      -> gpurun_out/TAG/   (e.g. r2_rcd rcd rcd_stream_kernel)
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$1; mkdir -p $O
W=$2; K=$3
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/bench.py --workload $W --no-cpu-baseline --sustained-seconds 0 --steps 4 --warmup 1 > $O/bench_${W}_traced.json 2> $O/err.txt) || true
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${W}_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_LDS"; do
  n=$(echo $c | cut -d' ' -f1)
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$n -- python $R/bench.py --workload $W --no-cpu-baseline --sustained-seconds 0 --steps 2 --warmup 1 > /dev/null 2>&1) || true
done
python scripts/pmc_summary.py $K $O/${W}_pmc_summary.json $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_INSTS_VALU $O/pmc_SQ_INSTS_LDS $O/pmc_SQ_WAIT_INST_ANY
rm -rf $O/trace $O/pmc_*
