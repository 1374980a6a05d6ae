#!/bin/bash
# Round-6 measurement pass (run on the GPU box from the repo root) -> gpurun_out/r6m/: bench lines of every workload, the config-3 kernel
# trace with FETCH_SIZE / WRITE_SIZE of EVERY kernel (per-frame traffic table; --no-extra-legs: STD frames of the fused tool only), counter
# summaries of the dominant kernels (stamped with the digest of their sources: bench.py's `traffic_stale`), per-kernel instruction tables of
# configs 3 - 5 (scripts/r6_insts.sh: issue-time model against duration).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r6m; mkdir -p $O
B="timeout 300 python $R/bench.py"
Q="--no-cpu-baseline --sustained-seconds 0"
$B > $O/bench_c3_n1.json 2> $O/err.txt
$B --workload amaze $Q > $O/bench_amaze_n1.json 2>> $O/err.txt
$B --workload amaze --width 8256 --height 5504 $Q > $O/bench_amaze_8256x5504.json 2>> $O/err.txt
$B --workload amaze --width 4000 --height 3000 $Q > $O/bench_amaze_4000x3000.json 2>> $O/err.txt
$B --workload amaze --width 6000 --height 4000 $Q > $O/bench_amaze_6000x4000.json 2>> $O/err.txt
$B --workload rcd $Q > $O/bench_rcd_n1.json 2>> $O/err.txt
$B --workload c4 $Q > $O/bench_c4_n1.json 2>> $O/err.txt
$B --workload c5 $Q --steps 5 > $O/bench_c5_n1.json 2>> $O/err.txt
$B --workload c5 --xtrans-passes 1 $Q --steps 5 > $O/bench_c5_one_pass.json 2>> $O/err.txt
$B --lanes 2 $Q > $O/bench_c3_lanes2.json 2>> $O/err.txt
$B --lanes 2 $Q --opt amaze_grid=256 > $O/bench_c3_lanes2_all_cus.json 2>> $O/err.txt
$B $Q --separate-stages > $O/bench_c3_separate_stages.json 2>> $O/err.txt
if [ "$1" = "lines" ]; then ls $O; exit 0; fi
# config 3, one kernel after the other (dn_streams=0: per-kernel durations and counters that do not overlap)
bash $R/scripts/dn_profile.sh r6m c3 --no-extra-legs > /dev/null 2>&1
prof() {  # workload kernel tag [bench args]
  W=$1; K=$2; T=$3; shift; shift; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$T -- python $R/bench.py --workload $W $Q --no-extra-legs --steps 4 --warmup 1 "$@" > /dev/null 2>&1) || true
  find $O/trace_$T -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${T}_kernel_stats.csv
  for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_LDS"; do
    n=$(echo $c | cut -d' ' -f1)
    (cd /tmp && timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${T}_$n -- python $R/bench.py --workload $W $Q --no-extra-legs --steps 2 --warmup 1 "$@" > /dev/null 2>&1) || true
  done
  python $R/scripts/pmc_summary.py $K $O/${K}_pmc_summary.json $O/pmc_${T}_FETCH_SIZE $O/pmc_${T}_WRITE_SIZE $O/pmc_${T}_SQ_INSTS_VALU $O/pmc_${T}_SQ_INSTS_LDS $O/pmc_${T}_SQ_WAIT_INST_ANY > /dev/null
  rm -rf $O/trace_$T $O/pmc_${T}_*
}
prof amaze amaze_stream_kernel amaze
prof c3 shrink_blur_kernel fused
prof rcd rcd_stream_kernel rcd
prof c4 nlm_group_kernel c4
prof c5 xtrans_tiles_kernel c5
nfr=$(python3 -c "
import csv
for r in csv.DictReader(open('$O/c3_kernel_stats.csv')):
    if 'amaze_stream' in r['Name']: print(r['Calls'])")
python $R/scripts/dn_table.py $O/c3_kernel_stats.csv $O/c3_pmc_all.json $nfr $O/c3_per_frame_table.md > /dev/null
# the same table (every kernel: time, bytes moved, the minimum it could move) for configs 4 and 5
for W in c4 c5; do
  bash $R/scripts/dn_profile.sh r6m $W --no-extra-legs --steps 4 > /dev/null 2>&1
  K=amaze_stream; PX="--px 44652904 --raw 44761088"; if [ $W = c5 ]; then K=xtrans_tiles; PX="--px 101471748 --raw 101756928"; fi
  n=$(python3 -c "
import csv
for r in csv.DictReader(open('$O/${W}_kernel_stats.csv')):
    if '$K' in r['Name']: print(r['Calls'])")
  python $R/scripts/dn_table.py $O/${W}_kernel_stats.csv $O/${W}_pmc_all.json $n $O/${W}_per_frame_table.md $PX > /dev/null
done
for W in c3 c4 c5; do bash $R/scripts/r6_insts.sh $W --no-extra-legs > /dev/null 2>&1; cp $R/gpurun_out/r6insts_$W/table.txt $O/${W}_instruction_table.txt; done
ls $O
