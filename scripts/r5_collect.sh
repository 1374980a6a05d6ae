#!/bin/bash
# gpurun_out/r5 (scripts/r5_measure.sh) -> profiles/r5
set -e
cd "$(dirname "$0")/.."
S=gpurun_out/r5; D=profiles/r5; mkdir -p $D
for f in $S/bench_*.json $S/*_kernel_stats.csv $S/c3_per_frame_table.md $S/c3_pmc_all.json $S/*_instruction_table.txt; do
  case $f in *bench_c3_traced.json) continue;; esac
  [ -e "$f" ] && cp $f $D/
done
for k in amaze_stream rcd_stream nlm_group xtrans_tiles shrink_blur; do [ -e $S/${k}_kernel_pmc_summary.json ] && cp $S/${k}_kernel_pmc_summary.json $D/${k}_pmc_summary.json; done
ls $D
