#!/bin/bash
# gpurun_out/r6m (scripts/r6_measure.sh) -> profiles/r6
set -e
cd "$(dirname "$0")/.."
S=gpurun_out/r6m; D=profiles/r6; mkdir -p $D
for f in $S/bench_*.json $S/*_kernel_stats.csv $S/c*_per_frame_table.md $S/c*_pmc_all.json $S/*_instruction_table.txt; do
  case $f in *_traced.json) continue;; esac
  [ -e "$f" ] && cp $f $D/
done
for k in amaze_stream rcd_stream nlm_group xtrans_tiles shrink_blur; do [ -e $S/${k}_kernel_pmc_summary.json ] && cp $S/${k}_kernel_pmc_summary.json $D/${k}_pmc_summary.json; done
ls $D
