#!/bin/bash
# gpurun_out/r4 (scripts/r4_measure.sh) -> profiles/r4
set -e
cd "$(dirname "$0")/.."
S=gpurun_out/r4; D=profiles/r4
for f in $S/bench_*.json $S/*_kernel_stats.csv $S/c3_per_frame_table.md $S/c3_pmc_all.json; do
  case $f in *bench_c3_traced.json) continue;; esac
  cp $f $D/
done
for k in amaze_stream rcd_stream nlm_group xtrans_tiles shrink_blur; do cp $S/${k}_kernel_pmc_summary.json $D/${k}_pmc_summary.json; done
