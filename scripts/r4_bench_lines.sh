#!/bin/bash
# The bench lines of scripts/r4_measure.sh alone (run it once more AFTER scripts/r4_collect.sh when a kernel source has changed: the lines compare
# the kernel's digest with the counter summaries under profiles/r4, which the first pass has only just produced)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r4; mkdir -p $O
B="timeout 300 python $R/bench.py"
Q="--no-cpu-baseline --sustained-seconds 0"
$B > $O/bench_c3_n1.json 2> $O/err.txt
$B --workload amaze $Q > $O/bench_amaze_n1.json 2>> $O/err.txt
$B --workload rcd $Q > $O/bench_rcd_n1.json 2>> $O/err.txt
$B --workload c4 $Q > $O/bench_c4_n1.json 2>> $O/err.txt
$B --workload c5 $Q --steps 5 > $O/bench_c5_n1.json 2>> $O/err.txt
$B --lanes 2 $Q > $O/bench_c3_lanes2.json 2>> $O/err.txt
$B $Q --opt dn_fused=0 > $O/bench_c3_three_kernel_shrink.json 2>> $O/err.txt
