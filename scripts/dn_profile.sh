#!/bin/bash
# kernel trace + FETCH_SIZE / WRITE_SIZE passes of one bench workload for EVERY kernel -> gpurun_out/TAG/{W}_kernel_stats.csv, {W}_pmc_all.json
# usage: bash scripts/dn_profile.sh TAG [WORKLOAD] [extra bench args]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$1; mkdir -p $O
W=${2:-c3}; shift; shift
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/bench.py --workload $W --no-cpu-baseline --sustained-seconds 0 --steps 10 --warmup 2 --opt dn_streams=0 "$@" > $O/bench_${W}_traced.json 2> $O/err.txt) || true
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${W}_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python $R/bench.py --workload $W --no-cpu-baseline --sustained-seconds 0 --steps 3 --warmup 1 --opt dn_streams=0 "$@" > /dev/null 2>&1) || true
done
python $R/scripts/pmc_all.py $O/${W}_pmc_all.json $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
rm -rf $O/trace $O/pmc_*
