#!/bin/bash
# round 5: is the AMaZE stream kernel (63 KB of code, sixteen waves per CU on different stages) short of instruction cache?
O=$PWD/gpurun_out/r5ic; mkdir -p $O; R=$PWD
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E 'icache|ifetch|inst_cache|SQC_' | head -60 > $O/counters.txt
for c in SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAIT_INST_ANY SQ_INST_LEVEL_LDS SQ_IFETCH_LEVEL; do
  (timeout 120 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python $R/bench.py --workload ${W:-amaze} --no-cpu-baseline --sustained-seconds 0 --steps 3 --warmup 1 > /dev/null 2>&1) || echo "failed $c" >> $O/counters.txt
done
cd $R
python scripts/pmc_summary.py ${K:-amaze_stream_kernel} $O/summary.json $O/pmc_* ; rm -rf $O/pmc_*
cat $O/counters.txt | head -40
