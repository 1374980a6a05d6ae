#!/bin/bash
# round 5, call 2: detail_blocks_kernel with the XCD-aware block order (default build) against the raster order (variants/libdetail_old.so)
mkdir -p gpurun_out/r5ab2
R=$PWD
{
echo "== c3 A/B"; bash scripts/ab_libs.sh 3 c3 default variants/libdetail_old.so
echo "== c5 A/B"; bash scripts/ab_libs.sh 2 c5 default variants/libdetail_old.so
echo "== parity"; timeout 300 python -m pytest tests/test_gpu_denoise.py -x -q -m gpu -k "detail or fused_equals" 2>&1 | tail -2
for v in new old; do
  if [ $v = old ]; then export ARTGPU_LIB=$R/variants/libdetail_old.so; else unset ARTGPU_LIB; fi
  cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/tr_$v /tmp/pf_$v /tmp/pw_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$v -- python $R/bench.py --no-cpu-baseline --sustained-seconds 0 --steps 6 --warmup 2 --opt dn_streams=0 > /dev/null 2>&1
  f=$(find /tmp/tr_$v -name "*kernel_stats.csv" | head -1)
  echo "== kernel stats ($v)"; grep -E 'detail|Name' $f | cut -d, -f1-4
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf_$v -- python $R/bench.py --no-cpu-baseline --sustained-seconds 0 --steps 3 --warmup 1 --opt dn_streams=0 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pw_$v -- python $R/bench.py --no-cpu-baseline --sustained-seconds 0 --steps 3 --warmup 1 --opt dn_streams=0 > /dev/null 2>&1
  cd $R
  echo "== counters ($v)"; python scripts/pmc_summary.py detail_blocks gpurun_out/r5ab2/detail_blocks_$v.json /tmp/pf_$v /tmp/pw_$v
  python scripts/pmc_summary.py detail_gather gpurun_out/r5ab2/detail_gather_$v.json /tmp/pf_$v /tmp/pw_$v
done
} > gpurun_out/r5ab2/log.txt 2>&1
tail -50 gpurun_out/r5ab2/log.txt
