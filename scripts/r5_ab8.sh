#!/bin/bash
# round 5: detail_blocks_kernel with -blur^2 / factor as a multiplication by the factor's reciprocal (default build) against the correctly rounded
# division and one row per iteration (variants/libdet_old.so = scripts/mkvariant.sh det_old detail.hip "-DDETAIL_EXACT_DIV -DDETAIL_ROW_AT_A_TIME") and against the
# groups of eight with the division kept (libdet_g8div.so: -DDETAIL_EXACT_DIV, the bits of det_old); other bits in the tolerance stage: checksums differ
mkdir -p gpurun_out/r5ab8
{
echo "== checksums"
python scripts/dn_checksum.py 2>/dev/null; ARTGPU_LIB=$PWD/variants/libdet_old.so python scripts/dn_checksum.py 2>/dev/null; ARTGPU_LIB=$PWD/variants/libdet_g8div.so python scripts/dn_checksum.py 2>/dev/null
echo "== kernel times"
for v in default variants/libdet_old.so variants/libdet_g8div.so; do
  if [ $v = default ]; then unset ARTGPU_LIB; else export ARTGPU_LIB=$PWD/$v; fi
  echo "-- $v"; KEYS=detail_blocks bash scripts/kernel_times.sh; KEYS=detail_blocks bash scripts/kernel_times.sh --workload c5 --steps 4
done
unset ARTGPU_LIB
echo "== c3"; bash scripts/ab_libs.sh 3 c3 default variants/libdet_old.so variants/libdet_g8div.so
echo "== c5"; bash scripts/ab_libs.sh 2 c5 default variants/libdet_old.so
echo "== parity"; timeout 600 python -m pytest tests/test_gpu_denoise.py tests/test_gpu_fullsize.py tests/test_gpu_cli.py -x -q -m gpu 2>&1 | grep -E 'passed|failed|rror' | tail -2
} > gpurun_out/r5ab8/log.txt 2>&1
cat gpurun_out/r5ab8/log.txt
