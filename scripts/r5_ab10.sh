#!/bin/bash
# round 5: detail_gather_kernel with its nine candidate terms as straight-line code, loads first (default build) against the nested loops with `continue`
# (variants/libgather_loops.so = scripts/mkvariant.sh gather_loops detail.hip "-DDETAIL_GATHER_LOOPS"): same bits (checksums), other speed
mkdir -p gpurun_out/r5ab10
{
echo "== checksums (have to be equal)"
python scripts/dn_checksum.py 2>/dev/null; ARTGPU_LIB=$PWD/variants/libgather_loops.so python scripts/dn_checksum.py 2>/dev/null
echo "== kernel times"
for v in default variants/libgather_loops.so; do
  if [ $v = default ]; then unset ARTGPU_LIB; else export ARTGPU_LIB=$PWD/$v; fi
  echo "-- $v"; KEYS=detail_gather bash scripts/kernel_times.sh
done
unset ARTGPU_LIB
echo "== c3"; bash scripts/ab_libs.sh 3 c3 default variants/libgather_loops.so
echo "== c5"; bash scripts/ab_libs.sh 2 c5 default variants/libgather_loops.so
echo "== parity"; timeout 600 python -m pytest tests/test_gpu_denoise.py tests/test_gpu_fullsize.py tests/test_gpu_cli.py -x -q -m gpu 2>&1 | grep -E 'passed|failed|rror' | tail -2
} > gpurun_out/r5ab10/log.txt 2>&1
cat gpurun_out/r5ab10/log.txt
