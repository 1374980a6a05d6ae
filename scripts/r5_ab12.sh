#!/bin/bash
# round 5: the box-blur pair of the guided filter / the three-kernel shrink passes -- hblur_kernel without workgroup barriers (one wave: LDS order is the
# hardware's) and with the next chunk's columns prefetched, vblur_combine_kernel<plain> with 24 rows per batch -- against the previous build (variants/libblur_head.so)
mkdir -p gpurun_out/r5ab12
{
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_denoise.py tests/test_gpu_guided.py tests/test_gpu_hsl.py tests/test_gpu_logenc.py tests/test_gpu_pipeline.py tests/test_gpu_cli.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | grep -E 'passed|failed|rror' | tail -2
echo "== kernel times (c4)"
for v in default variants/libblur_head.so; do
  if [ $v = default ]; then unset ARTGPU_LIB; else export ARTGPU_LIB=$PWD/$v; fi
  echo "-- $v"; KEYS=hblur,vblur bash scripts/kernel_times.sh --workload c4 --steps 4
  KEYS=hblur,vblur,shrink_sf bash scripts/kernel_times.sh --workload c3 --steps 4 --opt dn_fused=0
done
unset ARTGPU_LIB
echo "== c4"; bash scripts/ab_libs.sh 3 c4 default variants/libblur_head.so
} > gpurun_out/r5ab12/log.txt 2>&1
cat gpurun_out/r5ab12/log.txt
