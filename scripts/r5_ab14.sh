#!/bin/bash
# round 5: shrink_blur_kernel with the update's coefficients kept in registers (one ds_bpermute per value, a three-step queue; default build)
# against reading them a second time (variants/libsb_head.so = the previous commit's shrinkblur.hip)
mkdir -p gpurun_out/r5ab14
{
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_denoise.py tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | grep -E 'passed|failed|rror' | tail -2
echo "== kernel times"
for v in default variants/libsb_head.so; do
  if [ $v = default ]; then unset ARTGPU_LIB; else export ARTGPU_LIB=$PWD/$v; fi
  echo "-- $v"; KEYS=shrink_blur bash scripts/kernel_times.sh; KEYS=shrink_blur bash scripts/kernel_times.sh --workload c5 --steps 4
done
unset ARTGPU_LIB
echo "== c3"; bash scripts/ab_libs.sh 3 c3 default variants/libsb_head.so
} > gpurun_out/r5ab14/log.txt 2>&1
cat gpurun_out/r5ab14/log.txt
