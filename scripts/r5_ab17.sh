#!/bin/bash
# round 5: the level-0 analysis with its column stage walked down the rows in registers (default build) against six loads per tmp value
# (variants/liba0_plain.so, -DA0_PLAIN: the rounds-1..4 form, 64 x 16 tiles)
mkdir -p gpurun_out/r5ab17
{
echo "== parity"; timeout 600 python -m pytest tests/test_gpu_wavelet.py tests/test_gpu_denoise.py tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py tests/test_gpu_dninfo.py -x -q -m gpu 2>&1 | grep -E 'passed|failed|rror' | tail -3
for v in default variants/liba0_plain.so default variants/liba0_plain.so; do
  if [ $v = default ]; then unset ARTGPU_LIB; else export ARTGPU_LIB=$PWD/$v; fi
  echo "-- $v"; KEYS=wavelet_analysis0 bash scripts/kernel_times.sh --no-extra-legs
done
unset ARTGPU_LIB
} > gpurun_out/r5ab17/log.txt 2>&1
cat gpurun_out/r5ab17/log.txt
