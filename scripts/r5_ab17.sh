#!/bin/bash
# round 5: output tile of the level-0 analysis (default 64 x 16; variants/liba0_WxH.so)
mkdir -p gpurun_out/r5ab17
{
for v in default variants/liba0_64x4.so variants/liba0_32x8.so variants/liba0_128x8.so variants/liba0_128x4.so variants/liba0_64x8.so default; do
  if [ $v = default ]; then unset ARTGPU_LIB; else export ARTGPU_LIB=$PWD/$v; fi
  echo "-- $v"; KEYS=wavelet_analysis0 bash scripts/kernel_times.sh --no-extra-legs
done
unset ARTGPU_LIB
} > gpurun_out/r5ab17/log.txt 2>&1
cat gpurun_out/r5ab17/log.txt
