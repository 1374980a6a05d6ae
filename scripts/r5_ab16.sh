#!/bin/bash
# round 5: the undecimated Haar kernels' grid-stride loop against the number of workgroups (default: at most 16384; variants/libwv_capN.so: N;
# one pixel per thread, 43.7 k workgroups at 45 MP, measured before: analysis the same, synthesis 55.3 against 48.9 us)
mkdir -p gpurun_out/r5ab16
{
for v in default variants/libwv_cap2048.so variants/libwv_cap4096.so variants/libwv_cap8192.so default; do
  if [ $v = default ]; then unset ARTGPU_LIB; else export ARTGPU_LIB=$PWD/$v; fi
  echo "-- $v"; KEYS=wavelet_haar bash scripts/kernel_times.sh --no-extra-legs
done
unset ARTGPU_LIB
} > gpurun_out/r5ab16/log.txt 2>&1
cat gpurun_out/r5ab16/log.txt
