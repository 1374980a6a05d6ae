#!/bin/bash
# round 5, call 1: the SLP vectoriser per translation unit (packed fp32 instructions cost 1.8 plain ones: scripts/ubench/pk_rate.hip)
mkdir -p gpurun_out/r5ab1
{
echo "== amaze"; bash scripts/ab_libs.sh 3 amaze default variants/libamz_noslp.so
echo "== rcd"; bash scripts/ab_libs.sh 3 rcd default variants/librcd_noslp.so
echo "== c3"; bash scripts/ab_libs.sh 3 c3 default variants/libsb_noslp.so variants/libdn_noslp.so variants/libwv_noslp.so variants/libpx_noslp.so
echo "== c5"; bash scripts/ab_libs.sh 2 c5 default variants/libxt_noslp.so
echo "== parity of the variants"
ARTGPU_LIB=$PWD/variants/libamz_noslp.so timeout 300 python -m pytest tests/test_gpu_demosaic.py -x -q -m gpu 2>&1 | tail -2
ARTGPU_LIB=$PWD/variants/librcd_noslp.so timeout 300 python -m pytest tests/test_gpu_demosaic.py -x -q -m gpu -k rcd 2>&1 | tail -2
} > gpurun_out/r5ab1/log.txt 2>&1
tail -60 gpurun_out/r5ab1/log.txt
