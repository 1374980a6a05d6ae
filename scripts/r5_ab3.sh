#!/bin/bash
# round 5, call 3 (timing only, results differ): what the IEEE fp32 divisions cost each kernel -- builds with the approximate division
mkdir -p gpurun_out/r5ab3
{
echo "== amaze"; bash scripts/ab_libs.sh 2 amaze default variants/libamz_fastdiv.so
echo "== c3"; bash scripts/ab_libs.sh 2 c3 default variants/libsb_fastdiv.so variants/libdet_fastdiv.so
echo "== c4"; bash scripts/ab_libs.sh 2 c4 default variants/libnlm_fastdiv.so
echo "== c5"; bash scripts/ab_libs.sh 2 c5 default variants/libxt_fastdiv.so
} > gpurun_out/r5ab3/log.txt 2>&1
cat gpurun_out/r5ab3/log.txt
