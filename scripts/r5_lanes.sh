#!/bin/bash
# round 5: frames in flight -- can the other frame's LDS-free streaming kernels live on the CUs of an AMaZE stream kernel built with 87 registers
# (-fno-slp-vectorize: 4 x 88 of a SIMD's 512 registers instead of 4 x 128)?
mkdir -p gpurun_out/r5lanes
run() { timeout 120 python bench.py --workload c3 --lanes 2 --opt amaze_grid=$1 --opt lut_lds=$2 --no-cpu-baseline --sustained-seconds 0 --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$3 grid $1 lut_lds $2', d['value'], d['ms_per_step'])"; }
{
unset ARTGPU_LIB; run 160 1 default; run 256 1 default
export ARTGPU_LIB=$PWD/variants/libamz_noslp.so
for G in 256 224 192 160; do for LL in 1 0; do run $G $LL noslp87; done; done
} > gpurun_out/r5lanes/log3.txt 2>&1
cat gpurun_out/r5lanes/log3.txt
