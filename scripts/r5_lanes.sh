#!/bin/bash
# round 5: frames in flight with the AMaZE stream kernel on fewer than all CUs (option amaze_grid)
mkdir -p gpurun_out/r5lanes
{
for L in 2; do for G in 0 160 144; do for LL in 1 0; do
  timeout 120 python bench.py --workload c3 --lanes $L --opt amaze_grid=$G --opt lut_lds=$LL --no-cpu-baseline --sustained-seconds 0 --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('lanes $L grid $G lut_lds $LL', d['value'], d['ms_per_step'])"
done; done; done
} > gpurun_out/r5lanes/log2.txt 2>&1
cat gpurun_out/r5lanes/log2.txt
