#!/bin/bash
# round 6: two frames in flight -- the demosaic's share of the CUs again, now that the stream kernel takes 3.3 instead of 4.4 ms
mkdir -p gpurun_out/r6/lanes
run() { timeout 120 python bench.py --workload c3 --lanes 2 --opt amaze_grid=$1 --no-cpu-baseline --sustained-seconds 0 --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('grid $1', d['value'], d['ms_per_step'])"; }
{ for rep in 1 2; do for G in 256 192 176 160 144 128 112; do run $G; done; done; } > gpurun_out/r6/lanes/log.txt 2>&1
cat gpurun_out/r6/lanes/log.txt
