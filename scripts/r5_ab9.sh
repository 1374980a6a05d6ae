#!/bin/bash
# round 5: the faster detail_blocks_kernel (groups of eight rows + reciprocal factor: 0.91 -> 0.73 ms alone) inside the frame, with the DCT stage on its side stream
# beside the chroma reconstructions (dn_streams 1) and behind them (dn_streams 0); variants/libdet_head.so = the build before the change
mkdir -p gpurun_out/r5ab8
run() { timeout 200 python bench.py --workload $3 --no-cpu-baseline --sustained-seconds 0 --steps ${4:-10} --warmup 3 --opt dn_streams=$2 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$1] $3 dn_streams $2', d['ms_per_step'], list(d['config']['stage_ms'].values()))"; }
{
for w in c4 c5; do for i in 1 2; do for s in 1 0; do
  unset ARTGPU_LIB; run default $s $w 5
  export ARTGPU_LIB=$PWD/variants/libdet_head.so; run head $s $w 5
done; done; done
unset ARTGPU_LIB
for s in 1 0; do timeout 200 python bench.py --workload c3 --lanes 2 --no-cpu-baseline --sustained-seconds 0 --steps 10 --warmup 3 --opt dn_streams=$s 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('lanes 2 dn_streams $s', d['value'], d['ms_per_step'])"; done
} > gpurun_out/r5ab8/log4.txt 2>&1
cat gpurun_out/r5ab8/log4.txt
