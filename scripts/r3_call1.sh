#!/bin/bash
# round 3, GPU call 1: correctness of the forked denoise chain + A/B of dn_streams
mkdir -p gpurun_out/r3c1
python -m pytest tests/test_gpu_denoise.py tests/test_gpu_pipeline.py tests/test_gpu_pixelops.py tests/test_gpu_tonecurve.py tests/test_gpu_wavelet.py tests/test_gpu_lab.py -x -q -m gpu 2>&1 | tail -5
for i in 1 2; do
for s in 0 1; do
python bench.py --no-cpu-baseline --sustained-seconds 0 --steps 10 --warmup 3 --opt dn_streams=$s 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('dn_streams=$s', d['ms_per_step'], d['config']['stage_ms'])"
done
done
python bench.py --no-cpu-baseline --sustained-seconds 0 --steps 10 --warmup 3 --workload c4 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('c4', d['ms_per_step'], d['config']['stage_ms'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r3c1/trace -- python /root/repo/bench.py --no-cpu-baseline --sustained-seconds 0 --steps 10 --warmup 3 --opt dn_streams=0 > /dev/null 2>&1
find /root/repo/gpurun_out/r3c1/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} /root/repo/gpurun_out/r3c1/c3_serial_kernel_stats.csv
rm -rf /root/repo/gpurun_out/r3c1/trace
