#!/bin/bash
mkdir -p gpurun_out/r3c2
python -m pytest tests/test_gpu_denoise.py tests/test_gpu_pipeline.py tests/test_gpu_pixelops.py tests/test_gpu_tonecurve.py tests/test_gpu_wavelet.py tests/test_gpu_lab.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r3c2/pytest.log 2>&1
tail -3 gpurun_out/r3c2/pytest.log
for i in 1 2 3; do
for s in 0 1; do
python bench.py --no-cpu-baseline --sustained-seconds 0 --steps 10 --warmup 3 --opt dn_streams=$s 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('dn_streams=$s', d['ms_per_step'], d['config']['stage_ms'])"
done
done
