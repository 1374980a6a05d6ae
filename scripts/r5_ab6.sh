#!/bin/bash
mkdir -p gpurun_out/r5ab6
{
echo "== parity"; timeout 600 python -m pytest tests/test_gpu_denoise.py tests/test_gpu_pipeline.py -x -q -m gpu 2>&1 | tail -3
echo "== kernel times"; KEYS=chroma_map bash scripts/kernel_times.sh
KEYS=chroma_map bash scripts/kernel_times.sh --opt lut_lds=0
echo "== c3"; for i in 1 2 3; do for o in 1 0; do timeout 120 python bench.py --workload c3 --no-cpu-baseline --sustained-seconds 0 --steps 10 --warmup 3 --opt lut_lds=$o 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('lut_lds $o', d['ms_per_step'], list(d['config']['stage_ms'].values()))"; done; done
} > gpurun_out/r5ab6/log.txt 2>&1
cat gpurun_out/r5ab6/log.txt
