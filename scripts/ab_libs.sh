#!/bin/bash
# same-box A/B of library builds: bash scripts/ab_libs.sh REPS WORKLOAD default variants/libA.so variants/libB.so ...
R=$1; W=$2; shift; shift
for i in $(seq $R); do
  for v in "$@"; do
    if [ $v = default ]; then unset ARTGPU_LIB; else export ARTGPU_LIB=$PWD/$v; fi
    timeout 120 python bench.py --workload $W --no-cpu-baseline --sustained-seconds 0 --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$v]', d['ms_per_step'], list(d['config']['stage_ms'].values()))"
  done
done
