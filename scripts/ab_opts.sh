#!/bin/bash
# same-box A/B of option sets: bash scripts/ab_opts.sh REPS WORKLOAD "opts A" "opts B" ...   (each opts string: "name=value name=value")
R=$1; W=$2; shift; shift
for i in $(seq $R); do
  for o in "$@"; do
    a=""; for kv in $o; do a="$a --opt $kv"; done
    timeout 120 python bench.py --workload $W --no-cpu-baseline --sustained-seconds 0 --steps 10 --warmup 3 $a 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$o]', d['ms_per_step'], d['config']['stage_ms'].get('denoise'))"
  done
done
