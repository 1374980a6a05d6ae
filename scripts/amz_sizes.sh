#!/bin/bash
# AMaZE alone at sensor sizes whose width is / is not 32 + a multiple of 128, with and without the early arena launch
for o in 1 0; do
for s in "8192 5464" "8256 5504" "4000 3000" "6000 4000" "9504 6336"; do set -- $s
timeout 60 python bench.py --workload amaze --width $1 --height $2 --no-cpu-baseline --sustained-seconds 0 --steps 10 --warmup 3 --opt amaze_overlap=$o 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('overlap=$o', '$1 x $2', d['ms_per_step'], 'ms', d['value'], 'MP/s')"
done; done
