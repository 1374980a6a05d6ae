# same-box A/B of two builds of the library: bash scripts/ab_bench.sh variants/libA.so [workload] -- alternates A (the given .so) and B (the in-tree one)
A=$1; W=${2:-c3}
for i in 1 2 3; do
  for v in A B; do
    if [ $v = A ]; then export ARTGPU_LIB=$PWD/$A; else unset ARTGPU_LIB; fi
    python bench.py --workload $W --no-cpu-baseline --sustained-seconds 0 --steps 8 --warmup 2 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['config']['stage_ms'].get('denoise'))"
  done
done
