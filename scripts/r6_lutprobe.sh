for v in variants/liblut_full.so variants/liblut_half.so; do
export ARTGPU_LIB=$PWD/$v
python bench.py --no-cpu-baseline --sustained-seconds 0 --steps 10 --warmup 3 --separate-stages --no-extra-legs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['config']['stage_ms'])"
done
