cd $GRAFT_REPO_ROOT
run() { # workload lib
  if [ "$2" = default ]; then unset ARTGPU_LIB; else export ARTGPU_LIB=$PWD/$2; fi
  python bench.py --workload $1 --no-cpu-baseline --sustained-seconds 0 --steps ${3:-10} --warmup 3 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$1 $2]', d['ms_per_step'], d['roofline']['kernel_ms'])"
}
for i in 1 2; do
run amaze default; run amaze variants/libamaze_clause.so; run amaze variants/libamaze_default.so
run rcd default; run rcd variants/librcd_stream_ilp.so; run rcd variants/librcd_stream_clause.so
run c5 default 4; run c5 variants/libxtrans_ilp.so 4; run c5 variants/libxtrans_clause.so 4
run c4 default 6; run c4 variants/libnlm_sweep_ilp.so 6; run c4 variants/libnlm_sweep_clause.so 6
done
