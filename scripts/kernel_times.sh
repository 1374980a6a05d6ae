cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/tr
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --sustained-seconds 0 --steps 10 --warmup 2 --opt dn_streams=0 "$@" > /dev/null 2>&1
f=$(find /tmp/tr -name "*kernel_stats.csv" | head -1)
python3 -c "
import csv,sys
keys=sys.argv[1].split(',')
for r in csv.DictReader(open('$f')):
    n=r['Name']
    if any(k in n for k in keys): print(n[:70], r['Calls'], round(float(r['AverageNs'])/1e3,1))" "${KEYS:-rgb2yuv,yuv2rgb,tone_std}"
