cd $GRAFT_REPO_ROOT
cp art_amd/libartgpu.so /tmp/lib_base.so
for v in "128 32" "128 64" "256 32" "256 16" "64 64" "128 16"; do
  set -- $v
  (cd art_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DS0_TW_=$1 -DS0_TH_=$2 -c wavelet.hip -o wavelet.o 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libartgpu.so *.o) || { echo "build failed $v"; continue; }
  export TMPDIR=/tmp; rm -rf /tmp/tr; (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --sustained-seconds 0 --steps 6 --warmup 2 --opt dn_streams=0 > /dev/null 2>&1)
  f=$(find /tmp/tr -name "*kernel_stats.csv" | head -1)
  python3 -c "
import csv
for r in csv.DictReader(open('$f')):
    if 'synthesis0' in r['Name']: print('tile $1 x $2:', round(float(r['AverageNs'])/1e3,1), 'us')"
done
cp /tmp/lib_base.so art_amd/libartgpu.so
