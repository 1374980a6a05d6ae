#!/bin/bash
# round 6 (as round 5): instruction counters of EVERY kernel of a workload (one rocprofv3 --pmc pass per counter) next to its duration:
# which of the "memory-bound" passes are in fact bound by instruction issue?   usage: bash scripts/r5_insts.sh c3|c4|c5 [bench flags]
W=${1:-c3}; shift
O=$PWD/gpurun_out/r6insts_$W; mkdir -p $O; R=$PWD
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/ti_$W; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ti_$W -- python $R/bench.py --workload $W --no-cpu-baseline --sustained-seconds 0 --steps 4 --warmup 1 --opt dn_streams=0 "$@" > /dev/null 2>&1
cp $(find /tmp/ti_$W -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
D=""
for c in SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pi_${W}_$c
  (timeout 200 rocprofv3 --pmc $c --output-format csv -d /tmp/pi_${W}_$c -- python $R/bench.py --workload $W --no-cpu-baseline --sustained-seconds 0 --steps 2 --warmup 1 --opt dn_streams=0 "$@" > /dev/null 2>&1) || echo "failed $c"
  D="$D /tmp/pi_${W}_$c"
done
cd $R
python scripts/pmc_all.py $O/pmc_all.json $D > /dev/null
python3 - $O <<'PY'
import csv, json, re, sys
o = sys.argv[1]
pmc = json.load(open(o + "/pmc_all.json"))
rows = []
for r in csv.DictReader(open(o + "/kernel_stats.csv")):
    k = re.sub(r"\(.*", "", r["Name"].replace("(anonymous namespace)::", "")).replace("artgpu::", "").replace("void ", "")
    if k not in pmc: continue
    g = lambda c: pmc[k].get(c, {}).get("mean_per_launch", 0.0)
    us = float(r["AverageNs"]) / 1e3
    issue = (g("SQ_INSTS_VALU") * 1.19 + g("SQ_INSTS_SALU") * 1.3 + g("SQ_INSTS_LDS") * 1.06) / 1024 / 1e3
    gb = (2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024 / 1e9
    rows.append((float(r["TotalDurationNs"]), f"{k[:44]:44s} calls {int(r['Calls']):4d} avg {us:9.1f} us  issue-model {issue:9.1f} us ({issue/us*100:5.1f} %)  {gb:7.3f} GB = {gb/us*1e3 if us else 0:5.2f} TB/s  valu {g('SQ_INSTS_VALU')/1e6:8.1f}M salu {g('SQ_INSTS_SALU')/1e6:7.1f}M lds {g('SQ_INSTS_LDS')/1e6:6.1f}M vmem {(g('SQ_INSTS_VMEM_RD')+g('SQ_INSTS_VMEM_WR'))/1e6:6.1f}M"))
rows.sort(reverse=True)
open(o + "/table.txt", "w").write("\n".join(r[1] for r in rows) + "\n")
print("\n".join(r[1] for r in rows[:30]))
PY
