cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$n -- python $R/bench.py --no-cpu-baseline --sustained-seconds 0 --steps 2 --warmup 1 --opt dn_streams=0 > /dev/null 2>&1
done
python3 - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pmc_*/**/*counter_collection.csv', recursive=True):
    per=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if not any(x in k for x in ('rgb2yuv','yuv2rgb','tone_std','tone_neutral','chroma_map','mad_window','synthesis0','analysis0','detail_gather','haar_syn')): continue
        per[(k.split('(')[0][-40:], r['Counter_Name'], r['Dispatch_Id'])]+=float(r['Counter_Value'])
    for (k,c,d),v in per.items(): acc[k][c].append(v)
for k,cs in acc.items():
    print(k, {c: round(sum(v)/len(v)/1e6,2) for c,v in sorted(cs.items())})
PY
