#!/bin/bash
# round 5: where the waves of every kernel of a workload spend their cycles (one rocprofv3 --pmc pass per counter)
W=${1:-c3}; shift
O=$PWD/gpurun_out/r5waits_$W; mkdir -p $O; R=$PWD
cd /tmp; export TMPDIR=/tmp
D=""
for c in SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_WAVES GRBM_GUI_ACTIVE TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum; do
  rm -rf /tmp/pw_${W}_$c
  (timeout 200 rocprofv3 --pmc $c --output-format csv -d /tmp/pw_${W}_$c -- python $R/bench.py --workload $W --no-cpu-baseline --sustained-seconds 0 --steps 2 --warmup 1 --opt dn_streams=0 "$@" > /dev/null 2>&1) || echo "failed $c"
  D="$D /tmp/pw_${W}_$c"
done
cd $R
python scripts/pmc_all.py $O/pmc_waits.json $D > /dev/null
python3 - $O <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + "/pmc_waits.json"))
keys = ["SQ_WAVES","SQ_WAVE_CYCLES","SQ_BUSY_CYCLES","SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_WAIT_INST_LDS","SQ_ACTIVE_INST_ANY","SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_SCA","SQ_ACTIVE_INST_LDS","SQ_ACTIVE_INST_VMEM","SQ_INST_CYCLES_VMEM_RD","SQ_INST_CYCLES_VMEM_WR","SQ_LDS_BANK_CONFLICT","SQ_LDS_IDX_ACTIVE","SQ_INSTS_SMEM","SQ_INSTS_FLAT","GRBM_GUI_ACTIVE","TA_BUSY_avr","TCP_PENDING_STALL_CYCLES_sum"]
for k, v in d.items():
    if not any(s in k for s in ("lds_kernel", "detail", "mad_window", "exposure", "shrink", "haar", "amaze_stream", "chroma", "analysis0", "synthesis0")): continue
    print(k[:40], " ".join(f"{c.replace('SQ_','')}={v[c]['mean_per_launch']/1e6:.1f}M" for c in keys if c in v))
PY
