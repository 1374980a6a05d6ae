#!/bin/bash
# round 5: kernel timeline of one config-3 frame (start / end / gap to the previous kernel), streams on and off
O=$PWD/gpurun_out/r5tr; mkdir -p $O; R=$PWD
cd /tmp; export TMPDIR=/tmp
for m in 1 0; do
  rm -rf /tmp/tl_$m
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$m -- python $R/bench.py --no-cpu-baseline --sustained-seconds 0 --steps 4 --warmup 2 --opt dn_streams=$m > /dev/null 2>&1
  f=$(find /tmp/tl_$m -name "*kernel_trace.csv" | head -1)
  python3 - "$f" > $O/timeline_streams$m.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last full frame: from the last amaze_stream_kernel to the tone kernel behind it
idx = [i for i, r in enumerate(rows) if "amaze_stream_kernel" in r["Kernel_Name"]]
i0 = idx[3]      # a frame of the timed region (behind it: the NEUTRAL and separate-stage legs)
t0 = int(rows[i0]["Start_Timestamp"]); prev_end = None; busy = 0; tot_gap = 0
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"][:48]
    gap = (s - prev_end) if prev_end is not None else 0
    print(f"{(s-t0)/1e3:9.1f} {(e-t0)/1e3:9.1f} dur {(e-s)/1e3:8.1f} gap {gap/1e3:7.1f} q{r.get('Queue_Id','?')} {name}")
    if prev_end is not None and gap > 0: tot_gap += gap
    prev_end = max(prev_end or 0, e)
    if "tone_" in name: break
print("sum of positive gaps us:", tot_gap/1e3)
PY
done
for f in $O/timeline_streams1.txt $O/timeline_streams0.txt; do tail -n 2 $f; done
