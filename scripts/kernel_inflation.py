"""Per-kernel mean duration in two rocprofv3 --kernel-trace runs (csv): how much each kernel of a frame stretches when frames run beside each
other.  usage: kernel_inflation.py DIR_ALONE DIR_TOGETHER [frames_alone frames_together]"""
import csv, glob, sys
from collections import defaultdict
def load(d):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].split("(")[0][-44:]
            a = acc[n]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return acc
a, b = load(sys.argv[1]), load(sys.argv[2])
fa = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
fb = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
print(f"{'kernel':46s} {'alone us':>10s} {'together us':>12s} {'ratio':>6s}   ms/frame alone -> together")
ta = tb = 0.0
for n in sorted(a, key=lambda n: -a[n][1]):
    if n not in b or a[n][1] / fa < 20: continue
    ma, mb = a[n][1] / a[n][0], b[n][1] / b[n][0]
    print(f"{n:46s} {ma:10.1f} {mb:12.1f} {mb / ma:6.2f}   {a[n][1] / fa / 1e3:6.3f} -> {b[n][1] / fb / 1e3:6.3f}")
    ta += a[n][1] / fa / 1e3; tb += b[n][1] / fb / 1e3
print(f"sum of kernel time per frame: {ta:.2f} -> {tb:.2f} ms")
