"""Timeline of a rocprofv3 --kernel-trace --memory-copy-trace run (csv): every memory copy of at least 1 MB with its direction, size, duration
and rate, and what the kernels did meanwhile (busy time of the union of kernel intervals inside the copy's interval).  usage: copy_timeline.py DIR"""
import csv, glob, sys
d = sys.argv[1]
def rows(pat):
    out = []
    for f in glob.glob(f"{d}/**/*{pat}", recursive=True):
        out += list(csv.DictReader(open(f)))
    return out
ks = rows("kernel_trace.csv"); cs = rows("memory_copy_trace.csv")
kint = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in ks)
t0 = min([a for a, _ in kint] + [int(r["Start_Timestamp"]) for r in cs])
def busy(a, b):
    tot, cur = 0, a
    for s, e in kint:
        if e <= cur: continue
        if s >= b: break
        s = max(s, cur)
        if e > s: tot += min(e, b) - s; cur = max(cur, min(e, b))
    return tot
print(f"{len(ks)} kernels, {len(cs)} copies")
for r in sorted(cs, key=lambda r: int(r["Start_Timestamp"])):
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    sz = int(r.get("Size", r.get("Bytes", 0)) or 0)
    if sz < (1 << 20): continue
    print(f"{(a - t0) / 1e6:9.3f} ms  +{(b - a) / 1e6:7.3f} ms  {sz / 1e6:8.1f} MB  {sz / max(b - a, 1):6.1f} GB/s  {r.get('Direction', '?'):28s} kernels busy {100.0 * busy(a, b) / max(b - a, 1):5.1f} %")
