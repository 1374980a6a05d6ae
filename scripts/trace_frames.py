"""Frame timeline of a rocprofv3 --kernel-trace (+ --memory-copy-trace) csv: per marker kernel (default amaze_stream_kernel) the period between
its launches, and inside the last period every kernel / copy that took more than --min-us, with the idle time of the compute queue.
usage: trace_frames.py DIR [--marker NAME] [--min-us 50]"""
import argparse, csv, glob
ap = argparse.ArgumentParser(); ap.add_argument("dir"); ap.add_argument("--marker", default="amaze_stream_kernel"); ap.add_argument("--min-us", type=float, default=50)
a = ap.parse_args()
def rows(pat):
    out = []
    for f in glob.glob(f"{a.dir}/**/*{pat}", recursive=True):
        out += list(csv.DictReader(open(f)))
    return out
ks, cs = rows("kernel_trace.csv"), rows("memory_copy_trace.csv")
t0 = min(int(r["Start_Timestamp"]) for r in ks)
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-48:], "q" + r.get("Queue_Id", "?")) for r in ks]
ev += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r["Direction"][12:], "s" + r["Stream_Id"]) for r in cs]
ev.sort()
marks = [e[0] for e in ev if a.marker in e[2]]
print("periods between", a.marker, "launches (ms):", " ".join(f"{(y - x) / 1e6:.2f}" for x, y in zip(marks, marks[1:])))
lo, hi = marks[-2], marks[-1]
mq = [e[3] for e in ev if a.marker in e[2]][-1]
busy = 0; last = lo
for s, e, n, q in ev:
    if s < lo or s >= hi: continue
    if q == mq:
        if s > last: pass
        busy += max(0, min(e, hi) - max(s, last)); last = max(last, e)
    if (e - s) / 1e3 >= a.min_us: print(f"{(s - t0) / 1e6:10.3f} +{(e - s) / 1e6:7.3f}  {q:4s} {n}")
print(f"queue {mq}: busy {busy / 1e6:.2f} ms of the {(hi - lo) / 1e6:.2f} ms period")
