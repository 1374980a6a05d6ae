"""print per-kernel average durations (us) from a rocprofv3 --stats kernel_stats.csv; usage: kstat.py DIR [substring ...]"""
import csv, glob, os, sys
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if len(sys.argv) < 3 or any(k in r["Name"] for k in sys.argv[2:]):
            print(f"{r['Name'][:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us")
