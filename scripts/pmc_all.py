"""Per-kernel means of rocprofv3 --pmc counters for EVERY kernel of a run.
usage: python scripts/pmc_all.py OUT.json DIR [DIR ...]   (each DIR = the -d directory of one --pmc pass)"""
import csv, glob, json, os, re, sys
from collections import defaultdict
out = sys.argv[1]
per = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))       # kernel -> counter -> dispatch -> value
for d in sys.argv[2:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", row.get("Kernel_Name", "").replace("(anonymous namespace)::", "")).replace("artgpu::", "").replace("void ", "")
            per[k][row["Counter_Name"]][(f, row.get("Dispatch_Id", ""))] += float(row["Counter_Value"])
res = {k: {c: {"launches": len(v), "mean_per_launch": sum(v.values()) / len(v)} for c, v in cs.items()} for k, cs in per.items()}
json.dump(res, open(out, "w"), indent=1)
