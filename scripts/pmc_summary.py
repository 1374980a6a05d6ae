"""Mean per launch of rocprofv3 --pmc counters for one kernel.
usage: python scripts/pmc_summary.py KERNEL_SUBSTRING OUT.json DIR [DIR ...]     (each DIR = the -d directory of one --pmc pass)
Reads every *counter_collection.csv below the directories; counters that rocprofv3 reports per dimension (XCC, SE, ...) are summed
per dispatch first."""
import csv, glob, json, os, sys
from collections import defaultdict

kern, out = sys.argv[1], sys.argv[2]
per = defaultdict(lambda: defaultdict(float))          # counter -> dispatch id -> value
for d in sys.argv[3:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if kern not in row.get("Kernel_Name", ""):
                continue
            per[row["Counter_Name"]][(f, row.get("Dispatch_Id", ""))] += float(row["Counter_Value"])
res = {c: {"launches": len(v), "mean_per_launch": sum(v.values()) / len(v)} for c, v in sorted(per.items())}
# the digest of the kernel's sources as they are now (bench.py compares it with the sources of the day it runs: `traffic_stale`)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from art_amd import srchash
for k in srchash.KERNEL_SOURCES:
    if k in kern or kern in k:
        res["_kernel"] = k
        res["_source_sha256"] = srchash.kernel_source_sha256(k)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
