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
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
