"""Per-frame table of a bench workload: kernel, launches per frame, ms per frame, HBM traffic per frame (2 x FETCH_SIZE + WRITE_SIZE, the
gfx950 correction of MI355X_MICROARCH.md), TB/s.  usage: python scripts/dn_table.py KERNEL_STATS.csv PMC_ALL.json FRAMES [OUT.md]"""
import csv, json, re, sys
stats = list(csv.DictReader(open(sys.argv[1])))
pmc = json.load(open(sys.argv[2]))
frames = float(sys.argv[3])
rows = []
for r in stats:
    k = re.sub(r"\(.*", "", r["Name"].replace("(anonymous namespace)::", "")).replace("artgpu::", "").replace("void ", "").replace("(anonymous namespace)::", "")
    if k.startswith("at::") or "elementwise" in k:
        continue
    calls, tot = int(r["Calls"]), float(r["TotalDurationNs"])
    p = pmc.get(k, {})
    f, w = p.get("FETCH_SIZE", {}).get("mean_per_launch"), p.get("WRITE_SIZE", {}).get("mean_per_launch")
    gb = None if f is None or w is None else (2 * f + w) * 1024 / 1e9        # KB -> GB per launch
    rows.append((k, calls / frames, tot / frames / 1e6, None if gb is None else gb * calls / frames, None if gb is None else gb / (tot / calls / 1e9) / 1e3))
rows.sort(key=lambda x: -x[2])
lines = ["| kernel | launches / frame | ms / frame | GB / frame (2 x FETCH + WRITE) | TB/s |", "|---|---|---|---|---|"]
tms = tgb = 0.0
for k, n, ms, gb, tbs in rows:
    if ms < 0.002:
        continue
    tms += ms; tgb += gb or 0.0
    lines.append(f"| `{k}` | {n:.1f} | {ms:.3f} | {'' if gb is None else f'{gb:.2f}'} | {'' if tbs is None else f'{tbs:.2f}'} |")
lines.append(f"| **total** | | **{tms:.2f}** | **{tgb:.1f}** | **{tgb / tms:.2f}** |")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 4:
    open(sys.argv[4], "w").write(out + "\n")
