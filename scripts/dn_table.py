"""Per-frame table of a bench workload: kernel, launches per frame, ms per frame, HBM traffic per frame (2 x FETCH_SIZE + WRITE_SIZE, the
gfx950 correction of MI355X_MICROARCH.md), TB/s, and -- round 6, SURVEY.md section 8(d) last sentence -- the MINIMUM each kernel could move
(its inputs read once + its outputs written once, given the barriers between the kernels: the "barrier-aware" bytes) with the ratio
moved / minimum.  usage: python scripts/dn_table.py KERNEL_STATS.csv PMC_ALL.json FRAMES [OUT.md] [--px P --raw N]
(P: pixels of the frame behind getImage's crop, N: CFA pixels; defaults: the 45 MP benchmark frame)"""
import csv, json, re, sys
args = [a for a in sys.argv[1:] if not a.startswith("--")]
opt = {sys.argv[i][2:]: float(sys.argv[i + 1]) for i in range(1, len(sys.argv) - 1) if sys.argv[i].startswith("--")}
args = [a for a in args if a not in [str(v) for v in opt.values()] and a not in [sys.argv[i + 1] for i in range(1, len(sys.argv) - 1) if sys.argv[i].startswith("--")]]
sys.argv = [sys.argv[0]] + args
P = opt.get("px", 8184.0 * 5456.0)
N = opt.get("raw", 8192.0 * 5464.0)


def min_bytes(k, launches):
    """barrier-aware minimum of one FRAME's launches of kernel k, bytes.  Planes: a full-resolution plane is 4 P bytes, a wavelet band / low-pass
    plane (half resolution in both directions) is P bytes; FTblockDN decomposes L, a, b into 5 levels x 3 bands."""
    per = None                       # bytes per LAUNCH ...
    frame = None                     # ... or per FRAME for the kernels whose one launch (set) walks all 45 bands
    if k.startswith(("amaze_stream_kernel", "rcd_stream_kernel", "xtrans_tiles_kernel")): per = 16 * N        # CFA in, R G B out
    elif k.startswith("rgb2yuv"): per = 12 * N + 12 * P                         # (fused getImage: reads the demosaiced planes) -> L a b
    elif k.startswith(("yuv2rgb", "tone_std", "tone_neutral", "exposure_kernel", "yuv_mode", "get_image")): per = 24 * P
    elif k.startswith(("wavelet_analysis0", "wavelet_synthesis0")): per = 4 * P + 4 * P      # one plane in (out), low-pass + 3 bands out (in)
    elif k.startswith(("wavelet_haar_analysis", "wavelet_haar_synthesis")): per = 5 * P       # low-pass in (out), low-pass + 3 bands out (in)
    elif k.startswith("mad_window_kernel"): frame = 45 * P                      # every band once (counts taken where the bands are produced would need none)
    elif k.startswith("mad_sample_kernel"): frame = 45 * P / 32
    elif k.startswith("shrink_blur_kernel"): frame = 15 * 2 * P + 30 * 4 * P    # L: read + write; chroma: read + write + the L coefficient + the noise map
    elif k.startswith("detail_blocks"): per = 8 * P + (P / 625.0) * 4096 * 4 * 1.02      # Lin + L in, one 64 x 64 block per 25 x 25 pixels out
    elif k.startswith("detail_gather"): per = (P / 625.0) * 4096 * 4 * 1.02 + 4 * P
    elif k.startswith("chroma_map"): per = 12 * P / 4 + P                       # every other row / column of the frame in, the quarter-resolution map out
    elif k.startswith("nlm_group"): per = 12 * P                                # Y + mask in, Y out
    elif k.startswith("gauss_stream"): per = 8 * P
    if frame is not None:
        return frame
    return None if per is None else per * launches
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
    mb = min_bytes(k, calls / frames)
    rows.append((k, calls / frames, tot / frames / 1e6, None if gb is None else gb * calls / frames, None if gb is None else gb / (tot / calls / 1e9) / 1e3, None if mb is None else mb / 1e9))
rows.sort(key=lambda x: -x[2])
lines = ["| kernel | launches / frame | ms / frame | GB / frame (2 x FETCH + WRITE) | TB/s | minimum GB / frame | moved / minimum |", "|---|---|---|---|---|---|---|"]
tms = tgb = tmin = 0.0
for k, n, ms, gb, tbs, mb in rows:
    if ms < 0.002:
        continue
    tms += ms; tgb += gb or 0.0; tmin += mb or 0.0
    ratio = "" if gb is None or not mb else f"{gb / mb:.2f}"
    lines.append(f"| `{k}` | {n:.1f} | {ms:.3f} | {'' if gb is None else f'{gb:.2f}'} | {'' if tbs is None else f'{tbs:.2f}'} | {'' if mb is None else f'{mb:.2f}'} | {ratio} |")
lines.append(f"| **total** | | **{tms:.2f}** | **{tgb:.1f}** | **{tgb / tms:.2f}** | **{tmin:.1f}** (kernels with a model) | |")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 4:
    open(sys.argv[4], "w").write(out + "\n")
