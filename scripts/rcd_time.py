"""RCD at 8192x5464 on the device: stage time of the streaming kernel (rows per iteration 4 / 8) and of the arena kernel,
and a bit comparison between them (the arena kernel is parity-checked against the oracle at small sizes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from art_amd import capi, synth
W, H = 8192, 5464
raw = synth.bayer_frame(W, H, synth.FILTERS_RGGB, seed=0)
d_raw = torch.from_numpy(raw).cuda()
ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
res = {}
for name, path, rows in (("arena", 1, 4), ("stream4", 0, 4), ("stream8", 0, 8)):
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        continue
    ctx.set_option("rcd_path", path); ctx.set_option("rcd_rows", rows)
    d_out = [torch.full((H, W), float("nan"), dtype=torch.float32, device="cuda") for _ in range(3)]
    out = capi.RGB(*[capi.device_plane(t) for t in d_out])
    for _ in range(2):
        ctx.demosaic_bayer(capi.BAYER_RCD, capi.device_plane(d_raw), synth.FILTERS_RGGB, 1.0, 4, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 10
    for _ in range(n):
        ctx.demosaic_bayer(capi.BAYER_RCD, capi.device_plane(d_raw), synth.FILTERS_RGGB, 1.0, 4, out)
    e1.record(); torch.cuda.synchronize()
    res[name] = [t.clone() for t in d_out]
    print(f"{name}: {e0.elapsed_time(e1) / n:.3f} ms per frame (demosaic + border)", flush=True)
if "arena" in res:
    for name, planes in res.items():
        if name != "arena":
            print(name, "differs from arena in", [int((a.view(torch.int32) != b.view(torch.int32)).sum()) for a, b in zip(planes, res["arena"])], "values")
