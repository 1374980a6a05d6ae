"""RCD at 8192x5464 on the device: time per frame of the streaming kernel (+ border) with 4 and 8 rows per iteration."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from art_amd import capi, synth
W, H = 8192, 5464
raw = synth.bayer_frame(W, H, synth.FILTERS_RGGB, seed=0)
d_raw = torch.from_numpy(raw).cuda()
ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
d_out = [torch.empty((H, W), dtype=torch.float32, device="cuda") for _ in range(3)]
out = capi.RGB(*[capi.device_plane(t) for t in d_out])
for rows in (4, 8):
    ctx.set_option("rcd_rows", rows)
    for _ in range(2):
        ctx.demosaic_bayer(capi.BAYER_RCD, capi.device_plane(d_raw), synth.FILTERS_RGGB, 1.0, 4, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 10
    for _ in range(n):
        ctx.demosaic_bayer(capi.BAYER_RCD, capi.device_plane(d_raw), synth.FILTERS_RGGB, 1.0, 4, out)
    e1.record(); torch.cuda.synchronize()
    print(f"rows {rows}: {e0.elapsed_time(e1) / n:.3f} ms per frame (demosaic + border)", flush=True)
