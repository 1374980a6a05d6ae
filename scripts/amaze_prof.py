import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from art_amd import capi, synth
W, H = 8192, 5464
raw = synth.bayer_frame(W, H, synth.FILTERS_RGGB, seed=0)
d_raw = torch.from_numpy(raw).cuda()
d_out = [torch.empty((H, W), dtype=torch.float32, device="cuda") for _ in range(3)]
ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
out = capi.RGB(*[capi.device_plane(t) for t in d_out])
for _ in range(3):
    ctx.demosaic_bayer(capi.BAYER_AMAZE, capi.device_plane(d_raw), synth.FILTERS_RGGB, 1.0, 4, out)
torch.cuda.synchronize()
