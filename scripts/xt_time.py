import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from art_amd import capi, synth
W, H = 11648, 8736
raw = synth.xtrans_frame(W, H, seed=0)
d_raw = torch.from_numpy(raw).cuda()
d_out = [torch.empty((H, W), dtype=torch.float32, device="cuda") for _ in range(3)]
ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
out = capi.RGB(*[capi.device_plane(t) for t in d_out])
for passes, lab in ((3, True), (1, False)):
    f = lambda: ctx.demosaic_xtrans(passes, lab, capi.device_plane(d_raw), synth.XTRANS_FUJI, synth.XTRANS_RGB_CAM, out)
    f(); torch.cuda.synchronize(); t = time.time(); f(); f(); torch.cuda.synchronize()
    print("xtrans", passes, "pass ms", (time.time() - t) * 500)
