"""round 5: does the out-of-LDS path of the LUT-in-LDS pixel kernels cost what the counters suggest?  tone_curve (STD) on 45 MP frames whose
values all lie inside the LDS-resident part of the curve [0, 40704) / span the whole table / lie mostly above it."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from art_amd import capi
dev = torch.device("cuda:0")
W, H = 8184, 5456
ctx = capi.Context(0, torch.cuda.current_stream(dev).cuda_stream)
x = np.arange(65536, dtype=np.float64) / 65535.0
lut = ((1.0 - np.cos(np.pi * x ** 0.7)) / 2.0 * 65535.0).astype(np.float32)
for name, lo, hi in (("inside", 0.0, 40000.0), ("whole", 0.0, 65535.0), ("above", 41000.0, 65535.0), ("1% above", 0.0, 40000.0)):
    g = torch.Generator(device=dev); g.manual_seed(1)
    base = [torch.rand((H, W), device=dev, generator=g) * (hi - lo) + lo for _ in range(3)]
    if name == "1% above":
        for b in base:
            m = torch.rand((H, W), device=dev, generator=g) < 0.01
            b[m] = 60000.0
    for clip in (True, False):
        ts = []
        for rep in range(6):
            pl = [b.clone() for b in base]
            img = capi.RGB(*[capi.device_plane(t) for t in pl])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ctx.tone_curve(img, lut, 1.0, clip); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print(f"{name:9s} filmlike_clip={clip}: {min(ts[1:])*1e3:7.1f} us")
