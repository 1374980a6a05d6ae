"""The NEUTRAL tone curve (tonecurve.hip) alone on a device-resident 45 MP frame: ms per call, timed with events over REPS calls.
ARTGPU_LIB=... selects another build of the library; OPT=name=value,... sets context options (e.g. lut_lds=0).
Prints a checksum of the result so that two builds can be compared for equal bits."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from art_amd import capi
REPS = int(os.environ.get("REPS", "10"))
W, H = 8184, 5456
dev = torch.device("cuda:0")
ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
for kv in filter(None, os.environ.get("OPT", "").split(",")):
    k, v = kv.split("="); ctx.set_option(k, int(v))
g = torch.Generator(device=dev); g.manual_seed(5)
base = torch.rand((H, W), device=dev, generator=g) ** 2.2 * 60000.0          # linear data: most pixels in the lower part of the range
src = [(base * s + torch.rand((H, W), device=dev, generator=g) * 3000.0).clamp_(0, 65535) for s in (1.0, 0.8, 0.6)]
src[0][:64] *= 1.6                                                             # some super-white rows (the powf arm)
ws = np.array([[0.6734241, 0.1656411, 0.1251286], [0.2790177, 0.6753402, 0.0456377], [-0.0019300, 0.0299784, 0.7973330]])
iws = np.array([[1.6473376, -0.3935675, -0.2359961], [-0.6826036, 1.6475887, 0.0128190], [0.0296524, -0.0628993, 1.2531279]])
x = np.arange(65536, dtype=np.float64) / 65535.0
lut = ((1.0 - np.cos(np.pi * x ** 0.7)) / 2.0 * 65535.0).astype(np.float32)
work = [torch.empty_like(t) for t in src]
img = capi.RGB(*[capi.device_plane(t) for t in work])
def once():
    for d, s in zip(work, src): d.copy_(s)
    ctx.tone_curve_neutral(img, lut, 1.0, ws, iws)
once(); torch.cuda.synchronize()
chk = [int(t.view(torch.int32).to(torch.int64).sum().item()) for t in work]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
tot = 0.0
for _ in range(REPS):
    for d, s in zip(work, src): d.copy_(s)
    e0.record(); ctx.tone_curve_neutral(img, lut, 1.0, ws, iws); e1.record(); torch.cuda.synchronize()
    tot += e0.elapsed_time(e1)
print(f"[{os.environ.get('ARTGPU_LIB', 'default').split('/')[-1]} {os.environ.get('OPT', '')}] neutral 45 MP: {tot / REPS:.3f} ms  checksum {chk}")
