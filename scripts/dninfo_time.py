"""Time artgpu_denoise_compute_params (AUTOMATIC chrominance) on 45 MP planes resident on the device."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from art_amd import capi
ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
W, H = 8192, 5464
g = torch.Generator(device="cuda"); g.manual_seed(1)
yy, xx = torch.meshgrid(torch.arange(H, device="cuda", dtype=torch.float32), torch.arange(W, device="cuda", dtype=torch.float32), indexing="ij")
pl = [(b + a * torch.sin(f * xx) * torch.cos(0.011 * yy) + 800 * torch.randn((H, W), device="cuda", generator=g)).clamp_(min=0).contiguous()
      for b, a, f in ((9000, 6000, 0.013), (11000, 5000, 0.009), (7000, 4000, 0.015))]
rgb = capi.RGB(*[capi.device_plane(t) for t in pl])
MAT = np.array([[0.6325, 0.2312, 0.0921], [0.2198, 0.7712, 0.0090], [0.0166, 0.0713, 0.7514]])
WS = np.array([[0.6734241, 0.1656411, 0.1251286], [0.2790177, 0.6753402, 0.0456377], [-0.0019300, 0.0299784, 0.7973330]])
def fn():
    dn = capi.DenoiseParams(40.0, 50.0, 0, 15.0, 0.0, 0.0, 1.7, 0, 0, 1)
    st = ctx.denoise_compute_params(rgb, 4, (2.1, 1.0, 1.55), True, MAT, WS, dn)
    return dn.chrominance, st
fn(); torch.cuda.synchronize()
n = int(os.environ.get("N", "3"))
t = time.time()
for _ in range(n): c, st = fn()
torch.cuda.synchronize()
print("denoise_compute_params 45MP ms", (time.time() - t) * 1e3 / n, "chrominance", c)
