"""RGB_denoise with the DCT detail recovery on a seeded 12 MP frame: prints a checksum of the result's bits, so that two builds of the library
(ARTGPU_LIB=...) can be compared for identical output.  Not a test; run on an MI355X box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from art_amd import capi
W, H = 4000, 3000
dev = torch.device("cuda:0")
ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device=dev); g.manual_seed(9)
yy, xx = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
base = 20000.0 + 15000.0 * torch.sin(0.01 * xx) * torch.cos(0.013 * yy) + 6000.0 * (((xx // 64) + (yy // 64)) % 2)
planes = [(base * s + torch.randn((H, W), device=dev, generator=g) * 1800.0).clamp_(0, 65535).float().contiguous() for s in (1.0, 0.85, 0.6)]
ws = np.array([[0.6734241, 0.1656411, 0.1251286], [0.2790177, 0.6753402, 0.0456377], [-0.0019300, 0.0299784, 0.7973330]])
p = capi.DenoiseParams(40.0, 50.0, 0, 15.0, 0.0, 0.0, 1.7, 0, 0, 0)
ctx.rgb_denoise(capi.RGB(*[capi.device_plane(t) for t in planes]), p, ws.astype(np.float32), flags=0)      # flags 0: with the DCT stage
torch.cuda.synchronize()
print(os.environ.get("ARTGPU_LIB", "default").split("/")[-1], [int(t.view(torch.int32).to(torch.int64).sum().item()) for t in planes])
