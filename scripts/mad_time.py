import os, sys
ROOT = os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from art_amd import capi
ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
w, h = 8184, 5456
rng = np.random.default_rng(1)
img = (rng.normal(0, 1, (h, w)) * 3000 + 20000).astype(np.float32)
wv = ctx.wavelet_decompose(capi.host_plane(img), 5)
for _ in range(2): ctx.wavelet_mad(wv)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ctx.wavelet_mad(wv)
e1.record(); torch.cuda.synchronize()
print(os.environ.get("ARTGPU_LIB", "default"), "wavelet_mad (15 bands of 11 MP): %.3f ms" % (e0.elapsed_time(e1) / 10))
