"""How much does running S independent frames concurrently (one context + stream + host thread each) buy on one GPU?"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from art_amd import capi, synth
W, H, border = 8192, 5464, 4
S = int(os.environ.get("S", "2")); N = int(os.environ.get("N", "6"))
dev = torch.device("cuda:0")
raw = torch.from_numpy(synth.bayer_frame(W, H, synth.FILTERS_RGGB, seed=1)).to(dev)
mul = (2.1374, 1.0, 1.5918)
mat = np.array([[0.6325, 0.2312, 0.0921], [0.2198, 0.7712, 0.0090], [0.0166, 0.0713, 0.7514]])
ws = np.array([[0.6734241, 0.1656411, 0.1251286], [0.2790177, 0.6753402, 0.0456377], [-0.0019300, 0.0299784, 0.7973330]])
dn = capi.DenoiseToolParams(capi.DenoiseParams(40.0, 50.0, 0, 15.0, 0.0, 0.0, 1.7, 0, 0, 0), 0, 3, 0, 80)
ccurve, _ = capi.noise_curve_lut()
x = np.arange(65536, dtype=np.float64) / 65535.0
lut = ((1.0 - np.cos(np.pi * x ** 0.7)) / 2.0 * 65535.0).astype(np.float32)
es = float(np.float32(2.0 ** 0.3))
class Lane:
    def __init__(self):
        self.stream = torch.cuda.Stream(dev)
        self.ctx = capi.Context(0, self.stream.cuda_stream)
        self.out = [torch.empty((H, W), dtype=torch.float32, device=dev) for _ in range(3)]
        self.img = [torch.empty((H - 2 * border, W - 2 * border), dtype=torch.float32, device=dev) for _ in range(3)]
        self.p_out = capi.RGB(*[capi.device_plane(t) for t in self.out]); self.p_img = capi.RGB(*[capi.device_plane(t) for t in self.img])
        self.p_raw = capi.device_plane(raw)
    def frame(self):
        c = self.ctx
        c.demosaic_bayer(capi.BAYER_AMAZE, self.p_raw, synth.FILTERS_RGGB, 1.0, border, self.p_out)
        c.get_image(self.p_out, border, border, mul, True, mat, self.p_img)
        c.improc_denoise(self.p_img, dn, ws, ecomp=0.3, calclum_mat=mat, noise_c_curve=ccurve)
        c.exposure(self.p_img, es, 0.0)
        c.tone_curve(self.p_img, lut, 1.0, True)
    def run(self, n):
        for _ in range(n): self.frame()
        self.ctx.synchronize()
lanes = [Lane() for _ in range(S)]
for l in lanes: l.run(1)
torch.cuda.synchronize()
t = time.time()
th = [threading.Thread(target=l.run, args=(N,)) for l in lanes]
for x_ in th: x_.start()
for x_ in th: x_.join()
torch.cuda.synchronize()
dt = time.time() - t
print(f"S={S}: {S*N} frames in {dt*1e3:.1f} ms -> {dt*1e3/(S*N):.2f} ms/frame, {S*N*(W*H)/1e6/dt:.0f} MP/s")
