"""The PCIe-inclusive rate of config 3 when the boundary hands over host buffers: the CFA is uploaded from (pinned) host memory and the
final RGB planes are downloaded, around the same device work as bench.py.  Never the reported `value`; quoted in DESIGN.md section 6."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from art_amd import capi, synth
W, H, border = 8192, 5464, 4
dev = torch.device("cuda:0")
raw_h = torch.from_numpy(synth.bayer_frame(W, H, synth.FILTERS_RGGB, seed=1)).pin_memory()
out_h = [torch.empty((H - 2 * border, W - 2 * border), dtype=torch.float32).pin_memory() for _ in range(3)]
raw = torch.empty((H, W), dtype=torch.float32, device=dev)
mul = (2.1374, 1.0, 1.5918)
mat = np.array([[0.6325, 0.2312, 0.0921], [0.2198, 0.7712, 0.0090], [0.0166, 0.0713, 0.7514]])
ws = np.array([[0.6734241, 0.1656411, 0.1251286], [0.2790177, 0.6753402, 0.0456377], [-0.0019300, 0.0299784, 0.7973330]])
dn = capi.DenoiseToolParams(capi.DenoiseParams(40.0, 50.0, 0, 15.0, 0.0, 0.0, 1.7, 0, 0, 0), 0, 3, 0, 80)
ccurve, _ = capi.noise_curve_lut()
x = np.arange(65536, dtype=np.float64) / 65535.0
lut = ((1.0 - np.cos(np.pi * x ** 0.7)) / 2.0 * 65535.0).astype(np.float32)
es = float(np.float32(2.0 ** 0.3))
stream = torch.cuda.current_stream(dev)
ctx = capi.Context(0, stream.cuda_stream)
out = [torch.empty((H, W), dtype=torch.float32, device=dev) for _ in range(3)]
img = [torch.empty((H - 2 * border, W - 2 * border), dtype=torch.float32, device=dev) for _ in range(3)]
p_out = capi.RGB(*[capi.device_plane(t) for t in out]); p_img = capi.RGB(*[capi.device_plane(t) for t in img]); p_raw = capi.device_plane(raw)


def frame(pcie):
    if pcie:
        raw.copy_(raw_h, non_blocking=True)
    ctx.demosaic_bayer(capi.BAYER_AMAZE, p_raw, synth.FILTERS_RGGB, 1.0, border, p_out)
    ctx.get_image(p_out, border, border, mul, True, mat, p_img)
    ctx.improc_denoise(p_img, dn, ws, ecomp=0.3, calclum_mat=mat, noise_c_curve=ccurve)
    ctx.exposure(p_img, es, 0.0)
    ctx.tone_curve(p_img, lut, 1.0, True)
    if pcie:
        for d, s in zip(out_h, img):
            d.copy_(s, non_blocking=True)


for pcie in (False, True):
    frame(pcie); torch.cuda.synchronize()
    t = time.time()
    for _ in range(8):
        frame(pcie)
    torch.cuda.synchronize()
    dt = (time.time() - t) / 8
    print(f"{'host buffers (H2D 179 MB + D2H 536 MB per frame, pinned)' if pcie else 'device-resident'}: {dt*1e3:.2f} ms/frame = {W*H/1e6/dt:.0f} MP/s")
