"""The PCIe-inclusive rate of config 3 when the boundary hands over HOST buffers, frames in flight on several lanes.

Three forms of the same work per 8192 x 5464 frame, each timed over a batch of frames with 1, 2 and 3 lanes:
  fp32     artgpu_batch_run on host planes: 4 B/px of CFA up, 12 B/px of RGB down (what a drop-in that keeps rtengine's buffers pays)
  slim     the formats either side of the path: uint16 sensor data up through artgpu_scale_colors (2 B/px), 16-bit
           scanlines down through artgpu_rgb2out_matrix + artgpu_get_scanlines (6 B/px); one host thread, context and stream per lane
  io       artgpu_batch_run_io: the slim formats with a frame's upload and download on streams of their own beside its neighbours' kernels
           (two staging slots per lane; the host thread never waits for a copy inside the batch)
  device   the same batch with everything resident on the device (bench.py's --lanes figure, for reference)
Host buffers are pinned (hipHostMalloc through torch) unless --pageable.  Never the reported `value` of bench.py."""
import argparse
import ctypes as C
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from art_amd import capi, synth

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=12)
ap.add_argument("--width", type=int, default=8192)
ap.add_argument("--height", type=int, default=5464)
ap.add_argument("--pageable", action="store_true")
ap.add_argument("--lanes", default="1,2,3")
ap.add_argument("--io-direct", default="-1", help="values of the io_direct option to time the io form with (0: staged + runtime copy)")
ap.add_argument("--only", default="fp32,device,slim,io", help="which forms to time")
args = ap.parse_args()
W, H, border, NF = args.width, args.height, 4, args.frames
iw, ih = W - 2 * border, H - 2 * border
dev = torch.device("cuda:0")
MUL = (2.1374, 1.0, 1.5918)
MAT = np.array([[0.6325, 0.2312, 0.0921], [0.2198, 0.7712, 0.0090], [0.0166, 0.0713, 0.7514]])
WS = np.array([[0.6734241, 0.1656411, 0.1251286], [0.2790177, 0.6753402, 0.0456377], [-0.0019300, 0.0299784, 0.7973330]])
IWS = np.array([[1.6473376, -0.3935675, -0.2359961], [-0.6826036, 1.6475887, 0.0128190], [0.0296524, -0.0628993, 1.2531279]])
x = np.arange(65536, dtype=np.float64) / 65535.0
LUT = ((1.0 - np.cos(np.pi * x ** 0.7)) / 2.0 * 65535.0).astype(np.float32)


def host(shape, dtype):
    # (pageable buffers are touched once here: a fresh page's first write is a page fault, 15 - 25 ms per 268 MB buffer, which belongs to the
    # allocation and not to the copy)
    t = torch.zeros(shape, dtype=dtype) if args.pageable else torch.empty(shape, dtype=dtype).pin_memory()
    return t.numpy()


def params():
    p = capi.PipelineParams()
    p.sensor = 0; p.bayer_method = capi.BAYER_AMAZE; p.filters = synth.FILTERS_RGGB; p.initial_gain = 1.0
    p.xtrans_passes = 3; p.border = border
    p.mul[:] = MUL; p.do_clip = 1; p.has_cam_to_work = 1
    p.cam_to_work[:] = [float(v) for v in MAT.reshape(9)]
    p.ws[:] = [float(v) for v in WS.reshape(9)]; p.iws[:] = [float(v) for v in IWS.reshape(9)]
    p.denoise_enabled = 1
    p.denoise = capi.DenoiseToolParams(capi.DenoiseParams(40.0, 50.0, 0, 15.0, 0.0, 0.0, 1.7, 0, 0, 0), 0, 3, 0, 80)
    p.exposure_enabled = 1; p.expcomp = 0.3; p.black = 0.0
    p.tone_enabled = 1; p.tone_mode = 0
    p.tone_lut = LUT.ctypes.data_as(C.POINTER(C.c_float)); p.white_point = 1.0
    p.to_out[:] = [1, 0, 0, 0, 1, 0, 0, 0, 1]; p.to_work[:] = [1, 0, 0, 0, 1, 0, 0, 0, 1]
    p.scale = 1.0
    return p


P = params()
frame = synth.bayer_frame(W, H, synth.FILTERS_RGGB, seed=1)
raw_h = host((H, W), torch.float32); raw_h[:] = frame
raw16_h = host((H, W), torch.int16).view(np.uint16)
raw16_h[:] = np.clip(frame, 0, 65535).astype(np.uint16)
del frame
mode = "pageable" if args.pageable else "pinned"
MP = W * H / 1e6
results = {}
ONLY = args.only.split(",")
LANES = [int(v) for v in args.lanes.split(",")]


def report(name, lanes, dt, up, down):
    print(f"{name:7s} lanes={lanes}  {dt * 1e3:7.2f} ms/frame = {MP / dt:7.0f} MP/s   ({up} B/px up, {down} B/px down, {mode} host memory)", flush=True)
    results[(name, lanes)] = MP / dt


NF6 = min(NF, 6)       # (12 B/px of pinned output per frame)
# ---- fp32: artgpu_batch_run on host planes
outs_h = [[host((ih, iw), torch.float32) for _ in range(3)] for _ in range(NF6 if "fp32" in ONLY else 0)]
ctx = capi.Context(0, torch.cuda.current_stream(dev).cuda_stream)
for lanes in (LANES if "fp32" in ONLY else []):
    ctx.set_batch_lanes(lanes)
    raws = [capi.host_plane(raw_h) for _ in range(NF6)]
    outs = [capi.host_rgb(o) for o in outs_h]
    ctx.batch_run(raws[:lanes], P, outs[:lanes]); ctx.synchronize()          # warm-up: pools, tables
    t = time.perf_counter()
    ctx.batch_run(raws, P, outs); ctx.synchronize()
    report("fp32", lanes, (time.perf_counter() - t) / NF6, 4, 12)
ref_out = [o.copy() for o in outs_h[0]] if "fp32" in ONLY else None
del outs_h

# ---- device-resident batch for reference
d_raw = torch.from_numpy(raw_h).to(dev)
d_outs = [[torch.empty((ih, iw), dtype=torch.float32, device=dev) for _ in range(3)] for _ in range(NF6 if "device" in ONLY else 0)]
for lanes in (LANES if "device" in ONLY else []):
    ctx.set_batch_lanes(lanes)
    raws = [capi.device_plane(d_raw) for _ in range(NF6)]
    outs = [capi.RGB(*[capi.device_plane(t) for t in o]) for o in d_outs]
    ctx.batch_run(raws[:lanes], P, outs[:lanes]); ctx.synchronize()
    t = time.perf_counter()
    ctx.batch_run(raws, P, outs); ctx.synchronize()
    report("device", lanes, (time.perf_counter() - t) / NF6, 0, 0)
if "fp32" in ONLY and "device" in ONLY:
    same = all(np.array_equal(r.view(np.uint32), t.cpu().numpy().view(np.uint32)) for r, t in zip(ref_out, d_outs[0]))
    print("host-plane batch == device-plane batch, bit for bit:", same, flush=True)
del d_outs, d_raw
ctx.set_batch_lanes(1)

# ---- slim: uint16 up (scaleColors on the device), 16-bit scanlines down (rgb2out matrix path + getScanline)
OUTM = np.eye(3, dtype=np.float32)       # working space -> output profile matrix (identity: the arithmetic is what is timed)


class Lane:
    def __init__(self):
        self.stream = torch.cuda.Stream(dev)
        self.ctx = capi.Context(0, self.stream.cuda_stream)
        self.cfa = torch.empty((H, W), dtype=torch.float32, device=dev)
        self.img = [torch.empty((ih, iw), dtype=torch.float32, device=dev) for _ in range(3)]
        self.o = [torch.empty((ih, iw), dtype=torch.float32, device=dev) for _ in range(3)]
        self.p_cfa = capi.device_plane(self.cfa)
        self.p_img = capi.RGB(*[capi.device_plane(t) for t in self.img])
        self.p_o = capi.RGB(*[capi.device_plane(t) for t in self.o])
        self.scan = host((ih, iw, 3), torch.int16).view(np.uint16)

    def frame(self):
        c = self.ctx
        c.scale_colors(raw16_h, synth.FILTERS_RGGB, None, (0.0, 0.0, 0.0, 0.0), (1.0, 1.0, 1.0, 1.0), self.p_cfa)
        c.pipeline_run(self.p_cfa, P, self.p_img)
        c.rgb2out_matrix(self.p_img, self.p_o, OUTM, True)
        c._chk(capi.LIB.artgpu_get_scanlines(c._h, C.byref(self.p_o), 16, 0, self.scan.ctypes.data, self.scan.strides[0], 0))


for lanes in (LANES if "slim" in ONLY else []):
    ls = [Lane() for _ in range(lanes)]
    errs = []

    def work(ln, n):
        try:
            for _ in range(n):
                ln.frame()
        except BaseException as e:      # noqa: BLE001
            errs.append(e)

    def run(n_each):
        th = [threading.Thread(target=work, args=(ln, n_each)) for ln in ls]
        for t_ in th:
            t_.start()
        for t_ in th:
            t_.join()
        torch.cuda.synchronize(dev)
        if errs:
            raise errs[0]

    run(1)
    n_each = max(1, NF // lanes)
    t = time.perf_counter()
    run(n_each)
    report("slim", lanes, (time.perf_counter() - t) / (n_each * lanes), 2, 6)
    del ls
    torch.cuda.empty_cache()

# ---- io: the same formats through artgpu_batch_run_io
scan_h = [host((ih, iw, 3), torch.int16).view(np.uint16) for _ in range(NF)]
for direct in ([int(v) for v in args.io_direct.split(",")] if "io" in ONLY else []):
    ctx.set_option("io_direct", direct)
    for lanes in LANES:
        ctx.set_batch_lanes(lanes)
        ins = [capi.sensor_frame(raw16_h) for _ in range(NF)]
        outs = [capi.scanline_frame(o, OUTM) for o in scan_h]
        ctx.batch_run_io(ins[:2 * lanes], P, outs[:2 * lanes])          # warm-up: both staging slots of every lane
        t = time.perf_counter()
        ctx.batch_run_io(ins, P, outs)
        report(f"io/{direct}", lanes, (time.perf_counter() - t) / NF, 2, 6)
ctx.set_batch_lanes(1)
if "io" in ONLY:
    ln = Lane(); ln.frame()
    # every frame of the last batch (the same input each): a slot reused too early or a copy that overtook its kernel would show as a frame that differs
    print("batch_run_io == the four calls per frame, bit for bit, every frame of the batch:", bool(all(np.array_equal(ln.scan, o) for o in scan_h)), flush=True)

import json
print(json.dumps({"frame": f"{W}x{H}", "host_memory": mode, "frames": NF,
                  "mp_per_s": {f"{k[0]}_lanes{k[1]}": round(v, 1) for k, v in results.items()}}))
