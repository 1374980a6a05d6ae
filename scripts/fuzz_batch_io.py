"""Fuzz artgpu_batch_run_io against the four calls it stands for (artgpu_scale_colors -> artgpu_pipeline_run -> artgpu_rgb2out_matrix ->
artgpu_get_scanlines): random frame sizes (mixed within a batch), Bayer patterns, scanline formats, pinned / pageable / pitched buffers,
lanes 1-3, option io_direct.  SEED / N from the environment.  Last line: "failures: k"."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from art_amd import capi, synth
from test_gpu_pipeline import _lut, _params

rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
N = int(os.environ.get("N", "24"))
ctx = capi.Context(0)
lut = _lut()
trc = np.sqrt(np.arange(2048, dtype=np.float64) / 2047.0).astype(np.float32)
OUTM = np.array([[0.90, 0.06, 0.04], [0.05, 0.90, 0.05], [0.03, 0.07, 0.90]], np.float32)
FILTERS = [0x94949494, 0x16161616, 0x61616161, 0x49494949]
fails = 0
for case in range(N):
    lanes = int(rng.integers(1, 4)); nf = int(rng.integers(1, 7))
    bps, is_float = [(8, False), (16, False), (16, True), (32, True)][int(rng.integers(0, 4))]
    use_matrix = bool(rng.integers(0, 2)); use_trc = use_matrix and bool(rng.integers(0, 2))
    direct = int(rng.choice([-1, 0, 3, 8]))
    filt = FILTERS[int(rng.integers(0, 4))]
    p = _params(lut, 0); p.filters = filt; b = 4
    black = tuple(float(v) for v in rng.uniform(0, 200, 4)); scale = tuple(float(v) for v in rng.uniform(0.9, 1.1, 4))
    dt = np.float32 if bps == 32 else (np.uint8 if bps == 8 else np.uint16)
    sensors, wants, ins, outs, keep = [], [], [], [], []
    for f in range(nf):
        w, h = int(rng.integers(20, 90)) * 8, int(rng.integers(16, 70)) * 8
        s = np.clip(synth.bayer_frame(w, h, filt, seed=int(rng.integers(1, 1 << 30)), noise=1500), 0, 65535).astype(np.uint16)
        d_cfa = torch.empty((h, w), dtype=torch.float32, device="cuda")
        d_img = [torch.empty((h - 2 * b, w - 2 * b), dtype=torch.float32, device="cuda") for _ in range(3)]
        img = capi.RGB(*[capi.device_plane(t) for t in d_img])
        ctx.scale_colors(s, filt, None, black, scale, capi.device_plane(d_cfa))
        ctx.pipeline_run(capi.device_plane(d_cfa), p, img)
        if use_matrix:
            ctx.rgb2out_matrix(img, img, OUTM, not use_trc, trc if use_trc else None)
        wants.append(ctx.get_scanlines(img, bps, is_float))
        kind = int(rng.integers(0, 3))                                 # 0 pageable, 1 pinned, 2 pitched pageable
        iw, ih = w - 2 * b, h - 2 * b
        if kind == 1:
            ti = torch.from_numpy(s.view(np.int16)).pin_memory(); keep.append(ti)
            si = ti.numpy().view(np.uint16)
            to = torch.zeros((ih, iw * 3 * (bps // 8) + (-iw * 3 * (bps // 8)) % 16), dtype=torch.uint8).pin_memory(); keep.append(to)
            fr = capi.scanline_frame(np.zeros((1, 1, 3), dt), OUTM if use_matrix else None, trc if use_trc else None, is_float=is_float)
            fr.scanlines = to.data_ptr(); fr.row_stride_bytes = to.shape[1]
            outs.append((fr, to.numpy(), iw * 3 * (bps // 8)))
        else:
            pad = 7 if kind == 2 else 0
            wide = np.zeros((h, w + pad), np.uint16); wide[:, :w] = s; si = wide[:, :w]
            o = np.zeros((ih, iw + pad, 3), dt)[:, :iw, :]
            outs.append((capi.scanline_frame(o, OUTM if use_matrix else None, trc if use_trc else None, is_float=is_float), o, None))
        keep.append(si)
        ins.append(capi.sensor_frame(si, black, scale))
    ctx.set_batch_lanes(lanes); ctx.set_option("io_direct", direct)
    try:
        ctx.batch_run_io(ins, p, [o[0] for o in outs])
        ok = True
        for (fr, arr, nbytes), want in zip(outs, wants):
            got = arr[:, :nbytes] if nbytes is not None else np.ascontiguousarray(arr).reshape(want.shape[0], -1).view(np.uint8)
            ok = ok and np.array_equal(got, want.reshape(want.shape[0], -1).view(np.uint8))
    except Exception as e:      # noqa: BLE001
        ok = False
        print("exception:", e)
    if not ok:
        fails += 1
        print(f"case {case}: lanes {lanes} frames {nf} bps {bps} float {is_float} matrix {use_matrix} trc {use_trc} io_direct {direct} filters {filt:#x}: MISMATCH")
ctx.set_batch_lanes(1); ctx.set_option("io_direct", -1)
print("failures:", fails)
