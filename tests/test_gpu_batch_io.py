"""GPU: artgpu_batch_run_io (sensor data in, writers' scanlines out, a frame's copies beside its neighbours' kernels) returns the bits of the
chain it stands for -- artgpu_scale_colors -> artgpu_pipeline_run -> artgpu_rgb2out_matrix -> artgpu_get_scanlines, one frame after the
other -- whatever the number of lanes, the scanline format, the residency of the buffers and the mix of frame sizes."""
import ctypes as C

import numpy as np
import pytest
import torch

from art_amd import capi, synth
from test_gpu_pipeline import _lut, _params

pytestmark = pytest.mark.gpu
BLACK = (64.0, 60.0, 68.0, 62.0)
SCALE = (1.02, 1.0, 0.98, 1.01)
OUTM = np.array([[0.90, 0.06, 0.04], [0.05, 0.90, 0.05], [0.03, 0.07, 0.90]], np.float32)


def _sensor(w, h, seed, xtrans=False):
    f = synth.xtrans_frame(w, h, seed=seed, noise=1800) if xtrans else synth.bayer_frame(w, h, synth.FILTERS_RGGB, seed=seed, noise=1800)
    return np.clip(f, 0, 65535).astype(np.uint16)


def _trc(n=4096):
    x = np.arange(n, dtype=np.float64) / (n - 1)
    return np.where(x <= 0.0031308, 12.92 * x, 1.055 * x ** (1 / 2.4) - 0.055).astype(np.float32)


def _chain(ctx, sensor, p, b, fmt, matrix, trc, xtrans=False):
    """one frame through the four entry points, everything in between resident on the device"""
    h, w = sensor.shape
    d_cfa = torch.empty((h, w), dtype=torch.float32, device="cuda")
    d_img = [torch.empty((h - 2 * b, w - 2 * b), dtype=torch.float32, device="cuda") for _ in range(3)]
    img = capi.RGB(*[capi.device_plane(t) for t in d_img])
    chmax = ctx.scale_colors(sensor, synth.FILTERS_RGGB, synth.XTRANS_FUJI if xtrans else None, BLACK, SCALE, capi.device_plane(d_cfa))
    ctx.pipeline_run(capi.device_plane(d_cfa), p, img)
    if matrix is not None:
        ctx.rgb2out_matrix(img, img, matrix, trc is None, trc)
    bps, is_float = fmt
    return ctx.get_scanlines(img, bps, is_float), chmax


def _out_array(h, w, b, fmt):
    bps, is_float = fmt
    dt = np.float32 if bps == 32 else (np.uint8 if bps == 8 else np.uint16)
    return np.zeros((h - 2 * b, w - 2 * b, 3), dt)


@pytest.mark.parametrize("lanes", [1, 2, 3])
@pytest.mark.parametrize("fmt,use_matrix,use_trc", [((16, False), True, True), ((8, False), True, False), ((32, True), False, False), ((16, True), True, False)])
def test_batch_run_io_same_bits_as_the_four_calls(gpu_ctx, lanes, fmt, use_matrix, use_trc):
    w, h, b = 520, 392, 4
    lut = _lut()
    p = _params(lut, 0)
    trc = _trc() if use_trc else None
    matrix = OUTM if use_matrix else None
    sensors = [_sensor(w, h, 11 + k) for k in range(5)]
    want = [_chain(gpu_ctx, s, p, b, fmt, matrix, trc) for s in sensors]
    outs = [_out_array(h, w, b, fmt) for _ in sensors]
    gpu_ctx.set_batch_lanes(lanes)
    try:
        res = gpu_ctx.batch_run_io([capi.sensor_frame(s, BLACK, SCALE) for s in sensors], p,
                                   [capi.scanline_frame(o, matrix, trc, is_float=fmt[1]) for o in outs])
    finally:
        gpu_ctx.set_batch_lanes(1)
    for k, (o, (scan, chmax)) in enumerate(zip(outs, want)):
        assert res[k].status == 0
        assert [float(v) for v in res[k].chmax] == chmax, k
        assert np.array_equal(o.view(np.uint8), scan.view(np.uint8)), k


def test_batch_run_io_frames_of_different_sizes_and_a_repeat(gpu_ctx):
    """the staging slots are regrown between frames while other frames' copies are in flight; a second batch on the same context reuses them"""
    b = 4
    lut = _lut()
    p = _params(lut, 0)
    sizes = [(392, 296), (648, 488), (392, 296), (520, 392), (648, 488), (264, 200)]
    sensors = [_sensor(w, h, 30 + k) for k, (w, h) in enumerate(sizes)]
    fmt = (16, False)
    want = [_chain(gpu_ctx, s, p, b, fmt, OUTM, None) for s in sensors]
    gpu_ctx.set_batch_lanes(2)
    try:
        for rep in range(2):
            outs = [_out_array(s.shape[0], s.shape[1], b, fmt) for s in sensors]
            ins = list(sensors)
            if rep:       # pitched buffers on both sides of every other frame (rows of a larger allocation)
                for k in range(0, len(sensors), 2):
                    hh, ww = sensors[k].shape
                    wide = np.zeros((hh, ww + 9), np.uint16); wide[:, :ww] = sensors[k]; ins[k] = wide[:, :ww]
                    outs[k] = np.zeros((hh - 2 * b, ww - 2 * b + 5, 3), np.uint16)[:, :ww - 2 * b, :]
            gpu_ctx.batch_run_io([capi.sensor_frame(s, BLACK, SCALE) for s in ins], p, [capi.scanline_frame(o, OUTM) for o in outs])
            for k, (o, (scan, _)) in enumerate(zip(outs, want)):
                assert np.array_equal(o, scan), k
    finally:
        gpu_ctx.set_batch_lanes(1)


def test_batch_run_io_device_buffers_and_pinned_host_buffers(gpu_ctx):
    w, h, b = 520, 392, 4
    lut = _lut()
    p = _params(lut, 0)
    fmt = (16, False)
    sensors = [_sensor(w, h, 50 + k) for k in range(4)]
    want = [_chain(gpu_ctx, s, p, b, fmt, OUTM, None)[0] for s in sensors]
    # pinned host memory on both sides
    pin_in = [torch.from_numpy(s.view(np.int16)).pin_memory() for s in sensors]
    pin_out = [torch.zeros((h - 2 * b, w - 2 * b, 3), dtype=torch.int16).pin_memory() for _ in sensors]
    gpu_ctx.set_batch_lanes(2)
    try:
        # (pinned scanline buffers: written by the download stream's own kernel, option io_direct workgroups; 0: staged and copied by the runtime)
        for direct in (32, 3, 0):
            gpu_ctx.set_option("io_direct", direct)
            for t in pin_out:
                t.zero_()
            gpu_ctx.batch_run_io([capi.sensor_frame(t.numpy().view(np.uint16), BLACK, SCALE) for t in pin_in], p,
                                 [capi.scanline_frame(t.numpy().view(np.uint16), OUTM) for t in pin_out])
            for t, scan in zip(pin_out, want):
                assert np.array_equal(t.numpy().view(np.uint16), scan), direct
        gpu_ctx.set_option("io_direct", -1)
        # device memory on both sides: no copies at all
        d_in = [t.cuda() for t in pin_in]
        d_out = [torch.zeros((h - 2 * b, w - 2 * b, 3), dtype=torch.int16, device="cuda") for _ in sensors]
        torch.cuda.synchronize()
        ins = []
        outs = []
        for t, o in zip(d_in, d_out):
            f = capi.SensorFrame(t.data_ptr(), w, h, w * 2, 1, 1, (C.c_float * 4)(*BLACK), (C.c_float * 4)(*SCALE))
            ins.append(f)
            g = capi.scanline_frame(np.zeros((1, 1, 3), np.uint16), OUTM)
            g.scanlines = o.data_ptr(); g.row_stride_bytes = (w - 2 * b) * 6; g.on_device = 1
            outs.append(g)
        gpu_ctx.batch_run_io(ins, p, outs)
        for o, scan in zip(d_out, want):
            assert np.array_equal(o.cpu().numpy().view(np.uint16), scan)
    finally:
        gpu_ctx.set_batch_lanes(1)


@pytest.mark.parametrize("fmt", [(8, False), (16, True), (32, True)])
@pytest.mark.parametrize("w,h", [(520, 392), (1100, 264)])
def test_batch_run_io_pinned_scanlines_every_format(gpu_ctx, fmt, w, h):
    """the direct-to-host kernel: rows whose byte length is no multiple of 16, chunks shorter than 512 pixels, padded row strides"""
    b = 4
    lut = _lut()
    p = _params(lut, 0)
    sensors = [_sensor(w, h, 60 + k) for k in range(3)]
    want = [_chain(gpu_ctx, s, p, b, fmt, OUTM, None)[0] for s in sensors]
    bps = fmt[0]
    iw, ih = w - 2 * b, h - 2 * b
    pitch = ((iw * 3 * (bps // 8) + 15) // 16) * 16 + 32           # 16-byte aligned rows, padded
    pins = [torch.zeros((ih, pitch), dtype=torch.uint8).pin_memory() for _ in sensors]
    frames = []
    for t in pins:
        f = capi.scanline_frame(np.zeros((1, 1, 3), np.float32 if bps == 32 else (np.uint8 if bps == 8 else np.uint16)), OUTM, is_float=fmt[1])
        f.scanlines = t.data_ptr(); f.row_stride_bytes = pitch
        frames.append(f)
    gpu_ctx.set_batch_lanes(2)
    gpu_ctx.set_option("io_direct", 5)
    try:
        gpu_ctx.batch_run_io([capi.sensor_frame(s, BLACK, SCALE) for s in sensors], p, frames)
    finally:
        gpu_ctx.set_batch_lanes(1)
        gpu_ctx.set_option("io_direct", -1)
    for t, scan in zip(pins, want):
        got = t.numpy()[:, :iw * 3 * (bps // 8)]
        assert np.array_equal(got, scan.reshape(ih, -1).view(np.uint8))
        assert not t.numpy()[:, iw * 3 * (bps // 8):].any()        # the padding stays untouched


def test_batch_run_io_xtrans(gpu_ctx):
    w, h, b = 520, 392, 7
    lut = _lut()
    p = _params(lut, 0, xtrans=True)
    fmt = (16, False)
    sensors = [_sensor(w, h, 70 + k, xtrans=True) for k in range(3)]
    want = [_chain(gpu_ctx, s, p, b, fmt, OUTM, None, xtrans=True)[0] for s in sensors]
    outs = [_out_array(h, w, b, fmt) for _ in sensors]
    gpu_ctx.set_batch_lanes(2)
    try:
        gpu_ctx.batch_run_io([capi.sensor_frame(s, BLACK, SCALE) for s in sensors], p, [capi.scanline_frame(o, OUTM) for o in outs])
    finally:
        gpu_ctx.set_batch_lanes(1)
    for o, scan in zip(outs, want):
        assert np.array_equal(o, scan)


def test_batch_run_io_reports_what_rgb2out_cannot_do_and_bad_arguments(gpu_ctx):
    w, h, b = 392, 296, 4
    lut = _lut()
    p = _params(lut, 0)
    sensors = [_sensor(w, h, 90 + k) for k in range(2)]
    outs = [_out_array(h, w, b, (16, False)) for _ in sensors]
    trc = _trc()
    ins = (capi.SensorFrame * 2)(*[capi.sensor_frame(s, BLACK, SCALE) for s in sensors])
    # values above 1 behind a matrix that amplifies, with a non-linear TRC: the frame says so, the call returns the code
    big = 3.0 * np.eye(3, dtype=np.float32)
    fr = (capi.ScanlineFrame * 2)(capi.scanline_frame(outs[0], OUTM, trc), capi.scanline_frame(outs[1], big, trc))
    rc = capi.LIB.artgpu_batch_run_io(gpu_ctx._h, 2, ins, C.byref(p), fr)
    assert rc == -4      # ARTGPU_EUNSUPPORTED
    assert fr[0].status == 0 and fr[1].status == rc
    assert b"ARTOutputProfile::eval" in capi.LIB.artgpu_last_error(gpu_ctx._h)
    want0 = _chain(gpu_ctx, sensors[0], p, b, (16, False), OUTM, trc)[0]
    assert np.array_equal(outs[0], want0)
    # arguments
    bad = (capi.ScanlineFrame * 2)(capi.scanline_frame(outs[0], OUTM), capi.scanline_frame(outs[1], OUTM))
    bad[1].bps = 12
    assert capi.LIB.artgpu_batch_run_io(gpu_ctx._h, 2, ins, C.byref(p), bad) != 0
    bad[1].bps = 16; bad[1].row_stride_bytes = 10
    assert capi.LIB.artgpu_batch_run_io(gpu_ctx._h, 2, ins, C.byref(p), bad) != 0
    assert capi.LIB.artgpu_batch_run_io(gpu_ctx._h, 2, None, C.byref(p), bad) != 0
    assert capi.LIB.artgpu_batch_run_io(gpu_ctx._h, 0, None, None, None) == 0
    # the context is usable afterwards
    ok = (capi.ScanlineFrame * 2)(capi.scanline_frame(outs[0], OUTM, trc), capi.scanline_frame(outs[1], OUTM, trc))
    assert capi.LIB.artgpu_batch_run_io(gpu_ctx._h, 2, ins, C.byref(p), ok) == 0
    assert np.array_equal(outs[0], want0)
