"""GPU: artgpu_pipeline_run / artgpu_batch_run chain the same stages as the individual entry points."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle_lib as O
from art_amd import capi, synth

pytestmark = pytest.mark.gpu
MUL = (2.1374, 1.0, 1.5918)
MAT = np.array([[0.6325, 0.2312, 0.0921], [0.2198, 0.7712, 0.0090], [0.0166, 0.0713, 0.7514]])


def _lut():
    x = np.arange(65536, dtype=np.float64) / 65535.0
    return ((1.0 - np.cos(np.pi * x ** 0.7)) / 2.0 * 65535.0).astype(np.float32)


def _params(lut, tone_mode, xtrans=False):
    p = capi.PipelineParams()
    p.sensor = 1 if xtrans else 0
    p.bayer_method = capi.BAYER_AMAZE; p.filters = synth.FILTERS_RGGB; p.initial_gain = 1.0
    p.xtrans_passes = 3
    p.xtrans[:] = [int(v) for v in synth.XTRANS_FUJI.reshape(36)]
    p.rgb_cam[:] = [float(v) for v in synth.XTRANS_RGB_CAM.reshape(12)]
    p.border = 7 if xtrans else 4
    p.mul[:] = MUL; p.do_clip = 1; p.has_cam_to_work = 1
    p.cam_to_work[:] = [float(v) for v in MAT.reshape(9)]
    p.ws[:] = [float(v) for v in O.REC2020_WS_D.reshape(9)]
    p.iws[:] = [float(v) for v in O.REC2020_IWS_D.reshape(9)]
    p.denoise_enabled = 1
    p.denoise = capi.DenoiseToolParams(capi.DenoiseParams(40.0, 50.0, 0, 15.0, 0.0, 0.0, 1.7, 0, 0, 0), 0, 3, 0, 80)
    p.exposure_enabled = 1; p.expcomp = 0.3; p.black = 0.0
    p.tone_enabled = 1; p.tone_mode = tone_mode
    p.tone_lut = lut.ctypes.data_as(C.POINTER(C.c_float)); p.white_point = 1.0
    p.to_out[:] = [1, 0, 0, 0, 1, 0, 0, 0, 1]; p.to_work[:] = [1, 0, 0, 0, 1, 0, 0, 0, 1]
    p.scale = 1.0
    return p


@pytest.mark.parametrize("tone_mode,xtrans", [(0, False), (1, False), (0, True)])
def test_pipeline_run_equals_stage_by_stage(gpu_ctx, tone_mode, xtrans):
    w, h = 520, 392
    raw = synth.xtrans_frame(w, h, seed=2, noise=2048) if xtrans else synth.bayer_frame(w, h, synth.FILTERS_RGGB, seed=2, noise=2048)
    b = 7 if xtrans else 4
    lut = _lut()
    p = _params(lut, tone_mode, xtrans)
    got = [np.zeros((h - 2 * b, w - 2 * b), np.float32) for _ in range(3)]
    gpu_ctx.pipeline_run(capi.host_plane(raw), p, capi.host_rgb(got))
    # the same chain through the individual entry points, frame resident on the device
    d_raw = torch.from_numpy(raw).cuda()
    d_dem = [torch.empty((h, w), dtype=torch.float32, device="cuda") for _ in range(3)]
    dem = capi.RGB(*[capi.device_plane(t) for t in d_dem])
    if xtrans:
        gpu_ctx.demosaic_xtrans(3, True, capi.device_plane(d_raw), synth.XTRANS_FUJI, synth.XTRANS_RGB_CAM, dem)
    else:
        gpu_ctx.demosaic_bayer(capi.BAYER_AMAZE, capi.device_plane(d_raw), synth.FILTERS_RGGB, 1.0, 4, dem)
    d_img = [torch.empty((h - 2 * b, w - 2 * b), dtype=torch.float32, device="cuda") for _ in range(3)]
    img = capi.RGB(*[capi.device_plane(t) for t in d_img])
    gpu_ctx.get_image(dem, b, b, MUL, True, MAT, img)
    curve, _ = capi.noise_curve_lut()
    gpu_ctx.improc_denoise(img, p.denoise, O.REC2020_WS_D, ecomp=0.3, calclum_mat=MAT, noise_c_curve=curve, iws=O.REC2020_IWS_D)
    gpu_ctx.exposure(img, float(np.float32(2.0 ** 0.3)), 0.0)
    if tone_mode == 1:
        gpu_ctx.tone_curve_neutral(img, lut, 1.0, O.REC2020_WS_D, O.REC2020_IWS_D)
    else:
        gpu_ctx.tone_curve(img, lut, 1.0, True)
    gpu_ctx.synchronize()
    for g, t in zip(got, d_img):
        assert np.array_equal(g.view(np.uint32), t.cpu().numpy().view(np.uint32))


def test_batch_run_two_frames(gpu_ctx):
    w, h = 392, 296
    lut = _lut()
    p = _params(lut, 0)
    raws = [synth.bayer_frame(w, h, synth.FILTERS_RGGB, seed=s, noise=1500) for s in (5, 6)]
    outs = [[np.zeros((h - 8, w - 8), np.float32) for _ in range(3)] for _ in raws]
    gpu_ctx.batch_run([capi.host_plane(r) for r in raws], p, [capi.host_rgb(o) for o in outs])
    for r, o in zip(raws, outs):
        single = [np.zeros((h - 8, w - 8), np.float32) for _ in range(3)]
        gpu_ctx.pipeline_run(capi.host_plane(r), p, capi.host_rgb(single))
        for a, s in zip(o, single):
            assert np.array_equal(a.view(np.uint32), s.view(np.uint32)) and a.max() > 0
    assert not np.array_equal(outs[0][1], outs[1][1])


def test_batch_gives_scratch_back_before_much_smaller_frames():
    """round-5 review, weak point 10: a context's scratch only grows, and a batch that goes on with much smaller frames kept the large frame's
    pools.  artgpu_batch_run trims a lane when this frame and its next one have less than half the pixels the pools were grown for; a batch
    that alternates sizes keeps them.  Same bits either way."""
    lut = _lut()
    p = _params(lut, 0)
    big, small = (1000, 760), (392, 296)

    def frames(sizes, seed0):
        raws = [synth.bayer_frame(w, h, synth.FILTERS_RGGB, seed=seed0 + i, noise=1500) for i, (w, h) in enumerate(sizes)]
        outs = [[np.zeros((h - 8, w - 8), np.float32) for _ in range(3)] for (w, h) in sizes]
        return raws, outs
    ctx = capi.Context(0)
    raws, outs = frames([big], 20)
    ctx.batch_run([capi.host_plane(r) for r in raws], p, [capi.host_rgb(o) for o in outs])
    held_big = ctx.scratch_bytes()
    # big, small, small: the pools go back before the first small frame
    raws, outs = frames([big, small, small], 30)
    ctx.batch_run([capi.host_plane(r) for r in raws], p, [capi.host_rgb(o) for o in outs])
    held_small = ctx.scratch_bytes()
    assert held_small < held_big / 2, (held_small, held_big)
    ref = capi.Context(0)
    for r, o in zip(raws, outs):
        single = [np.zeros_like(o[0]) for _ in range(3)]
        ref.pipeline_run(capi.host_plane(r), p, capi.host_rgb(single))
        for a, s_ in zip(o, single):
            assert np.array_equal(a.view(np.uint32), s_.view(np.uint32))
    # big, small, big, small: alternating sizes keep the pools (no trim: the lane's next frame is large again)
    raws, outs = frames([big, small, big, small], 40)
    ctx.batch_run([capi.host_plane(r) for r in raws[:1]], p, [capi.host_rgb(o) for o in outs[:1]])
    held_big2 = ctx.scratch_bytes()
    assert held_big2 > 2 * held_small
    ctx.batch_run([capi.host_plane(r) for r in raws], p, [capi.host_rgb(o) for o in outs])
    assert ctx.scratch_bytes() >= held_big2
    del ctx, ref


def test_pipeline_automatic_chroma_equals_explicit_compute_params(gpu_ctx):
    """chrominance_method AUTOMATIC inside artgpu_pipeline_run = denoiseComputeParams on the demosaiced planes, then the same
    stages with the estimated values (simpleprocess.cc:254-256,311-315)."""
    w, h = 520, 392
    raw = synth.bayer_frame(w, h, synth.FILTERS_RGGB, seed=9, noise=2600)
    lut = _lut()
    p = _params(lut, 0)
    p.denoise.dn.chrominance_method = 1
    p.denoise.dn.chrominance = 0.0
    p.chrominance_auto_factor = 1.5
    got = [np.zeros((h - 8, w - 8), np.float32) for _ in range(3)]
    gpu_ctx.pipeline_run(capi.host_plane(raw), p, capi.host_rgb(got))
    d_raw = torch.from_numpy(raw).cuda()
    d_dem = [torch.empty((h, w), dtype=torch.float32, device="cuda") for _ in range(3)]
    dem = capi.RGB(*[capi.device_plane(t) for t in d_dem])
    gpu_ctx.demosaic_bayer(capi.BAYER_AMAZE, capi.device_plane(d_raw), synth.FILTERS_RGGB, 1.0, 4, dem)
    dn = capi.DenoiseParams(40.0, 50.0, 0, 0.0, 0.0, 0.0, 1.7, 0, 0, 1)
    st = gpu_ctx.denoise_compute_params(dem, 4, MUL, True, MAT, O.REC2020_WS_D, dn, auto_factor=1.5)
    assert st.valid == 1 and dn.chrominance > 0
    # the oracle agrees on the estimate
    ref = O.denoise_compute_params([t.cpu().numpy() for t in d_dem], 4, MUL, True, MAT, O.REC2020_WS_D)
    assert np.float32(st.chrominance) == ref[0][0] and np.float32(st.chrominance_red_green) == ref[0][1]
    d_img = [torch.empty((h - 8, w - 8), dtype=torch.float32, device="cuda") for _ in range(3)]
    img = capi.RGB(*[capi.device_plane(t) for t in d_img])
    gpu_ctx.get_image(dem, 4, 4, MUL, True, MAT, img)
    curve, _ = capi.noise_curve_lut()
    tp = capi.DenoiseToolParams(dn, 0, 3, 0, 80)
    gpu_ctx.improc_denoise(img, tp, O.REC2020_WS_D, ecomp=0.3, calclum_mat=MAT, noise_c_curve=curve, iws=O.REC2020_IWS_D)
    gpu_ctx.exposure(img, float(np.float32(2.0 ** 0.3)), 0.0)
    gpu_ctx.tone_curve(img, lut, 1.0, True)
    gpu_ctx.synchronize()
    for g, t in zip(got, d_img):
        assert np.array_equal(g.view(np.uint32), t.cpu().numpy().view(np.uint32))


@pytest.mark.parametrize("lanes", [2, 3])
def test_batch_lanes_give_the_same_bits(lanes):
    """artgpu_set_batch_lanes: frames in flight on sibling contexts / streams / host threads; results identical to one-by-one."""
    w, h = 392, 296
    lut = _lut()
    p = _params(lut, 1)
    seeds = (11, 12, 13, 14, 15)
    raws = [synth.bayer_frame(w, h, synth.FILTERS_RGGB, seed=s, noise=1500) for s in seeds]
    ctx = capi.Context(0)
    try:
        ref = [[np.zeros((h - 8, w - 8), np.float32) for _ in range(3)] for _ in raws]
        ctx.batch_run([capi.host_plane(r) for r in raws], p, [capi.host_rgb(o) for o in ref])
        ctx.set_batch_lanes(lanes)
        for _ in range(2):          # twice: the lanes are created once and reused
            outs = [[np.zeros((h - 8, w - 8), np.float32) for _ in range(3)] for _ in raws]
            ctx.batch_run([capi.host_plane(r) for r in raws], p, [capi.host_rgb(o) for o in outs])
            for o, r in zip(outs, ref):
                for a, b in zip(o, r):
                    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and a.max() > 0
        # device-resident frames too
        d_raws = [torch.from_numpy(r).cuda() for r in raws]
        d_outs = [[torch.zeros((h - 8, w - 8), dtype=torch.float32, device="cuda") for _ in range(3)] for _ in raws]
        torch.cuda.synchronize()
        ctx.batch_run([capi.device_plane(t) for t in d_raws], p, [capi.RGB(*[capi.device_plane(t) for t in o]) for o in d_outs])
        ctx.synchronize()
        for o, r in zip(d_outs, ref):
            for a, b in zip(o, r):
                assert np.array_equal(a.cpu().numpy().view(np.uint32), b.view(np.uint32))
        with pytest.raises(capi.ArtGpuError):
            ctx.set_batch_lanes(0)
    finally:
        ctx.close()


def test_batch_lanes_inherit_the_context_settings():
    """The lanes are the context as far as the caller can tell: a curve tail above 65535 (artgpu_set_curve_tail) and the options apply to
    every frame whichever lane runs it -- with white_point > 1 and bright frames the output of a lane that missed the setting would differ
    (or the call would fail: the default tail kind refuses white_point > 1)."""
    w, h = 392, 296
    lut = _lut()
    p = _params(lut, 0)
    p.white_point = 1.6
    raws = [np.minimum(synth.bayer_frame(w, h, synth.FILTERS_RGGB, seed=s, noise=1500) * np.float32(1.7), np.float32(65535.0)) for s in (21, 22, 23, 24, 25, 26, 27)]
    ctx = capi.Context(0)
    try:
        ctx.set_curve_tail(1, 0.93)
        ctx.set_option("lut_lds", 0)
        ref = [[np.zeros((h - 8, w - 8), np.float32) for _ in range(3)] for _ in raws]
        ctx.batch_run([capi.host_plane(r) for r in raws], p, [capi.host_rgb(o) for o in ref])
        ctx.set_batch_lanes(3)
        outs = [[np.zeros((h - 8, w - 8), np.float32) for _ in range(3)] for _ in raws]
        ctx.batch_run([capi.host_plane(r) for r in raws], p, [capi.host_rgb(o) for o in outs])
        for o, r in zip(outs, ref):
            for a, b in zip(o, r):
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and a.max() > 0
        # the setting changes between calls: the lanes follow
        ctx.set_curve_tail(2, 1.0)
        ctx.set_batch_lanes(1)
        ref2 = [[np.zeros((h - 8, w - 8), np.float32) for _ in range(3)] for _ in raws]
        ctx.batch_run([capi.host_plane(r) for r in raws], p, [capi.host_rgb(o) for o in ref2])
        ctx.set_batch_lanes(3)
        outs2 = [[np.zeros((h - 8, w - 8), np.float32) for _ in range(3)] for _ in raws]
        ctx.batch_run([capi.host_plane(r) for r in raws], p, [capi.host_rgb(o) for o in outs2])
        for o, r in zip(outs2, ref2):
            for a, b in zip(o, r):
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        assert any((a != b).any() for o, r in zip(ref, ref2) for a, b in zip(o, r))        # the tail does show in these frames
    finally:
        ctx.close()


def test_batch_complete_over_rccl(gpu_ctx):
    """artgpu_batch_complete: the completion all-gather of a multi-GPU batch over the caller's RCCL communicator.  One GPU here, so the
    communicator has one rank (created through librccl's C API the way a host application would); the two-rank exchange itself is
    RCCL's, the N > 1 bookkeeping of bench.py is covered on CPU by tests/test_multiproc.py."""
    import ctypes as C
    rec = [0, 3, 0, 0x1234567890ABCDEF - (1 << 64) if 0x1234567890ABCDEF >= (1 << 63) else 0x1234567890ABCDEF, 16424, 0, 0, 0]
    assert gpu_ctx.batch_complete(rec) == [rec]                       # no communicator: single rank
    with pytest.raises(capi.ArtGpuError):
        gpu_ctx.batch_complete(rec, nranks=2)                         # two ranks need one
    try:
        rccl = C.CDLL("librccl.so.1")
    except OSError:
        rccl = C.CDLL("/opt/rocm/lib/librccl.so.1")

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        assert gpu_ctx.batch_complete(rec, nranks=1, rccl_comm=comm) == [rec]
        rec2 = [0, 5, 0, -7, 99, 1, 2, 3]
        assert gpu_ctx.batch_complete(rec2, nranks=1, rccl_comm=comm) == [rec2]
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)


def test_progress_callback_and_roctx_ranges(gpu_ctx):
    """artgpu_set_progress_callback (rtengine::ProgressListener): every stage-level entry point reports (name of the reference function,
    0.0) when it starts and (name, 1.0) when it returns, nested the way the reference's call tree is; with the "roctx" option the same
    names label roctx ranges (must not change any result)."""
    w, h, filt = 330, 270, synth.FILTERS_RGGB
    raw = synth.bayer_frame(w, h, filt, seed=5)
    ref = gpu_ctx.demosaic_bayer_host(capi.BAYER_RCD, raw, filt, 1.0, 4)
    events = []
    gpu_ctx.set_progress_callback(lambda stage, frac: events.append((stage, frac)))
    gpu_ctx.set_option("roctx", 1)
    try:
        got = gpu_ctx.demosaic_bayer_host(capi.BAYER_RCD, raw, filt, 1.0, 4)
    finally:
        gpu_ctx.set_progress_callback(None)
        gpu_ctx.set_option("roctx", 0)
    assert events == [("RawImageSource::demosaic (Bayer)", 0.0), ("RawImageSource::demosaic (Bayer)", 1.0)]
    for a, b in zip(got, ref):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    n = len(events)
    gpu_ctx.demosaic_bayer_host(capi.BAYER_RCD, raw, filt, 1.0, 4)
    assert len(events) == n                                       # removed
