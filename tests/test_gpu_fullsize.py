"""GPU parity at BASELINE.json's full sizes (45 MP Bayer, 100 MP X-Trans): the whole frame against the oracle, bit for
bit (the oracle runs with OpenMP on the GPU box's host cores; a few tens of seconds per test)."""
import numpy as np
import pytest
import torch

import oracle_lib as O
from art_amd import capi, synth

pytestmark = pytest.mark.gpu
MUL = (2.1374, 1.0, 1.5918)
MAT = np.array([[0.6325, 0.2312, 0.0921], [0.2198, 0.7712, 0.0090], [0.0166, 0.0713, 0.7514]])


def _dev_planes(h, w):
    t = [torch.empty((h, w), dtype=torch.float32, device="cuda") for _ in range(3)]
    return t, capi.RGB(*[capi.device_plane(x) for x in t])


def test_config2_and_3_stages_45mp_bit_exact(gpu_ctx):
    W, H = 8192, 5464
    raw = synth.bayer_frame(W, H, synth.FILTERS_RGGB, seed=0)
    d_raw = torch.from_numpy(raw).cuda()
    d_out, out = _dev_planes(H, W)
    gpu_ctx.demosaic_bayer(capi.BAYER_AMAZE, capi.device_plane(d_raw), synth.FILTERS_RGGB, 1.0, 4, out)
    gpu_ctx.synchronize()
    ref = O.amaze(raw, synth.FILTERS_RGGB, 1.0, 4)
    for t, r in zip(d_out, ref):
        assert np.array_equal(t.cpu().numpy().view(np.uint32), r.view(np.uint32))          # BASELINE configs[1]
    # configs[2] stages on top (DCT detail recovery skipped: it is the one tolerance-checked stage)
    d_img, img = _dev_planes(H - 8, W - 8)
    gpu_ctx.get_image(out, 4, 4, MUL, True, MAT, img)
    curve, _ = capi.noise_curve_lut()
    tp = capi.DenoiseToolParams(capi.DenoiseParams(40.0, 50.0, 0, 15.0, 0.0, 0.0, 1.7, 0, 0, 0), 0, 3, 0, 80)
    gpu_ctx.improc_denoise(img, tp, O.REC2020_WS_D, ecomp=0.3, calclum_mat=MAT, noise_c_curve=curve, flags=capi.DN_SKIP_DETAIL_RECOVERY)
    gpu_ctx.exposure(img, float(np.float32(2.0 ** 0.3)), 0.0)
    x = np.arange(65536, dtype=np.float64) / 65535.0
    lut = ((1.0 - np.cos(np.pi * x ** 0.7)) / 2.0 * 65535.0).astype(np.float32)
    gpu_ctx.tone_curve(img, lut, 1.0, True)
    gpu_ctx.synchronize()
    o = O.get_image(ref, 4, 4, W - 8, H - 8, MUL, True)
    o = O.convert_color_space(o, MAT)
    o = O.improc_denoise(o, calclum_mat=MAT, noise_c_curve=curve, smoothing=False, ecomp=0.3, detail_recovery=False)
    o = O.exposure(o, float(np.float32(2.0 ** 0.3)), 0.0)
    o = O.tone_std(o, lut, 1.0, True)
    for t, r in zip(d_img, o):
        assert np.array_equal(t.cpu().numpy().view(np.uint32), r.view(np.uint32))
    del d_img, o

    # ... and the stage the benchmark times but the check above skips: ImProcFunctions::denoise WITH the DCT detail recovery at full
    # size, against the checker's double-accumulated DCT, under the bound of tests/test_gpu_denoise.py
    from test_gpu_denoise import DCT_ABS_BOUND, DCT_MEDIAN_BOUND
    d_img, img = _dev_planes(H - 8, W - 8)
    gpu_ctx.get_image(out, 4, 4, MUL, True, MAT, img)
    gpu_ctx.improc_denoise(img, tp, O.REC2020_WS_D, ecomp=0.3, calclum_mat=MAT, noise_c_curve=curve)
    gpu_ctx.synchronize()
    o = O.get_image(ref, 4, 4, W - 8, H - 8, MUL, True)
    o = O.convert_color_space(o, MAT)
    o = O.improc_denoise(o, calclum_mat=MAT, noise_c_curve=curve, smoothing=False, ecomp=0.3, detail_recovery=True)
    for t, r in zip(d_img, o):
        err = np.abs(t.cpu().numpy().astype(np.float64) - r.astype(np.float64))
        print(f"45 MP denoise incl. detail recovery: max |device - exact| {err.max():.4f}, p99.9 {np.percentile(err[::7, ::5], 99.9):.4f}, "
              f"bit-identical {100.0 * (err == 0).mean():.1f} %")
        assert err.max() <= DCT_ABS_BOUND and np.median(err[::7, ::5]) <= DCT_MEDIAN_BOUND


def test_rcd_45mp_bit_exact(gpu_ctx):
    """BASELINE configs[1] with RCD: the whole 8192x5464 frame (1504 reference tiles, partial ones at the right / bottom edge)"""
    W, H = 8192, 5464
    raw = synth.bayer_frame(W, H, synth.FILTERS_GRBG, seed=3)
    d_raw = torch.from_numpy(raw).cuda()
    d_out, out = _dev_planes(H, W)
    gpu_ctx.demosaic_bayer(capi.BAYER_RCD, capi.device_plane(d_raw), synth.FILTERS_GRBG, 1.0, 4, out)
    gpu_ctx.synchronize()
    ref = O.rcd(raw, synth.FILTERS_GRBG)
    for t, r in zip(d_out, ref):
        assert np.array_equal(t.cpu().numpy().view(np.uint32), r.view(np.uint32))


def test_config4_stages_45mp_bit_exact_without_dct(gpu_ctx):
    """BASELINE configs[3]'s per-frame pipe at full size -- AMaZE, getImage + matrix, ImProcFunctions::denoise with guided chroma smoothing
    and NL-means, exposure, tone curve -- with the DCT detail-recovery stage switched off on both sides: everything else is bit for bit."""
    W, H = 8192, 5464
    raw = synth.bayer_frame(W, H, synth.FILTERS_RGGB, seed=1)
    d_raw = torch.from_numpy(raw).cuda()
    d_out, out = _dev_planes(H, W)
    gpu_ctx.demosaic_bayer(capi.BAYER_AMAZE, capi.device_plane(d_raw), synth.FILTERS_RGGB, 1.0, 4, out)
    d_img, img = _dev_planes(H - 8, W - 8)
    gpu_ctx.get_image(out, 4, 4, MUL, True, MAT, img)
    curve, _ = capi.noise_curve_lut()
    tp = capi.DenoiseToolParams(capi.DenoiseParams(40.0, 50.0, 0, 15.0, 0.0, 0.0, 1.7, 0, 0, 0), 1, 3, 50, 80)
    gpu_ctx.improc_denoise(img, tp, O.REC2020_WS_D, ecomp=0.3, calclum_mat=MAT, noise_c_curve=curve, flags=capi.DN_SKIP_DETAIL_RECOVERY)
    gpu_ctx.exposure(img, float(np.float32(2.0 ** 0.3)), 0.0)
    x = np.arange(65536, dtype=np.float64) / 65535.0
    lut = ((1.0 - np.cos(np.pi * x ** 0.7)) / 2.0 * 65535.0).astype(np.float32)
    gpu_ctx.tone_curve(img, lut, 1.0, True)
    gpu_ctx.synchronize()
    ref = O.amaze(raw, synth.FILTERS_RGGB, 1.0, 4)
    o = O.get_image(ref, 4, 4, W - 8, H - 8, MUL, True)
    o = O.convert_color_space(o, MAT)
    o = O.improc_denoise(o, calclum_mat=MAT, noise_c_curve=curve, smoothing=True, radius=3, nl_strength=50, nl_detail=80, ecomp=0.3,
                         detail_recovery=False)
    o = O.exposure(o, float(np.float32(2.0 ** 0.3)), 0.0)
    o = O.tone_std(o, lut, 1.0, True)
    for t, r in zip(d_img, o):
        assert np.array_equal(t.cpu().numpy().view(np.uint32), r.view(np.uint32))


def test_config5_xtrans_and_ftblockdn_100mp_bit_exact(gpu_ctx):
    W, H = 11648, 8736
    raw = synth.xtrans_frame(W, H, seed=0)
    d_raw = torch.from_numpy(raw).cuda()
    d_out, out = _dev_planes(H, W)
    gpu_ctx.demosaic_xtrans(3, True, capi.device_plane(d_raw), synth.XTRANS_FUJI, synth.XTRANS_RGB_CAM, out)
    gpu_ctx.synchronize()
    ref = O.xtrans_demosaic(raw, synth.XTRANS_FUJI, synth.XTRANS_RGB_CAM, 3, True)
    for t, r in zip(d_out, ref):
        assert np.array_equal(t.cpu().numpy().view(np.uint32), r.view(np.uint32))
    # ... and the rest of BASELINE configs[4] on top, at full size: getImage + matrix (border 7), ImProcFunctions::denoise (FTblockDN
    # wavelet shrinkage luma + chroma; the tolerance-checked DCT stage off on both sides), exposure, tone curve -- bit for bit
    iw, ih = W - 14, H - 14
    d_img, img = _dev_planes(ih, iw)
    gpu_ctx.get_image(out, 7, 7, MUL, True, MAT, img)
    curve, _ = capi.noise_curve_lut()
    tp = capi.DenoiseToolParams(capi.DenoiseParams(40.0, 50.0, 0, 15.0, 0.0, 0.0, 1.7, 0, 0, 0), 0, 3, 0, 80)
    gpu_ctx.improc_denoise(img, tp, O.REC2020_WS_D, ecomp=0.3, calclum_mat=MAT, noise_c_curve=curve, flags=capi.DN_SKIP_DETAIL_RECOVERY)
    gpu_ctx.exposure(img, float(np.float32(2.0 ** 0.3)), 0.0)
    x = np.arange(65536, dtype=np.float64) / 65535.0
    lut = ((1.0 - np.cos(np.pi * x ** 0.7)) / 2.0 * 65535.0).astype(np.float32)
    gpu_ctx.tone_curve(img, lut, 1.0, True)
    gpu_ctx.synchronize()
    del d_out
    o = O.get_image(ref, 7, 7, iw, ih, MUL, True)
    del ref
    o = O.convert_color_space(o, MAT)
    o = O.improc_denoise(o, calclum_mat=MAT, noise_c_curve=curve, smoothing=False, ecomp=0.3, detail_recovery=False)
    o = O.exposure(o, float(np.float32(2.0 ** 0.3)), 0.0)
    o = O.tone_std(o, lut, 1.0, True)
    for t, r in zip(d_img, o):
        assert np.array_equal(t.cpu().numpy().view(np.uint32), r.view(np.uint32))


def test_config5_xtrans_one_pass_100mp_bit_exact(gpu_ctx):
    """BASELINE.md C5 names both Markesteijn methods: ONE_PASS (YPbPr homogeneity, xtrans_demosaic.cc:688-741) at the full 100 MP size
    (the THREE_PASS leg is the test above; until round 6 the 1-pass path was only exercised up to 2400 x 2350)."""
    W, H = 11648, 8736
    raw = synth.xtrans_frame(W, H, seed=1)
    d_raw = torch.from_numpy(raw).cuda()
    d_out, out = _dev_planes(H, W)
    gpu_ctx.demosaic_xtrans(1, False, capi.device_plane(d_raw), synth.XTRANS_FUJI, synth.XTRANS_RGB_CAM, out)
    gpu_ctx.synchronize()
    ref = O.xtrans_demosaic(raw, synth.XTRANS_FUJI, synth.XTRANS_RGB_CAM, 1, False)
    for t, r in zip(d_out, ref):
        assert np.array_equal(t.cpu().numpy().view(np.uint32), r.view(np.uint32))


def test_nlmeans_12mp_bit_exact(gpu_ctx):
    """NL-means (v3 kernel: workgroup per 150x150 reference tile, 27x20 tiles incl. partial ones) on a 12 MP luminance plane."""
    W, H = 4000, 3000
    rng = np.random.default_rng(17)
    y, x = np.mgrid[0:H, 0:W].astype(np.float32)
    img = (18000 + 9000 * np.sin(0.004 * x) * np.cos(0.005 * y) + 5000 * ((x.astype(np.int32) // 96 + y.astype(np.int32) // 96) % 2)
           + rng.normal(0, 900, (H, W))).clip(0, 65535).astype(np.float32)
    got = img.copy()
    gpu_ctx.nlmeans(capi.host_plane(got), 50, 80, 1.0)
    ref = O.nlmeans(img, 50, 80, 1.0)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    assert np.abs(got - img).mean() > 1.0
