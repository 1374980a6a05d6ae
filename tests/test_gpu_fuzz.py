"""Fixed-seed slices of the fuzzers under scripts/ (fuzz_demosaic.py, fuzz_xtrans.py, fuzz_sizes.py), small enough for the GPU test
run: random frame sizes -- sliver tiles and odd sizes included --, CFA phases, noise levels, pass counts and colour-map phases through
the streaming demosaicers, the X-Trans kernel, the YvV gaussian and the config-4 chain, device vs oracle bit for bit."""
import numpy as np
import pytest

import oracle_lib as O
from art_amd import capi, synth

pytestmark = pytest.mark.gpu
FILTERS = [synth.FILTERS_RGGB, synth.FILTERS_BGGR, synth.FILTERS_GRBG, synth.FILTERS_GBRG]
MUL = (2.1374, 1.0, 1.5918)
MAT = np.array([[0.6325, 0.2312, 0.0921], [0.2198, 0.7712, 0.0090], [0.0166, 0.0713, 0.7514]])


def _same(a, b):
    return all(np.array_equal(x.view(np.uint32), y.view(np.uint32)) for x, y in zip(a, b))


def _bayer_cases(n, seed):
    rng = np.random.default_rng(seed)
    cases = []
    for it in range(n):
        w, h = int(rng.integers(64, 900)), int(rng.integers(64, 700))
        if it % 3 == 0:                 # widths / heights that leave a sliver tile (RCD: 176 k + 17 .. 30; AMaZE: 128 k - 15 .. 16)
            w = 176 * int(rng.integers(1, 5)) + int(rng.integers(17, 31)) if it % 2 else 128 * int(rng.integers(1, 6)) + int(rng.integers(-15, 17))
        if it % 4 == 0:
            h = 176 * int(rng.integers(1, 4)) + int(rng.integers(17, 31)) if it % 2 else 128 * int(rng.integers(1, 5)) + int(rng.integers(-15, 17))
        cases.append((it, w, h, "rcd" if it % 2 else "amaze", FILTERS[it % 4], [0, 64, 1500, 6000][(it // 4) % 4]))
    return cases


@pytest.mark.parametrize("it,w,h,method,filt,noise", _bayer_cases(10, 20260927))
def test_fuzz_streaming_demosaicers(gpu_ctx, it, w, h, method, filt, noise):
    raw = synth.bayer_frame(w, h, filt, seed=500 + it, noise=noise)
    try:
        if method == "rcd":
            gpu_ctx.set_option("rcd_rows", 8 if it % 4 == 1 else 4)
        ref = O.rcd(raw, filt) if method == "rcd" else O.amaze(raw, filt, 1.0, 4)
        got = gpu_ctx.demosaic_bayer_host(capi.BAYER_RCD if method == "rcd" else capi.BAYER_AMAZE, raw, filt, 1.0, 4)
    finally:
        gpu_ctx.set_option("rcd_rows", 8)
    assert _same(got, ref), f"{w}x{h} {method} filters={filt:#x} noise {noise}"


def _xtrans_cases(n, seed):
    rng = np.random.default_rng(seed)
    cases = []
    for it in range(n):
        w, h = int(rng.integers(64, 800)), int(rng.integers(64, 700))
        if it % 3 == 0:                 # sizes that leave a sliver tile (tile stride 98, origin 3)
            w = 98 * int(rng.integers(1, 7)) + int(rng.integers(20, 40))
        if it % 4 == 0:
            h = 98 * int(rng.integers(1, 6)) + int(rng.integers(20, 40))
        roll = (int(rng.integers(0, 6)), int(rng.integers(0, 6))) if it % 3 == 0 else (0, 0)
        cases.append((it, w, h, [1, 3, 2, 3][it % 4], bool((it // 2) % 2), roll, [0, 300, 1500, 6000][(it // 4) % 4],
                      float(rng.choice([0.7, 1.0, 2.0, 3.3, 7.5, 12.0]))))
    return cases


@pytest.mark.parametrize("it,w,h,passes,lab,roll,noise,sigma", _xtrans_cases(6, 20260928))
def test_fuzz_xtrans_and_gaussian(gpu_ctx, it, w, h, passes, lab, roll, noise, sigma):
    xt = np.roll(np.roll(synth.XTRANS_FUJI, roll[0], axis=0), roll[1], axis=1)
    raw = synth.bayer_frame(w, h, 0, 700 + it, noise, True, True, xtrans=xt)
    out = [np.full((h, w), -1.0, np.float32) for _ in range(3)]
    gpu_ctx.demosaic_xtrans(passes, lab, capi.host_plane(raw), xt, synth.XTRANS_RGB_CAM, capi.host_rgb(out))
    ref = O.xtrans_demosaic(raw, xt, synth.XTRANS_RGB_CAM, passes, lab)
    assert _same(out, ref), f"xtrans {w}x{h} passes {passes} lab {lab} roll {roll} noise {noise}"
    img = np.ascontiguousarray(ref[1])      # the YvV gaussian on the green plane of the result
    got = img.copy()
    gpu_ctx.gaussian_blur(capi.host_plane(got), sigma)
    assert np.array_equal(got.view(np.uint32), O.gaussian_blur(img, sigma).view(np.uint32)), f"gaussian {w}x{h} sigma {sigma}"


def _chain_cases(n, seed):
    rng = np.random.default_rng(seed)
    return [(it, int(rng.integers(64, 500)) * 2 + (it % 2) * int(rng.integers(0, 2)), int(rng.integers(64, 400)) * 2 + (it % 2) * int(rng.integers(0, 2)),
             "rcd" if it % 3 == 2 else "amaze", FILTERS[it % 4]) for it in range(n)]


@pytest.mark.parametrize("it,w,h,method,filt", _chain_cases(4, 20260929))
def test_fuzz_config4_chain(gpu_ctx, it, w, h, method, filt):
    """demosaic -> getImage + matrix -> ImProcFunctions::denoise with guided smoothing and NL-means (DCT stage skipped) -> exposure -> tone"""
    x = np.arange(65536, dtype=np.float64) / 65535.0
    lut = ((1.0 - np.cos(np.pi * x ** 0.7)) / 2.0 * 65535.0).astype(np.float32)
    curve, _ = capi.noise_curve_lut()
    raw = synth.bayer_frame(w, h, filt, seed=100 + it, noise=1500)
    planes = O.rcd(raw, filt) if method == "rcd" else O.amaze(raw, filt, 1.0, 4)
    got_p = [np.zeros((h, w), np.float32) for _ in range(3)]
    gpu_ctx.demosaic_bayer(capi.BAYER_RCD if method == "rcd" else capi.BAYER_AMAZE, capi.host_plane(raw), filt, 1.0, 4, capi.host_rgb(got_p))
    assert _same(got_p, planes), "demosaic"
    iw, ih = w - 8, h - 8
    img = [np.zeros((ih, iw), np.float32) for _ in range(3)]
    gpu_ctx.get_image(capi.host_rgb(got_p), 4, 4, MUL, True, MAT, capi.host_rgb(img))
    tp = capi.DenoiseToolParams(capi.DenoiseParams(40.0, 50.0, 0, 15.0, 0.0, 0.0, 1.7, 0, 0, 0), 1, 3, 50, 80)
    gpu_ctx.improc_denoise(capi.host_rgb(img), tp, O.REC2020_WS_D, ecomp=0.3, calclum_mat=MAT, noise_c_curve=curve, flags=capi.DN_SKIP_DETAIL_RECOVERY)
    gpu_ctx.exposure(capi.host_rgb(img), float(np.float32(2.0 ** 0.3)), 0.0)
    gpu_ctx.tone_curve(capi.host_rgb(img), lut, 1.0, True)
    o = O.get_image(planes, 4, 4, iw, ih, MUL, True)
    o = O.convert_color_space(o, MAT)
    o = O.improc_denoise(o, calclum_mat=MAT, noise_c_curve=curve, smoothing=True, radius=3, nl_strength=50, nl_detail=80, ecomp=0.3, detail_recovery=False)
    o = O.exposure(o, float(np.float32(2.0 ** 0.3)), 0.0)
    o = O.tone_std(o, lut, 1.0, True)
    assert _same(img, o), f"{w}x{h} {method} filters={filt:#x}"


def test_fuzz_nlmeans_sizes_on_one_context(gpu_ctx):
    """A fixed-seed slice of scripts/fuzz_nlm.py plus the cases it found, all on ONE context in this order: a frame with a sliver tile at the
    right edge (padded width 274 = 2 x 136 + 2: the third tile column is narrower than two borders and writes nothing), then frames whose
    padded area is below the 8192 entries of the exp table (the table's tail used to keep whatever an earlier call had left at that place
    in the pool)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import fuzz_nlm
    fixed = [(260, 406, 20, 50, 1.0, 870025460), (269, 413, 50, 0, 1.0, 479761753), (72, 44, 100, 50, 1.0, 279563955), (34, 48, 50, 0, 1.0, 232415261),
             (40, 100, 50, 0, 1.0, 5), (100, 40, 50, 0, 1.0, 5), (60, 60, 50, 50, 1.0, 5), (399, 32, 50, 0, 1.0, 805697932)]
    for c in fixed + list(fuzz_nlm.cases(7, 10)):
        assert fuzz_nlm.run(gpu_ctx, *c) == 0, c


def _denoise_cases(n, seed):
    """scripts/fuzz_denoise.py's generator: sizes on both sides of what the fused shrink pass takes (bands of 64 x 64 and more), every strength
    incl. 0, preview scale 2 (smaller blur radii), QUALITY_HIGH (per-channel launches, radii above 7)"""
    rng = np.random.default_rng(seed)
    out = []
    for it in range(n):
        w = int(rng.choice([rng.integers(64, 200), rng.integers(200, 700), rng.integers(700, 1100)]))
        h = int(rng.choice([rng.integers(64, 200), rng.integers(200, 600)]))
        out.append((it, w, h, int(rng.choice([64, 2048, 6000])), float(rng.choice([0.0, 5.0, 40.0, 100.0])), float(rng.choice([0.0, 15.0, 60.0, 100.0])),
                    float(rng.choice([0.0, -40.0, 35.0])), float(rng.choice([0.0, 50.0, -25.0])), float(rng.choice([1.0, 1.7, 3.0])),
                    int(rng.integers(0, 2)), float(rng.choice([1.0, 1.0, 2.0]))))
    return out


@pytest.mark.parametrize("it,w,h,noise,lum,chrom,rg,by,gamma,aggressive,scale", _denoise_cases(16, 20260928))
def test_fuzz_rgb_denoise(gpu_ctx, it, w, h, noise, lum, chrom, rg, by, gamma, aggressive, scale):
    raw = synth.bayer_frame(w // 2 * 2, h // 2 * 2, synth.FILTERS_RGGB, seed=900 + it, noise=noise)
    img = [np.ascontiguousarray(p[:h, :w]) for p in O.amaze(raw, synth.FILTERS_RGGB, 1.0, 4)]
    got = [p.copy() for p in img]
    p = capi.DenoiseParams(lum, 50.0, 0, chrom, rg, by, gamma, aggressive, 0, 0)
    gpu_ctx.rgb_denoise(capi.host_rgb(got), p, O.REC2020_WS, scale=scale, flags=capi.DN_SKIP_DETAIL_RECOVERY)
    ref = O.rgb_denoise(img, O.default_denoise_params(luminance=lum, chrominance=chrom, chrominanceRedGreen=rg, chrominanceBlueYellow=by,
                                                      gamma=gamma, aggressive=aggressive, scale=scale))
    assert _same(got, ref), f"{w}x{h} lum {lum} chrom {chrom} aggressive {aggressive} scale {scale}"
