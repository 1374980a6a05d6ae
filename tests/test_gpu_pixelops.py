"""GPU parity of the per-pixel stages (getImage, matrix conversion, exposure, tone STD), bit-exact."""
import numpy as np
import pytest

import oracle_lib

pytestmark = pytest.mark.gpu

# Rec2020 working-space inverse times a plausible xyz_cam (inputs to the stage; any matrix will do)
MAT = np.array([[1.71665119, -0.35567078, -0.25336628],
                [-0.66668435, 1.61648124, 0.01576855],
                [0.01763986, -0.04277061, 0.94210312]]) @ np.array([[0.41, 0.36, 0.18], [0.21, 0.72, 0.07], [0.02, 0.12, 0.95]])


def _img(w, h, seed, lo=-500.0, hi=70000.0):
    rng = np.random.default_rng(seed)
    img = [rng.uniform(lo, hi, size=(h, w)).astype(np.float32) for _ in range(3)]
    # special values: zeros, negative zero, exact 65535, > 65535, equal channels
    img[0][0, :8] = [0.0, -0.0, 65535.0, 65536.0, 1e-30, 3.0, 3.0, 70000.0]
    img[1][0, :8] = [0.0, -0.0, 65535.0, 1.0, 0.0, 3.0, 2.0, 70000.0]
    img[2][0, :8] = [0.0, 0.0, 65535.0, 2.0, 0.0, 3.0, 3.0, 60000.0]
    return img


def _same(a, b):
    return [int((x.view(np.uint32) != y.view(np.uint32)).sum()) for x, y in zip(a, b)]


@pytest.mark.parametrize("w,h,border,clip,use_mat", [(333, 251, 4, False, True), (640, 480, 7, True, True), (257, 129, 0, True, False)])
def test_get_image(gpu_ctx, w, h, border, clip, use_mat):
    from art_amd import capi
    planes = _img(w, h, 1)
    ow, oh = w - 2 * border, h - 2 * border
    mul = (2.1374, 1.0, 1.5918)
    out = [np.full((oh, ow), np.nan, np.float32) for _ in range(3)]
    gpu_ctx.get_image(capi.host_rgb(planes), border, border, mul, clip, MAT if use_mat else None, capi.host_rgb(out))
    ref = oracle_lib.get_image(planes, border, border, ow, oh, mul, clip)
    if use_mat:
        ref = oracle_lib.convert_color_space(ref, MAT)
    assert _same(out, ref) == [0, 0, 0]


def test_convert_exposure_tone(gpu_ctx):
    from art_amd import capi
    w, h = 1027, 333  # W % 4 != 0 exercises the scalar tail of expcomp
    img = _img(w, h, 2)
    ref = [p.copy() for p in img]
    gpu_ctx.convert_color_space(capi.host_rgb(img), MAT)
    ref = oracle_lib.convert_color_space(ref, MAT)
    assert _same(img, ref) == [0, 0, 0]
    es, black = np.float32(2.0) ** np.float32(0.3), np.float32(0.01 * 2000.0)
    gpu_ctx.exposure(capi.host_rgb(img), float(es), float(black))
    ref = oracle_lib.exposure(ref, float(es), float(black))
    assert _same(img, ref) == [0, 0, 0]
    x = np.arange(65536, dtype=np.float64) / 65535.0
    lut = ((x ** 0.8) * (3 - 2 * x ** 0.8) * x ** 0.8 * 0 + (1 - np.cos(np.pi * x ** 0.7)) / 2).astype(np.float32) * np.float32(65535.0)
    gpu_ctx.tone_curve(capi.host_rgb(img), lut, 1.0, True)
    ref = oracle_lib.tone_std(ref, lut, 1.0, True)
    assert _same(img, ref) == [0, 0, 0]


@pytest.mark.parametrize("kind,y_last", [(0, 1.0), (1, 0.93), (2, 1.0)])
@pytest.mark.parametrize("w,h", [(333, 201), (2304, 1800)])          # plain kernel / the kernel with the curve in LDS
def test_tone_std_above_the_lut(gpu_ctx, kind, y_last, w, h):
    """curves::setLutVal for values above 65535 (whitePoint > 1): what the Curve object returns there -- nothing (no Curve: LUT clip),
    the last point's y (Linear / Spline / CatmullRom) or t (Empty / NURBS) -- reaches the device through artgpu_set_curve_tail."""
    from art_amd import capi
    import oracle_lib as O
    rng = np.random.default_rng(kind + w)
    img = [rng.uniform(-500, 110000, (h, w)).astype(np.float32) for _ in range(3)]
    img[0][0, :6] = [65535.0, 65535.004, 65536.0, 98302.5, 98303.0, 3e6]
    x = np.arange(65536, dtype=np.float64) / 65535.0
    lut = ((1.0 - np.cos(np.pi * x ** 0.7)) / 2.0 * 0.93 * 65535.0).astype(np.float32)
    for clip in (True, False):
        O.set_curve_tail(kind, y_last)
        try:
            ref = O.tone_std(img, lut, 1.5, clip)
        finally:
            O.set_curve_tail(0)
        got = [p.copy() for p in img]
        gpu_ctx.set_curve_tail(kind, y_last)
        try:
            gpu_ctx.tone_curve(capi.host_rgb(got), lut, 1.5, clip)
        finally:
            gpu_ctx.set_curve_tail(3)
        assert _same(got, ref) == [0, 0, 0]
        assert (np.array(ref) > 65535.0).any() == (kind == 2)        # only the identity tail leaves values above the LUT range


PARAMETRIC = [2.0, 0.25, 0.5, 0.75, 30.0, 20.0, -15.0, -25.0, 0.0]        # DCT_Parametric, zone boundaries, highlights / lights / darks / shadows


@pytest.mark.parametrize("w,h", [(333, 201), (2304, 1800)])          # plain kernel / the kernel with the curve in LDS
def test_tone_std_above_the_lut_parametric_curve(gpu_ctx, w, h):
    """A DCT_Parametric tone curve with whitePoint > 1: values above 65535 take DiagonalCurve::getVal's analytic form
    (diagonalcurves.cc:448-470, double-precision sleef) on the device, bit for bit with the oracle's restatement."""
    from art_amd import capi
    import oracle_lib as O
    rng = np.random.default_rng(w)
    img = [rng.uniform(-500, 160000, (h, w)).astype(np.float32) for _ in range(3)]
    img[0][0, :6] = [65535.0, 65535.004, 65536.0, 98302.5, 98303.0, 3e6]
    x = np.arange(65536, dtype=np.float64) / 65535.0
    lut = ((1.0 - np.cos(np.pi * x ** 0.7)) / 2.0 * 65535.0).astype(np.float32)
    for clip in (True, False):
        O.set_parametric_curve(PARAMETRIC)
        try:
            ref = O.tone_std(img, lut, 2.0, clip)
        finally:
            O.set_curve_tail(0)
        base = O.tone_std(img, lut, 2.0, clip)
        got = [p.copy() for p in img]
        gpu_ctx.set_curve_tail_parametric(PARAMETRIC)
        try:
            gpu_ctx.tone_curve(capi.host_rgb(got), lut, 2.0, clip)
        finally:
            gpu_ctx.set_curve_tail(3)
        assert _same(got, ref) == [0, 0, 0]
        assert any((b != r).any() for b, r in zip(base, ref))          # the tail was reached
    with pytest.raises(capi.ArtGpuError):
        gpu_ctx.set_curve_tail_parametric([2.0, 0.25, 0.5, 0.75, 0.0, 0.0, 0.0, 0.0])        # the identity: no Curve object
    gpu_ctx.set_curve_tail(3)


def test_tone_unsupported(gpu_ctx):
    from art_amd import capi
    img = _img(64, 64, 3)
    with pytest.raises(capi.ArtGpuError):
        gpu_ctx.tone_curve(capi.host_rgb(img), None, 1.5, True)      # whitept > 1 and a Curve the device cannot evaluate (default)
    with pytest.raises(capi.ArtGpuError):
        gpu_ctx.tone_curve(capi.host_rgb(img), None, 1.0, True, mode=6)  # NEUTRAL: not built yet


@pytest.mark.parametrize("kind", ["bayer_u16", "xtrans_u16", "bayer_f32"])
def test_scale_colors_bit_exact(gpu_ctx, kind):
    """N2: copyOriginalPixels + scaleColors (rawimagesource.cc:2739-2760,2806-2813): black subtraction, per-channel scale, maxima."""
    from art_amd import capi, synth
    import oracle_lib as O
    w, h = 515, 389
    rng = np.random.default_rng(7)
    data = rng.integers(0, 16384, (h, w)).astype(np.uint16)
    if kind == "bayer_f32":
        data = data.astype(np.float32) + np.float32(0.25)
    bayer = not kind.startswith("xtrans")
    filt = synth.FILTERS_GRBG
    cfa = synth.XTRANS_FUJI if not bayer else np.array([[synth.fc(filt, r, c) for c in range(6)] for r in range(6)], np.int32)
    black = (511.0, 512.5, 509.0, 513.0)
    mul = (2.31, 1.0, 1.57, 1.02)
    out = np.zeros((h, w), np.float32)
    mx = gpu_ctx.scale_colors(data, filt, None if bayer else synth.XTRANS_FUJI, black, mul, capi.host_plane(out))
    ref, rmx = O.scale_colors(data, cfa, bayer, black, mul)
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))
    assert [np.float32(v) for v in mx] == [np.float32(v) for v in rmx] and mx[3] == mx[1] and mx[0] > 0


def test_channel_mixer_and_rgb_curves_bit_exact(gpu_ctx):
    """N4: pixel loops of ImProcFunctions::channelMixer (ipchmixer.cc:200-230) and rgbCurves (iprgbcurves.cc:116-143)."""
    from art_amd import capi
    import oracle_lib as O
    w, h = 333, 127                                    # W % 4 != 0: vector groups + scalar tail
    rng = np.random.default_rng(3)
    img = [rng.uniform(-500.0, 70000.0, (h, w)).astype(np.float32) for _ in range(3)]
    m = np.array([1.1, -0.2, 0.1, -0.05, 1.2, -0.15, 0.02, -0.3, 1.28], np.float32)
    got = [p.copy() for p in img]
    gpu_ctx.channel_mixer(capi.host_rgb(got), m)
    ref = O.channel_mixer(img, m)
    for g, r in zip(got, ref):
        assert np.array_equal(g.view(np.uint32), r.view(np.uint32))
    x = np.arange(65536, dtype=np.float64) / 65535.0
    lr = (65535.0 * x ** 0.8).astype(np.float32)
    lb = (65535.0 * (1.0 - (1.0 - x) ** 1.3)).astype(np.float32)
    got = [p.copy() for p in img]
    gpu_ctx.rgb_curves(capi.host_rgb(got), lr, None, lb)
    ref = O.rgb_curves(img, (lr, None, lb))
    for g, r in zip(got, ref):
        assert np.array_equal(g.view(np.uint32), r.view(np.uint32))
    assert np.array_equal(got[1], img[1]) and not np.array_equal(got[0], img[0])


def test_rgb_curves_cache_survives_other_users_of_the_context(gpu_ctx):
    """The three rgbCurves tables are uploaded once and remembered on the host side (a curve that comes back unchanged is not sent
    again).  The per-image order of rtengine_gpu.h is labAdjustments, then rgbCurves, on ONE context, and rgb2out / the pipeline share the
    context too: none of them may disturb what the remembered tables describe (they once shared a pool slot with all three)."""
    from art_amd import capi
    import oracle_lib as O
    rng = np.random.default_rng(31)
    w, h = 257, 95
    img = [rng.uniform(-500.0, 70000.0, (h, w)).astype(np.float32) for _ in range(3)]
    x = np.arange(65536, dtype=np.float64) / 65535.0
    lr = (65535.0 * x ** 0.7).astype(np.float32)
    lg = (65535.0 * x ** 1.2).astype(np.float32)
    lb = (65535.0 * (1.0 - (1.0 - x) ** 1.4)).astype(np.float32)
    ref = O.rgb_curves(img, (lr, lg, lb))

    def run_and_check():
        got = [p.copy() for p in img]
        gpu_ctx.rgb_curves(capi.host_rgb(got), lr, lg, lb)
        for g, r in zip(got, ref):
            assert np.array_equal(g.view(np.uint32), r.view(np.uint32))

    run_and_check()
    # labAdjustments' three curves (the slot the tables used to live in)
    lab = O.image_rgb_to_lab([np.abs(p) for p in img])
    lc = np.arange(32770, dtype=np.float32) * np.float32(0.9)
    ac = (65535.0 * np.clip(x + 0.05 * np.sin(2 * np.pi * x), 0, 1)).astype(np.float32)
    lab_io = [p.copy() for p in lab]                    # host_rgb borrows the arrays: they have to outlive the call
    gpu_ctx.lab_adjustments(capi.host_rgb(lab_io), lc, ac, ac, 1.1)
    run_and_check()
    # rgb2out's TRC table
    m = np.array([[1.66, -0.59, -0.07], [-0.12, 1.13, -0.01], [-0.02, -0.10, 1.12]], np.float32)
    t = np.arange(1024, dtype=np.float64) / 1023.0
    trc = np.where(t <= 0.0031308, 12.92 * t, 1.055 * t ** (1 / 2.4) - 0.055).astype(np.float32)
    dim = [(np.abs(p) * 0.4).astype(np.float32) for p in img]
    out = [np.zeros((h, w), np.float32) for _ in range(3)]
    gpu_ctx.rgb2out_matrix(capi.host_rgb(dim), capi.host_rgb(out), m, False, trc)
    run_and_check()
    # the pipeline's planes (it regrows the slot: free + allocate)
    from art_amd import synth
    W, H = 640, 480
    raw = synth.bayer_frame(W, H, synth.FILTERS_RGGB, seed=5, noise=512)
    pp = capi.PipelineParams()
    pp.sensor = 0; pp.bayer_method = capi.BAYER_RCD; pp.filters = synth.FILTERS_RGGB; pp.initial_gain = 1.0; pp.border = 4
    pp.mul[:] = (2.0, 1.0, 1.5); pp.do_clip = 1; pp.scale = 1.0
    outp = [np.zeros((H - 8, W - 8), np.float32) for _ in range(3)]
    gpu_ctx.pipeline_run(capi.host_plane(raw), pp, capi.host_rgb(outp))
    assert float(outp[1].max()) > 0
    run_and_check()
    # a changed curve IS uploaded
    lr2 = (65535.0 * x ** 0.5).astype(np.float32)
    got = [p.copy() for p in img]
    gpu_ctx.rgb_curves(capi.host_rgb(got), lr2, lg, lb)
    for g, r in zip(got, O.rgb_curves(img, (lr2, lg, lb))):
        assert np.array_equal(g.view(np.uint32), r.view(np.uint32))


@pytest.mark.parametrize("skip,crop", [(2, (4, 4, 393, 289)), (3, (4, 4, 392, 288)), (4, (120, 60, 281, 233))])
def test_get_image_skip_bit_exact(gpu_ctx, skip, crop):
    """getImage with PreviewProps::skip > 1 (rawimagesource.cc:940-975): box sums in row-major order, window clamped at the edge."""
    from art_amd import capi
    import oracle_lib as O
    rng = np.random.default_rng(skip)
    H, W = 297, 401
    planes = [rng.uniform(0, 30000, (H, W)).astype(np.float32) for _ in range(3)]
    sx1, sy1, cw, ch = crop
    w, h = (cw + skip - 1) // skip, (ch + skip - 1) // skip        # transformRect L745-747
    mul = [m / (skip * skip) for m in (2.1374, 1.0, 1.5918)]
    got = [np.zeros((h, w), np.float32) for _ in range(3)]
    gpu_ctx.get_image(capi.host_rgb(planes), sx1, sy1, mul, True, None, capi.host_rgb(got), skip=skip)
    ref = O.get_image_skip(planes, sx1, sy1, w, h, skip, mul, True)
    for g, r in zip(got, ref):
        assert np.array_equal(g.view(np.uint32), r.view(np.uint32))


@pytest.mark.parametrize("sat,vib", [(35, 0), (0, 40), (-20, -60), (0, 0)])
def test_saturation_vibrance_bit_exact(gpu_ctx, sat, vib):
    """N4: ImProcFunctions::saturationVibrance (ipsaturation.cc:29-83), sleef pow_F on |chroma|, double luminance row."""
    from art_amd import capi
    import oracle_lib as O
    rng = np.random.default_rng(abs(sat * 100 + vib) + 7)
    img = [rng.uniform(0.0, 66000.0, (131, 203)).astype(np.float32) for _ in range(3)]
    img[0][:4, :4] = img[1][:4, :4] = img[2][:4, :4] = 1234.5          # grey: chroma exactly 0 (below the 2^-16 floor)
    got = [p.copy() for p in img]
    gpu_ctx.saturation_vibrance(capi.host_rgb(got), sat, vib, O.REC2020_WS_D)
    ref = O.saturation_vibrance(img, sat, vib)
    for g, r in zip(got, ref):
        assert np.array_equal(g.view(np.uint32), r.view(np.uint32))
    if sat or vib:
        assert not np.array_equal(got[0], img[0])


def test_rgb2out_fast_path_and_scanlines_bit_exact(gpu_ctx):
    """N1: ARTOutputProfile's matrix + TRC fast path (iprgb2out.cc:152-172) and Imagefloat::getScanline (imagefloat.cc:125-170)."""
    from art_amd import capi
    import oracle_lib as O
    rng = np.random.default_rng(21)
    h, w = 97, 211
    img = [rng.uniform(-200.0, 60000.0, (h, w)).astype(np.float32) for _ in range(3)]
    m = np.array([[1.66, -0.59, -0.07], [-0.12, 1.13, -0.01], [-0.02, -0.10, 1.12]], np.float32)
    # linear profile: exact for every value
    got = [np.zeros((h, w), np.float32) for _ in range(3)]
    gpu_ctx.rgb2out_matrix(capi.host_rgb(img), capi.host_rgb(got), m, True)
    ref, bad = O.rgb2out_matrix(img, m, True)
    assert bad == 0
    for g, r in zip(got, ref):
        assert np.array_equal(g.view(np.uint32), r.view(np.uint32))
    # TRC through a 1024-entry LUT (the preview pipeline's size); dim image so that nothing exceeds 1
    x = np.arange(1024, dtype=np.float64) / 1023.0
    lut = np.where(x <= 0.0031308, 12.92 * x, 1.055 * x ** (1 / 2.4) - 0.055).astype(np.float32)
    dim = [(p * 0.5).astype(np.float32) for p in img]
    got = [np.zeros((h, w), np.float32) for _ in range(3)]
    gpu_ctx.rgb2out_matrix(capi.host_rgb(dim), capi.host_rgb(got), m, False, lut)
    ref, bad = O.rgb2out_matrix(dim, m, False, lut)
    assert bad == 0
    for g, r in zip(got, ref):
        assert np.array_equal(g.view(np.uint32), r.view(np.uint32))
    # values above 1 with a non-linear TRC need lcms2 on the host: loud failure, not a silent approximation
    with pytest.raises(capi.ArtGpuError):
        gpu_ctx.rgb2out_matrix(capi.host_rgb([p * 3 for p in img]), capi.host_rgb(got), m, False, lut)
    # scanlines in all four sample formats
    test = [p.copy() for p in img]
    test[0][0, :8] = [np.nan, -1.0, 0.49, 0.5, 65534.6, 65535.0, 70000.0, 1e-3]
    for bps, fl in ((8, False), (16, False), (16, True), (32, True)):
        a = gpu_ctx.get_scanlines(capi.host_rgb(test), bps, fl)
        b = O.get_scanlines(test, bps, fl)
        if a.dtype == np.float32:
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        else:
            assert np.array_equal(a, b)


@pytest.mark.parametrize("w,h,clip", [(2304, 1900, True), (4099, 1031, False)])
def test_tone_std_large_frame_lds_curve_bit_exact(gpu_ctx, monkeypatch, w, h, clip):
    """frames of >= 4 Mpx take the kernel that keeps the lower 40704 curve entries in LDS: values on both sides of that split, at the
    curve's ends, negative, above 65535 and NaN; same bits as the oracle and as the plain kernel (option "lut_lds" 0)"""
    from art_amd import capi
    import oracle_lib as O
    rng = np.random.default_rng(w)
    img = [rng.uniform(-2000, 70000, (h, w)).astype(np.float32) for _ in range(3)]
    img[0][0, :8] = [40702.5, 40703.0, 40703.5, 40704.0, 65534.0, 65534.5, 65535.0, np.nan]
    img[1][1, :4] = [0.0, -0.0, 1e-30, 3e38]
    x = np.arange(65536, dtype=np.float64) / 65535.0
    lut = ((1.0 - np.cos(np.pi * x ** 0.7)) / 2.0 * 65535.0 + rng.normal(0, 3, x.size)).astype(np.float32)
    ref = O.tone_std(img, lut, 1.0, clip)
    got = [p.copy() for p in img]
    gpu_ctx.tone_curve(capi.host_rgb(got), lut, 1.0, clip)
    gpu_ctx.set_option("lut_lds", 0)
    plain = [p.copy() for p in img]
    try:
        gpu_ctx.tone_curve(capi.host_rgb(plain), lut, 1.0, clip)
    finally:
        gpu_ctx.set_option("lut_lds", 1)
    for g, p, r in zip(got, plain, ref):
        assert np.array_equal(g.view(np.uint32), r.view(np.uint32))
        assert np.array_equal(p.view(np.uint32), r.view(np.uint32))


@pytest.mark.parametrize("w,h", [(343, 211), (2304, 1800)])          # plain kernel / the kernel with the curve in LDS
def test_filmlike_clip_every_branch_of_the_reference_tree(gpu_ctx, w, h):
    """Color::filmlike_clip (color.cc:6650-6688) branches seven ways on the order of the channels.  Every order, every tie (r == g, g == b,
    r == b, all equal), values on both sides of the clip level, zeros, negatives, +-inf and NaN in each channel: the bits of the oracle.
    (Round 5 also measured a one-body form -- the tree's comparisons select a permutation, clip_rgb_tone runs once: same bits on this test,
    17 us faster on random data, 17 us SLOWER on the benchmark frame, whose waves mostly agree on the order: not kept)"""
    from art_amd import capi
    import oracle_lib as O
    rng = np.random.default_rng(7)
    vals = np.array([0.0, -3.0, 1.0, 500.0, 30000.0, 65535.0 * 0.9 - 1, 65535.0 * 0.9, 65535.0 * 0.9 + 1, 65535.0, 70000.0, 2.0e5, np.inf, -np.inf, np.nan], np.float32)
    n = len(vals)
    # the first rows: every (r, g, b) triple of the special values (ties and orders by construction); the rest: random neutral-ish data
    tri = np.array(np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")).reshape(3, -1)
    img = [rng.uniform(0.0, 75000.0, (h, w)).astype(np.float32) for _ in range(3)]
    base = img[0].copy()
    img[1] = (base + rng.normal(0, 900, (h, w))).astype(np.float32)          # close channels: the order changes from pixel to pixel
    img[2] = (base + rng.normal(0, 900, (h, w))).astype(np.float32)
    flat = [p.reshape(-1) for p in img]
    for c in range(3):
        flat[c][:tri.shape[1]] = vals[tri[c]]
    x = np.arange(65536, dtype=np.float64) / 65535.0
    lut = ((1.0 - np.cos(np.pi * x ** 0.7)) / 2.0 * 65535.0).astype(np.float32)

    def same(a, b):            # bit for bit; a NaN has to be a NaN (its payload is the host's / the device's own)
        bad = 0
        for p, q in zip(a, b):
            npn, nq = np.isnan(p), np.isnan(q)
            bad += int((npn != nq).sum()) + int((p.view(np.uint32)[~npn & ~nq] != q.view(np.uint32)[~npn & ~nq]).sum())
        return bad
    for whitept in (0.9, 1.0):
        ref = O.tone_std(img, lut, whitept, True)
        got = [p.copy() for p in img]
        gpu_ctx.tone_curve(capi.host_rgb(got), lut, whitept, True)
        assert same(got, ref) == 0
    # the seven cases really occur in the random part
    r, g, b = [p.reshape(-1)[tri.shape[1]:] for p in img]
    cases = np.where(r >= g, np.where(g > b, 0, np.where(b > r, 1, np.where(b > g, 2, 3))), np.where(r >= b, 4, np.where(b > g, 5, 6)))
    assert set(np.unique(cases)) >= {0, 1, 2, 4, 5, 6}
