"""GPU: NEUTRAL tone-curve mode (NeutralToneCurve::BatchApply, curves.cc:893-1038) vs the oracle."""
import numpy as np
import pytest

import oracle_lib as O
from art_amd import capi

pytestmark = pytest.mark.gpu


def s_curve():
    x = np.arange(65536, dtype=np.float64) / 65535.0
    return ((1.0 - np.cos(np.pi * x ** 0.7)) / 2.0 * 65535.0).astype(np.float32)


def frame(w, h, seed, top):
    rng = np.random.default_rng(seed)
    base = rng.uniform(0.0, top, (h, w)).astype(np.float32)
    img = [(base * rng.uniform(0.2, 1.0, (h, w))).astype(np.float32) for _ in range(3)]
    img[0][:8] = img[1][:8] = img[2][:8]          # neutral rows
    img[1][8:12] = 0.0
    img[0][12:14] = -50.0                          # negative input is clamped first
    img[2][14:16, ::2] = 0.0; img[0][14:16, ::2] = 0.0; img[1][14:16, ::2] = 0.0   # black
    return img


@pytest.mark.parametrize("whitecoeff", [1.0, 1.5])
def test_neutral_tone_curve_in_range_pixels_bit_exact(gpu_ctx, whitecoeff):
    w, h = 517, 203
    img = frame(w, h, 21, 65535.0)
    lut = s_curve()
    ref, oor = O.tone_neutral(img, lut, whitecoeff, want_oor=True)
    got = [p.copy() for p in img]
    gpu_ctx.tone_curve_neutral(capi.host_rgb(got), lut, whitecoeff, O.REC2020_WS_D, O.REC2020_IWS_D)
    assert oor.mean() < 0.02
    for g, r in zip(got, ref):
        assert np.array_equal(g[~oor].view(np.uint32), r[~oor].view(np.uint32))
        assert np.allclose(g[oor], r[oor], rtol=2e-4, atol=0.5)


def test_neutral_tone_curve_super_white_within_tolerance(gpu_ctx):
    """input above 65535 (after +EV exposure): LMS > 1 -> powf per pixel (host libm in the reference, ocml here)."""
    w, h = 320, 120
    img = frame(w, h, 22, 90000.0)
    lut = s_curve()
    out_m = np.array([[1.2, -0.15, -0.05], [-0.08, 1.1, -0.02], [0.0, -0.1, 1.1]], np.float32)   # a wider-than-output gamut pair
    in_m = np.linalg.inv(out_m.astype(np.float64)).astype(np.float32)
    st = O.neutral_state(to_out=out_m, to_work=in_m)
    ref, oor = O.tone_neutral(img, lut, 1.0, state=st, want_oor=True)
    got = [p.copy() for p in img]
    gpu_ctx.tone_curve_neutral(capi.host_rgb(got), lut, 1.0, O.REC2020_WS_D, O.REC2020_IWS_D, out_m, in_m)
    assert 0.01 < oor.mean() < 0.9
    for g, r in zip(got, ref):
        assert np.array_equal(g[~oor].view(np.uint32), r[~oor].view(np.uint32))
        assert np.allclose(g[oor], r[oor], rtol=2e-4, atol=0.5)


def test_neutral_large_frame_lds_pq_table_bit_exact(gpu_ctx, monkeypatch):
    """frames of >= 4 Mpx keep the lower 40704 entries of the forward PQ table in LDS: same bits as the plain kernel everywhere and as
    the oracle on the in-range pixels"""
    w, h = 2310, 1840
    img = frame(w, h, 5, 65535.0)
    lut = s_curve()
    ref, oor = O.tone_neutral(img, lut, 1.0, want_oor=True)
    got = [p.copy() for p in img]
    gpu_ctx.tone_curve_neutral(capi.host_rgb(got), lut, 1.0, O.REC2020_WS_D, O.REC2020_IWS_D)
    gpu_ctx.set_option("lut_lds", 0)
    plain = [p.copy() for p in img]
    try:
        gpu_ctx.tone_curve_neutral(capi.host_rgb(plain), lut, 1.0, O.REC2020_WS_D, O.REC2020_IWS_D)
    finally:
        gpu_ctx.set_option("lut_lds", 1)
    for g, p, r in zip(got, plain, ref):
        assert np.array_equal(g.view(np.uint32), p.view(np.uint32))
        assert np.array_equal(g[~oor].view(np.uint32), r[~oor].view(np.uint32))


@pytest.mark.parametrize("kind,y_last", [(1, 0.95), (2, 1.0)])      # (0.95: not the LUT's last entry, so the constant tail is visible)
def test_neutral_tone_curve_above_the_lut(gpu_ctx, kind, y_last):
    """NeutralToneCurve::BatchApply applies the curve through curves::setLutVal too (curves.cc:1003): with whitecoeff > 1 the values
    above 65535 take the Curve object's value (artgpu_set_curve_tail) instead of the LUT's last entry."""
    w, h = 389, 150
    img = frame(w, h, 33, 65535.0 * 1.4)
    lut = (s_curve() * np.float32(0.9)).astype(np.float32)
    O.set_curve_tail(kind, y_last)
    try:
        ref, oor = O.tone_neutral(img, lut, 1.5, want_oor=True)
    finally:
        O.set_curve_tail(0)
    base = O.tone_neutral(img, lut, 1.5)                         # LUT clip: must differ somewhere, or the tail was never reached
    got = [p.copy() for p in img]
    gpu_ctx.set_curve_tail(kind, y_last)
    try:
        gpu_ctx.tone_curve_neutral(capi.host_rgb(got), lut, 1.5, O.REC2020_WS_D, O.REC2020_IWS_D)
    finally:
        gpu_ctx.set_curve_tail(3)
    assert any((b != r).any() for b, r in zip(base, ref))
    for g, r in zip(got, ref):
        assert np.array_equal(g[~oor].view(np.uint32), r[~oor].view(np.uint32))
        assert np.allclose(g[oor], r[oor], rtol=2e-4, atol=0.5)


def test_neutral_tone_curve_above_the_lut_parametric_curve(gpu_ctx):
    """the same through NeutralToneCurve::BatchApply (curves.cc:1003) with a DCT_Parametric curve"""
    w, h = 389, 150
    img = frame(w, h, 34, 65535.0 * 1.4)
    lut = (s_curve() * np.float32(0.9)).astype(np.float32)
    par = [2.0, 0.25, 0.5, 0.75, 30.0, 20.0, -15.0, -25.0, 0.0]
    O.set_parametric_curve(par)
    try:
        ref, oor = O.tone_neutral(img, lut, 1.5, want_oor=True)
    finally:
        O.set_curve_tail(0)
    base = O.tone_neutral(img, lut, 1.5)
    got = [p.copy() for p in img]
    gpu_ctx.set_curve_tail_parametric(par)
    try:
        gpu_ctx.tone_curve_neutral(capi.host_rgb(got), lut, 1.5, O.REC2020_WS_D, O.REC2020_IWS_D)
    finally:
        gpu_ctx.set_curve_tail(3)
    assert any((b != r).any() for b, r in zip(base, ref))
    for g, r in zip(got, ref):
        assert np.array_equal(g[~oor].view(np.uint32), r[~oor].view(np.uint32))
        assert np.allclose(g[oor], r[oor], rtol=2e-4, atol=0.5)
