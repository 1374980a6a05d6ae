"""GPU: artgpu_hsl_equalizer against the oracle (ImProcFunctions::hslEqualizer, rtengine/iphsl.cc:29-221)."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

# FlatCurve control points: {FCT_MinMaxCPoints, x, y, left tangent, right tangent, ...}
S_CURVE = (1, 0.0, 0.5, 0.35, 0.35, 0.12, 0.72, 0.35, 0.35, 0.40, 0.30, 0.35, 0.35, 0.70, 0.55, 0.35, 0.35)
L_CURVE = (1, 0.05, 0.5, 0.35, 0.35, 0.30, 0.64, 0.35, 0.35, 0.62, 0.41, 0.35, 0.35)
H_CURVE = (1, 0.0, 0.5, 0.0, 0.0, 0.25, 0.58, 0.35, 0.35, 0.55, 0.44, 0.35, 0.35, 0.80, 0.5, 0.35, 0.35)
FLAT = (1, 0.0, 0.5, 0.35, 0.35, 0.5, 0.5, 0.35, 0.35)


def scene(w, h, seed):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    r = 20000 + 15000 * np.sin(0.013 * x) * np.cos(0.011 * y)
    g = 22000 + 12000 * np.cos(0.009 * x + 0.4) * np.sin(0.02 * y)
    b = 18000 + 14000 * np.sin(0.015 * y + 0.01 * x)
    return [np.maximum(p + rng.normal(0, 600, (h, w)), 10).astype(np.float32) for p in (r, g, b)]


@pytest.mark.parametrize("w,h,smoothing,scale,curves", [
    (420, 300, 0, 1.0, (H_CURVE, S_CURVE, L_CURVE)),     # no mask smoothing (radius 0)
    (700, 501, 5, 1.0, (H_CURVE, S_CURVE, L_CURVE)),     # smoothing 5: radii 9 and 54, subsampled guided filters
    (420, 300, 10, 2.0, (None, S_CURVE, FLAT)),          # preview scale, only the saturation curve (the flat one is an identity)
])
def test_hsl_equalizer_bit_exact(gpu_ctx, w, h, smoothing, scale, curves):
    from art_amd import capi
    img = scene(w, h, w)
    hc, sc, lc = curves
    for to_rgb in (True, False):
        got = [p.copy() for p in img]
        gpu_ctx.hsl_equalizer(capi.host_rgb(got), hc, sc, lc, smoothing, O.REC2020_WS_D, scale, to_rgb)
        ref = O.hsl_equalizer(img, hc, sc, lc, smoothing, scale=scale, to_rgb=to_rgb)
        for g, r in zip(got, ref):
            assert np.array_equal(g.view(np.uint32), r.view(np.uint32))
    assert not np.allclose(got[1], img[1])


def test_identity_curves_only_round_trip(gpu_ctx):
    """all three curves flat: the pixels only go RGB -> YUV/65535 -> (hue, saturation) -> YUV -> RGB, same bits as the oracle"""
    from art_amd import capi
    img = scene(200, 160, 3)
    got = [p.copy() for p in img]
    gpu_ctx.hsl_equalizer(capi.host_rgb(got), FLAT, None, FLAT, 3, O.REC2020_WS_D)
    ref = O.hsl_equalizer(img, FLAT, None, FLAT, 3)
    for g, r in zip(got, ref):
        assert np.array_equal(g.view(np.uint32), r.view(np.uint32))
    assert np.abs(got[0] - img[0]).max() < 1.0
