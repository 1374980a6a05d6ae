"""GPU parity: denoise::denoiseGuidedSmoothing (guided filter on the chroma, log domain) vs the oracle."""
import numpy as np
import pytest

import oracle_lib as O
from art_amd import synth

pytestmark = pytest.mark.gpu


def _same(a, b):
    return [int((x.view(np.uint32) != y.view(np.uint32)).sum()) for x, y in zip(a, b)]


@pytest.mark.parametrize("w,h,radius,scale", [
    (720, 640, 3, 1.0),     # max(w,h) > 600 -> subsampling 3, bilinear down/up
    (601, 451, 3, 1.0),     # odd sizes
    (480, 360, 3, 1.0),     # <= 600 -> subsampling 1
    (900, 700, 4, 1.0),     # r = 4 -> subsampling 4
    (900, 700, 5, 2.0),     # scale 2 -> r = round(2.5) = 3 (std::round: half away from zero)
])
def test_guided_smoothing_bit_exact(gpu_ctx, w, h, radius, scale):
    from art_amd import capi
    raw = synth.bayer_frame(w, h, synth.FILTERS_RGGB, seed=w, noise=2048)
    img = O.amaze(raw, synth.FILTERS_RGGB, 1.0, 4)
    got = [p.copy() for p in img]
    gpu_ctx.denoise_guided_smoothing(capi.host_rgb(got), O.REC2020_WS_D, radius, scale)
    ref = O.guided_smoothing(img, O.REC2020_WS_D, radius, scale)
    assert _same(got, ref) == [0, 0, 0]
    assert all(np.isfinite(p).all() for p in got)


def test_radius_zero_is_identity(gpu_ctx):
    from art_amd import capi
    img = [np.random.default_rng(c).uniform(0, 65535, (64, 80)).astype(np.float32) for c in range(3)]
    got = [p.copy() for p in img]
    gpu_ctx.denoise_guided_smoothing(capi.host_rgb(got), O.REC2020_WS_D, 0, 1.0)
    assert _same(got, img) == [0, 0, 0]


@pytest.mark.parametrize("w,h,r,eps", [(300, 200, 3, 0.001), (700, 500, 4, 0.001), (1203, 801, 25, 0.0001), (911, 640, 7, 0.01), (640, 912, 1, 0.001), (801, 603, 10, 0.001), (700, 501, 54, 0.0001), (500, 420, 40, 0.001), (900, 700, 97, 0.005), (1290, 610, 211, 0.005)])
def test_plain_guided_filter_bit_exact(gpu_ctx, w, h, r, eps):
    """rtengine::guidedFilter (guidedfilter.cc:78-241), single channel, automatic subsampling (1, 4, 5, 3, 1, 5, 3, 1 here; then box radii 18 and 40: the wide-window blur kernel; 97 and 211 are prime, so no subsampling: the dynamic-LDS blur kernel for image-sized radii)."""
    from art_amd import capi
    rng = np.random.default_rng(w + r)
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    guide = (0.4 + 0.3 * np.sin(0.02 * x) * np.cos(0.017 * y) + 0.15 * ((x.astype(np.int32) // 40 + y.astype(np.int32) // 40) % 2)).astype(np.float32)
    src = (guide * 0.8 + rng.normal(0, 0.05, (h, w))).astype(np.float32)
    ref = O.guided_filter(guide, src, r, eps)
    got = np.zeros_like(src)
    gpu_ctx.guided_filter(capi.host_plane(guide), capi.host_plane(src), capi.host_plane(got), r, eps)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    # in place on the source plane, as hslEqualizer calls it (mask -> mask)
    inplace = src.copy()
    pl = capi.host_plane(inplace)
    gpu_ctx.guided_filter(capi.host_plane(guide), pl, pl, r, eps)
    assert np.array_equal(inplace.view(np.uint32), ref.view(np.uint32))
    assert np.abs(ref - src).mean() > 1e-3
