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
