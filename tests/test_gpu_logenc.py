"""GPU: artgpu_log_encoding against the oracle (ImProcFunctions::logEncoding, rtengine/iplogenc.cc:132-316)."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def scene(w, h, seed):
    """linear scene-referred data with ~14 stops: deep shadows, negatives, zeros and values far above 65535"""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    ev = -13.0 + 15.0 * (0.5 + 0.5 * np.sin(0.011 * x + 0.3) * np.cos(0.013 * y)) + 0.8 * ((x.astype(np.int32) // 50 + y.astype(np.int32) // 60) % 2)
    lum = (0.18 * 65535.0 * np.exp2(ev)).astype(np.float32)
    tint = [1.0 + 0.5 * np.sin(0.02 * x), 1.0 + 0.3 * np.cos(0.017 * y), 1.0 + 0.6 * np.sin(0.015 * (x + y))]
    img = [(lum * t * rng.uniform(0.9, 1.1, (h, w))).astype(np.float32) for t in tint]
    img[0][5:9, 3:40] = 0.0
    img[1][5:9, 3:40] = 0.0
    img[2][5:9, 3:40] = 0.0
    img[2][20:24, 10:60] = -30.0          # negative channel (out-of-gamut after the working-space matrix)
    img[0][30:33, :50] *= 40.0            # far above white
    return img


CASES = [
    # w, h, kwargs
    (640, 400, dict(regularization=0)),
    (640, 400, dict(regularization=0, satcontrol=False, target_gray=1.0)),                 # no log2lin toe (linbase 0)
    (500, 333, dict(regularization=0, gain=1.5, black_ev=-10.0, white_ev=6.0, target_gray=30.0)),
    (720, 520, dict(regularization=60)),                                                    # the default: guided filter radius 24 -> subsampled by 4
    (901, 640, dict(regularization=100, gain=-0.7, full_width=8192, full_height=5464)),     # crop of a 45 MP frame: radius 273 = 3 x 91
    (610, 431, dict(regularization=25, satcontrol=False, full_width=8310, full_height=100)),  # radius 277 (prime): full-resolution box radius 214 after f_mean's clamp
]


@pytest.mark.parametrize("w,h,kw", CASES)
def test_log_encoding_bit_exact(gpu_ctx, w, h, kw):
    from art_amd import capi
    img = scene(w, h, w + h)
    got = [p.copy() for p in img]
    gpu_ctx.log_encoding(capi.host_rgb(got), O.REC2020_WS_D, **kw)
    ref = O.log_encoding(img, **kw)
    for g, r in zip(got, ref):
        assert np.array_equal(g.view(np.uint32), r.view(np.uint32))
    assert not np.allclose(got[1], img[1])
    assert np.isfinite(np.stack(got)).all()


def test_disabled_and_unsupported(gpu_ctx):
    from art_amd import capi
    img = scene(128, 96, 1)
    got = [p.copy() for p in img]
    gpu_ctx.log_encoding(capi.host_rgb(got), O.REC2020_WS_D, enabled=False)
    assert all(np.array_equal(g, p) for g, p in zip(got, img))


@pytest.mark.parametrize("w,h,kw", [
    (640, 400, dict(regularization=0, highlight_compression=40)),
    (500, 333, dict(regularization=0, highlight_compression=5, gain=1.0)),                   # factor < 0.1: blended with the identity
    (720, 520, dict(regularization=60, highlight_compression=100, white_ev=5.0)),
])
def test_log_encoding_highlight_compression(gpu_ctx, w, h, kw):
    """iplogenc.cc:148-170: the compression curve goes through the C library's powf (two calls per pixel above 0.8), which is not
    specified bit for bit; the device evaluates double-precision pow rounded to float.  Measured on MI355X against the oracle (which
    calls the host's powf; scripts/logenc_hl_stats.py): 0 - 8 values per million differ -- the pixels where glibc's powf is not the
    correctly rounded result --, by at most 1.2e-6 relative (one ULP of the curve value, amplified by the logarithm next to 1).
    Bound: at most 20 values per million differ, none by more than 4e-6 relative."""
    from art_amd import capi
    img = scene(w, h, w + h)
    got = [p.copy() for p in img]
    gpu_ctx.log_encoding(capi.host_rgb(got), O.REC2020_WS_D, **kw)
    ref = O.log_encoding(img, **kw)
    plain = O.log_encoding(img, **{**kw, "highlight_compression": 0})
    assert not np.array_equal(ref[1], plain[1])                      # the curve is active in this scene
    ndiff = 0
    for g, r in zip(got, ref):
        assert np.isfinite(g).all()
        bad = g.view(np.uint32) != r.view(np.uint32)
        ndiff += int(bad.sum())
        if bad.any():
            rel = np.abs(g[bad].astype(np.float64) - r[bad]) / np.maximum(np.abs(r[bad]), 1e-3)
            assert rel.max() <= 4e-6, f"max relative deviation {rel.max()}"
    assert ndiff <= max(6, 2e-5 * 3 * w * h), f"{ndiff} values differ"
