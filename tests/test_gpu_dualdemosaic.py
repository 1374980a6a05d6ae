"""GPU: artgpu_dual_demosaic_bayer against the oracle (RawImageSource::dual_demosaic_RT, rtengine/dual_demosaic_RT.cc:39-155, with
AMaZE / RCD first and the bilinear blend second; buildBlendMask rt_algo.cc:315-498)."""
import numpy as np
import pytest

import oracle_lib as O
from art_amd import capi, synth

pytestmark = pytest.mark.gpu


def frame(w, h, filt, seed, noise, flat=None):
    raw = synth.bayer_frame(w, h, filt, seed=seed, noise=noise)
    if flat is not None:
        # a mid-grey patch with very little noise: the flat tile the automatic threshold search looks for
        y0, x0, sz, amp = flat
        rng = np.random.default_rng(seed + 99)
        raw[y0:y0 + sz, x0:x0 + sz] = (9000.0 + rng.normal(0, amp, (sz, sz))).astype(np.float32)
    return raw


def run(gpu_ctx, raw, filt, method, contrast, auto, vng4=False):
    h, w = raw.shape
    out = [np.zeros((h, w), np.float32) for _ in range(3)]
    got_c = gpu_ctx.dual_demosaic_bayer(capi.BAYER_RCD if method == "rcd" else capi.BAYER_AMAZE, capi.host_plane(raw), filt, 1.0, 4, contrast, auto, capi.host_rgb(out),
                                        second=capi.DUAL_VNG4 if vng4 else capi.DUAL_BILINEAR)
    first = O.rcd(raw, filt) if method == "rcd" else O.amaze(raw, filt, 1.0, 4)
    ref, ref_c = O.dual_demosaic_blend(raw, first, filt, contrast, auto, vng4=vng4)
    return out, got_c, ref, ref_c, first


@pytest.mark.parametrize("w,h,filt,method,contrast", [
    (640, 480, synth.FILTERS_RGGB, "amaze", 20.0),
    (701, 523, 0x16161616, "amaze", 5.0),           # odd sizes: scalar tails of every 4-wide loop
    (802, 600, 0x61616161, "rcd", 35.0),
    (515, 398, 0x49494949, "rcd", 100.0),
])
def test_fixed_contrast_bit_exact(gpu_ctx, w, h, filt, method, contrast):
    raw = frame(w, h, filt, w, 800)
    raw[10:14, 20:60] = 70000.0           # L lookups above the table: the scalar form for those groups
    out, got_c, ref, ref_c, first = run(gpu_ctx, raw, filt, method, contrast, False)
    assert got_c == ref_c
    for o, r in zip(out, ref):
        assert np.array_equal(o.view(np.uint32), r.view(np.uint32))
    assert any(not np.array_equal(o, f) for o, f in zip(out, first))      # the blend did something


@pytest.mark.parametrize("w,h,noise,flat,expect", [
    (1280, 960, 300, (400, 640, 200, 150.0), "pass0"),    # an 80-pixel tile with normalised variance in [0.5, 1]: found in the first pass
    (1201, 900, 300, (333, 501, 70, 200.0), "pass1"),     # only a 40-pixel tile is flat enough: second pass + the +-10 pixel refinement
    (900, 700, 3000, None, "none"),                      # nothing flat: threshold 0, the first demosaicer is kept everywhere
])
def test_auto_contrast_bit_exact(gpu_ctx, w, h, noise, flat, expect):
    filt = synth.FILTERS_RGGB
    raw = frame(w, h, filt, w + 1, noise, flat)
    out, got_c, ref, ref_c, first = run(gpu_ctx, raw, filt, "amaze", 0.0, True)
    assert got_c == ref_c
    if expect == "none":
        assert got_c == 0.0
    else:
        assert 0.0 < got_c <= 100.0
    for o, r in zip(out, ref):
        assert np.array_equal(o.view(np.uint32), r.view(np.uint32))


def test_zero_contrast_is_the_first_demosaicer(gpu_ctx):
    filt = synth.FILTERS_RGGB
    raw = frame(320, 256, filt, 3, 500)
    out, got_c, ref, ref_c, first = run(gpu_ctx, raw, filt, "amaze", 0.0, False)
    assert got_c == 0.0
    for o, f in zip(out, first):
        assert np.array_equal(o.view(np.uint32), f.view(np.uint32))


@pytest.mark.parametrize("w,h,filt", [(640, 480, synth.FILTERS_RGGB), (701, 523, 0x16161616), (333, 802, 0x61616161), (515, 398, 0x49494949), (64, 64, synth.FILTERS_RGGB)])
def test_vng4_demosaic_bit_exact(gpu_ctx, w, h, filt):
    """ARTGPU_BAYER_VNG4: RawImageSource::vng4_demosaic (vng4_demosaic_RT.cc:62-397), all four CFA phases, odd sizes"""
    raw = frame(w, h, filt, w + 7, 1500)
    raw[5:9, 7:30] = 0.0
    raw[20:24, 10:50] = 65535.0
    out = [np.zeros((h, w), np.float32) for _ in range(3)]
    gpu_ctx.demosaic_bayer(capi.BAYER_VNG4, capi.host_plane(raw), filt, 1.0, 4, capi.host_rgb(out))
    ref = O.vng4(raw, filt)
    for o, r in zip(out, ref):
        assert np.array_equal(o.view(np.uint32), r.view(np.uint32))
    assert np.isfinite(np.stack(out)).all() and np.stack(out).min() >= 0.0


@pytest.mark.parametrize("w,h,filt,method,contrast,auto", [
    (640, 480, synth.FILTERS_RGGB, "amaze", 20.0, False),
    (701, 523, 0x49494949, "rcd", 8.0, False),
    (1201, 900, synth.FILTERS_RGGB, "amaze", 0.0, True),
])
def test_dual_with_vng4_bit_exact(gpu_ctx, w, h, filt, method, contrast, auto):
    """AMAZEVNG4 / RCDVNG4: the flat regions come from vng4_demosaic (dual_demosaic_RT.cc:128-148)"""
    raw = frame(w, h, filt, w + 3, 300, (333, 501, 70, 200.0) if auto else None)
    out, got_c, ref, ref_c, first = run(gpu_ctx, raw, filt, method, contrast, auto, vng4=True)
    assert got_c == ref_c
    for o, r in zip(out, ref):
        assert np.array_equal(o.view(np.uint32), r.view(np.uint32))
    assert any(not np.array_equal(o, f) for o, f in zip(out, first))
