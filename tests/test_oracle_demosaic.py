"""CPU: properties and regression fixtures of the demosaic oracle (AMaZE SSE semantics, RCD)."""
import hashlib
import json
import os

import numpy as np
import pytest

from art_amd import synth
import oracle_lib as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FILTERS = [synth.FILTERS_RGGB, synth.FILTERS_BGGR, synth.FILTERS_GRBG, synth.FILTERS_GBRG]


def digest(planes):
    h = hashlib.sha256()
    for p in planes:
        h.update(np.ascontiguousarray(p).tobytes())
    return h.hexdigest()


def test_fc_matches_bayer_layout():
    # RGGB: (0,0)=R (0,1)=G (1,0)=G (1,1)=B ; reference rawimage.h:186-189
    assert [int(synth.fc(synth.FILTERS_RGGB, r, c)) for r in (0, 1) for c in (0, 1)] == [0, 1, 1, 2]
    assert [int(synth.fc(synth.FILTERS_BGGR, r, c)) for r in (0, 1) for c in (0, 1)] == [2, 1, 1, 0]
    assert [int(synth.fc(synth.FILTERS_GRBG, r, c)) for r in (0, 1) for c in (0, 1)] == [1, 0, 2, 1]


@pytest.mark.parametrize("filt", FILTERS)
def test_rcd_keeps_native_samples_and_is_finite(filt):
    raw = synth.bayer_frame(400, 300, filt, seed=7)
    rgb = O.rcd(raw, filt)
    yy, xx = np.mgrid[0:300, 0:400]
    c = synth.fc(filt, yy, xx)
    for ch, p in enumerate(rgb):
        assert np.isfinite(p).all() and (p >= 0).all()
        assert np.array_equal(p[c == ch], raw[c == ch])  # LIM01(x/65536)*65536 is exact for integers < 65536


@pytest.mark.parametrize("filt", FILTERS)
def test_amaze_green_native_exact_and_finite(filt):
    raw = synth.bayer_frame(352, 288, filt, seed=3)
    r, g, b = O.amaze(raw, filt, 1.0, 4)
    yy, xx = np.mgrid[0:288, 0:352]
    c = synth.fc(filt, yy, xx)
    for p in (r, g, b):
        assert np.isfinite(p).all() and (p >= 0).all()
    assert np.array_equal(g[c == 1], raw[c == 1])
    assert np.abs(r[c == 0] - raw[c == 0]).max() < 0.01 and np.abs(b[c == 2] - raw[c == 2]).max() < 0.01


def test_amaze_flat_field_is_reproduced():
    raw = np.full((288, 352), 12000.0, np.float32)
    r, g, b = O.amaze(raw, synth.FILTERS_RGGB, 1.0, 4)
    for p in (r, g, b):
        assert np.abs(p - 12000.0).max() < 0.02


def test_amaze_thread_schedule_independent_on_full_tiles():
    """384x384: every tile with output is 160 wide/high, so a tile's result may not depend on
    what the arena held before (reference threads never clear it, amaze_demosaic_RT.cc:124)."""
    filt = synth.FILTERS_RGGB
    raw = synth.bayer_frame(384, 384, filt, seed=9)
    ref = O.amaze(raw, filt, 1.0, 4)
    for order in ("raster", "reverse"):
        out = O.amaze_tiles_stale(raw, filt, 1.0, order)
        assert [int((a.view(np.uint32) != b.view(np.uint32)).sum()) for a, b in zip(ref, out)] == [0, 0, 0]


def test_border_interpolate2_amaze_border_lt_4():
    filt = synth.FILTERS_GRBG
    raw = synth.bayer_frame(200, 180, filt, seed=1)
    a = O.amaze(raw, filt, 1.0, 4)
    b = O.amaze(raw, filt, 1.0, 0)
    for pa, pb in zip(a, b):
        assert np.array_equal(pa[3:-3, 3:-3], pb[3:-3, 3:-3])
    assert not np.array_equal(a[0][:3], b[0][:3])


def test_regression_digests():
    """Self-generated regression fixtures (NOT reference-derived: AMaZE/RCD parity is unpinned)."""
    path = os.path.join(G, "demosaic_digests.json")
    cur = {}
    for name, filt in (("rggb", synth.FILTERS_RGGB), ("gbrg", synth.FILTERS_GBRG)):
        raw = synth.bayer_frame(401, 331, filt, seed=5)
        cur[f"amaze_{name}"] = digest(O.amaze(raw, filt, 2.1, 4))
        cur[f"rcd_{name}"] = digest(O.rcd(raw, filt))
    xraw = synth.xtrans_frame(401, 331, seed=5)
    cur["xtrans_1pass"] = digest(O.xtrans_demosaic(xraw, synth.XTRANS_FUJI, synth.XTRANS_RGB_CAM, 1, False))
    cur["xtrans_3pass"] = digest(O.xtrans_demosaic(xraw, synth.XTRANS_FUJI, synth.XTRANS_RGB_CAM, 3, True))
    if not os.path.exists(path):
        json.dump(cur, open(path, "w"), indent=1)
        pytest.skip("fixture written")
    old = json.load(open(path))
    if set(cur) - set(old):                       # new rows are appended, existing rows must not move
        assert all(old[k] == cur[k] for k in old)
        json.dump(cur, open(path, "w"), indent=1)
        pytest.skip("fixture extended")
    assert old == cur


def test_xtrans_oracle_reconstructs_smooth_scene():
    """Markesteijn 1-pass and 3-pass on a noise-free synthetic scene: native samples kept, interpolated samples close to
    the ground truth (the scene is known for all three colours)."""
    from art_amd import synth as S
    import oracle_lib as O
    w, h = 360, 270
    truth = [S.bayer_frame(w, h, 0, 1, 0, False, False, xtrans=np.full((6, 6), k, np.int32)) for k in range(3)]
    raw = S.bayer_frame(w, h, 0, 1, 0, False, False, xtrans=S.XTRANS_FUJI)
    yy, xx = np.mgrid[0:h, 0:w]
    cmap = S.XTRANS_FUJI[yy % 6, xx % 6]
    for passes, lab in ((1, False), (3, True)):
        out = O.xtrans_demosaic(raw, S.XTRANS_FUJI, S.XTRANS_RGB_CAM, passes, lab)
        for k in range(3):
            inner = np.zeros((h, w), bool); inner[12:-12, 12:-12] = True
            assert np.array_equal(out[k][(cmap == k) & inner], raw[(cmap == k) & inner])
            err = np.abs(out[k] - truth[k])[16:-16, 16:-16]
            assert np.median(err) < 2.0 and np.percentile(err, 99) < 200.0
