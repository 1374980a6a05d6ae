"""The "fresh arena" definition as a checked contract (CPU).

The reference's demosaicers keep one work buffer per THREAD, calloc'ed once and never cleared (amaze_demosaic_RT.cc:124,
rcd_demosaic.cc:99-105, xtrans_demosaic.cc:295-315): positions a tile does not write keep what the thread's previous tile left there,
and a few in-range outputs at frame edges depend on them -- so the reference's result there depends on its OpenMP schedule.  Oracle
and device define every tile as starting from an all-zero buffer (what a thread sees on its first tile).  These tests evaluate the
other definition too -- ONE buffer, zeroed once, tiles in raster (or reverse) order, i.e. a single-threaded reference run -- and
assert exactly WHERE the two may differ; everything else has to be bit-identical.  """
import ctypes as C

import numpy as np
import pytest

from art_amd import synth
import oracle_lib as O


def _mag(a, b):
    """largest |a - b| over the three planes (NaN pairs apart), on the 0 .. 65535 scale"""
    m = 0.0
    for x, y in zip(a, b):
        d = np.abs(x.astype(np.float64) - y.astype(np.float64))
        d[np.isnan(d)] = 0.0
        m = max(m, float(d.max()))
    return m


# How MUCH the two definitions differ where they may (measured on these frames: AMaZE up to 709, RCD up to 42, X-Trans 3-pass up to 2425 of
# 65535; against the stub-compiled reference itself the round-2 review measured up to 638 for AMaZE): bounds at roughly twice that.
AMAZE_EDGE_BOUND, RCD_EDGE_BOUND, XTRANS_EDGE_BOUND = 1536.0, 128.0, 4096.0


def _diff(a, b):
    d = np.zeros(a[0].shape, bool)
    for x, y in zip(a, b):
        d |= (x.view(np.uint32) != y.view(np.uint32)) & ~(np.isnan(x) & np.isnan(y))
    return d


@pytest.mark.parametrize("w,h,filt,seed", [(1100, 870, synth.FILTERS_BGGR, 1), (1000, 777, synth.FILTERS_RGGB, 3),
                                           (1283, 901, synth.FILTERS_GRBG, 4), (1153, 907, synth.FILTERS_GBRG, 5)])
def test_amaze_stale_arena_differs_only_at_the_documented_positions(w, h, filt, seed):
    """AMaZE: (i) the last 8 columns / rows of the frame (right- and bottom-edge partial tiles: hcd[i+2], vcd[i+v2] read past the valid
    range, carried inward by the in-place passes), (ii) rows 142-143 of a tile (the rbp-on-vcd alias), nothing else."""
    raw = synth.bayer_frame(w, h, filt, seed=seed, noise=3000)
    fresh = O.amaze(raw, filt, 1.0, 4)
    total = 0
    for order in ("raster", "reverse"):
        stale = O.amaze_tiles_stale(raw, filt, 1.0, order)
        d = _diff(fresh, stale)
        assert _mag(fresh, stale) <= AMAZE_EDGE_BOUND
        yy, xx = np.nonzero(d)
        allowed = (xx >= w - 8) | (yy >= h - 8) | np.isin((yy + 16) % 128, (14, 15))
        assert allowed.all(), list(zip(yy[~allowed][:5], xx[~allowed][:5]))
        total += len(yy)
    assert 0 < total < 1e-3 * 2 * w * h        # the two definitions do differ, on fewer than 0.1 % of the values


@pytest.mark.parametrize("w,h,filt,seed", [(1100, 870, synth.FILTERS_BGGR, 1), (1000, 777, synth.FILTERS_RGGB, 3), (1283, 901, synth.FILTERS_GRBG, 4)])
def test_rcd_stale_buffer_differs_only_in_the_last_written_row_and_column(w, h, filt, seed):
    """RCD: only the last row / column the tiles write (the 9-pixel frame border behind it comes from border_interpolate2), in partial
    tiles: step 4.x reads one position past the rows / columns a partial tile recomputes."""
    raw = synth.bayer_frame(w, h, filt, seed=seed, noise=3000)
    fresh = O.rcd(raw, filt)
    flag = C.c_int.in_dll(O.lib(), "oracle_rcd_stale")
    flag.value = 1
    try:
        stale = O.rcd(raw, filt)
    finally:
        flag.value = 0
    yy, xx = np.nonzero(_diff(fresh, stale))
    assert ((xx == w - 10) | (yy == h - 10)).all()
    assert 0 < len(yy) < 2 * (w + h)
    assert _mag(fresh, stale) <= RCD_EDGE_BOUND


@pytest.mark.parametrize("w,h,seed,passes", [(750, 620, 1, 1), (750, 620, 2, 3), (1006, 800, 3, 3)])
def test_xtrans_stale_buffer_differs_only_in_the_bottom_tile_rows(w, h, seed, passes):
    """X-Trans: the 1-pass variant does not depend on the buffer's history at all; the 3-pass variant reads never-written
    homogeneity bytes in the bottom tile row (rows within 28 of the frame's lower edge), nowhere else."""
    raw = synth.xtrans_frame(w, h, seed=seed)
    fresh = O.xtrans_demosaic(raw, synth.XTRANS_FUJI, synth.XTRANS_RGB_CAM, passes, passes > 1)
    flag = C.c_int.in_dll(O.lib(), "oracle_xtrans_stale")
    flag.value = 1
    try:
        stale = O.xtrans_demosaic(raw, synth.XTRANS_FUJI, synth.XTRANS_RGB_CAM, passes, passes > 1)
    finally:
        flag.value = 0
    yy, xx = np.nonzero(_diff(fresh, stale))
    if passes == 1:
        assert len(yy) == 0
    else:
        assert len(yy) > 0 and (yy >= h - 28).all() and len(yy) < 4 * w
        assert _mag(fresh, stale) <= XTRANS_EDGE_BOUND
