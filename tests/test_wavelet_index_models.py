"""CPU model of the index arithmetic of wavelet.hip's level-0 analysis (round 5): the column stage of a tile is no longer six loads per tmp value
but a walk -- a thread takes sixteen output rows of ONE tmp column with the 36 input rows they read in registers, 128 columns x 2 row groups per
workgroup, the tile's six halo columns in the plain form.  The model fills a tile's tmp arrays both ways with the kernel's index expressions
(numpy float32, the reference's term order) and compares their bits, at the plane's edges (clamped rows / columns, tiles hanging over) as well."""
import numpy as np
import pytest

f32 = np.float32
LO = [f32(x) for x in (0.0, 0.0, 0.34150635, 0.59150635, 0.15849365, -0.091506351)]
HI = [f32(x) for x in (-0.091506351, -0.15849365, 0.59150635, -0.34150635, 0.0, 0.0)]
TW, TH, WR = 64, 32, 16
LW = 2 * TW + 6


def clampi(v, lo, hi):
    return lo if v < lo else (hi if v > hi else v)


def taps(vals):
    l, h = f32(0), f32(0)
    for j in range(6):
        l = l + LO[j] * vals[j]
        h = h + HI[j] * vals[j]
    return l, h


def tmp_plain(src, r0, c0, h2):
    """one tmp value at a time: rows 2 orow + 2 - j of column clamp(icol0 + cc)"""
    h, w = src.shape
    icol0 = 2 * c0 - 3
    out = {}
    for rr in range(TH):
        orow = r0 + rr
        if orow >= h2:
            continue
        for cc in range(LW):
            k = clampi(icol0 + cc, 0, w - 1)
            out[(rr, cc)] = taps([src[clampi(2 * orow + 2 - j, 0, h - 1), k] for j in range(6)])
    return out


def tmp_walk(src, r0, c0, h2):
    """the kernel's phase 1: threads 0 .. 255 = (row group g, column cc < 128) walkers + (hr, hc) halo values for threads < 6 TH"""
    h, w = src.shape
    icol0 = 2 * c0 - 3
    out = {}
    for tid in range(256):
        g, cc = tid >> 7, tid & 127
        orow0 = r0 + g * WR
        k = clampi(icol0 + cc, 0, w - 1)
        b0 = 2 * orow0 - 3
        v = [src[clampi(b0 + i, 0, h - 1), k] for i in range(2 * WR + 4)]
        for q in range(WR):
            if orow0 + q < h2:
                out[(g * WR + q, cc)] = taps([v[2 * q + 5 - j] for j in range(6)])
        hr = tid // 6
        hc = 128 + tid - hr * 6
        if hr < TH and r0 + hr < h2:
            hk = clampi(icol0 + hc, 0, w - 1)
            hrow = 2 * (r0 + hr)
            assert (hr, hc) not in out
            out[(hr, hc)] = taps([src[clampi(hrow + 2 - j, 0, h - 1), hk] for j in range(6)])
    return out


@pytest.mark.parametrize("h,w", [(70, 150), (64, 128), (37, 91), (129, 257)])
def test_walked_column_stage_is_the_plain_one(h, w):
    rng = np.random.default_rng(h * 1000 + w)
    src = rng.uniform(-5000, 60000, (h, w)).astype(f32)
    h2, w2 = (h + 1) // 2, (w + 1) // 2
    for r0 in range(0, h2, TH):
        for c0 in range(0, w2, TW):
            a, b = tmp_plain(src, r0, c0, h2), tmp_walk(src, r0, c0, h2)
            assert a.keys() == b.keys()
            for key in a:
                assert a[key][0].view(np.uint32) == b[key][0].view(np.uint32) and a[key][1].view(np.uint32) == b[key][1].view(np.uint32), (r0, c0, key)
