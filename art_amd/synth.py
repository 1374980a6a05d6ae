"""Deterministic synthetic CFA frames (integer-generated, identical on every host).

The reference ships no sample raws, so every test and benchmark
input comes from here.  Values are integers in [0, 65535] stored as float32, i.e. what
``RawImageSource::rawData`` holds after ``scaleColors`` for a 16-bit sensor
(reference: rtengine/rawimagesource.cc:2677-2859).

Scene = smooth gradient (integer sine table) + 64-px checker + a patch of 1-px
(Nyquist) stripes (exercises amaze_demosaic_RT.cc:846-954) + a clipped patch at 65535
(exercises the clip_pt paths, L417-425,565) + hash noise, multiplied by per-colour
gains R 0.6 / G 1.0 / B 0.7 through the CFA.
"""
from __future__ import annotations

import numpy as np

FILTERS_RGGB = 0x94949494
FILTERS_BGGR = 0x16161616
FILTERS_GRBG = 0x61616161
FILTERS_GBRG = 0x49494949

_SINE = np.round(32767.0 * np.sin(2.0 * np.pi * np.arange(4096) / 4096.0)).astype(np.int64)


def fc(filters: int, row, col):
    """RawImage::FC (reference: rtengine/rawimage.h:186-189)."""
    row = np.asarray(row, dtype=np.int64)
    col = np.asarray(col, dtype=np.int64)
    sh = ((((row << 1) & 14) + (col & 1)) << 1)
    return (np.int64(filters) >> sh) & 3


def _hash64(idx: np.ndarray, seed: int) -> np.ndarray:
    """splitmix64 finaliser on (index + seed * golden); uint64 wrap-around arithmetic."""
    with np.errstate(over="ignore"):
        z = idx.astype(np.uint64) + np.uint64((seed * 0x9E3779B97F4A7C15 + 0x632BE59BD9B4E019) & 0xFFFFFFFFFFFFFFFF)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def bayer_frame(width: int, height: int, filters: int = FILTERS_RGGB, seed: int = 0,
                noise: int = 1024, clip_patch: bool = True, nyquist_patch: bool = True, xtrans=None) -> np.ndarray:
    """Return a (height, width) float32 CFA frame with integer values in [0, 65535] (xtrans: 6x6 colour map instead of
    the Bayer `filters` word)."""
    out = np.empty((height, width), dtype=np.float32)
    x = np.arange(width, dtype=np.int64)[None, :]
    band = 512
    for y0 in range(0, height, band):
        y1 = min(height, y0 + band)
        y = np.arange(y0, y1, dtype=np.int64)[:, None]
        sx = _SINE[(x * 7) & 4095]
        sy = _SINE[(y * 9 + 1024) & 4095]
        v = 20000 + ((15000 * sx * sy) >> 30)
        v = v + np.where((((x >> 6) ^ (y >> 6)) & 1) == 1, 6000, -6000)
        if nyquist_patch:
            nx0, ny0, ns = width // 8, height // 8, max(16, min(512, width // 4))
            inside = (x >= nx0) & (x < nx0 + ns) & (y >= ny0) & (y < ny0 + ns)
            v = v + np.where(inside, np.where((x & 1) == 1, 8000, -8000), 0)
        idx = y * width + x
        if noise > 0:
            h = _hash64(idx, seed)
            v = v + (h & np.uint64(2 * noise - 1)).astype(np.int64) - noise
        c = fc(filters, y, x) if xtrans is None else xtrans[y % 6, x % 6]
        gain = np.where(c == 0, 614, np.where(c == 1, 1024, 717))
        v = (v * gain) >> 10
        if clip_patch:
            cx0, cy0, cs = width // 2, height // 8, max(8, min(256, width // 8))
            inside = (x >= cx0) & (x < cx0 + cs) & (y >= cy0) & (y < cy0 + cs)
            v = np.where(inside, 65535, v)
        out[y0:y1] = np.clip(v, 0, 65535).astype(np.float32)
    return out


# Fuji X-Trans 6x6 colour map (0 = R, 1 = G, 2 = B), SURVEY.md section 8d
XTRANS_FUJI = np.array([[1, 1, 0, 1, 1, 2],
                        [1, 1, 2, 1, 1, 0],
                        [2, 0, 1, 0, 2, 1],
                        [1, 1, 2, 1, 1, 0],
                        [1, 1, 0, 1, 1, 2],
                        [0, 2, 1, 2, 0, 1]], dtype=np.int32)
# camera -> sRGB matrix rows (RawImage::getRgbCam is 3x4; 4th column unused for 3-colour sensors)
XTRANS_RGB_CAM = np.array([[1.60, -0.45, -0.15, 0.0],
                           [-0.20, 1.45, -0.25, 0.0],
                           [0.02, -0.50, 1.48, 0.0]], dtype=np.float32)


def xtrans_frame(width: int, height: int, seed: int = 0, noise: int = 1024, clip_patch: bool = True) -> np.ndarray:
    return bayer_frame(width, height, 0, seed, noise, clip_patch, True, xtrans=XTRANS_FUJI)


def nyquist_patches_frame(width: int, height: int, filters: int = FILTERS_RGGB, step=(97, 89), size: int = 12, amp: int = 6000,
                          noise: int = 0, seed: int = 5) -> np.ndarray:
    """A smooth scene with small patches of 1-px stripes scattered over it, at positions that drift against AMaZE's 128-pixel tile grid:
    most tiles get a sparse set of Nyquist flags, i.e. a bounding box (amaze_demosaic_RT.cc:806-876) smaller than the tile -- the case in
    which the streaming kernel has to give a tile a second attempt (about a quarter of the tiles of a 13 MP frame)."""
    raw = bayer_frame(width, height, filters, seed=seed, noise=noise, clip_patch=False, nyquist_patch=False).astype(np.int64)
    yy, xx = np.mgrid[0:height, 0:width]
    py, px = yy % step[1], xx % step[0]
    oy = (7 * (yy // step[1]) + 3 * (xx // step[0])) % (step[1] - size)
    ox = (5 * (xx // step[0]) + 11 * (yy // step[1])) % (step[0] - size)
    inside = (py >= oy) & (py < oy + size) & (px >= ox) & (px < ox + size)
    raw = raw + np.where(inside, np.where((xx & 1) == 1, amp, -amp), 0)
    return np.clip(raw, 0, 65535).astype(np.float32)
