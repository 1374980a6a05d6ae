"""CPU: the log-encoding oracle (oracle/logenc.c, rtengine/iplogenc.cc) against an independent float64 model and its own invariants."""
import ctypes as C

import numpy as np

import oracle_lib as O


def find_gray(source, target):
    L = O.lib()
    L.oracle_logenc_find_gray.argtypes = [C.c_float, C.c_float]
    L.oracle_logenc_find_gray.restype = C.c_float
    return float(L.oracle_logenc_find_gray(source, target))


def test_find_gray_solves_log2lin():
    """find_gray (iplogenc.cc:38-91): (base^source - 1) / (base - 1) = target"""
    for black, white, target in [(-13.5, 2.5, 0.18), (-10.0, 6.0, 0.30), (-8.0, 4.0, 0.10)]:
        source = abs(black) / (white - black)
        base = find_gray(source, target)
        assert base > 1.0
        assert abs((base ** source - 1.0) / (base - 1.0) - target) < 2e-3
    assert find_gray(0.0, 0.18) == 0.0


def model(img, ws, gain=0.0, target_gray=18.0, black_ev=-13.5, white_ev=2.5, satcontrol=True):
    """regularization 0 in float64 numpy"""
    r, g, b = [p.astype(np.float64) for p in img]
    gray = 2.0 ** (-gain + np.log2(0.18))
    dr = max(white_ev - black_ev, 0.5)
    noise = 2.0 ** -16
    base = find_gray(abs(black_ev) / dr, target_gray / 100.0) if 1 < target_gray < 100 else 0.0
    wsr = np.asarray(ws, np.float64).reshape(3, 3)[1]

    def norm(r, g, b):
        ar, ag, ab = np.abs(r), np.abs(g), np.abs(b)
        pn = (ar ** 3 + ag ** 3 + ab ** 3) / np.maximum(ar ** 2 + ag ** 2 + ab ** 2, 1e-12)
        return pn / 2 + (r * wsr[0] + g * wsr[1] + b * wsr[2]) / 2

    def apply(x):
        x = np.maximum(np.maximum(x, noise) / gray, noise)
        x = np.maximum((np.log2(x) - black_ev) / dr, noise)
        return (base ** x - 1) / (base - 1) if base > 0 else x

    m = norm(r / 65535, g / 65535, b / 65535)
    ok = m > noise
    f = np.where(ok, apply(np.where(ok, m, 1.0)) / np.where(ok, m, 1.0), 1.0)
    r, g, b = r * f, g * f, b * f
    if satcontrol:
        ll = r * wsr[0] + g * wsr[1] + b * wsr[2]
        sf = lambda s, c: np.where(c > noise, 1 - np.minimum(np.abs(s) / np.where(c > noise, c, 1.0), 1.0), 0.0)
        mx = np.maximum(np.maximum(sf(r - ll, r), sf(g - ll, g)), sf(b - ll, b))
        s = mx * (f ** 0.3 * 0.6 + 0.4) + (1 - mx)
        sel = ok & (f < 1)
        r, g, b = [np.where(sel, ll + s * (c - ll), c) for c in (r, g, b)]
    return [r, g, b]


def test_direct_path_against_float64_model():
    rng = np.random.default_rng(5)
    ev = rng.uniform(-12, 3, (120, 160))
    img = [(0.18 * 65535 * np.exp2(ev) * rng.uniform(0.6, 1.4, ev.shape)).astype(np.float32) for _ in range(3)]
    for kw in (dict(), dict(satcontrol=False, target_gray=1.0), dict(gain=1.0, black_ev=-9.0, white_ev=5.0, target_gray=25.0)):
        ref = O.log_encoding(img, regularization=0, **kw)
        mod = model(img, O.REC2020_WS_D, **kw)
        for a, b in zip(ref, mod):
            assert np.allclose(a, b, rtol=2e-4, atol=0.05)


def test_mid_grey_maps_to_target_grey_and_regularised_path_is_close_to_direct():
    grey = np.full((64, 64), 0.18 * 65535, np.float32)
    out = O.log_encoding([grey, grey, grey], regularization=0)
    assert np.allclose(out[1], 0.18 * 65535, rtol=3e-3)          # find_gray's tolerance
    rng = np.random.default_rng(9)
    y, x = np.mgrid[0:200, 0:300].astype(np.float32)
    lum = (0.18 * 65535 * np.exp2(-6 + 8 * (0.5 + 0.5 * np.sin(0.02 * x) * np.cos(0.03 * y)))).astype(np.float32)
    img = [(lum * rng.uniform(0.95, 1.05, lum.shape)).astype(np.float32) for _ in range(3)]
    direct = O.log_encoding(img, regularization=0)
    reg = O.log_encoding(img, regularization=60)
    assert not np.array_equal(direct[1], reg[1])
    assert np.median(np.abs(reg[1] - direct[1]) / direct[1]) < 0.2
    assert np.isfinite(np.stack(reg)).all()
