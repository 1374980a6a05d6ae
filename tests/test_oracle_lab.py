"""CPU: the LAB-mode oracle (oracle/labadj.c; rtengine/imagefloat.cc:841-970, iplabadjustments.cc:236-327) against colour-science
identities in float64 and its own invariants."""
import numpy as np

import oracle_lib as O


def lab_float64(rgb, ws):
    """CIE L*a*b* (D50) of working-space RGB in [0, 65535], scaled by 327.68 like rtengine"""
    m = np.asarray(ws, np.float64).reshape(3, 3)
    xyz = np.tensordot(m, np.stack([p.astype(np.float64) for p in rgb]), 1) / 65535.0
    xyz[0] /= 0.9642
    xyz[2] /= 0.8249
    eps, kappa = 216.0 / 24389.0, 24389.0 / 27.0
    f = np.where(xyz > eps, np.cbrt(xyz), (kappa * xyz + 16.0) / 116.0)
    L = np.where(xyz[1] > eps, 116.0 * np.cbrt(xyz[1]) - 16.0, kappa * xyz[1])
    return [327.68 * 500.0 * (f[0] - f[1]), 327.68 * L, 327.68 * 200.0 * (f[1] - f[2])]


def test_rgb_to_lab_matches_cie_lab_and_round_trips():
    rng = np.random.default_rng(2)
    img = [rng.uniform(50, 60000, (67, 131)).astype(np.float32) for _ in range(3)]          # 131 = 4 * 32 + 3: vector groups and a tail
    lab = O.image_rgb_to_lab(img)
    ref = lab_float64(img, O.REC2020_WS_D)
    for a, b in zip(lab, ref):
        assert np.allclose(a, b, rtol=1e-4, atol=1.0)          # the 65536-entry LUT interpolation
    back = O.image_lab_to_rgb(lab, O.REC2020_IWS_D)
    for a, b in zip(back, img):
        assert np.allclose(a, b, rtol=1e-3, atol=1.5)
    # white maps to L = 100, a = b = 0
    white = [np.full((4, 8), 65535.0, np.float32)] * 3
    wl = O.image_rgb_to_lab(white)
    assert np.allclose(wl[1], 32768.0, atol=2.0) and np.abs(wl[0]).max() < 33 and np.abs(wl[2]).max() < 33   # < 0.1 Lab units (the matrix rows sum to the white point to 1e-5)


def test_vector_and_scalar_forms_agree_to_rounding_but_are_both_present():
    """the same pixel in a vector column and in the scalar tail: equal to a few ULP, not necessarily bit-equal (different association)"""
    rng = np.random.default_rng(4)
    row = [rng.uniform(100, 60000, (1, 7)).astype(np.float32) for _ in range(3)]
    wide = [np.concatenate([p, p[:, :1]], axis=1) for p in row]          # width 8: every column vector
    a = O.image_lab_to_rgb(O.image_rgb_to_lab(row), O.REC2020_IWS_D)
    b = O.image_lab_to_rgb(O.image_rgb_to_lab(wide), O.REC2020_IWS_D)
    for p, q in zip(a, b):
        assert np.allclose(p, q[:, :7], rtol=1e-5)


def test_histogram_and_identity_curves():
    rng = np.random.default_rng(6)
    L = rng.uniform(-100, 40000, (50, 61)).astype(np.float32)
    L[0, :4] = [np.nan, 1e12, -1e12, 65535.9]
    hist = O.lab_histogram(L)
    assert int(hist.sum()) == L.size
    idx = np.clip(np.nan_to_num(L, nan=0.0, posinf=0.0, neginf=0.0), 0, None)
    assert hist[65535] == 1 and hist[0] >= 3 + int((L < 1).sum()) - 3
    lc = np.arange(32770, dtype=np.float32)
    ac = np.arange(65536, dtype=np.float32)
    img = [rng.uniform(-20000, 20000, (9, 13)).astype(np.float32), rng.uniform(0, 32000, (9, 13)).astype(np.float32), rng.uniform(-20000, 20000, (9, 13)).astype(np.float32)]
    out = O.lab_adjustments(img, lc, ac, ac, 1.0)
    for p, q in zip(out, img):
        assert np.allclose(p, q, rtol=0, atol=0.01)
    half = O.lab_adjustments(img, lc, ac, ac, 0.5)
    assert np.allclose(half[0], img[0] * 0.5, atol=0.01) and np.allclose(half[1], img[1], atol=0.01)
