"""Host-side table construction for the chroma noise curve (NoiseCurve::Set / FlatCurve, ipdenoise.cc:684-716,
flatcurves.cc) -- library (C++) vs oracle (C) and the properties the reference's fixed curve must have."""
import ctypes as C
import os

import numpy as np

import oracle_lib as O

G = os.path.join(os.path.dirname(__file__), "golden")


def test_fixed_chroma_noise_curve_properties():
    lut, s = O.noise_curve()
    assert lut.shape == (501,) and s > 5.0                         # useNoiseCCurve is always on (FTblockDN.cc:1672)
    assert abs(lut[25] - 0.50) < 1e-6 and abs(lut[175] - 0.05) < 1e-6    # the two control points (x = 0.05, 0.35)
    assert np.all(lut[:26] == lut[0])                              # horizontal lead-in (flatcurves.cc:274-277)
    assert np.all(np.diff(lut[25:176]) <= 1e-7)                    # falls monotonically between the control points
    assert np.all(np.abs(lut[175:] - 0.05) < 1e-6) and lut.min() >= 0.01
    assert abs(s - float(np.cumsum(lut, dtype=np.float32)[-1])) < 1e-3


def test_library_noise_curve_matches_oracle_bit_for_bit():
    from art_amd import capi
    for pts in (O.NOISE_C_CURVE_POINTS,
                (1.0, 0.0, 0.2, 0.35, 0.35, 0.3, 0.9, 0.5, 0.2, 0.7, 0.4, 0.0, 0.35, 1.0, 0.1, 0.35, 0.35),
                (1.0, 0.1, 0.0, 0.35, 0.35, 0.6, 0.0, 0.35, 0.35),          # identity (all y == identity value 0)
                (1.0, 0.2, 0.3, 0.9, 0.8, 0.5, 0.6, 0.7, 0.6)):              # tangents summing above 1
        a, sa = capi.noise_curve_lut(pts)
        b, sb = O.noise_curve(pts)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and sa == sb


def test_flat_curve_periodic_and_identity():
    pts = (1.0, 0.1, 0.3, 0.35, 0.35, 0.5, 0.8, 0.35, 0.35, 0.8, 0.5, 0.35, 0.35)
    v, ident = O.flat_curve_sample(pts, True, 1000, 0.5, 257)
    assert not ident and abs(v[0] - v[-1]) < 1e-9 and 0.29 < v.min() and v.max() < 0.81     # periodic: wraps
    v2, ident2 = O.flat_curve_sample((1.0, 0.1, 0.5, 0.35, 0.35, 0.6, 0.5, 0.35, 0.35), True, 1000, 0.5, 33)
    assert ident2 and np.all(v2 == 0.5)


def test_xcbrtf_matches_reference():
    g = np.load(os.path.join(G, "sleef2.npz"))
    x = np.ascontiguousarray(g["xc"])
    y = np.empty_like(x)
    O.lib().oracle_t_xcbrtf(x.ctypes.data_as(C.POINTER(C.c_float)), y.ctypes.data_as(C.POINTER(C.c_float)), C.c_size_t(len(x)))
    assert np.array_equal(y.view(np.uint32), g["cbrt"].view(np.uint32))
