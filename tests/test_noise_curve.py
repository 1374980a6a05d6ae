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


def test_xatan2f_xsincosf_match_reference():
    g = np.load(os.path.join(G, "sleef2.npz"))
    P = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    ay, ax, sd = (np.ascontiguousarray(g[k]) for k in ("ay", "ax", "sd"))
    r = np.empty_like(ay)
    O.lib().oracle_t_xatan2f(P(ay), P(ax), P(r), C.c_size_t(len(ay)))
    same = (r.view(np.uint32) == g["atan2"].view(np.uint32)) | (np.isnan(r) & np.isnan(g["atan2"]))
    assert same.all()
    sn, cs = np.empty_like(sd), np.empty_like(sd)
    O.lib().oracle_t_xsincosf(P(sd), P(sn), P(cs), C.c_size_t(len(sd)))
    assert np.array_equal(sn.view(np.uint32), g["sin"].view(np.uint32)) and np.array_equal(cs.view(np.uint32), g["cos"].view(np.uint32))


def test_oracle_neutral_curve_keeps_greys_and_range():
    """identity LUT: greys come back (sat ~ 0, no luminance change); output stays inside [0, whitept]."""
    lut = np.arange(65536, dtype=np.float32)
    g = np.linspace(100.0, 60000.0, 256, dtype=np.float32)[None, :].repeat(4, 0)
    out = O.tone_neutral([g, g, g], lut, 1.0)
    for p in out:
        assert np.allclose(p, g, rtol=2e-3, atol=2.0)
    rng = np.random.default_rng(0)
    img = [rng.uniform(0, 70000, (32, 64)).astype(np.float32) for _ in range(3)]
    out = O.tone_neutral(img, lut, 1.0)
    assert all(np.isfinite(p).all() and p.min() >= 0 and p.max() <= 65535.0 for p in out)
    st = O.neutral_state()
    assert -np.pi < st.bhue < st.rhue < st.yhue < np.pi and st.rrange > 0 and st.yrange > 0
