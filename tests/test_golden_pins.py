"""CPU: the oracle's primitives against golden vectors produced by the REFERENCE's own headers
(tests/golden/make_golden.py -> oracle/_ref).  Bit-exact, NaN payloads excluded."""
import ctypes as C
import os

import numpy as np

import oracle_lib as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
fp = C.POINTER(C.c_float)


def P(a):
    return a.ctypes.data_as(fp)


def same_bits(a, b):
    nan = np.isnan(a) & np.isnan(b)
    return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | nan))


def test_helpers_match_reference_headers():
    g = np.load(os.path.join(G, "helpers.npz"))
    a, b, c = g["a"], g["b"], g["c"]
    n = len(a)
    L = O.lib()
    mn, mx, med, itp, d2, d4 = [np.empty(n, np.float32) for _ in range(6)]
    L.oracle_t_minmax(P(a), P(b), P(mn), P(mx), C.c_size_t(n))
    L.oracle_t_median3(P(a), P(b), P(c), P(med), C.c_size_t(n))
    L.oracle_t_intp(P(c), P(a), P(b), P(itp), C.c_size_t(n))
    L.oracle_t_xdiv2f(P(a), P(d2), C.c_size_t(n))
    L.oracle_t_xdivf2(P(a), P(d4), C.c_size_t(n))
    assert same_bits(mn, g["vmin"]) and same_bits(mx, g["vmax"])
    assert same_bits(med, g["median3"])
    assert same_bits(itp, g["vintpf"])
    assert same_bits(d2, g["xdiv2f"]) and same_bits(d4, g["xdivf2"])


def test_lutf_scalar_matches_reference_lut_h():
    g = np.load(os.path.join(G, "lutf.npz"))
    size = int(g["table_size"])
    x = np.arange(size, dtype=np.float64) / (size - 1)
    table = (np.sqrt(x) * 65535.0).astype(np.float32)
    got = O.lutf(table, g["index"])
    assert same_bits(got, g["scalar"])
    n4 = len(g["index"]) // 4 * 4
    gotv = O.lutf_vec(table, g["index"][:n4])
    assert same_bits(gotv, g["vector"][:n4])     # LUTf::operator[](vfloat)


def test_sleef_scalar_and_vector_forms_match_reference():
    g = np.load(os.path.join(G, "sleef.npz"))
    L = O.lib()
    n = len(g["xe"])
    for name, fn, src in (("exp_s", L.oracle_t_xexpf_s, g["xe"]), ("exp_v", L.oracle_t_xexpf_v, g["xe"]),
                          ("exp_vn", L.oracle_t_xexpf_vn, g["xe"]), ("log_s", L.oracle_t_xlogf_s, g["xl"]),
                          ("log_v", L.oracle_t_xlogf_v, g["xl"]), ("log_vn", L.oracle_t_xlogf_vn, g["xl"])):
        y = np.empty(n, np.float32)
        src = np.ascontiguousarray(src)
        fn(P(src), P(y), C.c_size_t(n))
        assert same_bits(y, g[name]), name
    # the scalar and vector exp really are different functions (guards against collapsing them)
    assert not np.array_equal(g["exp_s"].view(np.uint32), g["exp_v"].view(np.uint32))
    y = np.empty(n, np.float32)
    a, b = np.ascontiguousarray(g["pow_a"]), np.ascontiguousarray(g["pow_b"])
    L.oracle_t_pow_F(P(a), P(b), P(y), C.c_size_t(n))
    assert same_bits(y, g["pow_F"])
    x01 = np.ascontiguousarray(g["x01"])
    for base in (10, 101):
        L.oracle_t_xlin2log(P(x01), C.c_float(base), P(y), C.c_size_t(n))
        assert same_bits(y, g[f"lin2log_{base}"])
        L.oracle_t_xlog2lin(P(x01), C.c_float(base), P(y), C.c_size_t(n))
        assert same_bits(y, g[f"log2lin_{base}"])


def test_wavelet_matches_reference_decomposition():
    import hashlib
    import sys
    sys.path.insert(0, G)
    from make_golden_inputs import wavelet_input
    g = np.load(os.path.join(G, "wavelet.npz"))
    for (w, h, lv, full) in ((129, 97, 5, True), (258, 196, 6, False), (321, 255, 5, False)):
        key = f"{w}x{h}x{lv}"
        src = wavelet_input(w, h, w + h)
        d = O.wavelet_decompose(src, lv)
        bands, c0, views = O.wavelet_bands(d)
        assert [O.lib().oracle_wavelet_skip(l) for l in range(lv)] == list(g[key + "_strides"])
        i = 0
        for l in range(lv):
            for k in range(3):
                views[i] *= np.float32(0.5 + 0.1 * (l + k))
                i += 1
        rec = O.wavelet_reconstruct(d, h, w)
        if full:
            assert same_bits(bands, g[key + "_bands"])
            assert same_bits(c0, g[key + "_coeff0"])
            assert same_bits(rec, g[key + "_recon"])
        else:
            sha = np.frombuffer(hashlib.sha256(bands.tobytes() + c0.tobytes() + rec.tobytes()).digest(), dtype=np.uint8)
            assert np.array_equal(sha, g[key + "_sha"])


def test_rescale_bilinear_matches_reference_rescale_h():
    """oracle_rescale_bilinear against the reference's own rescaleBilinear (rescale.h:53-74, compiled in place into oracle/_ref)."""
    g = np.load(os.path.join(G, "rescale.npz"))
    L = O.lib()
    k = 0
    while f"src{k}" in g:
        src, ref = np.ascontiguousarray(g[f"src{k}"]), g[f"dst{k}"]
        dst = np.empty_like(ref)
        L.oracle_rescale_bilinear(P(src), src.shape[1], src.shape[0], P(dst), ref.shape[1], ref.shape[0])
        assert same_bits(dst, ref), k
        k += 1
    assert k == 5


def test_working_space_matrices_are_the_reference_constants():
    """The Rec2020 matrices the tests and bench.py feed to both sides are typed by hand; they have to be iccmatrices.h:151-161."""
    g = np.load(os.path.join(G, "rescale.npz"))
    assert np.array_equal(O.REC2020_WS.astype(np.float32), g["xyz_rec2020"])
    assert np.array_equal(O.REC2020_WS_D.astype(np.float32), g["xyz_rec2020"])
    assert np.array_equal(O.REC2020_IWS_D.astype(np.float32), g["rec2020_xyz"])
    import re
    src = open(os.path.join(os.path.dirname(G), "..", "bench.py")).read()
    for m, name in ((g["xyz_rec2020"], "ws"), (g["rec2020_xyz"], "iws_n")):
        txt = re.search(name + r" = np\.array\((\[\[.*?\]\])\)", src, re.S).group(1)
        assert np.array_equal(np.array(eval(txt), dtype=np.float32), m), name


def test_mat_vec_matches_reference_linalgebra_h():
    """the oracle's mat_vec (tonecurve.c: NEUTRAL's to_out / to_work / chromatic-adaptation products) against dot_product(Mat33, Vec3) of the
    reference's own linalgebra.h:226-239 (compiled in place into oracle/_ref)"""
    g = np.load(os.path.join(G, "linalgebra.npz"))
    m, v, ref = np.ascontiguousarray(g["m"]), np.ascontiguousarray(g["v"]), g["r"]
    out = np.empty_like(ref)
    O.lib().oracle_t_mat_vec(P(m), P(v), P(out), C.c_size_t(len(m)))
    assert same_bits(out, ref)


def test_double_xlog_xexp_match_reference_sleef_h():
    """the double-precision forms behind DiagonalCurve's parametric branch (oracle/sleef.c oracle_xlog / oracle_xexp)"""
    g = np.load(os.path.join(G, "sleef_d.npz"))
    L = O.lib()
    dp = C.POINTER(C.c_double)
    for src, want, fn in ((g["xl"], g["log"], L.oracle_t_xlog), (g["xe"], g["exp"], L.oracle_t_xexp)):
        x = np.ascontiguousarray(src)
        y = np.empty_like(x)
        fn(x.ctypes.data_as(dp), y.ctypes.data_as(dp), C.c_size_t(len(x)))
        nan = np.isnan(y) & np.isnan(want)
        assert bool(np.all((y.view(np.uint64) == want.view(np.uint64)) | nan))


def test_float_to_half_matches_reference_halffloat_h():
    """oracle/pixelops.c float_to_half_dng against DNG_FloatToHalf itself (halffloat.h:9-46 compiled in place): ~600 000 floats -- every half
    value, its neighbours, the rounding ties, every exponent, random patterns"""
    g = np.load(os.path.join(G, "halffloat.npz"))
    x = g["x_bits"].view(np.float32)
    assert x.size > 500000
    assert np.array_equal(O.float_to_half(x), g["half"])
    # and the scanline form that uses it: value / 65535 first (imagefloat.cc:150-158)
    v = (x[::16][:30000] * np.float32(65535.0)).reshape(100, 100, 3)
    planes = [np.ascontiguousarray(v[:, :, c]) for c in range(3)]
    want = O.float_to_half((v / np.float32(65535.0)).astype(np.float32))
    assert np.array_equal(O.get_scanlines(planes, 16, True), want)


def test_parametric_curve_is_continuous_with_its_lut_and_monotonic_above_one():
    """oracle_parametric_getval (diagonalcurves.cc:448-470, unpinned: curves.h needs glibmm): a float64 model of the same formulas
    with libm's log / exp agrees to rounding level, getVal(1) is 1, and the tail keeps rising above 1"""
    import math
    p = [2.0, 0.25, 0.5, 0.75, 30.0, 20.0, -15.0, -25.0, 0.0]
    O.set_parametric_curve(p)
    try:
        L = O.lib()
        dp = C.POINTER(C.c_double)
        t = np.concatenate([np.linspace(1e-6, 1.0, 500), np.linspace(1.0, 2.5, 300)])
        y = np.empty_like(t)
        L.oracle_t_parametric_getval(t.ctypes.data_as(dp), y.ctypes.data_as(dp), C.c_size_t(len(t)))

        def basel(x, m1, m2):
            if x == 0.0:
                return 0.0
            k = math.sqrt((m1 - 1.0) * (m1 - m2) * 0.5) / (1.0 - m2)
            l = (m1 - m2) / (1.0 - m2) + k
            lx = math.log(x) if x > 0 else float("nan")
            return m2 * x + (1.0 - m2) * (2.0 - math.exp(k * lx)) * math.exp(l * lx)

        def cupper(x, m, hr):
            if hr > 1.0:
                return 1.0 - basel(1.0 - x, m, 2.0 * (hr - 1.0) / m)
            x1 = (1.0 - hr) / m
            if x >= x1 + hr:
                return 1.0
            if x < x1:
                return x * m
            return 1.0 - hr + hr * (1.0 - basel(1.0 - (x - x1) / hr, m, 0))

        def clower(x, m, sr):
            return 1.0 - cupper(1.0 - x, m, sr)

        def pfull(x, prot, sh, hl):
            p01 = clower(x * 2, 2.0, prot) * 0.5 if x <= 0.5 else 0.5 + cupper((x - 0.5) * 2, 2.0, prot) * 0.5
            p10 = cupper(x * 2, 2.0, prot) * 0.5 if x <= 0.5 else 0.5 + clower((x - 0.5) * 2, 2.0, prot) * 0.5
            return (1 - sh) * (1 - hl) * clower(x, 2.0, prot) + sh * hl * cupper(x, 2.0, prot) + (1 - sh) * hl * p01 + sh * (1 - hl) * p10

        x = [p[0]] + [min(max(v, 0.001), 0.99) for v in p[1:4]] + [(v + 100.0) / 200.0 for v in p[4:8]] + [p[8] / 100.0]
        mc = -math.log(2.0) / math.log(x[2])
        mfc = math.exp(math.log(pfull(0.5, x[8], x[6], x[5])) / mc)
        msc = -math.log(2.0) / math.log(x[1] / x[2])
        mhc = -math.log(2.0) / math.log((x[3] - x[2]) / (1 - x[2]))

        def getval(tt):
            base = pfull(math.exp(mc * math.log(tt)), x[8], x[6], x[5])
            st = math.exp(math.log(base) / mc)
            if tt < x[2]:
                sb = pfull(math.exp(msc * math.log(st / mfc)), x[8], x[7], 0.5)
                return mfc * math.exp(math.log(sb) / msc)
            hb = pfull(math.exp(mhc * math.log((st - mfc) / (1 - mfc))), x[8], 0.5, x[4])
            return mfc + (1 - mfc) * math.exp(math.log(hb) / mhc)
        model = np.array([getval(float(v)) for v in t[:500]])
        assert np.allclose(y[:500], model, rtol=1e-9, atol=1e-12)
        assert abs(y[499] - 1.0) < 1e-9
        assert np.all(np.isfinite(y[500:])) and np.all(np.diff(y[500:]) >= 0) and y[-1] > y[500]      # the tail keeps rising above 1
    finally:
        O.set_curve_tail(0)
