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
