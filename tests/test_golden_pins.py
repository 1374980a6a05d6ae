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
