#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ from the REFERENCE itself.

Runs only in the build container (needs /root/reference): it builds oracle/_ref/libartref.so
from the reference headers where they lie (oracle/Makefile.ref) and records inputs + outputs
of the reference's own code as small .npz fixtures.  The fixtures are data; no reference
source is stored.  Re-run: `python tests/golden/make_golden.py`.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "-f", "Makefile.ref"])
R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libartref.so"))
fp = C.POINTER(C.c_float)


def P(a):
    return a.ctypes.data_as(fp)


def special_floats(rng, n):
    x = rng.standard_normal(n).astype(np.float32) * np.float32(3.0)
    sp = np.array([0.0, -0.0, 1.0, -1.0, 0.5, 2.0, 1e-38, -1e-38, 1e-45, 3e38, -3e38, np.inf, -np.inf, np.nan,
                   0.75, 0.25, 65535.0, 1e-5, 1e-10], dtype=np.float32)
    x[: len(sp)] = sp
    return x


def helpers():
    rng = np.random.default_rng(1)
    n = 4096
    a, b, c = special_floats(rng, n), special_floats(rng, n)[::-1].copy(), rng.uniform(-2, 2, n).astype(np.float32)
    b[:32] = a[:32]  # equal operands, incl. +-0 and NaN pairs
    mn, mx, med, itp, d2, d4 = [np.empty(n, np.float32) for _ in range(6)]
    R.ref_vminmax(P(a), P(b), P(mn), P(mx), C.c_size_t(n))
    R.ref_vmedian3(P(a), P(b), P(c), P(med), C.c_size_t(n))
    R.ref_vintpf(P(c), P(a), P(b), P(itp), C.c_size_t(n))
    R.ref_xdiv2f(P(a), P(d2), C.c_size_t(n))
    R.ref_xdivf2(P(a), P(d4), C.c_size_t(n))
    np.savez_compressed(os.path.join(HERE, "helpers.npz"), a=a, b=b, c=c, vmin=mn, vmax=mx, median3=med, vintpf=itp, xdiv2f=d2, xdivf2=d4)


def lutf():
    rng = np.random.default_rng(2)
    size = 65536
    x = np.arange(size, dtype=np.float64) / (size - 1)
    table = (np.sqrt(x) * 65535.0).astype(np.float32)
    n = 8192
    idx = rng.uniform(-10.0, 65600.0, n).astype(np.float32)
    idx[:12] = [0.0, -0.0, -1.0, 65534.0, 65534.5, 65535.0, 65535.5, 65536.0, 1e9, np.nan, 0.999999, 32767.5]
    ys, yv = np.empty(n, np.float32), np.empty(n, np.float32)
    R.ref_lutf(P(table), C.c_size_t(size), P(idx), P(ys), P(yv), C.c_size_t(n))
    np.savez_compressed(os.path.join(HERE, "lutf.npz"), table_size=size, index=idx, scalar=ys, vector=yv)


if __name__ == "__main__":
    helpers()
    lutf()
    print("golden vectors written to", HERE)
