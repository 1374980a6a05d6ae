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

import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
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


def sleef():
    rng = np.random.default_rng(3)
    n = 16384
    # exp arguments: the ranges the path uses (shrinkage: [-40, 0]; gamma LUT: [-12, 0.1]) plus extremes
    xe = np.concatenate([rng.uniform(-110, 5, n // 2), rng.uniform(-2, 2, n // 4), rng.uniform(-104.5, -103.5, n // 8),
                         rng.uniform(80, 90, n // 8)]).astype(np.float32)
    xe[:8] = [0.0, -0.0, -104.0, -103.99999, 88.0, -87.5, 1e-8, -1e-8]
    xl = np.concatenate([rng.uniform(1e-6, 70000, n // 2), np.exp(rng.uniform(-80, 80, n // 2))]).astype(np.float32)
    xl[:10] = [1.0, 0.5, 2.0, 65535.0, 0.7071, 1.4142135, 1e-38, 1e-42, 3e38, 10.0]
    out = {"xe": xe, "xl": xl}
    for name, fn, src in (("exp_s", R.ref_xexpf, xe), ("exp_v", R.ref_vexpf, xe), ("exp_vn", R.ref_vexpf_nocheck, xe),
                          ("log_s", R.ref_xlogf, xl), ("log_v", R.ref_vlogf, xl), ("log_vn", R.ref_vlogf_nocheck, xl)):
        y = np.empty(n, np.float32)
        fn(P(src), P(y), C.c_size_t(n))
        out[name] = y
    a = rng.uniform(0.001, 100.0, n).astype(np.float32)
    b = rng.uniform(-3.0, 3.0, n).astype(np.float32)
    y = np.empty(n, np.float32)
    R.ref_pow_F(P(a), P(b), P(y), C.c_size_t(n))
    out.update(pow_a=a, pow_b=b, pow_F=y)
    x01 = rng.uniform(0.0, 1.2, n).astype(np.float32)
    for base in (10.0, 101.0):
        y1, y2 = np.empty(n, np.float32), np.empty(n, np.float32)
        R.ref_xlin2log(P(x01), C.c_float(base), P(y1), C.c_size_t(n))
        R.ref_xlog2lin(P(x01), C.c_float(base), P(y2), C.c_size_t(n))
        out[f"lin2log_{int(base)}"] = y1
        out[f"log2lin_{int(base)}"] = y2
    out["x01"] = x01
    np.savez_compressed(os.path.join(HERE, "sleef.npz"), **out)


def sleef2():
    """functions added after sleef.npz was frozen (kept in a second file so the first stays byte-stable)"""
    rng = np.random.default_rng(11)
    n = 8192
    xc = np.concatenate([rng.uniform(1.0, 40.0, n // 2), np.exp(rng.uniform(-60, 60, n // 4)), -np.exp(rng.uniform(-20, 20, n // 4))]).astype(np.float32)
    xc[:6] = [1.0, 8.0, 27.0, 1.0000153, 0.0, 1e-40]
    y = np.empty(n, np.float32)
    R.ref_xcbrtf(P(xc), P(y), C.c_size_t(n))
    # xatan2f / xsincosf(float) as the Jzazbz hue code uses them (color.cc:6690-6703): small-magnitude az/bz, hue in [-pi, pi]
    ay = np.concatenate([rng.uniform(-0.2, 0.2, n // 2), rng.uniform(-1e-4, 1e-4, n // 4), rng.uniform(-50, 50, n // 4)]).astype(np.float32)
    ax = np.concatenate([rng.uniform(-0.2, 0.2, n // 2), rng.uniform(-1e-4, 1e-4, n // 4), rng.uniform(-50, 50, n // 4)]).astype(np.float32)
    ay[:8] = [0.0, -0.0, 1.0, -1.0, 0.0, 1e-30, np.inf, 0.3]
    ax[:8] = [1.0, -1.0, 0.0, 0.0, 0.0, -1e-30, 1.0, np.nan]
    at = np.empty(n, np.float32)
    R.ref_xatan2f(P(ay), P(ax), P(at), C.c_size_t(n))
    sd = np.concatenate([rng.uniform(-3.5, 3.5, n // 2), rng.uniform(-40, 40, n // 4), rng.uniform(-1e-3, 1e-3, n // 4)]).astype(np.float32)
    sd[:6] = [0.0, -0.0, np.pi / 2, np.pi, -np.pi, 0.7853982]
    sn, cs = np.empty(n, np.float32), np.empty(n, np.float32)
    R.ref_xsincosf(P(sd), P(sn), P(cs), C.c_size_t(n))
    np.savez_compressed(os.path.join(HERE, "sleef2.npz"), xc=xc, cbrt=y, ay=ay, ax=ax, atan2=at, sd=sd, sin=sn, cos=cs)


def sleef3():
    """4-lane xatan2f as RGB_denoise_info's hue map uses it (ipdenoise.cc:395-402): Lab a/b magnitudes, plus the special cases"""
    rng = np.random.default_rng(12)
    n = 8192
    ay = np.concatenate([rng.uniform(-30000, 30000, n // 2), rng.uniform(-300, 300, n // 4), rng.uniform(-1e-3, 1e-3, n // 4)]).astype(np.float32)
    ax = np.concatenate([rng.uniform(-30000, 30000, n // 2), rng.uniform(-300, 300, n // 4), rng.uniform(-1e-3, 1e-3, n // 4)]).astype(np.float32)
    ay[:12] = [0.0, -0.0, 1.0, -1.0, 0.0, 1e-30, np.inf, 0.3, -np.inf, 5.0, -0.0, 0.0]
    ax[:12] = [1.0, -1.0, 0.0, -0.0, 0.0, -1e-30, 1.0, np.nan, np.inf, -np.inf, -0.0, -2.0]
    at = np.empty(n, np.float32)
    R.ref_vatan2f(P(ay), P(ax), P(at), C.c_size_t(n))
    np.savez_compressed(os.path.join(HERE, "sleef3.npz"), ay=ay, ax=ax, atan2_v=at)


from make_golden_inputs import wavelet_input  # noqa: E402


def wavelet():
    import hashlib
    R.ref_wavelet_new.restype = C.c_void_p
    R.ref_wavelet_band.restype = fp
    R.ref_wavelet_coeff0.restype = fp
    out = {}
    for (w, h, lv, full) in ((129, 97, 5, True), (258, 196, 6, False), (321, 255, 5, False)):
        src = wavelet_input(w, h, w + h)
        d = C.c_void_p(R.ref_wavelet_new(P(src), w, h, lv))
        assert R.ref_wavelet_maxlevel(d) == lv
        w2, h2 = R.ref_wavelet_W(d, 0), R.ref_wavelet_H(d, 0)
        bands = np.empty((lv, 3, h2, w2), np.float32)
        for l in range(lv):
            assert (R.ref_wavelet_W(d, l), R.ref_wavelet_H(d, l)) == (w2, h2)
            for k in range(3):
                bands[l, k] = np.ctypeslib.as_array(R.ref_wavelet_band(d, l, k + 1), shape=(h2, w2))
        c0 = np.ctypeslib.as_array(R.ref_wavelet_coeff0(d), shape=(h2, w2)).copy()
        strides = [R.ref_wavelet_stride(d, l) for l in range(lv)]
        # perturb the coefficients like a shrinkage would, then reconstruct
        for l in range(lv):
            for k in range(3):
                b = np.ctypeslib.as_array(R.ref_wavelet_band(d, l, k + 1), shape=(h2, w2))
                b *= np.float32(0.5 + 0.1 * (l + k))
        rec = np.full((h, w), 7.0, np.float32)
        R.ref_wavelet_reconstruct(d, P(rec), C.c_float(1.0))
        R.ref_wavelet_delete(d)
        key = f"{w}x{h}x{lv}"
        out[key + "_strides"] = np.array(strides)
        if full:
            out[key + "_bands"] = bands
            out[key + "_coeff0"] = c0
            out[key + "_recon"] = rec
        else:
            out[key + "_sha"] = np.frombuffer(hashlib.sha256(bands.tobytes() + c0.tobytes() + rec.tobytes()).digest(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "wavelet.npz"), **out)


def rescale_and_matrices():
    """rescaleBilinear (rescale.h:53-74; guidedFilter's sub-sampling and detail_mask's quarter-resolution plane go through it) for
    down- and up-scaling with non-integer ratios, and the Rec2020 working-space matrices (iccmatrices.h:151-161)."""
    rng = np.random.default_rng(77)
    out = {}
    for k, (ws, hs, wd, hd) in enumerate(((97, 61, 33, 21), (33, 21, 97, 61), (128, 96, 32, 24), (50, 37, 149, 111), (40, 40, 40, 40))):
        src = (rng.uniform(0.0, 65535.0, (hs, ws)) * rng.uniform(0.2, 1.0, (hs, ws))).astype(np.float32)
        dst = np.empty((hd, wd), np.float32)
        R.ref_rescale_bilinear(P(src), ws, hs, P(dst), wd, hd)
        out[f"src{k}"] = src
        out[f"dst{k}"] = dst
    a, b = np.empty(9, np.float32), np.empty(9, np.float32)
    R.ref_rec2020_matrices(P(a), P(b))
    out["xyz_rec2020"] = a.reshape(3, 3)
    out["rec2020_xyz"] = b.reshape(3, 3)
    np.savez_compressed(os.path.join(HERE, "rescale.npz"), **out)


def linalgebra():
    """dot_product(Mat33<float>, Vec3<float>) of linalgebra.h:226-239: what NeutralToneCurve applies per pixel (to_out, to_work, the D50 / D65
    adaptation matrices of Jzazbz).  Random matrices and vectors incl. special values."""
    rng = np.random.default_rng(91)
    n = 4096
    m = rng.normal(0, 1.5, (n, 9)).astype(np.float32)
    v = (rng.normal(0, 1, (n, 3)) * np.exp2(rng.integers(-20, 20, (n, 1)))).astype(np.float32)
    v[:16, 0] = 0.0
    v[16:32, 1] = -0.0
    m[32:48, 4] = np.float32(1e-30)
    r = np.empty((n, 3), np.float32)
    for k in range(n):
        R.ref_mat33_dot_vec3(P(m[k]), P(v[k]), P(r[k]))
    np.savez_compressed(os.path.join(HERE, "linalgebra.npz"), m=m, v=v, r=r)


def sleef_double():
    """xlog / xexp in double (sleef.h:519-571) as DiagonalCurve's parametric form uses them (curves.h:92-156, diagonalcurves.cc:106-131,
    448-470): arguments around the curve's working range, the wide range, and the special cases"""
    rng = np.random.default_rng(13)
    n = 8192
    dp = C.POINTER(C.c_double)
    xl = np.concatenate([rng.uniform(1e-3, 4.0, n // 2), np.exp(rng.uniform(-700, 700, n // 4)), rng.uniform(-1.0, 1e-300, n // 4)])
    xl[:10] = [1.0, 2.0, 0.5, 0.0, -0.0, np.inf, -1.0, np.nan, 4.9e-324, 1.7e308]
    xe = np.concatenate([rng.uniform(-5.0, 5.0, n // 2), rng.uniform(-745.0, 709.0, n // 4), rng.uniform(-1e-8, 1e-8, n // 4)])
    xe[:8] = [0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 709.5]
    yl, ye = np.empty(n), np.empty(n)
    R.ref_xlog(xl.ctypes.data_as(dp), yl.ctypes.data_as(dp), C.c_size_t(n))
    R.ref_xexp(xe.ctypes.data_as(dp), ye.ctypes.data_as(dp), C.c_size_t(n))
    np.savez_compressed(os.path.join(HERE, "sleef_d.npz"), xl=xl, log=yl, xe=xe, exp=ye)


def halffloat():
    """DNG_FloatToHalf (halffloat.h:9-46): every half value as a float, its float neighbours, the midpoints between adjacent halfs (ties and
    both sides of them), exponents from the subnormal floats to +-inf / NaN, and random bit patterns"""
    rng = np.random.default_rng(7)
    h = np.arange(65536, dtype=np.uint16).view(np.float16).astype(np.float32)
    hb = h.view(np.uint32)
    fin = np.isfinite(h)
    cand = [hb, hb[fin] + 1, hb[fin] - 1]
    for off in (0x0FFF, 0x1000, 0x1001, 0x1FFF, 0x2000):          # around the rounding bit of a normal half's 13 dropped bits
        cand.append(hb[fin] + np.uint32(off))
    cand.append(rng.integers(0, 1 << 32, 120000, dtype=np.uint64).astype(np.uint32))
    for e in range(0, 256):                                       # every float exponent with a few mantissas
        cand.append((np.uint32(e) << np.uint32(23)) | np.array([0, 1, 0x1000, 0x7FFFFF, 0x400000, 0x3FF000, 0x3FF001], np.uint32))
        cand.append(np.uint32(0x80000000) | (np.uint32(e) << np.uint32(23)) | np.array([0, 0x1FFF, 0x2000, 0x7FE000], np.uint32))
    x = np.unique(np.concatenate([c.astype(np.uint32) for c in cand])).view(np.float32)
    y = np.empty(x.size, np.uint16)
    R.ref_float_to_half(P(x), y.ctypes.data_as(C.POINTER(C.c_uint16)), C.c_size_t(x.size))
    np.savez_compressed(os.path.join(HERE, "halffloat.npz"), x_bits=x.view(np.uint32), half=y)


if __name__ == "__main__":
    halffloat()
    sleef_double()
    linalgebra()
    rescale_and_matrices()
    wavelet()
    helpers()
    lutf()
    sleef()
    sleef2()
    sleef3()
    print("golden vectors written to", HERE)
