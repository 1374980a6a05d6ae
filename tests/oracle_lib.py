"""ctypes loader for the CPU oracle (oracle/liboracle.so).  Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")
_LIB = None

_fp = C.POINTER(C.c_float)


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(_fp)


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_ORACLE_DIR, "liboracle.so")
        srcs = [os.path.join(_ORACLE_DIR, f) for f in os.listdir(_ORACLE_DIR) if f.endswith((".c", ".h"))]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.check_call(["make", "-s", "-C", _ORACLE_DIR, "liboracle.so"])
        _LIB = C.CDLL(so)
        _LIB.oracle_amaze_arena_floats.restype = C.c_size_t
    return _LIB


def _planes(h, w):
    return [np.full((h, w), np.nan, dtype=np.float32) for _ in range(3)]


def rcd(raw: np.ndarray, filters: int):
    raw = np.ascontiguousarray(raw, dtype=np.float32)
    h, w = raw.shape
    r, g, b = _planes(h, w)
    rc = lib().oracle_rcd_demosaic(_ptr(raw), C.c_size_t(w), w, h, C.c_uint(filters), _ptr(r), _ptr(g), _ptr(b), C.c_size_t(w))
    if rc != 0:
        raise RuntimeError(f"oracle_rcd_demosaic failed: {rc}")
    return r, g, b


def amaze(raw: np.ndarray, filters: int, initial_gain: float = 1.0, border: int = 4):
    raw = np.ascontiguousarray(raw, dtype=np.float32)
    h, w = raw.shape
    r, g, b = _planes(h, w)
    rc = lib().oracle_amaze_demosaic(_ptr(raw), C.c_size_t(w), w, h, C.c_uint(filters), C.c_double(initial_gain), border,
                                     _ptr(r), _ptr(g), _ptr(b), C.c_size_t(w))
    if rc != 0:
        raise RuntimeError(f"oracle_amaze_demosaic failed: {rc}")
    return r, g, b


def amaze_tiles_stale(raw: np.ndarray, filters: int, initial_gain: float, order: str = "raster"):
    """AMaZE tile by tile on ONE arena that is zeroed once and never cleared again (what a
    reference thread does: calloc once, amaze_demosaic_RT.cc:124), visiting the tiles in
    raster or reverse order."""
    raw = np.ascontiguousarray(raw, dtype=np.float32)
    h, w = raw.shape
    r, g, b = _planes(h, w)
    L = lib()
    arena = np.zeros(L.oracle_amaze_arena_floats(), dtype=np.float32)
    clip_pt = np.float32(1.0 / initial_gain)
    clip_pt8 = np.float32(0.8 / initial_gain)
    tiles = [(top, left) for top in range(-16, h, 128) for left in range(-16, w, 128)]
    if order == "reverse":
        tiles = tiles[::-1]
    for top, left in tiles:
        L.oracle_amaze_tile(_ptr(raw), C.c_size_t(w), w, h, C.c_uint(filters), C.c_float(clip_pt), C.c_float(clip_pt8),
                            top, left, _ptr(r), _ptr(g), _ptr(b), C.c_size_t(w), _ptr(arena), 1)
    return r, g, b


def _p3(planes):
    arr = (_fp * 3)(*[_ptr(p) for p in planes])
    return arr


def get_image(planes, sx1, sy1, w, h, mul, do_clip):
    planes = [np.ascontiguousarray(p, dtype=np.float32) for p in planes]
    out = [np.full((h, w), np.nan, dtype=np.float32) for _ in range(3)]
    m = (C.c_float * 3)(*[float(v) for v in mul])
    lib().oracle_get_image(_p3(planes), C.c_size_t(planes[0].shape[1]), sx1, sy1, _p3(out), C.c_size_t(w), w, h, m, int(do_clip))
    return out


def convert_color_space(img, mat):
    img = [np.array(p, dtype=np.float32, order="C") for p in img]
    h, w = img[0].shape
    m = (C.c_double * 9)(*[float(v) for v in np.asarray(mat, dtype=np.float64).reshape(9)])
    lib().oracle_convert_color_space(_p3(img), C.c_size_t(w), w, h, m)
    return img


def exposure(img, exp_scale, black):
    img = [np.array(p, dtype=np.float32, order="C") for p in img]
    h, w = img[0].shape
    lib().oracle_exposure(_p3(img), C.c_size_t(w), w, h, C.c_float(exp_scale), C.c_float(black))
    return img


def set_parametric_curve(p):
    """DiagonalCurve(DCT_Parametric) behind setLutVal: values above 65535 take getVal's analytic form (oracle kind 4)"""
    arr = (C.c_double * len(p))(*[float(v) for v in p])
    lib().oracle_set_parametric_curve(arr, len(p))


def set_curve_tail(kind=0, y_last=1.0):
    """What the tone curve's Curve object returns above 1.0 (curves::setLutVal): 0 no Curve object (LUT clip), 1 constant y_last, 2 identity."""
    C.c_int.in_dll(lib(), "oracle_curve_tail_kind").value = int(kind)
    C.c_double.in_dll(lib(), "oracle_curve_tail_y").value = float(y_last)


def tone_std(img, lut, whitept=1.0, filmlike_clip=True):
    img = [np.array(p, dtype=np.float32, order="C") for p in img]
    h, w = img[0].shape
    if filmlike_clip:
        lib().oracle_filmlike_clip(_p3(img), C.c_size_t(w), w, h, C.c_float(whitept))
    if lut is not None:
        lut = np.ascontiguousarray(lut, dtype=np.float32)
        lib().oracle_tone_curve_std(_p3(img), C.c_size_t(w), w, h, _ptr(lut))
    return img


def lutf(table, x):
    table = np.ascontiguousarray(table, dtype=np.float32)
    L = lib()
    L.oracle_lutf.restype = C.c_float
    return np.array([L.oracle_lutf(_ptr(table), len(table), C.c_float(float(v))) for v in x], dtype=np.float32)


class _OW(C.Structure):
    _fields_ = [("w", C.c_int), ("h", C.c_int), ("w2", C.c_int), ("h2", C.c_int), ("nlevels", C.c_int),
                ("band", (_fp * 4) * 10), ("coeff0", _fp)]


def wavelet_decompose(src, maxlvl):
    """Returns (bands[l,3,h2,w2] view-copies, coeff0, handle). handle must be freed/reconstructed."""
    src = np.ascontiguousarray(src, dtype=np.float32)
    h, w = src.shape
    L = lib()
    L.oracle_wavelet_decompose.restype = C.POINTER(_OW)
    d = L.oracle_wavelet_decompose(_ptr(src), w, h, maxlvl)
    return d


def wavelet_bands(d):
    o = d.contents
    bands = np.empty((o.nlevels, 3, o.h2, o.w2), np.float32)
    views = []
    for l in range(o.nlevels):
        for k in range(3):
            v = np.ctypeslib.as_array(o.band[l][k + 1], shape=(o.h2, o.w2))
            bands[l, k] = v
            views.append(v)
    c0 = np.ctypeslib.as_array(o.coeff0, shape=(o.h2, o.w2)).copy()
    return bands, c0, views


def wavelet_reconstruct(d, h, w, blend=1.0, fill=7.0):
    rec = np.full((h, w), fill, np.float32)
    lib().oracle_wavelet_reconstruct(d, _ptr(rec), C.c_float(blend))
    lib().oracle_wavelet_free(d)
    return rec


class DenoiseParams(C.Structure):
    _fields_ = [("luminance", C.c_double), ("luminanceDetail", C.c_double), ("chrominance", C.c_double),
                ("chrominanceRedGreen", C.c_double), ("chrominanceBlueYellow", C.c_double), ("gamma", C.c_double),
                ("expcomp", C.c_double), ("scale", C.c_double), ("autoch", C.c_int), ("aggressive", C.c_int), ("detail_thresh", C.c_int), ("lab_mode", C.c_int), ("iws", C.c_float * 9)]


def default_denoise_params(**kw):
    p = DenoiseParams(40.0, 50.0, 15.0, 0.0, 0.0, 1.7, 0.0, 1.0, 0, 0, 0, 0)
    p.iws[:] = [float(v) for v in REC2020_IWS_D.astype(np.float32).reshape(9)]
    for k, v in kw.items():
        setattr(p, k, v)
    return p


REC2020_WS = np.array([[0.6734241, 0.1656411, 0.1251286],
                       [0.2790177, 0.6753402, 0.0456377],
                       [-0.0019300, 0.0299784, 0.7973330]], dtype=np.float32)


def rgb_denoise(img, params=None, wpi=REC2020_WS, noisevarchrom=None, want_L=False, detail_recovery=False, want_resid=False):
    img = [np.array(p, dtype=np.float32, order="C") for p in img]
    h, w = img[0].shape
    params = params or default_denoise_params()
    wp = np.ascontiguousarray(wpi, dtype=np.float32).reshape(9)
    nvc = None if noisevarchrom is None else _ptr(np.ascontiguousarray(noisevarchrom, dtype=np.float32))
    Lin = np.zeros((h, w), np.float32) if want_L else None
    Lden = np.zeros((h, w), np.float32) if want_L else None
    resid = np.zeros(2, np.float32)
    # detail_recovery="f32": the DCT of the detail-recovery stage in plain fp32 direct form instead of double accumulation
    C.c_int.in_dll(lib(), "oracle_detail_dct_f32").value = 1 if detail_recovery == "f32" else 0
    detail_recovery = bool(detail_recovery)
    rc = lib().oracle_rgb_denoise_ex(_p3(img), C.c_size_t(w), w, h, C.byref(params), _ptr(wp), nvc,
                                     _ptr(Lin) if want_L else None, _ptr(Lden) if want_L else None, int(detail_recovery),
                                     _ptr(resid) if want_resid else None)
    C.c_int.in_dll(lib(), "oracle_detail_dct_f32").value = 0
    assert rc == 0
    if want_resid:
        return img, float(resid[0]), float(resid[1])
    return (img, Lin, Lden) if want_L else img


def madrgb(a):
    a = np.ascontiguousarray(a, dtype=np.float32).ravel()
    L = lib()
    L.oracle_madrgb.restype = C.c_float
    return float(L.oracle_madrgb(_ptr(a), len(a)))


def boxblur_flat(src, radx, rady):
    src = np.ascontiguousarray(src, dtype=np.float32)
    h, w = src.shape
    dst = np.empty_like(src)
    tmp = np.empty_like(src)
    lib().oracle_boxblur_flat(_ptr(src), _ptr(dst), _ptr(tmp), radx, rady, w, h)
    return dst


REC2020_WS_D = np.array([[0.6734241, 0.1656411, 0.1251286], [0.2790177, 0.6753402, 0.0456377], [-0.0019300, 0.0299784, 0.7973330]], dtype=np.float64)


def guided_smoothing(img, ws=REC2020_WS_D, radius=3, scale=1.0):
    img = [np.array(p, dtype=np.float32, order="C") for p in img]
    h, w = img[0].shape
    m = (C.c_double * 9)(*[float(v) for v in np.asarray(ws, dtype=np.float64).reshape(9)])
    lib().oracle_denoise_guided_smoothing(_p3(img), w, h, m, radius, C.c_double(scale))
    return img


def boxblur_ring(a, radius):
    a = np.array(a, dtype=np.float32, order="C")
    lib().oracle_boxblur_ring(_ptr(a), radius, a.shape[1], a.shape[0])
    return a


def guided_filter(guide, src, r, eps, subsampling=0):
    guide = np.ascontiguousarray(guide, dtype=np.float32)
    src = np.ascontiguousarray(src, dtype=np.float32)
    dst = np.empty_like(src)
    lib().oracle_guided_filter(_ptr(guide), _ptr(src), _ptr(dst), src.shape[1], src.shape[0], r, C.c_float(eps), subsampling)
    return dst


def gaussian_blur(a, sigma):
    a = np.array(a, dtype=np.float32, order="C")
    lib().oracle_gaussian_blur(_ptr(a), a.shape[1], a.shape[0], C.c_double(sigma))
    return a


def detail_mask(src, scaling, threshold, ceiling, factor, blur):
    src = np.ascontiguousarray(src, dtype=np.float32)
    m = np.empty_like(src)
    lib().oracle_detail_mask(_ptr(src), _ptr(m), src.shape[1], src.shape[0], C.c_float(scaling), C.c_float(threshold),
                             C.c_float(ceiling), C.c_float(factor), C.c_float(blur))
    return m


def nlmeans(img, strength=50, detail=80, scale=1.0, normcoeff=65535.0):
    img = np.array(img, dtype=np.float32, order="C")
    lib().oracle_nlmeans(_ptr(img), img.shape[1], img.shape[0], C.c_float(normcoeff), strength, detail, C.c_float(scale))
    return img


def lutf_vec(table, x):
    table = np.ascontiguousarray(table, dtype=np.float32)
    L = lib()
    L.oracle_lutf_vec.restype = C.c_float
    return np.array([L.oracle_lutf_vec(_ptr(table), len(table), C.c_float(float(v))) for v in x], dtype=np.float32)


NOISE_C_CURVE_POINTS = (1.0, 0.05, 0.50, 0.35, 0.35, 0.35, 0.05, 0.35, 0.35)   # ipdenoise.cc:1139-1149


def flat_curve_sample(points, periodic, ppn, identity, nout):
    pts = (C.c_double * len(points))(*[float(p) for p in points])
    out = np.zeros(nout, np.float64)
    ident = lib().oracle_flat_curve_sample(pts, len(points), int(periodic), int(ppn), C.c_double(identity), nout,
                                           out.ctypes.data_as(C.POINTER(C.c_double)))
    return out, bool(ident)


def noise_curve(points=NOISE_C_CURVE_POINTS):
    pts = (C.c_double * len(points))(*[float(p) for p in points])
    lut = np.zeros(501, np.float32)
    lib().oracle_noise_curve.restype = C.c_float
    s = lib().oracle_noise_curve(pts, len(points), _ptr(lut))
    return lut, float(s)


def chroma_noise_map(img, mat, ws, curve):
    """calclum + ccalc (ipdenoise.cc:1113-1131, FTblockDN.cc:1716-1777); ws = working-space matrix (doubles)."""
    img = [np.ascontiguousarray(p, dtype=np.float32) for p in img]
    h, w = img[0].shape
    out = np.zeros(((h + 1) // 2, (w + 1) // 2), np.float32)
    m = None if mat is None else (C.c_double * 9)(*[float(v) for v in np.asarray(mat, dtype=np.float64).reshape(9)])
    wpi = np.ascontiguousarray(np.asarray(ws, dtype=np.float64).astype(np.float32)).reshape(9)
    cv = np.ascontiguousarray(curve, dtype=np.float32)
    lib().oracle_chroma_noise_map(_p3(img), C.c_size_t(w), w, h, m, _ptr(wpi), _ptr(cv), _ptr(out))
    return out


def improc_denoise(img, dn_kw=None, calclum_mat=None, noise_c_curve=None, smoothing=True, radius=3, nl_strength=50, nl_detail=80, ecomp=0.0, scale=1.0,
                   ws=REC2020_WS_D, detail_recovery=True):
    """ImProcFunctions::denoise (ipdenoise.cc:1096-1189) composed from the oracle stages."""
    wsf = np.asarray(ws, dtype=np.float64).astype(np.float32)
    ccalc = None
    if noise_c_curve is not None and float(np.cumsum(np.asarray(noise_c_curve, np.float32), dtype=np.float32)[-1]) > 5.0:
        ccalc = chroma_noise_map(img, calclum_mat, ws, noise_c_curve)
    if ecomp > 0:
        img = exposure(img, float(np.float32(2.0 ** ecomp)), 0.0)
    dnp = default_denoise_params(scale=scale, **(dn_kw or {}))
    if scale > 1.0:      # adjust_params (ipdenoise.cc:35-63)
        def c(x, f):
            y = min(max(abs(x) / 100.0, 0.0), 1.0)
            return ((0.0 < x) - (x < 0.0)) * (y * (y * f) + (1.0 - y) * y) * 100.0
        sf = 1.0 / scale
        nc, nl = sf ** 0.46, sf ** 0.62 * sf
        dnp.luminance = c(dnp.luminance, nl)
        dnp.luminanceDetail *= (1.0 + (1.0 - sf) ** 2.2)
        dnp.chrominance = c(dnp.chrominance, nc)
        dnp.chrominanceRedGreen = c(dnp.chrominanceRedGreen, nc)
        dnp.chrominanceBlueYellow = c(dnp.chrominanceBlueYellow, nc)
    img = rgb_denoise(img, dnp, wsf, detail_recovery=detail_recovery, noisevarchrom=ccalc)
    if smoothing:
        img = guided_smoothing(img, ws, radius, scale)
        if nl_strength:
            img = [np.array(p, dtype=np.float32, order="C") for p in img]
            h, w = img[0].shape
            wp = np.ascontiguousarray(wsf).reshape(9)
            lib().oracle_rgb_to_yuv(_p3(img), C.c_size_t(w), w, h, _ptr(wp))
            img[1] = nlmeans(img[1], nl_strength, nl_detail, scale)
            lib().oracle_yuv_to_rgb(_p3(img), C.c_size_t(w), w, h, _ptr(wp))
    if ecomp > 0:
        img = exposure(img, float(np.float32(2.0 ** -ecomp)), 0.0)
    return img


class NeutralState(C.Structure):
    _fields_ = [("ws", C.c_float * 9), ("iws", C.c_float * 9), ("to_out", C.c_float * 9), ("to_work", C.c_float * 9),
                ("rhue", C.c_float), ("bhue", C.c_float), ("yhue", C.c_float), ("rrange", C.c_float), ("brange", C.c_float),
                ("yrange", C.c_float)]


# rec2020_xyz (iccmatrices.h:156-160), the inverse working-space matrix
REC2020_IWS_D = np.array([[1.6473376, -0.3935675, -0.2359961], [-0.6826036, 1.6475887, 0.0128190], [0.0296524, -0.0628993, 1.2531279]], dtype=np.float64)


def neutral_state(ws=REC2020_WS_D, iws=REC2020_IWS_D, to_out=None, to_work=None):
    st = NeutralState()
    a = (C.c_double * 9)(*[float(v) for v in np.asarray(ws, np.float64).reshape(9)])
    b = (C.c_double * 9)(*[float(v) for v in np.asarray(iws, np.float64).reshape(9)])
    o = None if to_out is None else (C.c_float * 9)(*[float(v) for v in np.asarray(to_out, np.float32).reshape(9)])
    k = None if to_work is None else (C.c_float * 9)(*[float(v) for v in np.asarray(to_work, np.float32).reshape(9)])
    lib().oracle_neutral_state_init(C.byref(st), a, b, o, k)
    return st


def tone_neutral(img, lut, whitecoeff=1.0, state=None, want_oor=False):
    img = [np.array(p, dtype=np.float32, order="C") for p in img]
    h, w = img[0].shape
    st = state or neutral_state()
    lut = np.ascontiguousarray(lut, dtype=np.float32)
    oor = np.zeros((h, w), np.uint8)
    lib().oracle_tone_curve_neutral(_p3(img), C.c_size_t(w), w, h, _ptr(lut), C.c_float(whitecoeff), C.byref(st),
                                    oor.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return (img, oor.astype(bool)) if want_oor else img


def xtrans_demosaic(raw, xtrans, rgb_cam, passes=1, use_cielab=False):
    raw = np.ascontiguousarray(raw, dtype=np.float32)
    h, w = raw.shape
    xt = np.ascontiguousarray(xtrans, dtype=np.int32).reshape(36)
    cam = np.ascontiguousarray(rgb_cam, dtype=np.float32).reshape(12)
    out = [np.zeros((h, w), np.float32) for _ in range(3)]
    lib().oracle_xtrans_demosaic(_ptr(raw), w, h, xt.ctypes.data_as(C.POINTER(C.c_int)), _ptr(cam), int(passes), int(use_cielab),
                                 _ptr(out[0]), _ptr(out[1]), _ptr(out[2]))
    return out


def scale_colors(src, cfa36, bayer, cblacksom, scale_mul):
    src = np.ascontiguousarray(src)
    h, w = src.shape
    out = np.zeros((h, w), np.float32)
    mx = (C.c_float * 4)()
    lib().oracle_scale_colors(src.ctypes.data_as(C.c_void_p), 1 if src.dtype == np.uint16 else 0, w, h,
                              np.ascontiguousarray(cfa36, dtype=np.int32).reshape(36).ctypes.data_as(C.POINTER(C.c_int)), int(bayer),
                              (C.c_float * 4)(*[float(v) for v in cblacksom]), (C.c_float * 4)(*[float(v) for v in scale_mul]), _ptr(out), mx)
    return out, [float(v) for v in mx]


def channel_mixer(img, m):
    img = [np.array(p, dtype=np.float32, order="C") for p in img]
    h, w = img[0].shape
    lib().oracle_channel_mixer(_p3(img), C.c_size_t(w), w, h, (C.c_float * 9)(*[float(v) for v in np.asarray(m, np.float32).reshape(9)]))
    return img


def rgb_curves(img, luts):
    img = [np.array(p, dtype=np.float32, order="C") for p in img]
    h, w = img[0].shape
    keep = [None if l is None else np.ascontiguousarray(l, dtype=np.float32) for l in luts]
    arr = (_fp * 3)(*[None if k is None else _ptr(k) for k in keep])
    lib().oracle_rgb_curves(_p3(img), C.c_size_t(w), w, h, arr)
    return img


def denoise_compute_params(planes, border, mul, do_clip, mat, ws, gamma=1.7, aggressive=False):
    """ImProcFunctions::denoiseComputeParams (ipdenoise.cc:800-1093). Returns (store[30], info[9,16]) or None if too small."""
    planes = [np.ascontiguousarray(p, dtype=np.float32) for p in planes]
    h, w = planes[0].shape
    store = np.zeros(30, np.float32)
    info = np.zeros((9, 16), np.float32)
    m = (C.c_double * 9)(*[float(v) for v in np.asarray(mat, dtype=np.float64).reshape(9)])
    wp = np.ascontiguousarray(np.asarray(ws, dtype=np.float64).astype(np.float32)).reshape(9)
    L = lib()
    L.oracle_denoise_compute_params.argtypes = [C.POINTER(_fp), C.c_size_t, C.c_int, C.c_int, C.c_int, _fp, C.c_int, C.POINTER(C.c_double), _fp,
                                                C.c_double, C.c_int, _fp, _fp]
    rc = L.oracle_denoise_compute_params(_p3(planes), w, w, h, int(border), (C.c_float * 3)(*[float(v) for v in mul]), 1 if do_clip else 0, m,
                                         _ptr(wp), float(gamma), 1 if aggressive else 0, _ptr(store), _ptr(info))
    return None if rc else (store, info)


def get_image_skip(planes, sx1, sy1, w, h, skip, mul, do_clip):
    planes = [np.ascontiguousarray(p, dtype=np.float32) for p in planes]
    H, W = planes[0].shape
    out = _planes(h, w)
    lib().oracle_get_image_skip(_p3(planes), C.c_size_t(W), W, H, sx1, sy1, skip, _p3(out), C.c_size_t(w), w, h,
                                (C.c_float * 3)(*[float(v) for v in mul]), 1 if do_clip else 0)
    return out


def saturation_vibrance(img, saturation, vibrance, ws=None):
    img = [np.array(p, dtype=np.float32, order="C") for p in img]
    h, w = img[0].shape
    wsd = (C.c_double * 9)(*[float(v) for v in np.asarray(REC2020_WS_D if ws is None else ws, dtype=np.float64).reshape(9)])
    lib().oracle_saturation_vibrance(_p3(img), C.c_size_t(w), w, h, int(saturation), int(vibrance), wsd)
    return img


def rgb2out_matrix(img, m, linear, lut=None):
    img = [np.ascontiguousarray(p, dtype=np.float32) for p in img]
    h, w = img[0].shape
    out = _planes(h, w)
    la = None if lut is None else np.ascontiguousarray(lut, dtype=np.float32)
    L = lib(); L.oracle_rgb2out_matrix.restype = C.c_int
    bad = L.oracle_rgb2out_matrix(_p3(img), _p3(out), C.c_size_t(w), w, h, (C.c_float * 9)(*[float(v) for v in np.asarray(m, np.float32).reshape(9)]),
                                  1 if linear else 0, None if la is None else _ptr(la), 0 if la is None else la.size)
    return out, bad


def float_to_half(x):
    """DNG_FloatToHalf (halffloat.h:9-46) on a float32 array -> uint16"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty(x.size, np.uint16)
    lib().oracle_float_to_half(_ptr(x), y.ctypes.data_as(C.c_void_p), C.c_size_t(x.size))
    return y.reshape(x.shape)


def get_scanlines(img, bps, is_float=False):
    img = [np.ascontiguousarray(p, dtype=np.float32) for p in img]
    h, w = img[0].shape
    dt = np.float32 if (is_float and bps == 32) else (np.uint8 if bps == 8 else np.uint16)
    out = np.zeros((h, w, 3), dt)
    lib().oracle_get_scanlines(_p3(img), C.c_size_t(w), w, h, bps, 1 if is_float else 0, out.ctypes.data_as(C.c_void_p))
    return out


def hsl_equalizer(img, hcurve, scurve, lcurve, smoothing, ws=None, scale=1.0, to_rgb=True):
    img = [np.array(p, dtype=np.float32, order="C") for p in img]
    h, w = img[0].shape
    def arr(c):
        return (None, 0) if c is None else ((C.c_double * len(c))(*[float(v) for v in c]), len(c))
    (hc, nh), (sc, ns), (lc, nl) = arr(hcurve), arr(scurve), arr(lcurve)
    wsd = (C.c_double * 9)(*[float(v) for v in np.asarray(REC2020_WS_D if ws is None else ws, dtype=np.float64).reshape(9)])
    L = lib()
    L.oracle_hsl_equalizer.argtypes = [C.POINTER(_fp), C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.c_int,
                                       C.c_int, C.POINTER(C.c_double), C.c_double, C.c_int]
    L.oracle_hsl_equalizer(_p3(img), w, h, hc, nh, sc, ns, lc, nl, int(smoothing), wsd, float(scale), 1 if to_rgb else 0)
    return img


def log_encoding(img, ws=None, gain=0.0, target_gray=18.0, black_ev=-13.5, white_ev=2.5, regularization=60, satcontrol=True,
                 highlight_compression=0, full_width=0, full_height=0):
    """ImProcFunctions::logEncoding (iplogenc.cc:132-316) on three contiguous planes; returns new planes."""
    L = lib()
    out = [np.ascontiguousarray(p, dtype=np.float32).copy() for p in img]
    h, w = out[0].shape
    wsd = (C.c_double * 9)(*[float(v) for v in np.asarray(REC2020_WS_D if ws is None else ws, dtype=np.float64).reshape(9)])
    L.oracle_log_encoding.argtypes = [C.POINTER(_fp), C.c_int, C.c_int, C.POINTER(C.c_double), C.c_double, C.c_double, C.c_double, C.c_double,
                                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.oracle_log_encoding.restype = None
    L.oracle_log_encoding(_p3(out), w, h, wsd, float(gain), float(target_gray), float(black_ev), float(white_ev), int(regularization),
                          1 if satcontrol else 0, int(highlight_compression), int(full_width), int(full_height))
    return out


def image_rgb_to_lab(img, ws=None):
    """Imagefloat::rgb_to_lab (imagefloat.cc:841-876): returns [a, L, b] planes in the r, g, b slots."""
    out = [np.ascontiguousarray(p, dtype=np.float32).copy() for p in img]
    h, w = out[0].shape
    wsd = (C.c_double * 9)(*[float(v) for v in np.asarray(REC2020_WS_D if ws is None else ws, dtype=np.float64).reshape(9)])
    L = lib()
    L.oracle_image_rgb_to_lab.argtypes = [C.POINTER(_fp), C.c_int, C.c_int, C.POINTER(C.c_double)]
    L.oracle_image_rgb_to_lab.restype = None
    L.oracle_image_rgb_to_lab(_p3(out), w, h, wsd)
    return out


def image_lab_to_rgb(img, iws):
    out = [np.ascontiguousarray(p, dtype=np.float32).copy() for p in img]
    h, w = out[0].shape
    m = (C.c_double * 9)(*[float(v) for v in np.asarray(iws, dtype=np.float64).reshape(9)])
    L = lib()
    L.oracle_image_lab_to_rgb.argtypes = [C.POINTER(_fp), C.c_int, C.c_int, C.POINTER(C.c_double)]
    L.oracle_image_lab_to_rgb.restype = None
    L.oracle_image_lab_to_rgb(_p3(out), w, h, m)
    return out


def lab_histogram(Lplane):
    Lp = np.ascontiguousarray(Lplane, dtype=np.float32)
    hist = np.zeros(65536, np.uint32)
    L = lib()
    L.oracle_lab_histogram.argtypes = [_fp, C.c_int, C.c_int, C.POINTER(C.c_uint32)]
    L.oracle_lab_histogram.restype = None
    L.oracle_lab_histogram(_ptr(Lp), Lp.shape[1], Lp.shape[0], hist.ctypes.data_as(C.POINTER(C.c_uint32)))
    return hist


def lab_adjustments(img, lcurve, acurve, bcurve, chroma):
    out = [np.ascontiguousarray(p, dtype=np.float32).copy() for p in img]
    h, w = out[0].shape
    cs = [np.ascontiguousarray(c, dtype=np.float32) for c in (lcurve, acurve, bcurve)]
    L = lib()
    L.oracle_lab_adjustments.argtypes = [C.POINTER(_fp), C.c_int, C.c_int, _fp, _fp, _fp, C.c_float]
    L.oracle_lab_adjustments.restype = None
    L.oracle_lab_adjustments(_p3(out), w, h, _ptr(cs[0]), _ptr(cs[1]), _ptr(cs[2]), C.c_float(chroma))
    return out


def dual_demosaic_blend(raw, planes, filters, contrast, auto_contrast=False, vng4=False):
    """the blend half of RawImageSource::dual_demosaic_RT (dual_demosaic_RT.cc:73-152, bilinear second demosaicer) on demosaiced
    planes; returns (planes, contrast in percent)."""
    raw = np.ascontiguousarray(raw, dtype=np.float32)
    out = [np.ascontiguousarray(p, dtype=np.float32).copy() for p in planes]
    h, w = raw.shape
    c = C.c_double(float(contrast))
    L = lib()
    L.oracle_dual_demosaic_blend2.argtypes = [_fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_uint, C.POINTER(C.c_double), C.c_int, C.c_int]
    L.oracle_dual_demosaic_blend2.restype = None
    L.oracle_dual_demosaic_blend2(_ptr(raw), _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), w, h, C.c_uint(filters), C.byref(c), 1 if auto_contrast else 0, 1 if vng4 else 0)
    return out, c.value


def blend_mask(planes, contrast, auto_contrast=False):
    """Color::RGB2L + buildBlendMask: returns (L, blend, threshold)"""
    pl = [np.ascontiguousarray(p, dtype=np.float32) for p in planes]
    h, w = pl[0].shape
    Lp = np.zeros((h, w), np.float32); bl = np.zeros((h, w), np.float32)
    L = lib()
    L.oracle_rgb2l.argtypes = [_fp, _fp, _fp, _fp, C.c_int, C.c_int]; L.oracle_rgb2l.restype = None
    L.oracle_build_blend_mask.argtypes = [_fp, _fp, C.c_int, C.c_int, C.c_float, C.c_int]; L.oracle_build_blend_mask.restype = C.c_float
    L.oracle_rgb2l(_ptr(pl[0]), _ptr(pl[1]), _ptr(pl[2]), _ptr(Lp), w, h)
    thr = L.oracle_build_blend_mask(_ptr(Lp), _ptr(bl), w, h, C.c_float(contrast), 1 if auto_contrast else 0)
    return Lp, bl, float(thr)


def vng4(raw, filters, prefilters=0):
    """RawImageSource::vng4_demosaic (vng4_demosaic_RT.cc:62-397)"""
    raw = np.ascontiguousarray(raw, dtype=np.float32)
    h, w = raw.shape
    out = [np.zeros((h, w), np.float32) for _ in range(3)]
    L = lib()
    L.oracle_vng4_demosaic.argtypes = [_fp, C.c_int, C.c_int, C.c_uint, C.c_uint, _fp, _fp, _fp]
    L.oracle_vng4_demosaic.restype = None
    L.oracle_vng4_demosaic(_ptr(raw), w, h, C.c_uint(filters), C.c_uint(prefilters), _ptr(out[0]), _ptr(out[1]), _ptr(out[2]))
    return out
