"""GPU: the device-side math primitives (art_amd/csrc/devmath.h, devsleef.h, paramcurve.h), evaluated ON THE GPU through
artgpu_eval_primitive, against the fixtures generated from the REFERENCE's own headers compiled in place (tests/golden/*.npz <-
oracle/_ref; make_golden.py).  Bit for bit, NaN payloads excluded.  tests/test_golden_pins.py pins the CPU oracle with the same
fixtures; this file is the device's own pin -- until round 6 the device functions were only pinned through the oracle.

Reference: rtengine/sleef.h:938-966,1198-1313; sleefsseavx.h:978-1000,1232-1372,1435-1442; helpersse2.h:168-179; median.h; LUT.h:349-459."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def same_bits(a, b):
    v = a.view(np.uint32 if a.dtype == np.float32 else np.uint64)
    w = b.view(np.uint32 if b.dtype == np.float32 else np.uint64)
    return bool(np.all((v == w) | (np.isnan(a) & np.isnan(b))))


@pytest.fixture(scope="module")
def ctx():
    from art_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


def test_device_exp_log_pow_match_reference_sleef(ctx):
    from art_amd import capi
    g = np.load(os.path.join(G, "sleef.npz"))
    for key, prim, src in (("exp_s", capi.PRIM_XEXPF_S, "xe"), ("exp_v", capi.PRIM_XEXPF_V, "xe"), ("exp_vn", capi.PRIM_XEXPF_VN, "xe"),
                           ("log_s", capi.PRIM_XLOGF_S, "xl"), ("log_v", capi.PRIM_XLOGF_V, "xl"), ("log_vn", capi.PRIM_XLOGF_VN, "xl")):
        assert same_bits(ctx.eval_primitive(prim, g[src]), g[key]), key
    # the scalar and the 4-lane exp really are different functions on the device too
    assert not np.array_equal(ctx.eval_primitive(capi.PRIM_XEXPF_S, g["xe"]).view(np.uint32), ctx.eval_primitive(capi.PRIM_XEXPF_V, g["xe"]).view(np.uint32))
    assert same_bits(ctx.eval_primitive(capi.PRIM_POW_F, g["pow_a"], g["pow_b"]), g["pow_F"])
    for base in (10, 101):
        assert same_bits(ctx.eval_primitive(capi.PRIM_XLIN2LOG, g["x01"], param=float(base)), g[f"lin2log_{base}"])
        assert same_bits(ctx.eval_primitive(capi.PRIM_XLOG2LIN, g["x01"], param=float(base)), g[f"log2lin_{base}"])


def test_device_ldexp_form_of_the_vector_exp_equals_it_where_the_result_is_normal(ctx):
    """xexpf_v_ldexp (one v_ldexp_f32 for sleef's five multiplications by powers of two) is what the shrink kernels call: it has to return
    the reference's bits wherever exp(d) is a normal number (devsleef.h states the range; tests/test_exp_ldexp.py walks it on the CPU)."""
    from art_amd import capi
    g = np.load(os.path.join(G, "sleef.npz"))
    x = g["xe"]
    got, ref = ctx.eval_primitive(capi.PRIM_XEXPF_V_LDEXP, x), g["exp_v"]
    ok = np.isfinite(x) & (x > -87.0) & (x < 88.0)
    assert ok.sum() > 1000
    assert same_bits(got[ok], ref[ok])


def test_device_cbrt_atan2_sincos_match_reference_sleef(ctx):
    from art_amd import capi
    g = np.load(os.path.join(G, "sleef2.npz"))
    assert same_bits(ctx.eval_primitive(capi.PRIM_XCBRTF, g["xc"]), g["cbrt"])
    assert same_bits(ctx.eval_primitive(capi.PRIM_XATAN2F, g["ay"], g["ax"]), g["atan2"])
    sn, cs = ctx.eval_primitive(capi.PRIM_XSINCOSF, g["sd"])
    assert same_bits(sn, g["sin"]) and same_bits(cs, g["cos"])
    # the 4-lane xatan2f of the reference (sleefsseavx.h) returns the scalar one's bits: the device has one form for both
    g3 = np.load(os.path.join(G, "sleef3.npz"))
    assert same_bits(ctx.eval_primitive(capi.PRIM_XATAN2F, g3["ay"], g3["ax"]), g3["atan2_v"])


def test_device_helpers_match_reference_headers(ctx):
    from art_amd import capi
    g = np.load(os.path.join(G, "helpers.npz"))
    a, b, c = g["a"], g["b"], g["c"]
    assert same_bits(ctx.eval_primitive(capi.PRIM_VMINF, a, b), g["vmin"])
    assert same_bits(ctx.eval_primitive(capi.PRIM_VMAXF, a, b), g["vmax"])
    assert same_bits(ctx.eval_primitive(capi.PRIM_MEDIAN3, a, b, c), g["median3"])
    assert same_bits(ctx.eval_primitive(capi.PRIM_VINTPF, c, a, b), g["vintpf"])
    assert same_bits(ctx.eval_primitive(capi.PRIM_XDIV2F, a), g["xdiv2f"])
    assert same_bits(ctx.eval_primitive(capi.PRIM_XDIVF2, a), g["xdivf2"])


def test_device_lutf_lookups_match_reference_lut_h(ctx):
    from art_amd import capi
    g = np.load(os.path.join(G, "lutf.npz"))
    size = int(g["table_size"])
    x = np.arange(size, dtype=np.float64) / (size - 1)
    table = (np.sqrt(x) * 65535.0).astype(np.float32)
    assert same_bits(ctx.eval_primitive(capi.PRIM_LUTF_SCALAR, g["index"], table=table), g["scalar"])      # LUTf::operator[](float)
    n4 = len(g["index"]) // 4 * 4
    assert same_bits(ctx.eval_primitive(capi.PRIM_LUTF_VECTOR, g["index"][:n4], table=table), g["vector"][:n4])    # operator[](vfloat)


def test_device_double_log_exp_match_reference_sleef(ctx):
    """the double-precision xlog / xexp the DCT_Parametric curve tail is evaluated with per super-white pixel (paramcurve.h)"""
    from art_amd import capi
    g = np.load(os.path.join(G, "sleef_d.npz"))
    assert same_bits(ctx.eval_primitive(capi.PRIM_XLOG_D, g["xl"]), g["log"])
    assert same_bits(ctx.eval_primitive(capi.PRIM_XEXP_D, g["xe"]), g["exp"])


def test_missing_operand_is_an_error(ctx):
    from art_amd import capi
    with pytest.raises(Exception):
        ctx.eval_primitive(capi.PRIM_POW_F, np.ones(4, np.float32))


def test_device_float_to_half_matches_reference_halffloat_h(ctx):
    """DNG_FloatToHalf (halffloat.h:9-46 compiled in place -> tests/golden/halffloat.npz): every half value, its float neighbours, the rounding
    ties, every float exponent, random patterns -- through the primitive and through the half-float scanlines that use it"""
    from art_amd import capi
    g = np.load(os.path.join(G, "halffloat.npz"))
    x = g["x_bits"].view(np.float32)
    got = ctx.eval_primitive(capi.PRIM_FLOAT_TO_HALF, x)
    assert np.array_equal(got.astype(np.uint16), g["half"]) and int(got.max()) <= 0xFFFF
    import torch
    v = (x[::16][:30000] * np.float32(65535.0)).reshape(100, 100, 3)
    planes = [torch.from_numpy(np.ascontiguousarray(v[:, :, c])).cuda() for c in range(3)]
    scan = ctx.get_scanlines(capi.RGB(*[capi.device_plane(t) for t in planes]), 16, True)
    lut = dict(zip(g["x_bits"].tolist(), g["half"].tolist()))
    back = (v / np.float32(65535.0)).astype(np.float32)          # the scanline's own division
    known = np.array([lut.get(int(b), -1) for b in back.view(np.uint32).ravel()]).reshape(back.shape)
    sel = known >= 0
    assert sel.mean() > 0.5 and np.array_equal(scan[sel], known[sel].astype(np.uint16))
