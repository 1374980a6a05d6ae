"""Random sizes / parameters through the remaining per-frame operators on ONE context: guided chroma smoothing, detail mask, getImage + matrix,
exposure, STD / NEUTRAL tone curves, chroma noise map.  GPU vs oracle, bit for bit.  Not a test; run on an MI355X box (env SEED, N)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from art_amd import capi, synth

MUL = (2.1374, 1.0, 1.5918)
MAT = np.array([[0.6325, 0.2312, 0.0921], [0.2198, 0.7712, 0.0090], [0.0166, 0.0713, 0.7514]])


def same(a, b):
    return sum(int((np.asarray(x).view(np.uint32) != np.asarray(y).view(np.uint32)).sum()) for x, y in zip(a, b))


if __name__ == "__main__":
    rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
    ctx = capi.Context(0)
    x = np.linspace(0.0, 1.0, 65536, dtype=np.float64)
    lut = (65535.0 * (x ** 0.7 * (1.0 - 0.25 * np.sin(3.0 * x)))).astype(np.float32)
    bad = 0
    for it in range(int(os.environ.get("N", "16"))):
        w, h = int(rng.integers(40, 1200)), int(rng.integers(40, 800))
        raw = synth.bayer_frame(w // 2 * 2 + 2, h // 2 * 2 + 2, synth.FILTERS_RGGB, seed=int(rng.integers(0, 1 << 30)), noise=int(rng.choice([64, 2048, 6000])))
        planes = O.amaze(raw, synth.FILTERS_RGGB, 1.0, 4)
        res = []
        # getImage + matrix
        iw, ih = w, h
        img = [np.zeros((ih, iw), np.float32) for _ in range(3)]
        ctx.get_image(capi.host_rgb(planes), 1, 1, MUL, True, MAT, capi.host_rgb(img))
        o = O.convert_color_space(O.get_image(planes, 1, 1, iw, ih, MUL, True), MAT)
        res.append(("getimage", same(img, o)))
        # guided smoothing
        radius, scale = int(rng.integers(1, 8)), float(rng.choice([1.0, 2.0]))
        got = [p.copy() for p in o]
        ctx.denoise_guided_smoothing(capi.host_rgb(got), O.REC2020_WS_D, radius, scale)
        res.append((f"guided(r={radius},s={scale})", same(got, O.guided_smoothing(o, O.REC2020_WS_D, radius, scale))))
        # detail mask
        m = np.empty_like(o[1]); fac, blur = float(rng.uniform(0.1, 0.99)), float(rng.choice([2.0, 1.0, 0.67]))
        ctx.detail_mask(capi.host_plane(np.ascontiguousarray(o[1])), capi.host_plane(m), 65535.0, 65.535, 65535.0, fac, blur)
        res.append((f"mask(b={blur})", same([m], [O.detail_mask(o[1], 65535.0, np.float32(1e-3) * np.float32(65535.0), 65535.0, fac, blur)])))
        # exposure + tone
        es, black = float(np.float32(2.0 ** rng.uniform(-1, 1.5))), float(rng.choice([0.0, 200.0]))
        got = [p.copy() for p in o]
        ctx.exposure(capi.host_rgb(got), es, black)
        e = O.exposure(o, es, black)
        res.append(("exposure", same(got, e)))
        wp = 1.0          # (a white point above 1 needs the curve's continuation: artgpu_set_curve_tail, tests/test_gpu_pixelops.py)
        got = [p.copy() for p in e]
        ctx.tone_curve(capi.host_rgb(got), lut, wp, True)
        res.append((f"tone_std(wp={wp})", same(got, O.tone_std(e, lut, wp, True))))
        st = O.neutral_state()
        got = [p.copy() for p in e]
        ctx.tone_curve_neutral(capi.host_rgb(got), lut, wp, O.REC2020_WS_D, O.REC2020_IWS_D)
        # (pixels the gamut compression takes out of range follow libm's pow in the reference: tolerance there, tests/test_gpu_tonecurve.py)
        refn, oor = O.tone_neutral(e, lut, wp, st, want_oor=True)
        dn = sum(int((g_[~oor].view(np.uint32) != r_[~oor].view(np.uint32)).sum()) + (0 if np.allclose(g_[oor], r_[oor], rtol=2e-4, atol=0.5) else 1) for g_, r_ in zip(got, refn))
        res.append((f"tone_neutral(oor {oor.mean():.2f})", dn))
        ok = all(d == 0 for _, d in res)
        bad += not ok
        print(it, f"{w}x{h}", ", ".join(f"{n} {'ok' if d == 0 else 'DIFF %d' % d}" for n, d in res), flush=True)
    print("failures:", bad)
    sys.exit(1 if bad else 0)
