"""Random sizes through the N4 tools (log encoding, Lab adjustments, hsl equaliser, dual demosaic, VNG4): GPU vs oracle, bit for bit.
Not a test (takes a while); run on an MI355X box: `python scripts/fuzz_tools.py` (env SEED, N)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from art_amd import capi, synth

rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
ctx = capi.Context(0)
bad = 0


def same(a, b):
    return all(np.array_equal(np.asarray(x).view(np.uint32), np.asarray(y).view(np.uint32)) for x, y in zip(a, b))


def scene(w, h):
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    ev = -12.0 + 14.0 * (0.5 + 0.5 * np.sin(0.05 * x + 0.3) * np.cos(0.04 * y))
    lum = (0.18 * 65535.0 * np.exp2(ev)).astype(np.float32)
    return [(lum * rng.uniform(0.5, 1.5, (h, w))).astype(np.float32) for _ in range(3)]


S_CURVE = (1, 0.0, 0.5, 0.35, 0.35, 0.12, 0.72, 0.35, 0.35, 0.40, 0.30, 0.35, 0.35, 0.70, 0.55, 0.35, 0.35)
L_CURVE = (1, 0.05, 0.5, 0.35, 0.35, 0.30, 0.64, 0.35, 0.35, 0.62, 0.41, 0.35, 0.35)
H_CURVE = (1, 0.0, 0.5, 0.0, 0.0, 0.25, 0.58, 0.35, 0.35, 0.55, 0.44, 0.35, 0.35, 0.80, 0.5, 0.35, 0.35)
for it in range(int(os.environ.get("N", "12"))):
    w, h = int(rng.integers(9, 900)), int(rng.integers(9, 700))
    img = scene(w, h)
    res = []
    # log encoding
    kw = dict(regularization=int(rng.choice([0, 30, 60, 100])), satcontrol=bool(rng.integers(0, 2)), gain=float(rng.uniform(-1, 1)),
              full_width=int(rng.choice([0, 3000, 9000])), full_height=0)
    got = [p.copy() for p in img]
    ctx.log_encoding(capi.host_rgb(got), O.REC2020_WS_D, **kw)
    res.append(("logenc", same(got, O.log_encoding(img, **kw))))
    # Lab round trip + curves
    got = [p.copy() for p in img]
    ctx.rgb_to_lab(capi.host_rgb(got), O.REC2020_WS_D)
    lab = O.image_rgb_to_lab(img)
    ok = same(got, lab)
    t = np.arange(32770, dtype=np.float64) / 32767.0
    lc = (32767.0 * np.clip(t ** 0.9, 0, 1.0002)).astype(np.float32)
    ac = (np.arange(65536, dtype=np.float64) * 0.98 + 300).astype(np.float32)
    ctx.lab_adjustments(capi.host_rgb(got), lc, ac, ac, 1.2)
    lab2 = O.lab_adjustments(lab, lc, ac, ac, 1.2)
    ok = ok and same(got, lab2)
    ctx.lab_to_rgb(capi.host_rgb(got), O.REC2020_IWS_D)
    res.append(("lab", ok and same(got, O.image_lab_to_rgb(lab2, O.REC2020_IWS_D))))
    # hsl equaliser
    sm = int(rng.integers(0, 11))
    got = [np.clip(p, 10, 60000) for p in img]
    src = [p.copy() for p in got]
    ctx.hsl_equalizer(capi.host_rgb(got), H_CURVE, S_CURVE, L_CURVE, sm, O.REC2020_WS_D, 1.0, True)
    res.append((f"hsl(sm={sm})", same(got, O.hsl_equalizer(src, H_CURVE, S_CURVE, L_CURVE, sm, scale=1.0, to_rgb=True))))
    # demosaicers
    if w >= 96 and h >= 96:
        filt = [synth.FILTERS_RGGB, 0x16161616, 0x61616161, 0x49494949][it % 4]
        raw = synth.bayer_frame(w, h, filt, seed=200 + it, noise=int(rng.choice([200, 1500])))
        out = [np.zeros((h, w), np.float32) for _ in range(3)]
        ctx.demosaic_bayer(capi.BAYER_VNG4, capi.host_plane(raw), filt, 1.0, 4, capi.host_rgb(out))
        res.append(("vng4", same(out, O.vng4(raw, filt))))
        vng = bool(rng.integers(0, 2)); auto = bool(rng.integers(0, 2)); con = float(rng.choice([0.0, 10.0, 40.0]))
        out = [np.zeros((h, w), np.float32) for _ in range(3)]
        c = ctx.dual_demosaic_bayer(capi.BAYER_RCD, capi.host_plane(raw), filt, 1.0, 4, con, auto, capi.host_rgb(out), second=capi.DUAL_VNG4 if vng else capi.DUAL_BILINEAR)
        ref, rc = O.dual_demosaic_blend(raw, O.rcd(raw, filt), filt, con, auto, vng4=vng)
        res.append((f"dual(vng4={vng},auto={auto},c={con}->{c})", same(out, ref) and c == rc))
    bad += sum(0 if ok else 1 for _, ok in res)
    print(f"{it}: {w}x{h} " + ", ".join(f"{nm} {'ok' if ok else 'MISMATCH'}" for nm, ok in res), flush=True)
print("mismatches:", bad)
