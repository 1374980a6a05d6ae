"""Random frame sizes through the config-4 chain (AMaZE or RCD -> getImage+matrix -> denoise incl. guided smoothing and NL-means,
DCT stage skipped -> exposure -> tone): GPU vs oracle, bit for bit.  Not a test (takes a while); run on an MI355X box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from art_amd import capi, synth

MUL = (2.1374, 1.0, 1.5918)
MAT = np.array([[0.6325, 0.2312, 0.0921], [0.2198, 0.7712, 0.0090], [0.0166, 0.0713, 0.7514]])
x = np.arange(65536, dtype=np.float64) / 65535.0
LUT = ((1.0 - np.cos(np.pi * x ** 0.7)) / 2.0 * 65535.0).astype(np.float32)
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
ctx = capi.Context(0)
curve, _ = capi.noise_curve_lut()
bad = 0
for it in range(int(os.environ.get("N", "10"))):
    w, h = int(rng.integers(64, 720)) * 2, int(rng.integers(64, 560)) * 2
    if os.environ.get("ODD"):
        w, h = w + int(rng.integers(0, 2)), h + int(rng.integers(0, 2))
    method = "rcd" if it % 3 == 2 else "amaze"
    filt = [synth.FILTERS_RGGB, 0x16161616, 0x61616161, 0x49494949][it % 4]
    raw = synth.bayer_frame(w, h, filt, seed=100 + it, noise=1500)
    planes = O.rcd(raw, filt) if method == "rcd" else O.amaze(raw, filt, 1.0, 4)
    got_p = [np.zeros((h, w), np.float32) for _ in range(3)]
    ctx.demosaic_bayer(capi.BAYER_RCD if method == "rcd" else capi.BAYER_AMAZE, capi.host_plane(raw), filt, 1.0, 4, capi.host_rgb(got_p))
    ok_dem = all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(got_p, planes))
    iw, ih = w - 8, h - 8
    img = [np.zeros((ih, iw), np.float32) for _ in range(3)]
    ctx.get_image(capi.host_rgb(got_p), 4, 4, MUL, True, MAT, capi.host_rgb(img))
    tp = capi.DenoiseToolParams(capi.DenoiseParams(40.0, 50.0, 0, 15.0, 0.0, 0.0, 1.7, 0, 0, 0), 1, 3, 50, 80)
    ctx.improc_denoise(capi.host_rgb(img), tp, O.REC2020_WS_D, ecomp=0.3, calclum_mat=MAT, noise_c_curve=curve, flags=capi.DN_SKIP_DETAIL_RECOVERY)
    ctx.exposure(capi.host_rgb(img), float(np.float32(2.0 ** 0.3)), 0.0)
    ctx.tone_curve(capi.host_rgb(img), LUT, 1.0, True)
    o = O.get_image(planes, 4, 4, iw, ih, MUL, True)
    o = O.convert_color_space(o, MAT)
    o = O.improc_denoise(o, calclum_mat=MAT, noise_c_curve=curve, smoothing=True, radius=3, nl_strength=50, nl_detail=80, ecomp=0.3, detail_recovery=False)
    o = O.exposure(o, float(np.float32(2.0 ** 0.3)), 0.0)
    o = O.tone_std(o, LUT, 1.0, True)
    ok = all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(img, o))
    print(f"{it}: {w}x{h} {method} filters={filt:#x}: demosaic {'ok' if ok_dem else 'MISMATCH'}, chain {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += (not ok) + (not ok_dem)
print("mismatches:", bad)
