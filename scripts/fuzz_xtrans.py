"""Random frame sizes, pass counts, colour-map phases and both perceptual spaces through the X-Trans kernel, and random sizes / sigmas through the
YvV gaussian: GPU vs oracle, bit for bit.  Not a test (the oracle takes a while); run on an MI355X box:  N=40 SEED=3 python scripts/fuzz_xtrans.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from art_amd import capi, synth

rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
ctx = capi.Context(0)
bad = 0
for it in range(int(os.environ.get("N", "40"))):
    w, h = int(rng.integers(64, 800)), int(rng.integers(64, 700))
    if it % 4 == 0:                 # sizes that leave a sliver tile (tile stride 98, origin 3)
        w = 98 * int(rng.integers(1, 7)) + int(rng.integers(20, 40))
    if it % 6 == 0:
        h = 98 * int(rng.integers(1, 6)) + int(rng.integers(20, 40))
    passes = [1, 3, 2, 3][it % 4]
    lab = bool((it // 2) % 2)
    xt = np.roll(np.roll(synth.XTRANS_FUJI, int(rng.integers(0, 6)), axis=0), int(rng.integers(0, 6)), axis=1) if it % 3 == 0 else synth.XTRANS_FUJI
    noise = [0, 300, 1500, 6000][(it // 4) % 4]
    raw = synth.bayer_frame(w, h, 0, 700 + it, noise, True, True, xtrans=xt)
    out = [np.full((h, w), -1.0, np.float32) for _ in range(3)]
    ctx.demosaic_xtrans(passes, lab, capi.host_plane(raw), xt, synth.XTRANS_RGB_CAM, capi.host_rgb(out))
    ref = O.xtrans_demosaic(raw, xt, synth.XTRANS_RGB_CAM, passes, lab)
    ok = all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(out, ref))
    print(f"{it}: xtrans {w}x{h} passes {passes} lab {lab} noise {noise}: {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += not ok
    # the YvV gaussian on the green plane of the result
    sigma = float(rng.choice([0.7, 1.0, 2.0, 3.3, 7.5, 12.0]))
    img = np.ascontiguousarray(ref[1])
    got = img.copy()
    ctx.gaussian_blur(capi.host_plane(got), sigma)
    ok = np.array_equal(got.view(np.uint32), O.gaussian_blur(img, sigma).view(np.uint32))
    print(f"{it}: gaussian {w}x{h} sigma {sigma}: {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += not ok
print("mismatches:", bad)
