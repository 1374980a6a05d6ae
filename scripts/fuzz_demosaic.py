"""Random frame sizes (odd ones included) and all four CFA phases through the two streaming demosaicers: GPU vs oracle, bit for bit.
Not a test (the oracle takes a while); run on an MI355X box:  N=60 SEED=3 python scripts/fuzz_demosaic.py"""
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
    w, h = int(rng.integers(64, 900)), int(rng.integers(64, 700))
    if it % 5 == 0:                 # widths / heights that leave a sliver tile (RCD: 176 k + 19 .. 176 k + 30)
        w = 176 * int(rng.integers(1, 5)) + int(rng.integers(17, 31))
    if it % 7 == 0:
        h = 176 * int(rng.integers(1, 4)) + int(rng.integers(17, 31))
    method = "rcd" if it % 2 else "amaze"
    filt = [synth.FILTERS_RGGB, synth.FILTERS_BGGR, synth.FILTERS_GRBG, synth.FILTERS_GBRG][it % 4]
    noise = [0, 64, 1500, 6000][(it // 4) % 4]
    raw = synth.bayer_frame(w, h, filt, seed=500 + it, noise=noise)
    if method == "rcd":
        ctx.set_option("rcd_rows", 8 if it % 4 == 1 else 4)
    ref = O.rcd(raw, filt) if method == "rcd" else O.amaze(raw, filt, 1.0, 4)
    got = ctx.demosaic_bayer_host(capi.BAYER_RCD if method == "rcd" else capi.BAYER_AMAZE, raw, filt, 1.0, 4)
    ok = all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(got, ref))
    print(f"{it}: {w}x{h} {method} filters={filt:#x} noise {noise}: {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += not ok
print("mismatches:", bad)
