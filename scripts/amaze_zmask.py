"""Which AMaZE arena regions must be cleared per tile?  Poison the arena with NaN, clear all regions but one, compare with
the oracle.  A region whose un-cleared run still matches has no read-before-write position on full tiles."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from art_amd import capi, synth
names = ["rgbgreen", "delhvsqsum", "dirwts0", "dirwts1", "vcd", "hcd", "vcdalt", "hcdalt", "cddiffsq", "hvwt", "dgintv", "dginth", "Dgrbsq1m",
         "Dgrbsq1p", "cfa", "nyquist", "nyqutest"]
ctx = capi.Context(0)
frames = []
for (w, h, filt, seed, gain, noise) in ((1152, 896, synth.FILTERS_RGGB, 11, 1.0, 3000), (768, 640, synth.FILTERS_GBRG, 12, 2.1, 3000),
                                         (1024, 768, synth.FILTERS_RGGB, 13, 1.0, 0), (896, 640, synth.FILTERS_BGGR, 14, 1.3, 200)):
    raw = synth.bayer_frame(w, h, filt, seed=seed, noise=noise)
    frames.append((raw, filt, gain, O.amaze(raw, filt, gain, 4)))
def run(mask, poison="0xFF"):
    os.environ["ARTGPU_AMAZE_POISON"] = poison; os.environ["ARTGPU_AMAZE_ZMASK"] = hex(mask)
    bad = 0
    for raw, filt, gain, ref in frames:
        h, w = raw.shape
        out = [np.zeros((h, w), np.float32) for _ in range(3)]
        ctx.demosaic_bayer(capi.BAYER_AMAZE, capi.host_plane(raw), filt, gain, 4, capi.host_rgb(out))
        bad += sum(int((o.view(np.uint32) != r.view(np.uint32)).sum()) for o, r in zip(out, ref))
    return bad
full = (1 << 17) - 1
print("all cleared:", run(full))
need = 0x81f0
print("needed mask", hex(need), "->", run(need))
for pz in ("0xFF", "0x7F", "0xC0", "0x3F", "0x80"):
    print("0x81f0 poison", pz, "->", run(0x81f0, pz))
