"""Random sizes and parameters through RGB_denoise (gamma / YUV, MadRgb, the shrinkage passes with their box blurs, reconstruction; the DCT
stage off, so bit for bit) on ONE context: GPU vs oracle.  Not a test; run on an MI355X box: `python scripts/fuzz_denoise.py` (env SEED, N)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from art_amd import capi, synth


def cases(seed, n):
    rng = np.random.default_rng(seed)
    for _ in range(n):
        w = int(rng.choice([rng.integers(64, 200), rng.integers(200, 700), rng.integers(700, 1100)]))
        h = int(rng.choice([rng.integers(64, 200), rng.integers(200, 600)]))
        yield dict(w=w, h=h, seed=int(rng.integers(0, 1 << 30)), noise=int(rng.choice([64, 2048, 6000])),
                   luminance=float(rng.choice([0.0, 5.0, 40.0, 100.0])), chrominance=float(rng.choice([0.0, 15.0, 60.0, 100.0])),
                   rg=float(rng.choice([0.0, -40.0, 35.0])), by=float(rng.choice([0.0, 50.0, -25.0])), gamma=float(rng.choice([1.0, 1.7, 3.0])),
                   aggressive=int(rng.integers(0, 2)), scale=float(rng.choice([1.0, 1.0, 2.0])))


def run(ctx, c):
    raw = synth.bayer_frame(c["w"] // 2 * 2, c["h"] // 2 * 2, synth.FILTERS_RGGB, seed=c["seed"], noise=c["noise"])
    img = [np.ascontiguousarray(p[:c["h"], :c["w"]]) for p in O.amaze(raw, synth.FILTERS_RGGB, 1.0, 4)]
    got = [p.copy() for p in img]
    p = capi.DenoiseParams(c["luminance"], 50.0, 0, c["chrominance"], c["rg"], c["by"], c["gamma"], c["aggressive"], 0, 0)
    ctx.rgb_denoise(capi.host_rgb(got), p, O.REC2020_WS, scale=c["scale"], flags=capi.DN_SKIP_DETAIL_RECOVERY)
    ref = O.rgb_denoise(img, O.default_denoise_params(luminance=c["luminance"], chrominance=c["chrominance"], chrominanceRedGreen=c["rg"],
                                                      chrominanceBlueYellow=c["by"], gamma=c["gamma"], aggressive=c["aggressive"], scale=c["scale"]))
    return sum(int((a.view(np.uint32) != b.view(np.uint32)).sum()) for a, b in zip(got, ref))


if __name__ == "__main__":
    ctx = capi.Context(0)
    bad = 0
    for c in cases(int(os.environ.get("SEED", "1")), int(os.environ.get("N", "30"))):
        d = run(ctx, c)
        bad += d != 0
        print(c, "ok" if d == 0 else f"DIFF {d}", flush=True)
    print("failures:", bad)
    sys.exit(1 if bad else 0)
