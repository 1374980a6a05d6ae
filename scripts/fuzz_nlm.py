"""Random sizes, strengths and scales through NL-means (tile edges, sliver tiles, the three search / patch radii): GPU vs oracle, bit for bit.
Not a test; run on an MI355X box: `python scripts/fuzz_nlm.py` (env SEED, N).  tests/test_gpu_fuzz.py holds a fixed-seed slice."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from art_amd import capi, synth


def cases(seed, n):
    rng = np.random.default_rng(seed)
    for _ in range(n):
        # widths / heights around the tile step (136 = 150 - 2 * 7 at scale 1) and tiny frames
        w = int(rng.choice([rng.integers(32, 140), rng.integers(130, 160), rng.integers(260, 300), rng.integers(300, 520)]))
        h = int(rng.choice([rng.integers(32, 140), rng.integers(130, 160), rng.integers(260, 300), rng.integers(300, 420)]))
        yield w, h, int(rng.choice([20, 50, 100])), int(rng.choice([0, 50, 100])), float(rng.choice([1.0, 1.0, 2.0, 3.0, 5.0])), int(rng.integers(0, 1 << 30))


def run(ctx, w, h, strength, detail, scale, seed):
    raw = synth.bayer_frame(max(w, 32) // 2 * 2, max(h, 32) // 2 * 2, synth.FILTERS_RGGB, seed=seed, noise=2048)
    img = np.ascontiguousarray(O.amaze(raw, synth.FILTERS_RGGB, 1.0, 4)[1][:h, :w])
    got = img.copy()
    ctx.nlmeans(capi.host_plane(got), strength, detail, scale)
    ref = O.nlmeans(img, strength, detail, scale)
    return int((got.view(np.uint32) != ref.view(np.uint32)).sum())


if __name__ == "__main__":
    ctx = capi.Context(0)
    bad = 0
    for c in cases(int(os.environ.get("SEED", "1")), int(os.environ.get("N", "40"))):
        d = run(ctx, *c)
        bad += d != 0
        print(c, "ok" if d == 0 else f"DIFF {d}", flush=True)
    print("failures:", bad)
    sys.exit(1 if bad else 0)
