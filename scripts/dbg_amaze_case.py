"""debug helper: repeated AMaZE calls on one frame, diff against the oracle per call"""
import sys, numpy as np, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import oracle_lib as O
from art_amd import capi, synth
ctx = capi.Context(0)
w, h, filt, gain, noise = 1296, 1040, synth.FILTERS_RGGB, 2.5, 32
raw = synth.bayer_frame(w, h, filt, seed=w + noise, noise=noise)
ref = O.amaze(raw, filt, gain, 4)
for it in range(8):
    got = ctx.demosaic_bayer_host(capi.BAYER_AMAZE, raw, filt, gain, 4)
    d = np.zeros((h, w), bool)
    for g, r in zip(got, ref):
        d |= g.view(np.uint32) != r.view(np.uint32)
    yy, xx = np.nonzero(d)
    print("counters", [ctx.get_option(f"amaze_counter{k}") for k in range(7)])
    print("call", it, "ndiff", len(yy), (yy.min(), yy.max(), xx.min(), xx.max()) if len(yy) else None)
    if len(yy):
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        np.save(os.path.join(ROOT, "gpurun_out", "dbg_wrong.npy"), np.stack(got)[:, 96:160, 560:620])
