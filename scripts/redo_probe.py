"""how many second attempts does a frame provoke?  (counters of the AMaZE redo queue; picks the stress frame of tests/test_gpu_demosaic.py)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
from art_amd import capi, synth

def patches(w, h, filt, step, size, amp, noise, seed=5):
    raw = synth.bayer_frame(w, h, filt, seed=seed, noise=noise, clip_patch=False, nyquist_patch=False).astype(np.int64)
    yy, xx = np.mgrid[0:h, 0:w]
    py, px = yy % step[1], xx % step[0]
    # patch position drifts with the cell index so that patches fall everywhere relative to the 128-pixel tile grid
    oy, ox = (7 * (yy // step[1]) + 3 * (xx // step[0])) % (step[1] - size), (5 * (xx // step[0]) + 11 * (yy // step[1])) % (step[0] - size)
    inside = (py >= oy) & (py < oy + size) & (px >= ox) & (px < ox + size)
    raw = raw + np.where(inside, np.where((xx & 1) == 1, amp, -amp), 0)
    return np.clip(raw, 0, 65535).astype(np.float32)

ctx = capi.Context(0)
filt = synth.FILTERS_RGGB
for (w, h, step, size, amp, noise) in [(4224, 3168, (97, 89), 12, 6000, 0), (4224, 3168, (151, 139), 20, 6000, 0), (4224, 3168, (97, 89), 12, 6000, 64), (4224, 3168, (211, 197), 40, 8000, 16)]:
    raw = patches(w, h, filt, step, size, amp, noise)
    for it in range(2):
        ctx.demosaic_bayer_host(capi.BAYER_AMAZE, raw, filt, 1.0, 4)
        print((w, h, step, size, amp, noise), "counters", [ctx.get_option(f"amaze_counter{k}") for k in range(7)], flush=True)
