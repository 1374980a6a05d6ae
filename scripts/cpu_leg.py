"""bench.py's cpu_baseline leg alone (host cores only): the checker and its timing variant on the c3 sample."""
import os, sys, time, statistics, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import oracle_lib as O
from art_amd import synth, capi
os.environ.setdefault("OMP_NUM_THREADS", str(os.cpu_count()))
W, H = 4096, 2732
MUL = (2.1374, 1.0, 1.5918)
MAT = np.array([[0.6325, 0.2312, 0.0921], [0.2198, 0.7712, 0.0090], [0.0166, 0.0713, 0.7514]])
raw = np.ascontiguousarray(synth.bayer_frame(8192, 5464, synth.FILTERS_RGGB, seed=0)[:H, :W])
curve, _ = capi.noise_curve_lut()
x = np.arange(65536, dtype=np.float64) / 65535.0
lut = ((1.0 - np.cos(np.pi * x ** 0.7)) / 2.0 * 65535.0).astype(np.float32)
flag = ctypes.c_int.in_dll(O.lib(), "oracle_fast")
def fn(parts):
    t = [time.perf_counter()]
    pl = O.amaze(raw, synth.FILTERS_RGGB, 1.0, 4); t.append(time.perf_counter())
    im = O.get_image(pl, 4, 4, W - 8, H - 8, MUL, True); im = O.convert_color_space(im, MAT); t.append(time.perf_counter())
    im = O.improc_denoise(im, calclum_mat=MAT, noise_c_curve=curve, smoothing=False, ecomp=0.3, detail_recovery=True); t.append(time.perf_counter())
    im = O.exposure(im, float(np.float32(2.0 ** 0.3)), 0.0); im = O.tone_std(im, lut, 1.0, True); t.append(time.perf_counter())
    parts.append([b - a for a, b in zip(t, t[1:])])
for fast in (0, 1, 0, 1):
    flag.value = fast
    parts = []
    fn(parts); parts = []
    for _ in range(3): fn(parts)
    med = [statistics.median(p[k] for p in parts) for k in range(4)]
    print("fast" if fast else "checker", "total %.3f s = %.2f MP/s" % (sum(med), W * H / 1e6 / sum(med)), "amaze %.3f getimage %.3f denoise %.3f tone %.3f" % tuple(med), flush=True)
flag.value = 0
