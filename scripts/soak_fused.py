"""Soak of the fused ShrinkAll pass (shrinkblur.hip): the denoise tool on a device-resident 45 MP frame N times in a row while a second
context on a second host thread keeps the device busy with frames of another size (uneven load: workgroups of the persistent kernel come
and go at different times); every result is compared ON THE DEVICE with the first one, which in turn is compared with the three-kernel form.
A stale or torn hand-over between two strips shows as a difference.  Not a test; run on an MI355X box: python scripts/soak_fused.py [N] [detail]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from art_amd import capi, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
FLAGS = 0 if (len(sys.argv) > 2 and sys.argv[2] == "detail") else capi.DN_SKIP_DETAIL_RECOVERY   # "detail": with the DCT stage (on its side stream: option dn_streams 1 below)
W, H = 8184, 5456
WS = np.array([[0.6734241, 0.1656411, 0.1251286], [0.2790177, 0.6753402, 0.0456377], [-0.0019300, 0.0299784, 0.7973330]])
MAT = np.array([[0.6325, 0.2312, 0.0921], [0.2198, 0.7712, 0.0090], [0.0166, 0.0713, 0.7514]])
dev = torch.device("cuda:0")


def frame(w, h, seed):
    rng = np.random.default_rng(seed)
    base = rng.uniform(0, 1, (h // 8 + 1, w // 8 + 1)).astype(np.float32)
    big = np.kron(base, np.ones((8, 8), np.float32))[:h, :w] * 30000.0
    return [torch.from_numpy((big * s + rng.normal(0, 900, (h, w)).astype(np.float32)).clip(0, 65535)).to(dev) for s in (1.0, 0.9, 0.7)]


def run(ctx, planes, tp, curve, out):
    for d, s in zip(out, planes):
        d.copy_(s)
    ctx.improc_denoise(capi.RGB(*[capi.device_plane(t) for t in out]), tp, WS, ecomp=0.3, calclum_mat=MAT, noise_c_curve=curve, flags=FLAGS)


tp = capi.DenoiseToolParams(capi.DenoiseParams(40.0, 50.0, 0, 15.0, 0.0, 0.0, 1.7, 0, 0, 0), 0, 3, 0, 80)
curve, _ = capi.noise_curve_lut()
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
c1, c2 = capi.Context(0, s1.cuda_stream), capi.Context(0, s2.cuda_stream)
if FLAGS == 0:
    c1.set_option("dn_streams", 1)      # the side stream is what this mode soaks (off by default since round 5)
src = frame(W, H, 1)
with torch.cuda.stream(s1):
    first = [torch.empty_like(t) for t in src]
    run(c1, src, tp, curve, first)
    c1.set_option("dn_fused", 0)
    three = [torch.empty_like(t) for t in src]
    run(c1, src, tp, curve, three)
    c1.set_option("dn_fused", 1)
    s1.synchronize()
assert all(torch.equal(a, b) for a, b in zip(first, three)), "fused != three-kernel form"
stop = False


def noise_maker():
    small = frame(2600, 1900, 2)
    outb = [torch.empty_like(t) for t in small]
    k = 0
    with torch.cuda.stream(s2):
        while not stop:
            run(c2, small, tp, curve, outb)
            k += 1
            if k % 7 == 0:
                time.sleep(0.003)          # uneven on purpose
            s2.synchronize()


th = threading.Thread(target=noise_maker)
th.start()
bad = 0
t0 = time.time()
with torch.cuda.stream(s1):
    out = [torch.empty_like(t) for t in src]
    for it in range(N):
        run(c1, src, tp, curve, out)
        if not all(torch.equal(a, b) for a, b in zip(out, first)):
            bad += 1
            print("iteration", it, "differs", flush=True)
    s1.synchronize()
stop = True
th.join()
print(f"{N} frames of {W}x{H} beside a second context in {time.time() - t0:.1f} s: {bad} differed")
sys.exit(1 if bad else 0)
