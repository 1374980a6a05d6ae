"""Random frame sizes / borders / noise levels through the AUTOMATIC chroma estimation (denoiseComputeParams: nine crops, their wavelet
statistics, the ordered sums) on ONE context: GPU vs oracle, every stored value bit for bit.  Not a test; run on an MI355X box (env SEED, N)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from art_amd import capi
from test_oracle_dninfo import MAT, planes

if __name__ == "__main__":
    rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
    ctx = capi.Context(0)
    bad = 0
    for it in range(int(os.environ.get("N", "16"))):
        w, h = int(rng.integers(200, 1400)), int(rng.integers(160, 1000))
        border, noise, agg, gamma = int(rng.choice([0, 4])), int(rng.choice([60, 400, 1200, 4000])), bool(rng.integers(0, 2)), float(rng.choice([1.0, 1.7, 3.0]))
        clip = bool(rng.integers(0, 2))
        pl = planes(w, h, noise, seed=int(rng.integers(0, 1 << 30)))
        mul = (float(rng.uniform(1.5, 2.5)), 1.0, float(rng.uniform(1.2, 2.0)))
        dn = capi.DenoiseParams(40.0, 50.0, 0, 15.0, 0.0, 0.0, gamma, 1 if agg else 0, 0, 1)
        d = [p.copy() for p in pl]
        st = ctx.denoise_compute_params(capi.host_rgb(d), border, mul, clip, MAT, O.REC2020_WS_D, dn, auto_factor=1.0)
        ref = O.denoise_compute_params(pl, border, mul, clip, MAT, O.REC2020_WS_D, gamma, agg)
        if ref is None:
            ok = st.valid == 0
        else:
            store, info = ref
            got_info = np.array([list(st.crop_info[k]) for k in range(9)], np.float32)
            ok = (st.valid == 1 and np.array_equal(got_info[:, :11].view(np.uint32), info[:, :11].view(np.uint32))
                  and np.float32(st.chrominance) == store[0] and np.float32(st.chrominance_red_green) == store[1] and np.float32(st.chrominance_blue_yellow) == store[2]
                  and np.array_equal(np.array(list(st.ch_M), np.float32), store[3:12]) and np.array_equal(np.array(list(st.max_r), np.float32), store[12:21])
                  and np.array_equal(np.array(list(st.max_b), np.float32), store[21:30]))
        bad += not ok
        print(it, f"{w}x{h} border {border} noise {noise} aggressive {agg} gamma {gamma} clip {clip}:", "ok" if ok else "DIFF", flush=True)
    print("failures:", bad)
    sys.exit(1 if bad else 0)
