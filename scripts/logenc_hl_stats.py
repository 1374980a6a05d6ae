import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import oracle_lib as O
from art_amd import capi
from test_gpu_logenc import scene
ctx = capi.Context(0)
for w, h, kw in [(640, 400, dict(regularization=0, highlight_compression=40)), (500, 333, dict(regularization=0, highlight_compression=5, gain=1.0)), (720, 520, dict(regularization=60, highlight_compression=100, white_ev=5.0)), (1600, 1200, dict(regularization=0, highlight_compression=70))]:
    img = scene(w, h, w + h)
    got = [p.copy() for p in img]
    ctx.log_encoding(capi.host_rgb(got), O.REC2020_WS_D, **kw)
    ref = O.log_encoding(img, **kw)
    nd = 0; mu = 0; mr = 0.0
    for g, r in zip(got, ref):
        d = np.abs(g.view(np.int32).astype(np.int64) - r.view(np.int32).astype(np.int64))
        nd += int((d != 0).sum()); mu = max(mu, int(d.max()))
        rel = np.abs(g.astype(np.float64) - r) / np.maximum(np.abs(r), 1e-3)
        mr = max(mr, float(rel.max()))
    print(kw, "ndiff", nd, "of", 3 * w * h, "max ulp", mu, "max rel", mr)
