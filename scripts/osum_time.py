"""Time artgpu_ordered_sum_f32 (orderedsum.hip) on device arrays: one workgroup, wave 0 runs the chain."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from art_amd import capi
ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
n = 2_800_000
cases = {"uniform": torch.rand(n, device="cuda") * 1e4 + 100, "zeros": torch.zeros(n, device="cuda"), "const100": torch.full((n,), 100.0, device="cuda")}
for name, x in cases.items():
    ctx.ordered_sum_f32(x)
    t = time.time()
    for _ in range(5): r = ctx.ordered_sum_f32(x)
    dt = (time.time() - t) / 5
    print(f"{name}: {dt*1e3:.3f} ms  ({dt/ (n/2048) * 1e6:.2f} us per 2048-chunk)  sum={r}")
