import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from art_amd import capi
ctx=capi.Context(0, torch.cuda.current_stream().cuda_stream)
W,H=8184,5456
Y=torch.rand((H,W),device="cuda")*65535
pl=capi.device_plane(Y)
fn=lambda: ctx.nlmeans(pl,50,80,1.0)
fn(); torch.cuda.synchronize(); t=time.time(); fn(); torch.cuda.synchronize(); print("nlmeans 45MP ms", (time.time()-t)*1e3)
