"""Which sources a kernel is built from, and their digest: the rocprofv3 counter summaries under profiles/ carry the digest of the day they
were taken, and bench.py marks a traffic figure whose kernel has changed since (`traffic_stale`)."""
from __future__ import annotations

import hashlib
import os

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
_COMMON = ["kernels.h", "devmath.h", "devsleef.h", "paramcurve.h"]      # paramcurve.h: ParamCurve is embedded by value in kernels.h' argument structs
KERNEL_SOURCES = {
    "amaze_stream_kernel": ["amaze_stream.hip", "amaze_stream_core.h"],
    "rcd_stream_kernel": ["rcd_stream.hip", "rcd_stream_core.h"],
    "xtrans_tiles_kernel": ["xtrans.hip"],
    "nlm_group_kernel": ["nlm_sweep.hip"],
    "shrink_blur_kernel": ["shrinkblur.hip"],
}


def kernel_source_sha256(kernel: str) -> str:
    h = hashlib.sha256()
    for name in KERNEL_SOURCES[kernel] + _COMMON:
        with open(os.path.join(_CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()
