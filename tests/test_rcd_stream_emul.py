"""RCD row streaming (art_amd/csrc/rcd_stream_core.h): the stage code and its schedule, compiled for the host and executed thread
by thread between barriers (tests/emul/rcd_stream_emul.cc), against the CPU oracle.

No GPU needed: every cross-thread dependency has to go through a barrier, so any thread order between two barriers must give the
oracle's bits; ring-slot tags prove that every consumed read finds the tile row it expects (ring depths / stage lags); LDS starts
as NaN or arbitrary garbage and is not cleared between tiles, so a read of a never-written or stale slot shows."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from art_amd import synth
import oracle_lib

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "emul", "rcd_stream_emul.cc")
CORE = os.path.join(HERE, "..", "art_amd", "csrc", "rcd_stream_core.h")
SO = os.path.join(HERE, "emul", "librcd_stream_emul.so")
_fp = C.POINTER(C.c_float)


@pytest.fixture(scope="module")
def emul():
    if not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in (SRC, CORE)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-math-errno", "-msse2",
                               "-Wno-unknown-pragmas", "-o", SO, SRC])
    return C.CDLL(SO)


def _run(lib, raw, filt, R, order):
    h, w = raw.shape
    out = [np.full((h, w), np.nan, np.float32) for _ in range(3)]
    info = (C.c_longlong * 8)()
    rc = lib.rcd_stream_emul(raw.ctypes.data_as(_fp), C.c_long(w), w, h, C.c_uint(filt), *[o.ctypes.data_as(_fp) for o in out], C.c_long(w),
                             R, order, info)
    assert rc == 0
    assert info[1] == 0, f"{info[1]} ring tag errors; first: core line {info[2]} wanted row {info[3]} found tag {info[4]} at float {info[5]}"
    assert info[6] <= 80 * 1024, "two workgroups per CU"
    return out, info


@pytest.mark.parametrize("w,h,filt,noise,R,order", [
    (400, 380, synth.FILTERS_RGGB, 1024, 4, 0),       # 3 x 3 tiles with partial right / bottom tiles; thread order 0..NT-1
    (388, 370, synth.FILTERS_BGGR, 64, 8, 1),         # reverse order
    (547, 231, synth.FILTERS_GRBG, 4096, 4, 2 + 512), # odd width, shuffled order, garbage LDS
    (371, 563, synth.FILTERS_GBRG, 0, 8, 2),          # odd height, noise-free
    (194, 194, synth.FILTERS_RGGB, 512, 8, 1),        # one full tile and slivers
    (64, 64, synth.FILTERS_RGGB, 512, 4, 0),          # a single small tile
    (195, 204, synth.FILTERS_BGGR, 512, 4, 2),        # second tile column / row narrower than two borders: writes nothing
])
def test_stream_schedule_matches_oracle(emul, w, h, filt, noise, R, order):
    raw = synth.bayer_frame(w, h, filt, seed=w + h + R, noise=noise)
    ref = oracle_lib.rcd(raw, filt)
    b = 9
    for o in sorted({order, (order + 1) % 3 + (order & 512)}):     # the listed thread order and one more
        out, info = _run(emul, raw, filt, R, o)
        for k in range(3):
            a, r = out[k][b:h - b, b:w - b], ref[k][b:h - b, b:w - b]
            bad = a.view(np.uint32) != r.view(np.uint32)
            assert not bad.any(), f"order {o} plane {k}: {bad.sum()} of {bad.size} differ, first at {tuple(np.argwhere(bad)[0] + b)}"
