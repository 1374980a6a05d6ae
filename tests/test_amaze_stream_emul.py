"""AMaZE v2 (LDS row streaming, art_amd/csrc/amaze_stream_core.h): the stage code and its schedule, compiled for the host and
executed thread by thread between the two barriers of a step (tests/emul/amaze_stream_emul.cc), against the CPU oracle.

No GPU needed: every cross-thread dependency of the schedule has to go through a barrier, so any thread order between two
barriers must give the oracle's bits; the ring-slot tags prove that every consumed read finds the tile row it expects (ring
depths / stage offsets), and LDS starts as NaN so that a consumed read of a never-written slot shows up in the output."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from art_amd import synth
import oracle_lib

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "emul", "amaze_stream_emul.cc")
CORE = os.path.join(HERE, "..", "art_amd", "csrc", "amaze_stream_core.h")
SO = os.path.join(HERE, "emul", "libamaze_stream_emul.so")
_fp = C.POINTER(C.c_float)


@pytest.fixture(scope="module")
def emul():
    if not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in (SRC, CORE)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-math-errno", "-msse2",
                               "-Wno-unknown-pragmas", "-o", SO, SRC])
    lib = C.CDLL(SO)
    assert lib.amaze_stream_emul_lds_bytes() <= 160 * 1024
    return lib


def _streamable(top, left, w, h):
    """Same classification as artgpu_api.hip: full 160x160 tiles whose mirrored bottom / right fill does not over-run."""
    rr1 = min(top + 160, h + 16) - top
    cc1 = min(left + 160, w + 16) - left
    return rr1 == 160 and cc1 == 160 and (top + 160 <= h or top + 160 == h + 16) and (left + 160 <= w or left + 160 == w + 16)


def _run(lib, raw, filt, gain, order):
    h, w = raw.shape
    ref = oracle_lib.amaze(raw, filt, gain, 4)
    out = [np.full((h, w), np.nan, np.float32) for _ in range(3)]
    info = (C.c_longlong * 8)()
    ntiles = nvalid = 0
    for top in range(-16, h, 128):
        for left in range(-16, w, 128):
            if not _streamable(top, left, w, h):
                continue
            ntiles += 1
            lib.amaze_stream_emul_tile(raw.ctypes.data_as(_fp), C.c_long(w), w, h, C.c_uint(filt), C.c_float(np.float32(1.0 / gain)),
                                       C.c_float(np.float32(0.8 / gain)), top, left, *[o.ctypes.data_as(_fp) for o in out], C.c_long(w), order, info)
            assert info[1] == 0, f"ring tag errors in tile ({top},{left}): ring {info[2]} wanted row {info[3]} found {info[4]}"
            if not info[0]:
                continue            # Nyquist sites outside the tile's bounding box: the arena kernel redoes the tile
            nvalid += 1
            ys, xs = slice(top + 16, top + 144), slice(left + 16, left + 144)
            for k in range(3):
                assert np.array_equal(out[k][ys, xs].view(np.uint32), ref[k][ys, xs].view(np.uint32)), f"tile ({top},{left}) plane {k}"
    return ntiles, nvalid


@pytest.mark.parametrize("w,h,filt,gain,noise,order", [
    (656, 528, synth.FILTERS_RGGB, 1.0, 1024, 0),     # interior + mirrored top/left tiles, thread order 0..1023
    (640, 512, synth.FILTERS_BGGR, 1.0, 1024, 1),     # exactly aligned mirrored right/bottom tiles, reverse order
    (656, 528, synth.FILTERS_GRBG, 2.5, 0, 2),        # noise-free: partial Nyquist boxes (some tiles handed back), shuffled order
    (784, 656, synth.FILTERS_GBRG, 0.7, 4096, 3),
    (1040, 400, synth.FILTERS_RGGB, 1.0, 64, 4),      # dense Nyquist patch (more than 64 sites per step)
])
def test_stream_schedule_matches_oracle(emul, w, h, filt, gain, noise, order):
    raw = synth.bayer_frame(w, h, filt, seed=w + order, noise=noise)
    ntiles, nvalid = _run(emul, raw, filt, gain, order)
    assert ntiles >= 8 and nvalid >= ntiles // 2
