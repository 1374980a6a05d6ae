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
    """Same classification as artgpu_api.hip: tiles that are 160 wide (any height that writes a pixel) whose mirrored bottom /
    right border fill does not over-run the row / the cfa plane in the reference."""
    rr1 = min(top + 160, h + 16) - top
    cc1 = min(left + 160, w + 16) - left
    return (cc1 == 160 and rr1 > 32 and (left + 160 <= w or left + 160 == w + 16)
            and (rr1 < 160 or top + 160 <= h or top + 160 == h + 16))


def _stream(lib, raw, filt, gain, order, seq, out, redo=None, boxes=None):
    h, w = raw.shape
    n = len(seq)
    tops = (C.c_int * n)(*[t for t, _ in seq])
    lefts = (C.c_int * n)(*[l for _, l in seq])
    valid = (C.c_int * n)()
    box_out = (C.c_int * (4 * n))()
    info = (C.c_longlong * 8)()
    redo_a = (C.c_int * n)(*redo) if redo is not None else None
    boxes_a = (C.c_int * (4 * n))(*boxes) if boxes is not None else None
    lib.amaze_stream_emul_seq(raw.ctypes.data_as(_fp), C.c_long(w), w, h, C.c_uint(filt), C.c_float(np.float32(1.0 / gain)),
                              C.c_float(np.float32(0.8 / gain)), n, tops, lefts, redo_a, boxes_a, *[o.ctypes.data_as(_fp) for o in out], C.c_long(w),
                              order, valid, box_out, info)
    assert info[1] == 0, f"ring tag errors: ring {info[2]} wanted row {info[3]} found {info[4]}"
    return list(valid), list(box_out)


def _check(out, ref, top, left, h):
    rr1 = min(top + 160, h + 16) - top
    ys, xs = slice(top + 16, top + rr1 - 16), slice(left + 16, left + 144)
    for k in range(3):
        assert np.array_equal(out[k][ys, xs].view(np.uint32), ref[k][ys, xs].view(np.uint32)), f"tile ({top},{left}) plane {k}"


def _run(lib, raw, filt, gain, order, nseq=2):
    """All streamable tiles of the frame, dealt to `nseq` workgroup sequences (tile k of a sequence starts while tile k-1 is
    still in the late stages).  Tiles whose Nyquist sites do not all lie inside the tile's bounding box are streamed a second time
    with the true box (what the kernel's redo queue does) and must then match as well."""
    h, w = raw.shape
    ref = oracle_lib.amaze(raw, filt, gain, 4)
    out = [np.full((h, w), np.nan, np.float32) for _ in range(3)]
    tiles = [(top, left) for top in range(-16, h, 128) for left in range(-16, w, 128) if _streamable(top, left, w, h)]
    nvalid = 0
    again = []
    for q in range(nseq):
        seq = tiles[q::nseq]
        if not seq:
            continue
        valid, boxes = _stream(lib, raw, filt, gain, order, seq, out)
        for k, ((top, left), ok) in enumerate(zip(seq, valid)):
            if ok:
                nvalid += 1
                _check(out, ref, top, left, h)
            else:
                again.append(((top, left), boxes[4 * k:4 * k + 4]))
    if again:
        seq = [t for t, _ in again]
        _stream(lib, raw, filt, gain, order, seq, out, redo=[1] * len(seq), boxes=[v for _, b in again for v in b])
        for top, left in seq:
            _check(out, ref, top, left, h)
    return len(tiles), nvalid, len(again)


@pytest.mark.parametrize("w,h,filt,gain,noise,order", [
    (656, 528, synth.FILTERS_RGGB, 1.0, 1024, 0),     # interior + mirrored top/left tiles, thread order 0..1023
    (640, 512, synth.FILTERS_BGGR, 1.0, 1024, 1),     # exactly aligned mirrored right/bottom tiles, reverse order
    (912, 400, synth.FILTERS_RGGB, 0.7, 0, 2),        # noise-free: partial Nyquist boxes (some tiles need the second attempt), shuffled order
    (1296, 1040, synth.FILTERS_RGGB, 1.0, 32, 1),
    (784, 656, synth.FILTERS_GBRG, 0.7, 4096, 3),
    (1040, 400, synth.FILTERS_RGGB, 1.0, 64, 4),      # dense Nyquist patch (more than 64 sites per step)
    (656, 471, synth.FILTERS_GBRG, 1.0, 1024, 5),     # partial-height bottom tiles (odd number of rows)
    (656, 600, synth.FILTERS_RGGB, 1.0, 512, 0),      # partial-height bottom tiles with mirrored rows
])
def test_stream_schedule_matches_oracle(emul, w, h, filt, gain, noise, order):
    raw = synth.bayer_frame(w, h, filt, seed={912: 6, 1296: 9}.get(w, w + order), noise=noise)
    ntiles, nvalid, nredo = _run(emul, raw, filt, gain, order)
    assert ntiles >= 8 and nvalid + nredo == ntiles
    if w in (912, 1296):
        assert nredo > 0        # these frames have tiles with partial Nyquist boxes: the second attempt is exercised
