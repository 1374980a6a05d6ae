"""art_amd -- MI355X-native raw-development hot path (drop-in for ART's rtengine demosaic /
denoise / tone stages).  The compute lives in ``libartgpu.so`` (hand-written HIP for gfx950
behind the C ABI of ``include/artgpu.h``); this package is the thin Python binding used by
the tests and ``bench.py``.  There is no CPU fallback: importing :mod:`art_amd.capi` fails
loudly when the HIP library has not been built.
"""
__all__ = ["synth"]
