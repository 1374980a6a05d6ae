"""Deterministic inputs shared by make_golden.py (container only) and the tests (everywhere)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def wavelet_input(w, h, seed):
    """Integer-valued synthetic frame scaled into [0, ~40000) (values exactly representable products)."""
    from art_amd import synth
    return (synth.bayer_frame(w, h, synth.FILTERS_RGGB, seed=seed) * np.float32(0.61)).astype(np.float32)
