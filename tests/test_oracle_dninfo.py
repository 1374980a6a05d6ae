"""AUTOMATIC chrominance estimation, oracle side (ipdenoise.cc:66-206,227-669,800-1093): properties the reference's formulas
imply, checked on the CPU restatement (the GPU parity tests are in test_gpu_dninfo.py)."""
import numpy as np

import oracle_lib as O

MAT = np.array([[0.6325, 0.2312, 0.0921], [0.2198, 0.7712, 0.0090], [0.0166, 0.0713, 0.7514]])


def planes(w, h, noise, seed=1):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    base = [9000 + 6000 * np.sin(0.013 * x) * np.cos(0.011 * y), 11000 + 5000 * np.cos(0.009 * x + 0.4), 7000 + 4000 * np.sin(0.015 * y)]
    return [np.maximum(b + rng.normal(0, noise, (h, w)), 0).astype(np.float32) for b in base]


def test_estimate_grows_with_chroma_noise_and_is_deterministic():
    lo = O.denoise_compute_params(planes(328, 264, 150), 4, (1.9, 1.0, 1.5), True, MAT, O.REC2020_WS_D)
    hi = O.denoise_compute_params(planes(328, 264, 900), 4, (1.9, 1.0, 1.5), True, MAT, O.REC2020_WS_D)
    again = O.denoise_compute_params(planes(328, 264, 900), 4, (1.9, 1.0, 1.5), True, MAT, O.REC2020_WS_D)
    assert lo is not None and hi is not None
    assert np.array_equal(hi[0], again[0]) and np.array_equal(hi[1], again[1])
    assert hi[0][0] > lo[0][0] > 0                      # store.chrominance
    info = hi[1]
    assert np.all(info[:, 10] == 15)                    # Nb: 5 levels x 3 directions
    assert np.all(info[:, 1] >= info[:, 3]) and np.all(info[:, 2] >= info[:, 4])     # max >= min of the band MADs
    assert np.all(info[:, 5] >= 100.0)                  # chroma floor of the map
    assert np.all((info[:, 6] >= 2.0) & (info[:, 6] <= 32768.0))
    assert np.all((info[:, 9] >= 0) & (info[:, 9] <= 1))


def test_too_small_image_is_rejected():
    assert O.denoise_compute_params(planes(90, 90, 100), 4, (1, 1, 1), True, MAT, O.REC2020_WS_D) is None
