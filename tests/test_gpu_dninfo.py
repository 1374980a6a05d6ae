"""artgpu_denoise_compute_params against the oracle: bit-exact store and per-crop values
(ImProcFunctions::denoiseComputeParams, rtengine/ipdenoise.cc:800-1093)."""
import numpy as np
import pytest

import oracle_lib as O
from test_oracle_dninfo import MAT, planes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu_ctx():
    from art_amd import capi
    ctx = capi.Context(0)
    yield ctx
    ctx.close()


def run(gpu_ctx, pl, border, mul, clip, gamma=1.7, aggressive=False, factor=1.0, device=False):
    from art_amd import capi
    dn = capi.DenoiseParams(40.0, 50.0, 0, 15.0, 0.0, 0.0, gamma, 1 if aggressive else 0, 0, 1)
    if device:
        import torch
        d = [torch.from_numpy(p.copy()).cuda() for p in pl]
        rgb = capi.RGB(*[capi.device_plane(t) for t in d])
    else:
        d = [p.copy() for p in pl]          # host_rgb borrows the arrays: keep them alive
        rgb = capi.host_rgb(d)
    st = gpu_ctx.denoise_compute_params(rgb, border, mul, clip, MAT, O.REC2020_WS_D, dn, auto_factor=factor)
    return st, dn


@pytest.mark.parametrize("w,h,border,noise,aggressive,device", [
    (418, 322, 4, 400, False, False),      # odd crop sizes (205 x 157): 4-lane groups + scalar tails, odd half-res
    (424, 344, 4, 1200, True, True),       # even sizes, aggressive (reduc 0.9), planes resident on the device
    (300, 240, 0, 60, False, False),       # nearly clean: the low branches of calcautodn_info
])
def test_compute_params_bit_exact(gpu_ctx, w, h, border, noise, aggressive, device):
    pl = planes(w, h, noise, seed=w)
    mul = (2.1, 1.0, 1.55)
    st, dn = run(gpu_ctx, pl, border, mul, True, aggressive=aggressive, factor=1.25, device=device)
    ref = O.denoise_compute_params(pl, border, mul, True, MAT, O.REC2020_WS_D, 1.7, aggressive)
    assert ref is not None and st.valid == 1
    store, info = ref
    got_info = np.array([list(st.crop_info[k]) for k in range(9)], np.float32)
    assert np.array_equal(got_info[:, :11].view(np.uint32), info[:, :11].view(np.uint32)), (got_info[:, :11], info[:, :11])
    assert np.float32(st.chrominance) == store[0] and np.float32(st.chrominance_red_green) == store[1] and np.float32(st.chrominance_blue_yellow) == store[2]
    assert np.array_equal(np.array(list(st.ch_M), np.float32), store[3:12])
    assert np.array_equal(np.array(list(st.max_r), np.float32), store[12:21])
    assert np.array_equal(np.array(list(st.max_b), np.float32), store[21:30])
    assert dn.chrominance == float(store[0]) * 1.25 and dn.chrominance_red_green == float(store[1]) * 1.25
    assert store[0] > 0


def test_valid_store_is_reused_and_manual_is_untouched(gpu_ctx):
    from art_amd import capi
    pl = planes(300, 240, 300)
    st, dn = run(gpu_ctx, pl, 0, (1, 1, 1), False)
    c = dn.chrominance
    # a valid store short-circuits (ipdenoise.cc:802-809): even nonsense planes are not looked at
    dn2 = capi.DenoiseParams(40.0, 50.0, 0, 0.0, 0.0, 0.0, 1.7, 0, 0, 1)
    z = [np.zeros((240, 300), np.float32) for _ in range(3)]
    gpu_ctx.denoise_compute_params(capi.host_rgb(z), 0, (1, 1, 1), False, MAT, O.REC2020_WS_D, dn2, auto_factor=2.0, store=st)
    assert dn2.chrominance == 2.0 * c
    dn3 = capi.DenoiseParams(40.0, 50.0, 0, 7.0, 1.0, 2.0, 1.7, 0, 0, 0)     # MANUAL
    st3 = gpu_ctx.denoise_compute_params(capi.host_rgb(pl), 0, (1, 1, 1), False, MAT, O.REC2020_WS_D, dn3)
    assert st3.valid == 0 and (dn3.chrominance, dn3.chrominance_red_green, dn3.chrominance_blue_yellow) == (7.0, 1.0, 2.0)


def test_too_small_fails_loudly(gpu_ctx):
    from art_amd import capi
    dn = capi.DenoiseParams(40.0, 50.0, 0, 15.0, 0.0, 0.0, 1.7, 0, 0, 1)
    z = [np.zeros((90, 90), np.float32) for _ in range(3)]
    with pytest.raises(capi.ArtGpuError):
        gpu_ctx.denoise_compute_params(capi.host_rgb(z), 0, (1, 1, 1), False, MAT, O.REC2020_WS_D, dn)


def seq_sum(x):
    """acc = 0; for v in x: acc += v in fp32 (numpy's accumulate is strictly sequential)"""
    x = np.asarray(x, np.float32)
    return np.float32(0.0) if x.size == 0 else np.add.accumulate(x, dtype=np.float32)[-1]


def _cases():
    rng = np.random.default_rng(5)
    yield "uniform", rng.uniform(100, 1e5, 300_001).astype(np.float32)
    yield "const100_ties", np.full(2_000_003, 100.0, np.float32)          # 100/8 = 12.5: a tie at every step while the sum is in [2^26, 2^27)
    yield "const_2_and_32768", np.where(rng.random(700_000) < 0.5, 2.0, 32768.0).astype(np.float32)
    yield "half_ulps", (np.float32(1.0) + rng.integers(0, 4, 500_000).astype(np.float32) * np.float32(0.5)).astype(np.float32)
    z = rng.uniform(1e4, 2e5, 400_000).astype(np.float32)
    z[rng.random(z.size) < 0.97] = 0.0
    z[:50_000] = 0.0
    yield "sparse_leading_zeros", z
    yield "all_zero", np.zeros(10_000, np.float32)
    yield "empty", np.zeros(0, np.float32)
    yield "one", np.array([3.25], np.float32)
    yield "wide_range", np.exp(rng.uniform(np.log(1e-3), np.log(1e6), 250_000)).astype(np.float32)
    yield "growing", np.sort(np.exp(rng.uniform(0, 20, 100_000))).astype(np.float32)
    yield "big_then_small", np.concatenate([[1e9], rng.uniform(0, 3, 200_000)]).astype(np.float32)
    yield "lum_like", np.clip(rng.normal(9000, 6000, 1_000_000), 2, 32768).astype(np.float32)
    neg = rng.uniform(-50, 100, 6000).astype(np.float32)
    yield "with_negatives", neg
    n = rng.uniform(1, 10, 5000).astype(np.float32); n[3000] = np.nan
    yield "nan_inside", n
    i = rng.uniform(1, 10, 5000).astype(np.float32); i[100] = np.inf
    yield "inf_inside", i
    i2 = i.copy(); i2[4000] = np.nan
    yield "inf_then_nan", i2
    yield "overflow_to_inf", np.full(3000, 3e38, np.float32)
    yield "denormals", np.full(5000, 1e-41, np.float32)


@pytest.mark.parametrize("name", [n for n, _ in _cases()])
def test_ordered_sum_equals_sequential_fp32(gpu_ctx, name):
    """The exact scan behind the image statistics (orderedsum.hip) against a strictly sequential fp32 accumulation."""
    x = dict(_cases())[name]
    with np.errstate(all="ignore"):
        ref = seq_sum(x)
    got = gpu_ctx.ordered_sum_f32(x)
    assert got.view(np.uint32) == ref.view(np.uint32) or (np.isnan(got) and np.isnan(ref)), (name, got, ref)
