"""GPU parity: libartgpu.so (through the C ABI) vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

from art_amd import synth
import oracle_lib

pytestmark = pytest.mark.gpu


def _diff(got, ref):
    return [int((g.view(np.uint32) != r.view(np.uint32)).sum()) for g, r in zip(got, ref)]


CASES = [
    (352, 288, synth.FILTERS_RGGB, 1.0, 4),
    (352, 288, synth.FILTERS_BGGR, 1.0, 0),   # border<4 -> border_interpolate2(3)
    (401, 331, synth.FILTERS_GRBG, 2.1, 4),   # odd sizes, partial tiles, clip_pt != 1
    (400, 400, synth.FILTERS_GBRG, 1.0, 4),
    (640, 480, synth.FILTERS_RGGB, 1.0, 4),
]


@pytest.mark.parametrize("w,h,filt,gain,border", CASES)
def test_amaze_bit_exact(gpu_ctx, w, h, filt, gain, border):
    from art_amd import capi
    raw = synth.bayer_frame(w, h, filt, seed=w + h)
    got = gpu_ctx.demosaic_bayer_host(capi.BAYER_AMAZE, raw, filt, gain, border)
    ref = oracle_lib.amaze(raw, filt, gain, border)
    assert _diff(got, ref) == [0, 0, 0]
    assert not any(np.isnan(p).any() for p in got)


@pytest.fixture
def rcd_options(gpu_ctx):
    yield gpu_ctx
    gpu_ctx.set_option("rcd_rows", 8)


@pytest.mark.parametrize("rows", [4, 8])
@pytest.mark.parametrize("w,h,filt,noise", [(400, 300, synth.FILTERS_RGGB, 1024), (401, 331, synth.FILTERS_GRBG, 64),
                                            (64, 64, synth.FILTERS_BGGR, 1024), (64, 64, synth.FILTERS_GBRG, 1024),
                                            (640, 480, synth.FILTERS_GBRG, 0), (1233, 907, synth.FILTERS_RGGB, 4096),
                                            (195, 204, synth.FILTERS_BGGR, 512)])
def test_rcd_bit_exact(rcd_options, w, h, filt, noise, rows):
    """the LDS streaming kernel (rcd_stream.hip) with 4 / 8 rows per iteration, called three times in a row (the rings and the tile
    counter carry over)"""
    from art_amd import capi
    ctx = rcd_options
    ctx.set_option("rcd_rows", rows)
    raw = synth.bayer_frame(w, h, filt, seed=w * 3 + h, noise=noise)
    ref = oracle_lib.rcd(raw, filt)
    for _ in range(3):
        got = ctx.demosaic_bayer_host(capi.BAYER_RCD, raw, filt, 1.0, 4)
        assert _diff(got, ref) == [0, 0, 0]


@pytest.mark.parametrize("method,pad,shift", [("amaze", 16, 0), ("rcd", 16, 0), ("rcd", 13, 1), ("rcd", 7, 3)])
def test_device_pointers_and_strides(gpu_ctx, method, pad, shift):
    """Device-resident planes with a padded row stride (PlanarRGBData layout, iimage.h:653-720).  RCD stores column pairs as one
    64-bit word when the output rows allow it: odd strides and planes that start on an odd float take the scalar path."""
    import torch
    from art_amd import capi
    w, h, filt = 330, 270, synth.FILTERS_RGGB
    raw = synth.bayer_frame(w, h, filt, seed=11)
    stride = (w + 15) // 16 * 16 + pad
    d_raw = torch.zeros((h, stride), device="cuda:0")
    d_raw[:, :w] = torch.from_numpy(raw).cuda()
    d_out = [torch.full((h, stride + shift), float("nan"), device="cuda:0") for _ in range(3)]
    out = capi.RGB(*[capi.device_plane(t[:, shift:shift + w]) for t in d_out])
    m = capi.BAYER_AMAZE if method == "amaze" else capi.BAYER_RCD
    gpu_ctx.demosaic_bayer(m, capi.device_plane(d_raw[:, :w]), filt, 1.0, 4, out)
    gpu_ctx.synchronize()
    ref = oracle_lib.amaze(raw, filt, 1.0, 4) if method == "amaze" else oracle_lib.rcd(raw, filt)
    got = [t[:, shift:shift + w].cpu().numpy() for t in d_out]
    assert _diff(got, ref) == [0, 0, 0]
    for t in d_out:  # padding untouched
        assert torch.isnan(t[:, shift + w:]).all() and torch.isnan(t[:, :shift]).all()


def test_deterministic(gpu_ctx):
    from art_amd import capi
    raw = synth.bayer_frame(512, 384, synth.FILTERS_RGGB, seed=5)
    a = gpu_ctx.demosaic_bayer_host(capi.BAYER_AMAZE, raw, synth.FILTERS_RGGB)
    b = gpu_ctx.demosaic_bayer_host(capi.BAYER_AMAZE, raw, synth.FILTERS_RGGB)
    assert _diff(a, b) == [0, 0, 0]


def test_errors(gpu_ctx):
    from art_amd import capi
    raw = synth.bayer_frame(128, 128, synth.FILTERS_RGGB)
    with pytest.raises(capi.ArtGpuError):
        gpu_ctx.demosaic_bayer_host(7, raw, synth.FILTERS_RGGB)          # unknown method
    with pytest.raises(capi.ArtGpuError):
        gpu_ctx.demosaic_bayer_host(capi.BAYER_RCD, raw, 0xFFFFFFFF)     # 4-colour CFA


@pytest.fixture
def amaze_options(gpu_ctx):
    """artgpu_set_option switches for one test; the shared context goes back to its defaults afterwards."""
    def set_(**kw):
        for k, v in kw.items():
            gpu_ctx.set_option(k, v)
    yield set_
    for k, v in (("amaze_path", 0), ("amaze_split", 0), ("amaze_poison", -1), ("amaze_zero_mask", 0x81f0), ("amaze_zero_frame", 16)):
        gpu_ctx.set_option(k, v)


@pytest.mark.parametrize("filt", [synth.FILTERS_RGGB, synth.FILTERS_BGGR, synth.FILTERS_GRBG, synth.FILTERS_GBRG])
def test_amaze_selective_arena_clear_is_exact(gpu_ctx, amaze_options, filt):
    """Arena kernel: only six of the 17 arena regions are cleared per full tile (artgpu_api.hip: zero_mask), and of the five full-size
    planes among them only a 16-pixel frame (zero_frame).  With the arenas pre-filled with different byte patterns (0xFF.. = NaN) the
    result must not move, for every CFA phase and with partial and full tiles: nothing else is read before it is written."""
    from art_amd import capi
    w, h = 1100, 870
    raw = synth.bayer_frame(w, h, filt, seed=21, noise=3000)
    ref = oracle_lib.amaze(raw, filt, 1.0, 4)
    for pattern in (0xFF, 0x7F, 0xC0):
        amaze_options(amaze_path=1, amaze_poison=pattern)
        out = [np.zeros((h, w), np.float32) for _ in range(3)]
        gpu_ctx.demosaic_bayer(capi.BAYER_AMAZE, capi.host_plane(raw), filt, 1.0, 4, capi.host_rgb(out))
        for o, r in zip(out, ref):
            assert np.array_equal(o.view(np.uint32), r.view(np.uint32))


STREAM_CASES = [
    # w, h, filters, gain, noise: full interior tiles, mirrored top/left tiles, exactly aligned mirrored right/bottom tiles (640x512),
    # tiles whose border fill over-runs in the reference (650x520 -> arena kernel), noise-free frames (partial Nyquist boxes ->
    # tiles handed back to the arena kernel)
    (1040, 784, synth.FILTERS_RGGB, 1.0, 1024),
    (640, 512, synth.FILTERS_BGGR, 1.0, 1024),
    (650, 520, synth.FILTERS_GRBG, 1.0, 512),
    (1296, 1040, synth.FILTERS_GBRG, 1.0, 0),
    (1296, 1040, synth.FILTERS_RGGB, 2.5, 32),
    (784, 656, synth.FILTERS_GBRG, 0.7, 4096),
]


@pytest.mark.parametrize("w,h,filt,gain,noise", STREAM_CASES)
def test_amaze_stream_matches_oracle_and_arena_kernel(gpu_ctx, amaze_options, w, h, filt, gain, noise):
    """The LDS streaming kernel (default path) and the arena kernel (amaze_path 1) give the oracle's bits."""
    from art_amd import capi
    raw = synth.bayer_frame(w, h, filt, seed=w + noise, noise=noise)
    ref = oracle_lib.amaze(raw, filt, gain, 4)
    for _ in range(4):      # repeated: a second attempt at a tile runs on whichever workgroup is free first (cross-XCD hand-over)
        got = gpu_ctx.demosaic_bayer_host(capi.BAYER_AMAZE, raw, filt, gain, 4)
        assert _diff(got, ref) == [0, 0, 0]
    amaze_options(amaze_path=1)
    got1 = gpu_ctx.demosaic_bayer_host(capi.BAYER_AMAZE, raw, filt, gain, 4)
    assert _diff(got1, ref) == [0, 0, 0]


def test_amaze_phase_per_kernel_path_is_identical(gpu_ctx, amaze_options):
    """amaze_path 1 + amaze_split 1 runs the arena kernel's 20 phases as one kernel launch each (profiling path): same bits."""
    from art_amd import capi
    w, h, filt = 904, 648, synth.FILTERS_GRBG
    raw = synth.bayer_frame(w, h, filt, seed=23, noise=2500)
    ref = oracle_lib.amaze(raw, filt, 1.0, 4)
    amaze_options(amaze_path=1, amaze_split=1)
    out = [np.zeros((h, w), np.float32) for _ in range(3)]
    gpu_ctx.demosaic_bayer(capi.BAYER_AMAZE, capi.host_plane(raw), filt, 1.0, 4, capi.host_rgb(out))
    for o, r in zip(out, ref):
        assert np.array_equal(o.view(np.uint32), r.view(np.uint32))


@pytest.mark.parametrize("noise", [0, 64])
def test_amaze_redo_queue_under_load(gpu_ctx, noise):
    """Hundreds of second attempts in one frame, several tiles per workgroup, ten calls in a row: a 13 MP scene of scattered Nyquist
    patches (synth.nyquist_patches_frame) gives most tiles a partial Nyquist box, so about a quarter of them go through the redo queue
    -- published by one workgroup, streamed again by whichever runs out of tiles first, usually on another XCD -- while the tiles
    themselves are handed out by the shared counter.  Every call has to give the oracle's bits, and the counters have to show that the
    path was really taken (0: entries pulled, 2: entries published, 3: handed to the arena kernel instead)."""
    from art_amd import capi
    w, h, filt = 4224, 3168, synth.FILTERS_RGGB
    raw = synth.nyquist_patches_frame(w, h, filt, noise=noise)
    ref = oracle_lib.amaze(raw, filt, 1.0, 4)
    for call in range(10):
        got = gpu_ctx.demosaic_bayer_host(capi.BAYER_AMAZE, raw, filt, 1.0, 4)
        assert _diff(got, ref) == [0, 0, 0], f"call {call}"
        pulled, published, to_arena = (gpu_ctx.get_option(f"amaze_counter{k}") for k in (0, 2, 3))
        assert published >= 100, published
        assert pulled + to_arena >= published           # every published entry was taken by a stream workgroup or by the arena kernel
