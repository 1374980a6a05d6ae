"""GPU: X-Trans Markesteijn demosaic (xtrans_demosaic.cc:181-969) bit-exact vs the oracle."""
import numpy as np
import pytest

import oracle_lib as O
from art_amd import capi, synth

pytestmark = pytest.mark.gpu


def run(gpu_ctx, raw, passes, lab, xt=synth.XTRANS_FUJI):
    h, w = raw.shape
    out = [np.full((h, w), -1.0, np.float32) for _ in range(3)]
    gpu_ctx.demosaic_xtrans(passes, lab, capi.host_plane(raw), xt, synth.XTRANS_RGB_CAM, capi.host_rgb(out))
    return out


@pytest.mark.parametrize("passes,lab", [(1, False), (3, True), (1, True), (2, False)])
@pytest.mark.parametrize("size", [(330, 250), (417, 309)])
def test_xtrans_bit_exact(gpu_ctx, passes, lab, size):
    w, h = size                                   # 3-4 tile columns incl. partial edge tiles
    raw = synth.xtrans_frame(w, h, seed=7, noise=1500)
    got = run(gpu_ctx, raw, passes, lab)
    ref = O.xtrans_demosaic(raw, synth.XTRANS_FUJI, synth.XTRANS_RGB_CAM, passes, lab)
    for g, r in zip(got, ref):
        assert np.array_equal(g.view(np.uint32), r.view(np.uint32))


@pytest.mark.parametrize("w,h,passes,lab", [(64, 64, 3, True), (70, 66, 1, False), (115, 117, 3, True), (131, 120, 4, True), (122, 230, 3, False)])
def test_xtrans_small_frames(gpu_ctx, w, h, passes, lab):
    """one or two tiles per direction, the second one a sliver: every bound of the LDS phases at its smallest"""
    raw = synth.xtrans_frame(w, h, seed=w + h, noise=1200)
    got = run(gpu_ctx, raw, passes, lab)
    ref = O.xtrans_demosaic(raw, synth.XTRANS_FUJI, synth.XTRANS_RGB_CAM, passes, lab)
    for g, r in zip(got, ref):
        assert np.array_equal(g.view(np.uint32), r.view(np.uint32))


@pytest.mark.parametrize("passes,lab", [(1, False), (2, True)])
def test_xtrans_more_tiles_than_workgroups(gpu_ctx, passes, lab):
    """25 x 24 tiles for 512 workgroups: some walk two tiles, with the LDS buffer and the arena as the first tile left them"""
    w, h = 2400, 2350
    raw = synth.xtrans_frame(w, h, seed=11, noise=900)
    got = run(gpu_ctx, raw, passes, lab)
    ref = O.xtrans_demosaic(raw, synth.XTRANS_FUJI, synth.XTRANS_RGB_CAM, passes, lab)
    for g, r in zip(got, ref):
        assert np.array_equal(g.view(np.uint32), r.view(np.uint32))


def test_xtrans_shifted_pattern_and_device_planes(gpu_ctx):
    """a sensor whose pattern phase differs (rolled colour map), device-resident planes with a padded row stride"""
    import torch
    w, h = 390, 280
    xt = np.roll(np.roll(synth.XTRANS_FUJI, 2, axis=0), 1, axis=1)
    raw = synth.bayer_frame(w, h, 0, 3, 1024, True, True, xtrans=xt)
    ref = O.xtrans_demosaic(raw, xt, synth.XTRANS_RGB_CAM, 3, True)
    d_raw = torch.from_numpy(raw).cuda()
    d_out = [torch.empty((h, w), dtype=torch.float32, device="cuda") for _ in range(3)]
    gpu_ctx.demosaic_xtrans(3, True, capi.device_plane(d_raw), xt, synth.XTRANS_RGB_CAM, capi.RGB(*[capi.device_plane(t) for t in d_out]))
    gpu_ctx.synchronize()
    for t, r in zip(d_out, ref):
        assert np.array_equal(t.cpu().numpy().view(np.uint32), r.view(np.uint32))


def test_xtrans_rejects_non_xtrans_maps(gpu_ctx):
    raw = synth.xtrans_frame(128, 128, seed=1)
    out = [np.zeros((128, 128), np.float32) for _ in range(3)]
    bad = np.zeros((6, 6), np.int32)
    with pytest.raises(capi.ArtGpuError):
        gpu_ctx.demosaic_xtrans(1, False, capi.host_plane(raw), bad, synth.XTRANS_RGB_CAM, capi.host_rgb(out))
