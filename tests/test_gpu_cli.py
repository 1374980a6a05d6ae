"""GPU: the C++ front end (art_amd/artgpu-cli over rtengine_gpu.h) end to end against the oracle,
on BASELINE config 1 (RCD, 4000x3000 RGGB) and a small AMaZE + denoise run."""
import json
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from art_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "art_amd", "artgpu-cli")
MUL = (2.1374, 1.0, 1.5918)
MAT = np.array([[0.6325, 0.2312, 0.0921], [0.2198, 0.7712, 0.0090], [0.0166, 0.0713, 0.7514]])


def tone_lut():
    x = np.arange(65536, dtype=np.float64) / 65535.0
    return ((1.0 - np.cos(np.pi * x ** 0.7)) / 2.0 * 65535.0).astype(np.float32)


def read_ppm16(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"P6"
        w, h = map(int, f.readline().split())
        assert f.readline().strip() == b"65535"
        data = np.frombuffer(f.read(), dtype=">u2").reshape(h, w, 3)
    return data


def oracle_pipeline(raw, filt, method, border, denoise=None, smoothing=None, expcomp=0.0, dct=True):
    planes = O.rcd(raw, filt) if method == "rcd" else O.amaze(raw, filt, 1.0, border)
    h, w = raw.shape
    img = O.get_image(planes, border, border, w - 2 * border, h - 2 * border, MUL, True)
    img = O.convert_color_space(img, MAT)
    if denoise:
        curve, _ = O.noise_curve()
        img = O.improc_denoise(img, dict(luminance=denoise[0], chrominance=denoise[1]), calclum_mat=MAT, noise_c_curve=curve,
                               smoothing=smoothing is not None, radius=(smoothing or (3, 0, 0))[0], nl_strength=(smoothing or (3, 0, 0))[1],
                               nl_detail=(smoothing or (3, 0, 80))[2], ecomp=expcomp, detail_recovery=dct)
    img = O.exposure(img, float(np.float32(2.0 ** expcomp)), 0.0)
    img = O.tone_std(img, tone_lut(), 1.0, True)
    return img


def run_cli(tmp_path, raw, method, extra=()):
    inp = tmp_path / "frame.f32"
    out = tmp_path / "out.ppm"
    raw.astype("<f4").tofile(inp)
    h, w = raw.shape
    res = subprocess.run([CLI, "--in", str(inp), "--width", str(w), "--height", str(h), "--method", method, "--out", str(out), *extra],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr
    run_cli.last_stderr = res.stderr
    return json.loads(res.stdout.strip().splitlines()[-1]), read_ppm16(out)


def test_config1_rcd_12mp_through_cli(tmp_path):
    w, h, filt = 4000, 3000, synth.FILTERS_RGGB
    raw = synth.bayer_frame(w, h, filt, seed=1)
    info, ppm = run_cli(tmp_path, raw, "rcd")
    assert (info["out_width"], info["out_height"]) == (w - 8, h - 8)
    ref = oracle_pipeline(raw, filt, "rcd", 4)
    q = np.stack([np.rint(np.clip(p, 0, 65535)).astype(np.uint16) for p in ref], axis=-1)
    assert np.array_equal(ppm, q)          # bit-exact fp32 pipeline -> identical 16-bit output


def test_amaze_denoise_through_cli(tmp_path):
    w, h, filt = 648, 488, synth.FILTERS_RGGB
    raw = synth.bayer_frame(w, h, filt, seed=2, noise=2048)
    info, ppm = run_cli(tmp_path, raw, "amaze", ("--denoise", "40,15"))
    ref = oracle_pipeline(raw, filt, "amaze", 4, denoise=(40.0, 15.0))
    q = np.stack([np.rint(np.clip(p, 0, 65535)).astype(np.int32) for p in ref], axis=-1)
    # the DCT detail-recovery stage is tolerance-checked (third-party FFTW in the reference): allow +-3 counts
    assert np.abs(ppm.astype(np.int32) - q).max() <= 3


def test_config4_stages_through_cli(tmp_path):
    """BASELINE config 4's per-frame pipe: AMaZE + FTblockDN + guided smoothing + NL-means + exposure + tone."""
    w, h, filt = 648, 488, synth.FILTERS_RGGB
    raw = synth.bayer_frame(w, h, filt, seed=3, noise=2048)
    info, ppm = run_cli(tmp_path, raw, "amaze", ("--denoise", "40,15", "--smoothing", "3,50,80", "--expcomp", "0.3"))
    ref = oracle_pipeline(raw, filt, "amaze", 4, denoise=(40.0, 15.0), smoothing=(3, 50, 80), expcomp=0.3)
    q = np.stack([np.rint(np.clip(p, 0, 65535)).astype(np.int32) for p in ref], axis=-1)
    err = np.abs(ppm.astype(np.int32) - q)
    # The DCT stage's round-off (<= 0.0625 on this scale, tests/test_gpu_denoise.py) passes through NL-means, whose weights are
    # exp(-distance) table look-ups of patch distances: a rounding-level change of L flips table indices, and the differences are no
    # longer rounding-sized.  That is a property of the pipeline, not of the device: the checker itself, run with a plain fp32
    # direct-form DCT instead of the double-accumulated one, moves the 16-bit output by the same amount -- that run is the yardstick
    # here.  (Everything but the DCT stage is compared bit for bit at full size in tests/test_gpu_fullsize.py.)
    ref32 = oracle_pipeline(raw, filt, "amaze", 4, denoise=(40.0, 15.0), smoothing=(3, 50, 80), expcomp=0.3, dct="f32")
    q32 = np.stack([np.rint(np.clip(p, 0, 65535)).astype(np.int32) for p in ref32], axis=-1)
    err32 = np.abs(q32 - q)
    print(f"config-4 CLI, 16-bit output vs the double-DCT checker: device max {err.max()} p99.9 {np.percentile(err, 99.9):.0f} differing {100.0 * (err > 0).mean():.1f} % | "
          f"checker with fp32 direct-form DCT max {err32.max()} p99.9 {np.percentile(err32, 99.9):.0f} differing {100.0 * (err32 > 0).mean():.1f} %")
    assert np.median(err) <= 1
    assert np.percentile(err, 99.9) <= 2 * np.percentile(err32, 99.9) + 2 and err.max() <= 2 * err32.max() + 8, (err.max(), err32.max())
    # ... and absolute caps beside the relative ones (measured: max 117 of 65535 on this frame, 99.9 % within 32: DESIGN.md section 3)
    assert err.max() <= 256 and np.percentile(err, 99.9) <= 64, (err.max(), np.percentile(err, 99.9))


def test_neutral_tone_mode_through_cli(tmp_path):
    """ART's default curve mode end to end: RCD + NEUTRAL tone curve (in-range pixels are bit-exact, so the 16-bit output
    may differ only where the input exceeded the PQ LUT range -- none here after the getImage clip at 65535)."""
    w, h, filt = 1000, 600, synth.FILTERS_RGGB
    raw = synth.bayer_frame(w, h, filt, seed=5)
    info, ppm = run_cli(tmp_path, raw, "rcd", ("--tone", "neutral"))
    planes = O.rcd(raw, filt)
    img = O.get_image(planes, 4, 4, w - 8, h - 8, MUL, True)
    img = O.convert_color_space(img, MAT)
    img = O.exposure(img, 1.0, 0.0)
    ref, oor = O.tone_neutral(img, tone_lut(), 1.0, want_oor=True)
    q = np.stack([np.rint(np.clip(p, 0, 65535)).astype(np.int32) for p in ref], axis=-1)
    err = np.abs(ppm.astype(np.int32) - q)
    assert err[~oor].max() == 0 and err.max() <= 8


def test_xtrans_three_pass_through_cli(tmp_path):
    """BASELINE config 5's stages at a small size: X-Trans 3-pass + denoise + exposure + tone through the C++ front end."""
    w, h = 600, 450
    raw = synth.xtrans_frame(w, h, seed=6, noise=2048)
    info, ppm = run_cli(tmp_path, raw, "amaze", ("--xtrans", "3", "--denoise", "40,15"))
    assert (info["out_width"], info["out_height"]) == (w - 14, h - 14)
    planes = O.xtrans_demosaic(raw, synth.XTRANS_FUJI, synth.XTRANS_RGB_CAM, 3, True)
    img = O.get_image(planes, 7, 7, w - 14, h - 14, MUL, True)
    img = O.convert_color_space(img, MAT)
    curve, _ = O.noise_curve()
    img = O.improc_denoise(img, dict(luminance=40.0, chrominance=15.0), calclum_mat=MAT, noise_c_curve=curve, smoothing=False, detail_recovery=True)
    img = O.exposure(img, 1.0, 0.0)
    img = O.tone_std(img, tone_lut(), 1.0, True)
    q = np.stack([np.rint(np.clip(p, 0, 65535)).astype(np.int32) for p in img], axis=-1)
    assert np.abs(ppm.astype(np.int32) - q).max() <= 3      # DCT detail recovery tolerance only


def test_automatic_chroma_through_cli(tmp_path):
    """--chroma-auto: denoiseComputeParams on the demosaiced planes (the default chrominanceMethod of ART), then the same pipe."""
    w, h, filt = 648, 488, synth.FILTERS_RGGB
    raw = synth.bayer_frame(w, h, filt, seed=5, noise=2600)
    info, ppm = run_cli(tmp_path, raw, "amaze", ("--denoise", "40,15", "--chroma-auto"))
    planes = O.amaze(raw, filt, 1.0, 4)
    store, _ = O.denoise_compute_params(planes, 4, MUL, True, MAT, O.REC2020_WS_D)
    line = [l for l in run_cli.last_stderr.splitlines() if l.startswith("auto chrominance")][0].split()
    assert abs(float(line[2]) - float(store[0])) < 1e-5 and abs(float(line[4]) - float(store[1])) < 1e-5 and abs(float(line[6]) - float(store[2])) < 1e-5
    img = O.get_image(planes, 4, 4, w - 8, h - 8, MUL, True)
    img = O.convert_color_space(img, MAT)
    curve, _ = O.noise_curve()
    img = O.improc_denoise(img, dict(luminance=40.0, chrominance=float(store[0]), chrominanceRedGreen=float(store[1]), chrominanceBlueYellow=float(store[2]), autoch=1),
                           calclum_mat=MAT, noise_c_curve=curve, smoothing=False, detail_recovery=True)
    img = O.exposure(img, 1.0, 0.0)
    img = O.tone_std(img, tone_lut(), 1.0, True)
    q = np.stack([np.rint(np.clip(p, 0, 65535)).astype(np.int32) for p in img], axis=-1)
    assert np.abs(ppm.astype(np.int32) - q).max() <= 3


def test_n4_tools_through_the_cpp_mirror(tmp_path):
    """dual demosaic (RCD + VNG4, automatic contrast), log encoding, saturation / vibrance and Lab chromaticity driven through
    rtengine_gpu.h's RawImageSource::demosaic / ImProcFunctions::process, against the same chain of oracle calls"""
    w, h, filt = 1000, 760, synth.FILTERS_RGGB
    raw = synth.bayer_frame(w, h, filt, seed=31, noise=300)
    rng = np.random.default_rng(8)
    raw[300:370, 400:470] = (9000.0 + rng.normal(0, 200.0, (70, 70))).astype(np.float32)      # a flat tile for the threshold search
    info, got = run_cli(tmp_path, raw, "rcd", ["--dual", "vng4", "--logenc", "60", "--saturation", "25,-30", "--labchroma", "20", "--expcomp", "0.2"])
    planes, contrast = O.dual_demosaic_blend(raw, O.rcd(raw, filt), filt, 20.0, True, vng4=True)
    assert 0.0 < contrast < 100.0
    img = O.get_image(planes, 4, 4, w - 8, h - 8, MUL, True)
    img = O.convert_color_space(img, MAT)
    img = O.exposure(img, float(np.float32(2.0 ** 0.2)), 0.0)                                   # STAGE_1
    img = O.log_encoding(img, regularization=60, full_width=0, full_height=0)                  # STAGE_3: logEncoding, saturationVibrance, toneCurve, labAdjustments
    img = O.saturation_vibrance(img, 25, -30)
    img = O.tone_std(img, tone_lut(), 1.0, True)
    lab = O.image_rgb_to_lab(img)
    ident_l = np.arange(32770, dtype=np.float32)
    ident = np.arange(65536, dtype=np.float32)
    lab = O.lab_adjustments(lab, ident_l, ident, ident, np.float32((20 + 100.0) / 100.0))
    img = O.image_lab_to_rgb(lab, O.REC2020_IWS_D)
    ref = np.stack([np.rint(np.clip(p, 0, 65535)).astype(np.uint16) for p in img], axis=-1)
    assert got.shape == ref.shape
    assert np.array_equal(got, ref)


def test_batch_queue_through_cli(tmp_path):
    """artgpu-cli --batch: the batch queue's loop over BatchQueue (rtengine_gpu.h) -> artgpu_batch_run_io -- uint16 sensor frames in, the
    writers' 16-bit scanlines out -- against the oracle's scaleColors + pipeline + getScanline, two lanes"""
    w, h, filt, b, black = 520, 392, synth.FILTERS_RGGB, 4, 64.0
    frames = [np.clip(synth.bayer_frame(w, h, filt, seed=40 + k, noise=1500), 0, 65535).astype(np.uint16) for k in range(3)]
    names = []
    for k, f in enumerate(frames):
        n = tmp_path / f"f{k}.u16"
        f.astype("<u2").tofile(n)
        names.append(str(n))
    res = subprocess.run([CLI, "--batch", ",".join(names), "--width", str(w), "--height", str(h), "--lanes", "2", "--black", str(black),
                          "--denoise", "30,12", "--expcomp", "0.3", "--out", str(tmp_path / "o")], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr
    info = json.loads(res.stdout.strip().splitlines()[-1])
    assert info["frames"] == 3 and (info["out_width"], info["out_height"]) == (w - 2 * b, h - 2 * b)
    for k, f in enumerate(frames):
        raw = np.maximum(f.astype(np.float32) - np.float32(black), np.float32(0.0))          # scaleColors with scale_mul 1
        ref = oracle_pipeline(raw, filt, "amaze", b, denoise=(30, 12), expcomp=0.3, dct=True)
        want = O.get_scanlines(ref, 16, False)
        got = read_ppm16(tmp_path / f"o.{k}.ppm")
        d = np.abs(got.astype(np.int32) - want.astype(np.int32))
        # the DCT detail-recovery stage is the path's one tolerance: a count of 65535 at most
        assert d.max() <= 1 and (d > 0).mean() < 0.02, (k, int(d.max()), float((d > 0).mean()))
    r0 = np.maximum(frames[0].astype(np.float32) - np.float32(black), np.float32(0.0))
    assert info["chmax0"] == [float(r0[0::2, 0::2].max()), float(max(r0[0::2, 1::2].max(), r0[1::2, 0::2].max())), float(r0[1::2, 1::2].max())]     # RGGB
