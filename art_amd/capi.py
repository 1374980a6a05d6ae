"""ctypes binding of ``libartgpu.so`` (the C ABI in ``include/artgpu.h``).

This is plumbing for the tests and ``bench.py``: device memory comes from torch (ROCm),
pointers and sizes go straight through the C ABI, the same entry points a C++ adapter inside
ART would call.  No CPU fallback: if the library is missing, import fails.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ARTGPU_LIB", os.path.join(_HERE, "libartgpu.so"))  # ARTGPU_LIB: kernel-variant experiments

BAYER_AMAZE = 0
BAYER_RCD = 1
BAYER_VNG4 = 2
DUAL_BILINEAR = 0
DUAL_VNG4 = 1


class Plane(C.Structure):
    _fields_ = [("p", C.c_void_p), ("w", C.c_int32), ("h", C.c_int32),
                ("row_stride_bytes", C.c_int64), ("on_device", C.c_int32)]


class RGB(C.Structure):
    _fields_ = [("r", Plane), ("g", Plane), ("b", Plane)]


class Timings(C.Structure):
    _fields_ = [("demosaic_ms", C.c_float), ("border_ms", C.c_float), ("total_ms", C.c_float)]


class DenoiseParams(C.Structure):
    _fields_ = [("luminance", C.c_double), ("luminance_detail", C.c_double), ("luminance_detail_threshold", C.c_int32),
                ("chrominance", C.c_double), ("chrominance_red_green", C.c_double), ("chrominance_blue_yellow", C.c_double),
                ("gamma", C.c_double), ("aggressive", C.c_int32), ("color_space", C.c_int32), ("chrominance_method", C.c_int32)]


class LogEncParams(C.Structure):
    _fields_ = [("enabled", C.c_int32), ("regularization", C.c_int32), ("satcontrol", C.c_int32), ("highlight_compression", C.c_int32),
                ("gain", C.c_double), ("target_gray", C.c_double), ("black_ev", C.c_double), ("white_ev", C.c_double)]


class DenoiseInfoStore(C.Structure):
    """DenoiseInfoStore (improcfun.h:117-131) + per-crop diagnostics"""
    _fields_ = [("valid", C.c_int32), ("ch_M", C.c_float * 9), ("max_r", C.c_float * 9), ("max_b", C.c_float * 9),
                ("chrominance", C.c_double), ("chrominance_red_green", C.c_double), ("chrominance_blue_yellow", C.c_double),
                ("crop_info", (C.c_float * 16) * 9)]


class NeutralState(C.Structure):
    _fields_ = [("ws", C.c_double * 9), ("iws", C.c_double * 9), ("to_out", C.c_float * 9), ("to_work", C.c_float * 9)]


class PipelineParams(C.Structure):
    pass


class DenoiseToolParams(C.Structure):
    _fields_ = [("dn", DenoiseParams), ("smoothing_enabled", C.c_int32), ("guided_chroma_radius", C.c_int32),
                ("nl_strength", C.c_int32), ("nl_detail", C.c_int32)]


class DenoiseFusion(C.Structure):
    _fields_ = [("demosaiced", C.POINTER(RGB)), ("sx1", C.c_int32), ("sy1", C.c_int32), ("mul", C.c_float * 3), ("do_clip", C.c_int32),
                ("cam_to_work", C.POINTER(C.c_double)), ("exposure_enabled", C.c_int32), ("exp_scale", C.c_float), ("black", C.c_float)]


PipelineParams._fields_ = [
    ("sensor", C.c_int32), ("bayer_method", C.c_int32), ("filters", C.c_uint32), ("initial_gain", C.c_double),
    ("xtrans_passes", C.c_int32), ("xtrans", C.c_int32 * 36), ("rgb_cam", C.c_float * 12), ("border", C.c_int32),
    ("mul", C.c_float * 3), ("do_clip", C.c_int32), ("has_cam_to_work", C.c_int32), ("cam_to_work", C.c_double * 9),
    ("ws", C.c_double * 9), ("iws", C.c_double * 9), ("denoise_enabled", C.c_int32), ("denoise", DenoiseToolParams),
    ("exposure_enabled", C.c_int32), ("expcomp", C.c_double), ("black", C.c_double), ("tone_enabled", C.c_int32),
    ("tone_mode", C.c_int32), ("tone_lut", C.POINTER(C.c_float)), ("white_point", C.c_float), ("to_out", C.c_float * 9),
    ("to_work", C.c_float * 9), ("scale", C.c_double), ("chrominance_auto_factor", C.c_double)]



class SensorFrame(C.Structure):
    """artgpu_sensor_frame: a frame as the decoder hands it over (uint16 / float sensor data + scaleColors' constants)"""
    _fields_ = [("data", C.c_void_p), ("w", C.c_int32), ("h", C.c_int32), ("row_stride_bytes", C.c_int64), ("is_u16", C.c_int32),
                ("on_device", C.c_int32), ("cblacksom", C.c_float * 4), ("scale_mul", C.c_float * 4)]


class ScanlineFrame(C.Structure):
    """artgpu_scanline_frame: a frame as the writers take it (rgb2out matrix path + getScanline)"""
    _fields_ = [("scanlines", C.c_void_p), ("row_stride_bytes", C.c_int64), ("bps", C.c_int32), ("is_float", C.c_int32),
                ("on_device", C.c_int32), ("rgb2out_enabled", C.c_int32), ("out_matrix", C.c_float * 9), ("trc_linear", C.c_int32),
                ("trc_lut", C.POINTER(C.c_float)), ("trc_lutsz", C.c_int32), ("chmax", C.c_float * 4), ("status", C.c_int32)]


DN_SKIP_DETAIL_RECOVERY = 1
# the chroma noise curve ImProcFunctions::denoise always installs (ipdenoise.cc:1139-1149)
DEFAULT_NOISE_C_CURVE_POINTS = (1.0, 0.05, 0.50, 0.35, 0.35, 0.35, 0.05, 0.35, 0.35)


def noise_curve_lut(points=DEFAULT_NOISE_C_CURVE_POINTS):
    """NoiseCurve::Set -> (501-entry float32 LUT, sum)."""
    pts = (C.c_double * len(points))(*[float(p) for p in points])
    lut = np.zeros(501, np.float32)
    s = C.c_float(0)
    rc = LIB.artgpu_noise_curve_lut(pts, len(points), lut.ctypes.data_as(C.POINTER(C.c_float)), C.byref(s))
    if rc:
        raise ArtGpuError(f"artgpu_noise_curve_lut: error {rc}")
    return lut, float(s.value)



class ArtGpuError(RuntimeError):
    pass


PROGRESS_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p, C.c_double)


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback for the device path)")
    # When this process also uses torch (device buffers, torch.distributed), torch must load its bundled HIP/HSA runtime
    # FIRST: loading libartgpu.so (linked against /opt/rocm's libamdhip64) before torch leaves two runtimes in the process
    # and hipGetDeviceCount fails.  Stand-alone users of the library (artgpu-cli, a C++ host) are unaffected.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    lib.artgpu_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    lib.artgpu_destroy.argtypes = [C.c_void_p]
    lib.artgpu_last_error.argtypes = [C.c_void_p]
    lib.artgpu_last_error.restype = C.c_char_p
    lib.artgpu_version.restype = C.c_char_p
    lib.artgpu_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    lib.artgpu_synchronize.argtypes = [C.c_void_p]
    lib.artgpu_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_long]
    lib.artgpu_get_option.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_long)]
    lib.artgpu_set_curve_tail.argtypes = [C.c_void_p, C.c_int, C.c_double]
    lib.artgpu_set_curve_tail_parametric.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
    lib.artgpu_enable_timing.argtypes = [C.c_void_p, C.c_int]
    lib.artgpu_get_timings.argtypes = [C.c_void_p, C.POINTER(Timings)]
    lib.artgpu_scratch_bytes.argtypes = [C.c_void_p]
    lib.artgpu_scratch_bytes.restype = C.c_size_t
    lib.artgpu_trim_scratch.argtypes = [C.c_void_p]
    lib.artgpu_demosaic_bayer.argtypes = [C.c_void_p, C.c_int, C.POINTER(Plane), C.c_uint32, C.c_double, C.c_int, C.POINTER(RGB)]
    lib.artgpu_border_interpolate2.argtypes = [C.c_void_p, C.POINTER(Plane), C.c_uint32, C.c_int, C.POINTER(RGB)]
    lib.artgpu_wavelet_decompose.argtypes = [C.c_void_p, C.POINTER(Plane), C.c_int, C.POINTER(C.c_void_p)]
    lib.artgpu_wavelet_mad.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
    lib.artgpu_wavelet_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.artgpu_wavelet_get_band.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
    lib.artgpu_wavelet_set_band.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
    lib.artgpu_wavelet_reconstruct.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Plane), C.c_float]
    lib.artgpu_wavelet_free.argtypes = [C.c_void_p, C.c_void_p]
    lib.artgpu_rgb_denoise.argtypes = [C.c_void_p, C.POINTER(RGB), C.POINTER(DenoiseParams), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_double,
                                       C.c_double, C.POINTER(Plane), C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.artgpu_denoise_guided_smoothing.argtypes = [C.c_void_p, C.POINTER(RGB), C.POINTER(C.c_double), C.c_int, C.c_double]
    lib.artgpu_gaussian_blur.argtypes = [C.c_void_p, C.POINTER(Plane), C.c_double]
    lib.artgpu_detail_mask.argtypes = [C.c_void_p, C.POINTER(Plane), C.POINTER(Plane), C.c_float, C.c_float, C.c_float, C.c_float, C.c_float]
    lib.artgpu_nlmeans.argtypes = [C.c_void_p, C.POINTER(Plane), C.c_float, C.c_int, C.c_int, C.c_float]
    lib.artgpu_improc_denoise.argtypes = [C.c_void_p, C.POINTER(RGB), C.POINTER(DenoiseToolParams), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double,
                                          C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_float), C.c_uint32]
    lib.artgpu_improc_denoise_fused.argtypes = [C.c_void_p, C.POINTER(RGB), C.POINTER(DenoiseFusion), C.POINTER(DenoiseToolParams), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                                C.c_double, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_float), C.c_uint32]
    lib.artgpu_scale_colors.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_uint32,
                                        C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(Plane), C.POINTER(C.c_float)]
    lib.artgpu_denoise_compute_params.argtypes = [C.c_void_p, C.POINTER(RGB), C.c_int, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_double),
                                                  C.POINTER(C.c_double), C.c_double, C.POINTER(DenoiseInfoStore), C.POINTER(DenoiseParams)]
    lib.artgpu_ordered_sum_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_float)]
    lib.artgpu_eval_primitive.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float,
                                          C.POINTER(C.c_float), C.c_int]
    lib.artgpu_saturation_vibrance.argtypes = [C.c_void_p, C.POINTER(RGB), C.c_int, C.c_int, C.POINTER(C.c_double)]
    lib.artgpu_set_batch_lanes.argtypes = [C.c_void_p, C.c_int]
    lib.artgpu_set_progress_callback.argtypes = [C.c_void_p, PROGRESS_FN, C.c_void_p]
    lib.artgpu_batch_complete.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.artgpu_rgb2out_matrix.argtypes = [C.c_void_p, C.POINTER(RGB), C.POINTER(RGB), C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float), C.c_int]
    lib.artgpu_get_scanlines.argtypes = [C.c_void_p, C.POINTER(RGB), C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_int]
    lib.artgpu_guided_filter.argtypes = [C.c_void_p, C.POINTER(Plane), C.POINTER(Plane), C.POINTER(Plane), C.c_int, C.c_float, C.c_int]
    lib.artgpu_hsl_equalizer.argtypes = [C.c_void_p, C.POINTER(RGB), C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.c_int,
                                         C.c_int, C.POINTER(C.c_double), C.c_double, C.c_int]
    lib.artgpu_log_encoding.argtypes = [C.c_void_p, C.POINTER(RGB), C.POINTER(LogEncParams), C.POINTER(C.c_double), C.c_int, C.c_int]
    lib.artgpu_rgb_to_lab.argtypes = [C.c_void_p, C.POINTER(RGB), C.POINTER(C.c_double)]
    lib.artgpu_lab_to_rgb.argtypes = [C.c_void_p, C.POINTER(RGB), C.POINTER(C.c_double)]
    lib.artgpu_lab_histogram.argtypes = [C.c_void_p, C.POINTER(RGB), C.POINTER(C.c_uint32)]
    lib.artgpu_lab_adjustments.argtypes = [C.c_void_p, C.POINTER(RGB), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float]
    lib.artgpu_dual_demosaic_bayer.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(Plane), C.c_uint32, C.c_double, C.c_int, C.POINTER(C.c_double), C.c_int, C.POINTER(RGB)]
    lib.artgpu_channel_mixer.argtypes = [C.c_void_p, C.POINTER(RGB), C.POINTER(C.c_float)]
    lib.artgpu_rgb_curves.argtypes = [C.c_void_p, C.POINTER(RGB), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.artgpu_pipeline_run.argtypes = [C.c_void_p, C.POINTER(Plane), C.POINTER(PipelineParams), C.POINTER(RGB)]
    lib.artgpu_batch_run.argtypes = [C.c_void_p, C.c_int, C.POINTER(Plane), C.POINTER(PipelineParams), C.POINTER(RGB)]
    lib.artgpu_batch_run_io.argtypes = [C.c_void_p, C.c_int, C.POINTER(SensorFrame), C.POINTER(PipelineParams), C.POINTER(ScanlineFrame)]
    lib.artgpu_demosaic_xtrans.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(Plane), C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(RGB)]
    lib.artgpu_tone_curve_neutral.argtypes = [C.c_void_p, C.POINTER(RGB), C.POINTER(C.c_float), C.c_float, C.POINTER(NeutralState)]
    lib.artgpu_noise_curve_lut.argtypes = [C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.artgpu_denoise_chroma_map.argtypes = [C.c_void_p, C.POINTER(RGB), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_float),
                                              C.POINTER(Plane)]
    lib.artgpu_get_image_skip.argtypes = [C.c_void_p, C.POINTER(RGB), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int,
                                          C.POINTER(C.c_double), C.POINTER(RGB)]
    lib.artgpu_get_image.argtypes = [C.c_void_p, C.POINTER(RGB), C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int,
                                     C.POINTER(C.c_double), C.POINTER(RGB)]
    lib.artgpu_convert_color_space.argtypes = [C.c_void_p, C.POINTER(RGB), C.POINTER(C.c_double)]
    lib.artgpu_exposure.argtypes = [C.c_void_p, C.POINTER(RGB), C.c_float, C.c_float]
    lib.artgpu_tone_curve.argtypes = [C.c_void_p, C.POINTER(RGB), C.c_int, C.POINTER(C.c_float), C.c_float, C.c_int]
    return lib


LIB = _load()

# artgpu_eval_primitive ids (include/artgpu.h: ARTGPU_PRIM_*)
PRIM_XEXPF_S = 0
PRIM_XEXPF_V = 1
PRIM_XEXPF_VN = 2
PRIM_XEXPF_V_LDEXP = 3
PRIM_XLOGF_S = 4
PRIM_XLOGF_V = 5
PRIM_XLOGF_VN = 6
PRIM_POW_F = 7
PRIM_XLIN2LOG = 8
PRIM_XLOG2LIN = 9
PRIM_XCBRTF = 10
PRIM_XATAN2F = 11
PRIM_XSINCOSF = 12
PRIM_LUTF_SCALAR = 13
PRIM_LUTF_VECTOR = 14
PRIM_MEDIAN3 = 15
PRIM_VMINF = 16
PRIM_VMAXF = 17
PRIM_VINTPF = 18
PRIM_XDIV2F = 19
PRIM_XDIVF2 = 20
PRIM_XLOG_D = 21
PRIM_XEXP_D = 22
PRIM_FLOAT_TO_HALF = 23     # out: float32-sized words, the half in the low 16 bits (eval_primitive returns them as uint32)

EXPORTS = ["artgpu_eval_primitive", "artgpu_set_progress_callback", "artgpu_set_option", "artgpu_get_option", "artgpu_set_curve_tail", "artgpu_set_curve_tail_parametric", "artgpu_create", "artgpu_destroy", "artgpu_last_error", "artgpu_version", "artgpu_set_stream",
           "artgpu_synchronize", "artgpu_enable_timing", "artgpu_get_timings", "artgpu_scratch_bytes", "artgpu_trim_scratch",
           "artgpu_demosaic_bayer", "artgpu_border_interpolate2", "artgpu_get_image",
           "artgpu_convert_color_space", "artgpu_exposure", "artgpu_tone_curve",
           "artgpu_wavelet_decompose", "artgpu_wavelet_mad", "artgpu_wavelet_info", "artgpu_wavelet_get_band", "artgpu_wavelet_set_band",
           "artgpu_wavelet_reconstruct", "artgpu_wavelet_free", "artgpu_rgb_denoise", "artgpu_denoise_guided_smoothing",
           "artgpu_gaussian_blur", "artgpu_detail_mask", "artgpu_nlmeans", "artgpu_improc_denoise", "artgpu_improc_denoise_fused", "artgpu_noise_curve_lut", "artgpu_denoise_chroma_map", "artgpu_tone_curve_neutral", "artgpu_demosaic_xtrans", "artgpu_pipeline_run", "artgpu_batch_run", "artgpu_batch_run_io", "artgpu_scale_colors", "artgpu_channel_mixer", "artgpu_rgb_curves", "artgpu_denoise_compute_params", "artgpu_ordered_sum_f32", "artgpu_get_image_skip", "artgpu_saturation_vibrance", "artgpu_set_batch_lanes", "artgpu_batch_complete", "artgpu_rgb2out_matrix", "artgpu_get_scanlines", "artgpu_guided_filter", "artgpu_hsl_equalizer", "artgpu_log_encoding", "artgpu_rgb_to_lab", "artgpu_lab_to_rgb", "artgpu_lab_histogram", "artgpu_lab_adjustments", "artgpu_dual_demosaic_bayer"]


def host_plane(a: np.ndarray) -> Plane:
    assert a.dtype == np.float32 and a.ndim == 2 and a.strides[1] == 4
    return Plane(a.ctypes.data, a.shape[1], a.shape[0], a.strides[0], 0)


def sensor_frame(a: np.ndarray, cblacksom=(0.0, 0.0, 0.0, 0.0), scale_mul=(1.0, 1.0, 1.0, 1.0)) -> SensorFrame:
    assert a.dtype in (np.uint16, np.float32) and a.ndim == 2 and a.strides[1] == a.itemsize
    return SensorFrame(a.ctypes.data, a.shape[1], a.shape[0], a.strides[0], 1 if a.dtype == np.uint16 else 0, 0,
                       (C.c_float * 4)(*[float(v) for v in cblacksom]), (C.c_float * 4)(*[float(v) for v in scale_mul]))


def scanline_frame(a: np.ndarray, matrix=None, trc_lut=None, is_float: bool = False) -> ScanlineFrame:
    """`a`: (h, w, 3) uint8 / uint16 / float32 host array the scanlines land in (uint16 + is_float: half floats).  matrix: rgb2out's
    3 x 3 (None: no rgb2out); trc_lut: float32 array the caller keeps alive until the batch call has returned (None: linear TRC)."""
    assert a.ndim == 3 and a.shape[2] == 3 and a.strides[2] == a.itemsize and a.strides[1] == 3 * a.itemsize
    f = ScanlineFrame()
    f.scanlines = a.ctypes.data; f.row_stride_bytes = a.strides[0]; f.bps = 8 * a.itemsize
    f.is_float = 1 if (is_float or a.dtype == np.float32) else 0
    f.on_device = 0
    f.rgb2out_enabled = 0 if matrix is None else 1
    if matrix is not None:
        f.out_matrix[:] = [float(v) for v in np.asarray(matrix, np.float32).reshape(9)]
    f.trc_linear = 1 if trc_lut is None else 0
    if trc_lut is not None:
        assert trc_lut.dtype == np.float32 and trc_lut.flags.c_contiguous
        f.trc_lut = trc_lut.ctypes.data_as(C.POINTER(C.c_float)); f.trc_lutsz = trc_lut.size
    return f


def host_rgb(planes) -> RGB:
    return RGB(*[host_plane(p) for p in planes])


def device_plane(t) -> Plane:
    """`t`: a 2-D float32 torch tensor on the context's device with unit column stride."""
    assert t.dim() == 2 and t.stride(1) == 1 and t.element_size() == 4
    return Plane(t.data_ptr(), t.shape[1], t.shape[0], t.stride(0) * 4, 1)


class Context:
    """One device context (include/artgpu.h: artgpu_create / artgpu_destroy)."""

    def __init__(self, device: int = 0, stream=None):
        self._h = C.c_void_p()
        rc = LIB.artgpu_create(device, C.byref(self._h))
        if rc != 0:
            raise ArtGpuError(f"artgpu_create(device={device}) failed: {rc}")
        if stream is not None:
            self.set_stream(stream)

    def _chk(self, rc):
        if rc != 0:
            raise ArtGpuError(f"[{rc}] {LIB.artgpu_last_error(self._h).decode()}")

    def set_stream(self, stream_handle: int):
        self._chk(LIB.artgpu_set_stream(self._h, C.c_void_p(stream_handle)))

    def synchronize(self):
        self._chk(LIB.artgpu_synchronize(self._h))

    def set_option(self, name: str, value: int):
        self._chk(LIB.artgpu_set_option(self._h, name.encode(), C.c_long(int(value))))

    def get_option(self, name: str) -> int:
        v = C.c_long(0)
        self._chk(LIB.artgpu_get_option(self._h, name.encode(), C.byref(v)))
        return int(v.value)

    def set_curve_tail(self, kind: int, y_last: float = 1.0):
        """0 LUT clip (no Curve object), 1 constant y_last, 2 identity, 3 host (default): curves::setLutVal above 65535"""
        self._chk(LIB.artgpu_set_curve_tail(self._h, int(kind), float(y_last)))

    def set_curve_tail_parametric(self, p):
        """DiagonalCurve's DCT_Parametric parameter vector (8 or 9 doubles, p[0] = the kind): getVal above 1.0 on the device"""
        arr = (C.c_double * len(p))(*[float(v) for v in p])
        self._chk(LIB.artgpu_set_curve_tail_parametric(self._h, arr, len(p)))

    def enable_timing(self, on: bool = True):
        self._chk(LIB.artgpu_enable_timing(self._h, int(on)))

    def timings(self) -> Timings:
        t = Timings()
        self._chk(LIB.artgpu_get_timings(self._h, C.byref(t)))
        return t

    def scratch_bytes(self) -> int:
        return LIB.artgpu_scratch_bytes(self._h)

    def trim_scratch(self):
        self._chk(LIB.artgpu_trim_scratch(self._h))

    def demosaic_bayer(self, method: int, raw: Plane, filters: int, initial_gain: float, border: int, out: RGB):
        self._chk(LIB.artgpu_demosaic_bayer(self._h, method, C.byref(raw), filters, initial_gain, border, C.byref(out)))

    def border_interpolate2(self, raw: Plane, filters: int, lborders: int, out: RGB):
        self._chk(LIB.artgpu_border_interpolate2(self._h, C.byref(raw), filters, lborders, C.byref(out)))

    def get_image(self, planes: RGB, sx1: int, sy1: int, mul, do_clip: bool, mat, image: RGB, skip: int = 1):
        m = (C.c_float * 3)(*[float(v) for v in mul])
        mp = None if mat is None else (C.c_double * 9)(*[float(v) for v in np.asarray(mat, dtype=np.float64).reshape(9)])
        if skip == 1:
            self._chk(LIB.artgpu_get_image(self._h, C.byref(planes), sx1, sy1, m, int(do_clip), mp, C.byref(image)))
        else:
            self._chk(LIB.artgpu_get_image_skip(self._h, C.byref(planes), sx1, sy1, int(skip), m, int(do_clip), mp, C.byref(image)))

    def convert_color_space(self, image: RGB, mat):
        mp = (C.c_double * 9)(*[float(v) for v in np.asarray(mat, dtype=np.float64).reshape(9)])
        self._chk(LIB.artgpu_convert_color_space(self._h, C.byref(image), mp))

    def exposure(self, image: RGB, exp_scale: float, black: float):
        self._chk(LIB.artgpu_exposure(self._h, C.byref(image), exp_scale, black))

    def tone_curve(self, image: RGB, lut, whitept: float = 1.0, filmlike_clip: bool = True, mode: int = 0):
        lp = None
        if lut is not None:
            lut = np.ascontiguousarray(lut, dtype=np.float32)
            assert lut.shape == (65536,)
            lp = lut.ctypes.data_as(C.POINTER(C.c_float))
        self._chk(LIB.artgpu_tone_curve(self._h, C.byref(image), mode, lp, whitept, int(filmlike_clip)))

    def rgb_denoise(self, image: RGB, params: DenoiseParams, ws, expcomp: float = 0.0, scale: float = 1.0,
                    ccalc: Plane = None, flags: int = DN_SKIP_DETAIL_RECOVERY, want_resid: bool = False, iws=None):
        wsf = (C.c_float * 9)(*[float(v) for v in np.asarray(ws, dtype=np.float32).reshape(9)])
        iwsf = None if iws is None else (C.c_float * 9)(*[float(v) for v in np.asarray(iws, dtype=np.float32).reshape(9)])
        nresi, highresi = C.c_float(0), C.c_float(0)
        self._chk(LIB.artgpu_rgb_denoise(self._h, C.byref(image), C.byref(params), wsf, iwsf, expcomp, scale,
                                         None if ccalc is None else C.byref(ccalc), flags,
                                         C.byref(nresi) if want_resid else None, C.byref(highresi) if want_resid else None))
        if want_resid:
            return float(nresi.value), float(highresi.value)

    def denoise_guided_smoothing(self, image: RGB, ws, radius: int = 3, scale: float = 1.0):
        m = (C.c_double * 9)(*[float(v) for v in np.asarray(ws, dtype=np.float64).reshape(9)])
        self._chk(LIB.artgpu_denoise_guided_smoothing(self._h, C.byref(image), m, radius, scale))

    def gaussian_blur(self, img: Plane, sigma: float):
        self._chk(LIB.artgpu_gaussian_blur(self._h, C.byref(img), sigma))

    def detail_mask(self, src: Plane, mask: Plane, scaling, threshold, ceiling, factor, blur):
        self._chk(LIB.artgpu_detail_mask(self._h, C.byref(src), C.byref(mask), scaling, threshold, ceiling, factor, blur))

    def nlmeans(self, img: Plane, strength: int = 50, detail: int = 80, scale: float = 1.0, normcoeff: float = 65535.0):
        self._chk(LIB.artgpu_nlmeans(self._h, C.byref(img), normcoeff, strength, detail, scale))

    def scale_colors(self, src: np.ndarray, filters: int, xtrans, cblacksom, scale_mul, dst: Plane):
        """host uint16 / float32 sensor data -> scaled float CFA plane; returns chmax[4]"""
        assert src.dtype in (np.uint16, np.float32) and src.flags.c_contiguous
        h, w = src.shape
        xt = None if xtrans is None else np.ascontiguousarray(xtrans, dtype=np.int32).reshape(36).ctypes.data_as(C.POINTER(C.c_int32))
        cb = (C.c_float * 4)(*[float(v) for v in cblacksom]); sm = (C.c_float * 4)(*[float(v) for v in scale_mul])
        mx = (C.c_float * 4)()
        self._chk(LIB.artgpu_scale_colors(self._h, src.ctypes.data, w, h, src.strides[0], 1 if src.dtype == np.uint16 else 0, 0, filters, xt,
                                          cb, sm, C.byref(dst), mx))
        return [float(v) for v in mx]

    def denoise_compute_params(self, planes: RGB, border: int, mul, do_clip: bool, cam_to_work, ws, dn: DenoiseParams,
                               auto_factor: float = 1.0, store: "DenoiseInfoStore" = None) -> "DenoiseInfoStore":
        """ImProcFunctions::denoiseComputeParams; updates dn.chrominance* in place and returns the store"""
        store = store if store is not None else DenoiseInfoStore()
        d9 = lambda m: (C.c_double * 9)(*[float(v) for v in np.asarray(m, dtype=np.float64).reshape(9)])
        self._chk(LIB.artgpu_denoise_compute_params(self._h, C.byref(planes), int(border), (C.c_float * 3)(*[float(v) for v in mul]),
                                                    1 if do_clip else 0, d9(cam_to_work), d9(ws), float(auto_factor), C.byref(store), C.byref(dn)))
        return store

    def ordered_sum_f32(self, x) -> np.float32:
        """acc = 0; for v in x: acc += v  (fp32), x a numpy array (host) or a CUDA torch tensor"""
        r = C.c_float(0)
        if isinstance(x, np.ndarray):
            a = np.ascontiguousarray(x, dtype=np.float32)
            self._chk(LIB.artgpu_ordered_sum_f32(self._h, a.ctypes.data, a.size, 0, C.byref(r)))
        else:
            assert x.is_contiguous() and x.dtype.itemsize == 4
            self._chk(LIB.artgpu_ordered_sum_f32(self._h, x.data_ptr(), x.numel(), 1, C.byref(r)))
        return np.float32(r.value)

    def eval_primitive(self, prim: int, a, b=None, c=None, param: float = 0.0, table=None):
        """out0[i] = f(a[i] [, b[i] [, c[i]]]) with the device's own math primitive `prim` (PRIM_*); host numpy arrays in and out.
        PRIM_XSINCOSF returns (sin, cos); PRIM_XLOG_D / PRIM_XEXP_D work on float64."""
        dt = np.float64 if prim in (PRIM_XLOG_D, PRIM_XEXP_D) else np.float32
        arrs = [None if x is None else np.ascontiguousarray(x, dtype=dt) for x in (a, b, c)]
        n = arrs[0].size
        out0, out1 = np.empty(n, dt), np.empty(n, dt)
        tab = None if table is None else np.ascontiguousarray(table, dtype=np.float32)
        ptr = lambda x: None if x is None else x.ctypes.data_as(C.c_void_p)
        self._chk(LIB.artgpu_eval_primitive(self._h, int(prim), ptr(arrs[0]), ptr(arrs[1]), ptr(arrs[2]), ptr(out0), ptr(out1), C.c_int64(n),
                                            C.c_float(param), None if tab is None else tab.ctypes.data_as(C.POINTER(C.c_float)),
                                            0 if tab is None else tab.size))
        if prim == PRIM_FLOAT_TO_HALF:
            return out0.view(np.uint32)
        return (out0, out1) if prim == PRIM_XSINCOSF else out0

    def saturation_vibrance(self, image: RGB, saturation: int, vibrance: int, ws):
        self._chk(LIB.artgpu_saturation_vibrance(self._h, C.byref(image), int(saturation), int(vibrance),
                                                 (C.c_double * 9)(*[float(v) for v in np.asarray(ws, dtype=np.float64).reshape(9)])))

    def rgb2out_matrix(self, src: RGB, dst: RGB, matrix, trc_linear: bool, lut=None):
        m = (C.c_float * 9)(*[float(v) for v in np.asarray(matrix, np.float32).reshape(9)])
        la = None if lut is None else np.ascontiguousarray(lut, dtype=np.float32)
        self._chk(LIB.artgpu_rgb2out_matrix(self._h, C.byref(src), C.byref(dst), m, 1 if trc_linear else 0,
                                            None if la is None else la.ctypes.data_as(C.POINTER(C.c_float)), 0 if la is None else la.size))

    def get_scanlines(self, image: RGB, bps: int, is_float: bool = False) -> np.ndarray:
        """Imagefloat::getScanline for every row -> (h, w, 3) array of uint8 / uint16 / float16-bits (uint16) / float32"""
        dt = np.float32 if (is_float and bps == 32) else (np.uint8 if bps == 8 else np.uint16)
        out = np.zeros((image.r.h, image.r.w, 3), dt)
        self._chk(LIB.artgpu_get_scanlines(self._h, C.byref(image), bps, 1 if is_float else 0, out.ctypes.data, out.strides[0], 0))
        return out

    def guided_filter(self, guide: Plane, src: Plane, dst: Plane, r: int, epsilon: float, subsampling: int = 0):
        self._chk(LIB.artgpu_guided_filter(self._h, C.byref(guide), C.byref(src), C.byref(dst), int(r), float(epsilon), int(subsampling)))

    def hsl_equalizer(self, image: RGB, hcurve, scurve, lcurve, smoothing: int, ws, scale: float = 1.0, to_rgb: bool = True):
        def arr(c):
            if c is None:
                return None, 0
            a = (C.c_double * len(c))(*[float(v) for v in c])
            return a, len(c)
        (h, nh), (s, ns), (l, nl) = arr(hcurve), arr(scurve), arr(lcurve)
        self._chk(LIB.artgpu_hsl_equalizer(self._h, C.byref(image), h, nh, s, ns, l, nl, int(smoothing),
                                           (C.c_double * 9)(*[float(v) for v in np.asarray(ws, dtype=np.float64).reshape(9)]), float(scale), 1 if to_rgb else 0))

    def log_encoding(self, image: RGB, ws, gain=0.0, target_gray=18.0, black_ev=-13.5, white_ev=2.5, regularization=60, satcontrol=True,
                     highlight_compression=0, full_width=0, full_height=0, enabled=True):
        p = LogEncParams(1 if enabled else 0, int(regularization), 1 if satcontrol else 0, int(highlight_compression),
                         float(gain), float(target_gray), float(black_ev), float(white_ev))
        self._chk(LIB.artgpu_log_encoding(self._h, C.byref(image), C.byref(p), (C.c_double * 9)(*[float(v) for v in np.asarray(ws, dtype=np.float64).reshape(9)]),
                                          int(full_width), int(full_height)))

    def rgb_to_lab(self, image: RGB, ws):
        self._chk(LIB.artgpu_rgb_to_lab(self._h, C.byref(image), (C.c_double * 9)(*[float(v) for v in np.asarray(ws, dtype=np.float64).reshape(9)])))

    def lab_to_rgb(self, image: RGB, iws):
        self._chk(LIB.artgpu_lab_to_rgb(self._h, C.byref(image), (C.c_double * 9)(*[float(v) for v in np.asarray(iws, dtype=np.float64).reshape(9)])))

    def lab_histogram(self, image: RGB):
        hist = np.zeros(65536, np.uint32)
        self._chk(LIB.artgpu_lab_histogram(self._h, C.byref(image), hist.ctypes.data_as(C.POINTER(C.c_uint32))))
        return hist

    def lab_adjustments(self, image: RGB, lcurve, acurve, bcurve, chroma: float):
        keep = [np.ascontiguousarray(c, dtype=np.float32) for c in (lcurve, acurve, bcurve)]
        assert keep[0].size == 32770 and keep[1].size == 65536 and keep[2].size == 65536
        self._chk(LIB.artgpu_lab_adjustments(self._h, C.byref(image), *[k.ctypes.data_as(C.POINTER(C.c_float)) for k in keep], float(chroma)))

    def dual_demosaic_bayer(self, method: int, raw: Plane, filters: int, initial_gain: float, border: int, contrast: float, auto_contrast: bool, out: RGB,
                            second: int = DUAL_BILINEAR):
        """returns the contrast threshold in percent (searched when auto_contrast)"""
        c = C.c_double(float(contrast))
        self._chk(LIB.artgpu_dual_demosaic_bayer(self._h, method, second, C.byref(raw), filters, float(initial_gain), border, C.byref(c), 1 if auto_contrast else 0, C.byref(out)))
        return c.value

    def channel_mixer(self, image: RGB, m):
        self._chk(LIB.artgpu_channel_mixer(self._h, C.byref(image), (C.c_float * 9)(*[float(v) for v in np.asarray(m, np.float32).reshape(9)])))

    def rgb_curves(self, image: RGB, rcurve=None, gcurve=None, bcurve=None):
        ptr = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float32).ctypes.data_as(C.POINTER(C.c_float))
        keep = [None if a is None else np.ascontiguousarray(a, dtype=np.float32) for a in (rcurve, gcurve, bcurve)]
        self._chk(LIB.artgpu_rgb_curves(self._h, C.byref(image), *[None if k is None else k.ctypes.data_as(C.POINTER(C.c_float)) for k in keep]))

    def set_progress_callback(self, fn):
        """fn(stage: str, fraction: float) or None"""
        self._progress = PROGRESS_FN(lambda user, stage, frac: fn(stage.decode(), frac)) if fn else PROGRESS_FN()
        self._chk(LIB.artgpu_set_progress_callback(self._h, self._progress, None))

    def set_batch_lanes(self, lanes: int):
        self._chk(LIB.artgpu_set_batch_lanes(self._h, int(lanes)))

    def batch_complete(self, record, nranks: int = 1, rccl_comm=None):
        """All-gather of the ranks' 8-word completion records over an RCCL communicator (None: single rank)."""
        rec = (C.c_int64 * 8)(*[int(v) for v in record])
        out = (C.c_int64 * (8 * nranks))()
        self._chk(LIB.artgpu_batch_complete(self._h, rccl_comm, int(nranks), rec, out))
        return [list(out[8 * r:8 * r + 8]) for r in range(nranks)]

    def pipeline_run(self, raw: Plane, params: PipelineParams, out: RGB):
        self._chk(LIB.artgpu_pipeline_run(self._h, C.byref(raw), C.byref(params), C.byref(out)))

    def batch_run(self, raws, params: PipelineParams, outs):
        n = len(raws)
        self._chk(LIB.artgpu_batch_run(self._h, n, (Plane * n)(*raws), C.byref(params), (RGB * n)(*outs)))

    def batch_run_io(self, sensor_frames, params: PipelineParams, scanline_frames):
        """artgpu_batch_run_io: sensor data in, writers' scanlines out, copies beside the kernels.  `sensor_frames`: numpy arrays (uint16 /
        float32, host) or ready SensorFrame structures; `scanline_frames`: ScanlineFrame structures (scanline_frame() builds one around a
        numpy array).  Returns the array of ScanlineFrame (chmax / status filled in)."""
        n = len(sensor_frames)
        ins = (SensorFrame * n)()
        for k, f in enumerate(sensor_frames):
            ins[k] = f if isinstance(f, SensorFrame) else sensor_frame(f)
        outs = (ScanlineFrame * n)(*scanline_frames)
        self._chk(LIB.artgpu_batch_run_io(self._h, n, ins, C.byref(params), outs))
        return outs

    def demosaic_xtrans(self, passes: int, use_cielab: bool, raw: Plane, xtrans, rgb_cam, out: RGB):
        xt = np.ascontiguousarray(xtrans, dtype=np.int32).reshape(36)
        cam = np.ascontiguousarray(rgb_cam, dtype=np.float32).reshape(12)
        self._chk(LIB.artgpu_demosaic_xtrans(self._h, passes, 1 if use_cielab else 0, C.byref(raw), xt.ctypes.data_as(C.POINTER(C.c_int32)),
                                             cam.ctypes.data_as(C.POINTER(C.c_float)), C.byref(out)))

    def tone_curve_neutral(self, image: RGB, lut65536, whitecoeff, ws, iws, to_out=None, to_work=None):
        lut = np.ascontiguousarray(lut65536, dtype=np.float32)
        assert lut.shape == (65536,)
        ident = np.eye(3, dtype=np.float32)
        st = NeutralState()
        st.ws[:] = [float(v) for v in np.asarray(ws, np.float64).reshape(9)]
        st.iws[:] = [float(v) for v in np.asarray(iws, np.float64).reshape(9)]
        st.to_out[:] = [float(v) for v in np.asarray(ident if to_out is None else to_out, np.float32).reshape(9)]
        st.to_work[:] = [float(v) for v in np.asarray(ident if to_work is None else to_work, np.float32).reshape(9)]
        self._chk(LIB.artgpu_tone_curve_neutral(self._h, C.byref(image), lut.ctypes.data_as(C.POINTER(C.c_float)), whitecoeff, C.byref(st)))

    def denoise_chroma_map(self, image: RGB, calclum_mat, ws, curve, ccalc: Plane):
        m = None if calclum_mat is None else (C.c_double * 9)(*[float(v) for v in np.asarray(calclum_mat, dtype=np.float64).reshape(9)])
        wsd = (C.c_double * 9)(*[float(v) for v in np.asarray(ws, dtype=np.float64).reshape(9)])
        cv = np.ascontiguousarray(curve, dtype=np.float32)
        assert cv.shape == (501,)
        self._chk(LIB.artgpu_denoise_chroma_map(self._h, C.byref(image), m, wsd, cv.ctypes.data_as(C.POINTER(C.c_float)), C.byref(ccalc)))

    def improc_denoise(self, image: RGB, params: DenoiseToolParams, ws, ecomp: float = 0.0, scale: float = 1.0, calclum_mat=None,
                       noise_c_curve=None, flags: int = 0, iws=None):
        wsd = (C.c_double * 9)(*[float(v) for v in np.asarray(ws, dtype=np.float64).reshape(9)])
        m = None if calclum_mat is None else (C.c_double * 9)(*[float(v) for v in np.asarray(calclum_mat, dtype=np.float64).reshape(9)])
        cv = None
        if noise_c_curve is not None:
            cva = np.ascontiguousarray(noise_c_curve, dtype=np.float32)
            assert cva.shape == (501,)
            cv = cva.ctypes.data_as(C.POINTER(C.c_float))
        iwsd = None if iws is None else (C.c_double * 9)(*[float(v) for v in np.asarray(iws, dtype=np.float64).reshape(9)])
        self._chk(LIB.artgpu_improc_denoise(self._h, C.byref(image), C.byref(params), wsd, iwsd, ecomp, scale, m, cv, flags))

    def improc_denoise_fused(self, image: RGB, params: DenoiseToolParams, ws, demosaiced: RGB = None, sx1: int = 0, sy1: int = 0, mul=(1.0, 1.0, 1.0),
                             do_clip: bool = True, cam_to_work=None, exposure=None, ecomp: float = 0.0, scale: float = 1.0, calclum_mat=None,
                             noise_c_curve=None, flags: int = 0, iws=None):
        """artgpu_improc_denoise_fused: getImage + convertColorSpace in front (demosaiced != None) and ImProcFunctions::exposure behind
        (exposure = (exp_scale, black)) inside the tool's own pixel passes"""
        fu = DenoiseFusion()
        keep = []
        if demosaiced is not None:
            keep.append(demosaiced)
            fu.demosaiced = C.pointer(demosaiced)
            fu.sx1, fu.sy1 = int(sx1), int(sy1)
            fu.mul[:] = [float(v) for v in mul]
            fu.do_clip = 1 if do_clip else 0
            if cam_to_work is not None:
                cm = (C.c_double * 9)(*[float(v) for v in np.asarray(cam_to_work, dtype=np.float64).reshape(9)])
                keep.append(cm)
                fu.cam_to_work = C.cast(cm, C.POINTER(C.c_double))
        if exposure is not None:
            fu.exposure_enabled = 1
            fu.exp_scale, fu.black = float(exposure[0]), float(exposure[1])
        wsd = (C.c_double * 9)(*[float(v) for v in np.asarray(ws, dtype=np.float64).reshape(9)])
        m = None if calclum_mat is None else (C.c_double * 9)(*[float(v) for v in np.asarray(calclum_mat, dtype=np.float64).reshape(9)])
        cv = None
        if noise_c_curve is not None:
            cva = np.ascontiguousarray(noise_c_curve, dtype=np.float32)
            assert cva.shape == (501,)
            cv = cva.ctypes.data_as(C.POINTER(C.c_float))
        iwsd = None if iws is None else (C.c_double * 9)(*[float(v) for v in np.asarray(iws, dtype=np.float64).reshape(9)])
        self._chk(LIB.artgpu_improc_denoise_fused(self._h, C.byref(image), C.byref(fu), C.byref(params), wsd, iwsd, C.c_double(ecomp), C.c_double(scale), m, cv, flags))

    # ---- wavelet_decomposition ----
    def wavelet_decompose(self, src: Plane, maxlvl: int):
        h = C.c_void_p()
        self._chk(LIB.artgpu_wavelet_decompose(self._h, C.byref(src), maxlvl, C.byref(h)))
        return h

    def wavelet_info(self, wv):
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        self._chk(LIB.artgpu_wavelet_info(wv, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def wavelet_get_band(self, wv, level: int, direction: int) -> np.ndarray:
        w2, h2, _ = self.wavelet_info(wv)
        out = np.empty((h2, w2), np.float32)
        self._chk(LIB.artgpu_wavelet_get_band(self._h, wv, level, direction, out.ctypes.data, 0))
        return out

    def wavelet_set_band(self, wv, level: int, direction: int, data: np.ndarray):
        data = np.ascontiguousarray(data, dtype=np.float32)
        self._chk(LIB.artgpu_wavelet_set_band(self._h, wv, level, direction, data.ctypes.data, 0))

    def wavelet_mad(self, wv) -> np.ndarray:
        """SQR(MadRgb) of every detail band: [3 * level + dir - 1]"""
        _, _, lv = self.wavelet_info(wv)
        out = (C.c_float * (3 * lv))()
        self._chk(LIB.artgpu_wavelet_mad(self._h, wv, out))
        return np.array(out[:], dtype=np.float32)

    def wavelet_reconstruct(self, wv, dst: Plane, blend: float = 1.0):
        self._chk(LIB.artgpu_wavelet_reconstruct(self._h, wv, C.byref(dst), blend))

    def wavelet_free(self, wv):
        LIB.artgpu_wavelet_free(self._h, wv)

    # convenience for tests: host numpy in, host numpy out (staged through the library)
    def demosaic_bayer_host(self, method: int, raw: np.ndarray, filters: int, initial_gain: float = 1.0, border: int = 4):
        raw = np.ascontiguousarray(raw, dtype=np.float32)
        h, w = raw.shape
        planes = [np.full((h, w), np.nan, dtype=np.float32) for _ in range(3)]
        out = RGB(host_plane(planes[0]), host_plane(planes[1]), host_plane(planes[2]))
        self.demosaic_bayer(method, host_plane(raw), filters, initial_gain, border, out)
        return planes

    def close(self):
        if self._h:
            LIB.artgpu_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
