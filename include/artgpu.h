/*
 * include/artgpu.h -- C ABI of libartgpu.so: the MI355X (gfx950) implementation of ART's
 * raw-development hot path.  Plain pointers and sizes only; no C++ or torch types.
 *
 * The reference (artpixls/ART) has no FFI or plugin seam: the boundary this ABI replaces is
 * the C++ member/free-function layer of rtengine.  Each entry point names the reference
 * function whose body it stands in for; INTEGRATION.md shows the adapter a maintainer adds.
 *
 * Conventions
 *   - every function returns 0 on success, a negative ARTGPU_E* code otherwise;
 *     artgpu_last_error(ctx) gives the message.  The reference's functions return void and
 *     cannot fail (allocation failures are compiled out, rtengine/FTblockDN.cc:859); an
 *     adapter falls through to the CPU code on a non-zero return.
 *   - images are caller-owned.  artgpu_plane.on_device != 0 means `p` is a device (HBM)
 *     pointer usable on ctx's device; otherwise it is host memory and the call stages it
 *     through the context's device buffers (H2D before, D2H after).
 *   - a context is bound to one HIP device and one stream; calls on one context are
 *     serialised by the caller (the reference runs one pipeline thread per ImageProcessor,
 *     rtengine/improccoordinator.cc:192).  Device-pointer calls are asynchronous on the
 *     context's stream; host-pointer calls return after the D2H copy has completed.
 *   - there is no CPU fallback inside the library.
 */
#ifndef ARTGPU_H
#define ARTGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ARTGPU_OK            0
#define ARTGPU_EINVAL       -1  /* bad argument */
#define ARTGPU_EHIP         -2  /* HIP runtime error (message has the hipError string) */
#define ARTGPU_ENOMEM       -3  /* device allocation failed */
#define ARTGPU_EUNSUPPORTED -4  /* valid in the reference, not implemented on the device path */

/* RAWParams::BayerSensor::Method values handled here (rtengine/procparams.h; dispatch at
 * rtengine/rawimagesource.cc:1862-1912). */
#define ARTGPU_BAYER_AMAZE 0
#define ARTGPU_BAYER_RCD   1
#define ARTGPU_BAYER_VNG4  2
/* second demosaicer of artgpu_dual_demosaic_bayer */
#define ARTGPU_DUAL_BILINEAR 0
#define ARTGPU_DUAL_VNG4     1

typedef struct artgpu_ctx artgpu_ctx;

/* One fp32 plane.  Mirrors array2D<float> (rtengine/array2D.h:74-296: row pointers into one
 * block, row stride W*4 bytes) and one channel of PlanarRGBData<float>
 * (rtengine/iimage.h:653-720: row stride = ceil16(W*4) bytes). */
typedef struct {
    float   *p;
    int32_t  w, h;
    int64_t  row_stride_bytes;
    int32_t  on_device;
} artgpu_plane;

typedef struct { artgpu_plane r, g, b; } artgpu_rgb;

/* Per-call device timings in milliseconds (hipEvents on the context's stream); filled only
 * when timing is enabled with artgpu_enable_timing(). */
typedef struct {
    float demosaic_ms;
    float border_ms;
    float total_ms;
} artgpu_timings;

int artgpu_create(int hip_device, artgpu_ctx **out);
int artgpu_destroy(artgpu_ctx *ctx);
const char *artgpu_last_error(const artgpu_ctx *ctx);
const char *artgpu_version(void);

/* Launch all work of this context on `hip_stream` (a hipStream_t; NULL = default stream).  Work the context has queued on its previous
 * stream is ordered before whatever it queues on the new one (its scratch memory and cached tables are shared between the two). */
int artgpu_set_stream(artgpu_ctx *ctx, void *hip_stream);
int artgpu_synchronize(artgpu_ctx *ctx);
/* Test / profiling switches of the context (they never change what a call computes, only which of two bit-identical device
 * paths runs or how scratch memory is prepared).  Unknown names return ARTGPU_EINVAL.
 *   "amaze_path"        0 (default): full AMaZE tiles are streamed through LDS (amaze_stream.hip), partial tiles and tiles the
 *                       stream hands back run on the per-tile arena kernel (amaze.hip); 1: arena kernel for every tile
 *   "amaze_split"       1 (with amaze_path 1): one kernel launch per AMaZE phase (per-phase profile)
 *   "amaze_zero_mask" / "amaze_zero_frame" / "amaze_poison"   arena-clearing experiments of tests/test_gpu_demosaic.py
 *   "amaze_overlap"     0: the arena kernel's tiles behind the stream kernel instead of beside it
 *   "amaze_grid"        n > 0: at most n stream workgroups (one per CU); 0 (default): every CU, or 5/8 of them while artgpu_batch_run has several
 *                       frames in flight (the other frames' memory-bound passes get the rest)
 *   "dn_fused"          1 (default): ShrinkAllL / ShrinkAllAB as one pass over the coefficients, one launch for the three channels where nothing has
 *                       to happen between them; 2: one launch per channel; 0: the three-kernel form (factors, row sums, column sums + update)
 *   "dn_streams"        0 (default since round 5): RGB_denoise's kernels one after the other on the context's stream; 1: the DCT detail recovery of L on a
 *                       side stream beside the reconstructions of a and b (the default of rounds 3 and 4); "lut_lds" 0: never the LUT-in-LDS shape
 *                       of the pixel passes; "rcd_rows" 4 | 8; "roctx" 1: roctx ranges named after the reference functions
 *   "io_direct"         artgpu_batch_run_io, scanlines that go to PINNED host memory: n > 0: n persistent workgroups on the download stream write
 *                       them there directly (no staging plane, no copy); 0: staged on the device and copied by the runtime; -1 (default): 8 with
 *                       two lanes, 0 otherwise (measured: DESIGN.md section 17); "cu_reserve" n: the persistent one-workgroup-per-CU pixel passes launch
 *                       n workgroups fewer (experiments with kernels running beside a frame; artgpu_batch_run_io sets it for its direct downloads) */
int artgpu_set_option(artgpu_ctx *ctx, const char *name, long value);
/* read-only counterparts: "amaze_counter0" .. "amaze_counter7" = bookkeeping of the last AMaZE call (how many tiles were streamed a
 * second time, handed to the arena kernel, ...); synchronises the context's stream */
int artgpu_get_option(artgpu_ctx *ctx, const char *name, long *value);
int artgpu_enable_timing(artgpu_ctx *ctx, int enable);
int artgpu_get_timings(const artgpu_ctx *ctx, artgpu_timings *out);

/* Replaces the Bayer branch of RawImageSource::demosaic (rtengine/rawimagesource.cc:1854-1912):
 *   ARTGPU_BAYER_AMAZE -> amaze_demosaic_RT(0,0,W,H,rawData,red,green,blue)
 *                         (rtengine/amaze_demosaic_RT.cc:41-1595) including its
 *                         border_interpolate2(W,H,3,...) when border < 4 (L1587-1589);
 *   ARTGPU_BAYER_RCD   -> rcd_demosaic() (rtengine/rcd_demosaic.cc:51-347) including its
 *                         border_interpolate2(W,H,9,...) (L342).
 *   ARTGPU_BAYER_VNG4  -> vng4_demosaic(rawData, red, green, blue) (rtengine/vng4_demosaic_RT.cc:62-397) including its
 *                         border_interpolate2(W,H,3,...) (L384).  The four-colour pattern it works on (RawImage::prefilters: the second
 *                         green is colour 3) is derived from `filters` the way dcraw's identify() does.
 * raw      : the CFA plane (RawImageSource::rawData), values 0..65535
 * filters  : RawImage::filters bit pattern (rtengine/rawimage.h:186-189), RGB Bayer only
 * initial_gain : RawImageSource::initialGain (clip_pt = 1/initialGain, amaze L53-54)
 * border   : RawImageSource::border (simpleprocess.cc:138-146)
 * out      : red, green, blue (same w,h as raw), fully overwritten. */
int artgpu_demosaic_bayer(artgpu_ctx *ctx, int method, const artgpu_plane *raw, uint32_t filters,
                          double initial_gain, int border, artgpu_rgb *out);

/* border_interpolate2 alone (rtengine/demosaic_algos.cc:200-353). */
int artgpu_border_interpolate2(artgpu_ctx *ctx, const artgpu_plane *raw, uint32_t filters,
                               int lborders, artgpu_rgb *out);

/* Replaces RawImageSource::xtrans_interpolate(passes, useCieLab) (rtengine/xtrans_demosaic.cc:181-969) including the
 * final xtransborder_interpolate(passes > 1 ? 8 : 11) (L968).  ART calls it as (1, false) for ONE_PASS and (3, true)
 * for THREE_PASS (rawimagesource.cc:1920-1925).
 * xtrans : RawImage::getXtransMatrix, 6x6 row-major, 0 = R, 1 = G, 2 = B.
 * rgb_cam: RawImage::getRgbCam, 3x4 row-major (only used to build xyz_cam for the CIELab path, L219-230). */
int artgpu_demosaic_xtrans(artgpu_ctx *ctx, int passes, int use_cielab, const artgpu_plane *raw, const int32_t xtrans[36],
                           const float rgb_cam[12], artgpu_rgb *out);

/* Replaces RawImageSource::getImage for tran=0, skip=1, no highlight recovery
 * (rtengine/rawimagesource.cc:781-1104; pixel loop L940-1025), optionally fused with the matrix
 * branch of colorSpaceConversion_ (L3184-3213) when `mat` is not NULL:
 *   image(y,x) = CLIP?( planes(sy1+y, sx1+x) * mul[c] ),  then  image = (float)(mat * image)
 * planes : red/green/blue produced by artgpu_demosaic_bayer (full sensor size)
 * sx1,sy1: crop origin = RawImageSource::border (transformRect, L664-700)
 * mul    : rm, gm, bm computed by the caller as in L790-928
 * do_clip: the reference's doClip (L964-973)
 * mat    : 9 doubles row-major = work^-1 * xyz_cam (L3187-3195), or NULL
 * image  : destination Imagefloat planes (w,h = cropped size), may not alias `planes`. */
int artgpu_get_image(artgpu_ctx *ctx, const artgpu_rgb *planes, int sx1, int sy1, const float mul[3],
                     int do_clip, const double *mat, artgpu_rgb *image);
/* The same with PreviewProps::skip > 1 (the editor's zoomed-out crops, dcrop.cc:204-205): every output pixel is the sum of a
 * skip x skip window in row-major order (L944-957) times mul -- the caller divides rm/gm/bm by skip*skip as L922-926 does --
 * with the window origin clamped to (W - skip, H - skip) (L945,949).  image is ceil(crop/skip) in each direction (L745-747). */
int artgpu_get_image_skip(artgpu_ctx *ctx, const artgpu_rgb *planes, int sx1, int sy1, int skip, const float mul[3],
                          int do_clip, const double *mat, artgpu_rgb *image);

/* RawImageSource::convertColorSpace, default-camera-profile (matrix) branch, in place
 * (rtengine/rawimagesource.cc:1128-1143,3184-3213): double accumulation, float store. */
int artgpu_convert_color_space(artgpu_ctx *ctx, artgpu_rgb *image, const double mat[9]);

/* ImProcFunctions::exposure -> expcomp (rtengine/ipexposure.cc:28-79), in place:
 *   v = max(v * exp_scale - black, 0), exp_scale = pow(2.f, expcomp), black = params.black*2000
 * (both computed by the caller, L39-40). */
int artgpu_exposure(artgpu_ctx *ctx, artgpu_rgb *image, float exp_scale, float black);

/* Device part of ImProcFunctions::toneCurve (rtengine/iptonecurve.cc:553-716) for
 * curveMode == STD with a single curve: optional filmlike_clip(img, whitept) (L214-231; taken
 * when basecurve is LINEAR, L593) followed by StandardToneCurve::Apply with the 65536-entry
 * LUT that ToneCurve::Set built on the host (rtengine/curves.cc:221-231; curves.h:224-231,
 * 360-368).  `lut65536` is a HOST pointer (copied to the device), NULL = clip only.
 * whitept > 1 needs the analytic curve beyond the LUT (setLutVal's else branch) and returns
 * ARTGPU_EUNSUPPORTED; other curve modes (NEUTRAL, PERCEPTUAL, ...) likewise. */
#define ARTGPU_TONE_STD 0
#define ARTGPU_TONE_NEUTRAL 1   /* through artgpu_tone_curve_neutral */
int artgpu_tone_curve(artgpu_ctx *ctx, artgpu_rgb *image, int mode, const float *lut65536,
                      float whitept, int filmlike_clip);

/* Host look-up tables.  One lifetime rule for every entry point that takes a LUT (artgpu_tone_curve, artgpu_tone_curve_neutral,
 * artgpu_rgb_curves, artgpu_rgb2out_matrix, artgpu_lab_adjustments, ...): the caller's array is free again when the call returns.
 * The tone and rgb curves are kept in a context-owned copy (the same curve call after call is uploaded once); the others are
 * copied on the context's stream, which is drained before the call returns if the array is pinned host memory (hipHostMalloc /
 * hipHostRegister: the DMA engine reads it when the stream gets there; a pageable array has been staged by then). */

/* curves::setLutVal (rtengine/curves.h:224-231): a value above 65535 does not go through the LUT but through the Curve object,
 * curve->getVal(val / 65535.f) * 65535.f.  The LUT is all that crosses this boundary, so the adapter states what its Curve returns
 * above 1.0 (rtengine/diagonalcurves.cc:443-561); the setting applies to the following artgpu_tone_curve / artgpu_tone_curve_neutral
 * calls of the context:
 *   ARTGPU_CURVE_TAIL_LUT       no Curve object (ToneCurve::curve == nullptr): the LUT's last entry
 *   ARTGPU_CURVE_TAIL_CONSTANT  DCT_Linear / DCT_Spline / DCT_CatmullRom: the last point's y (`y_last`, L476-477, L514-515)
 *   ARTGPU_CURVE_TAIL_IDENTITY  DCT_Empty, DCT_NURBS beyond its hash table: t itself (L529-535, L557-560)
 *   ARTGPU_CURVE_TAIL_PARAMETRIC  DCT_Parametric: the analytic form (L448-470), evaluated per pixel on the device; set with
 *                               artgpu_set_curve_tail_parametric, which takes the curve's parameter vector
 *   ARTGPU_CURVE_TAIL_HOST      (default) the adapter has not said: artgpu_tone_curve then returns ARTGPU_EUNSUPPORTED for
 *                               whitept > 1; values above 65535 that reach the curve because filmlike_clip is off take the LUT's
 *                               last entry (set the tail if the image can hold such values). */
#define ARTGPU_CURVE_TAIL_LUT 0
#define ARTGPU_CURVE_TAIL_CONSTANT 1
#define ARTGPU_CURVE_TAIL_IDENTITY 2
#define ARTGPU_CURVE_TAIL_HOST 3
#define ARTGPU_CURVE_TAIL_PARAMETRIC 4
int artgpu_set_curve_tail(artgpu_ctx *ctx, int kind, double y_last);
/* DiagonalCurve(p) with p[0] == DCT_Parametric (rtengine/diagonalcurves.cc:106-131): `p` is the curve's parameter vector as the
 * reference's constructor receives it -- p[0] the kind, p[1..3] the three zone boundaries, p[4..7] highlights / lights / darks /
 * shadows (-100 .. 100), optionally p[8] -- and `np` its length (8 or 9).  The derived constants (mc, mfc, msc, mhc) are computed here
 * with the reference's double-precision sleef forms; getVal above 1.0 then runs on the device.  ARTGPU_EINVAL for an all-zero
 * slider set: that curve is the identity in the reference and has no Curve object (ARTGPU_CURVE_TAIL_LUT). */
int artgpu_set_curve_tail_parametric(artgpu_ctx *ctx, const double *p, int np);

/* rtengine::ProgressListener (rtengine/rtengine.h:165-181; the demosaicers report through it, amaze_demosaic_RT.cc:1567-1580).
 * `fn(user, stage, fraction)` is called on the calling thread by every stage-level entry point: fraction 0.0 when the stage starts
 * queueing its work, 1.0 when the entry point returns (device-resident outputs may still be in flight on the stream; host-resident
 * ones are complete).  `stage` is the name of the reference function the entry point replaces.  NULL removes the callback.
 * The same names label roctx ranges when artgpu_set_option(ctx, "roctx", 1) is set (rocprofv3 --marker-trace). */
typedef void (*artgpu_progress_fn)(void *user, const char *stage, double fraction);
int artgpu_set_progress_callback(artgpu_ctx *ctx, artgpu_progress_fn fn, void *user);

/* rtengine::wavelet_decomposition with subsampling == 1 and the 6-tap Daub4 filters, the only
 * configuration the denoise path uses (rtengine/cplx_wavelet_dec.h:97-270; constructed at
 * rtengine/FTblockDN.cc:2296,2328,2365).  The object owns its coefficients in HBM.
 *   decompose   == the constructor (level 0 decimated Daub4, levels >= 1 undecimated Haar)
 *   band access == level_coeffs(level)[dir], dir 1..3; dir 0 = coeff0 (final low-pass)
 *   reconstruct == reconstruct(dst, blend); like the reference it consumes the object's
 *                  coefficients (the low-pass is overwritten level by level).
 * Requires min(w2,h2) >= 2^maxlvl (no Haar level wider than half the plane). */
typedef struct artgpu_wavelet artgpu_wavelet;
int artgpu_wavelet_decompose(artgpu_ctx *ctx, const artgpu_plane *src, int maxlvl, artgpu_wavelet **out);
/* SQR(MadRgb(band)) of every detail band (rtengine/FTblockDN.cc:569-603: the median of (int)min(|x|, 65535) found by a histogram
 * walk, / 0.6745): mad_sqr[3 * level + dir - 1], host memory, 3 * nlevels floats.  RGB_denoise calls it on the L and ab
 * decompositions (L2307-2320, L1000-1010); exposed so that the median search can be tested on arbitrary band contents. */
int artgpu_wavelet_mad(artgpu_ctx *ctx, const artgpu_wavelet *wv, float *mad_sqr);
int artgpu_wavelet_info(const artgpu_wavelet *wv, int32_t *w2, int32_t *h2, int32_t *nlevels);
/* copy one subband out of / into the object; `host_or_device` follows `on_device` */
int artgpu_wavelet_get_band(artgpu_ctx *ctx, const artgpu_wavelet *wv, int level, int dir, float *dst, int on_device);
int artgpu_wavelet_set_band(artgpu_ctx *ctx, artgpu_wavelet *wv, int level, int dir, const float *src, int on_device);
int artgpu_wavelet_reconstruct(artgpu_ctx *ctx, artgpu_wavelet *wv, artgpu_plane *dst, float blend);
int artgpu_wavelet_free(artgpu_ctx *ctx, artgpu_wavelet *wv);

/* The fields of procparams::DenoiseParams the path reads (rtengine/procparams.cc:1901-1918). */
typedef struct {
    double  luminance;                 /* 0..100 */
    double  luminance_detail;          /* 0..100 */
    int32_t luminance_detail_threshold;
    double  chrominance;
    double  chrominance_red_green;
    double  chrominance_blue_yellow;
    double  gamma;
    int32_t aggressive;                /* 0 = standard, 1 = QUALITY_HIGH (BiShrinkL / BiShrinkAB, FTblockDN.cc:2406-2449): both on the device */
    int32_t color_space;               /* 0 = RGB, 1 = LAB (needs the inverse working-space matrix `iwpi`): both on the device */
    int32_t chrominance_method;        /* 0 = MANUAL, 1 = AUTOMATIC: artgpu_rgb_denoise only takes `autoch` from it; the estimation itself
                                          (denoiseComputeParams) is artgpu_denoise_compute_params, which artgpu_pipeline_run calls */
} artgpu_denoise_params;

#define ARTGPU_DN_SKIP_DETAIL_RECOVERY 1u  /* leave out detail_recovery (FTblockDN.cc:1479-1635) */

/* Replaces denoise::RGB_denoise(im, kall=0, src=img, dst=img, calclum, ..., isRAW=true, dnparams,
 * expcomp, noiseLCurve(empty), noiseCCurve, nresi, highresi) (rtengine/FTblockDN.cc:1638-2689) for
 * colorSpace RGB or LAB, QUALITY_STANDARD or HIGH (aggressive), one tile (Tile_calc always yields one, L442-480).
 * img      : Imagefloat planes, denoised in place
 * ws       : ICCStore::workingSpaceMatrix(params->icm.workingProfile) as 9 floats, row-major (wpi)
 * iws      : workingSpaceInverseMatrix as 9 floats (wpi_inverse, L1702-1706); only read in LAB mode, may be NULL otherwise
 * expcomp  : RGB_denoise's `expcomp` argument (gain = 2^expcomp; 0 from ImProcFunctions::denoise)
 * scale    : ImProcData::scale (1 for full-resolution export)
 * ccalc    : the quarter-resolution chroma noise-curve map `ccalc` (L1707-1777), ((w+1)/2 x (h+1)/2),
 *            or NULL when the chroma curve is off (noisevarchrom = 1)
 * flags    : ARTGPU_DN_* ; nresi/highresi: nullable, only computed when not NULL. */
int artgpu_rgb_denoise(artgpu_ctx *ctx, artgpu_rgb *img, const artgpu_denoise_params *params, const float ws[9], const float *iws,
                       double expcomp, double scale, const artgpu_plane *ccalc, uint32_t flags,
                       float *nresi, float *highresi);

/* Replaces denoise::denoiseGuidedSmoothing (rtengine/ipsmoothing.cc:875-897): normalise to [0,1],
 * guided_smoothing(R,G,B, ws, iws, Channel::C, guidedChromaRadius, 0.001, scale) (L334-409) built on
 * guidedFilterLog / guidedFilter (rtengine/guidedfilter.cc:58-265), back to [0,65535].  In place.
 * ws: ICCStore::workingSpaceMatrix as 9 DOUBLES (TMatrix), row-major. */
int artgpu_denoise_guided_smoothing(artgpu_ctx *ctx, artgpu_rgb *img, const double ws[9], int guided_chroma_radius, double scale);

/* gaussianBlur(src, src, W, H, sigma) in place for 0.6 <= sigma < 25, GAUSS_STANDARD (rtengine/gauss.cc:1387-1574:
 * gaussHorizontalSse + gaussVerticalSse, L554-665,716-856).  Other sigma ranges: ARTGPU_EUNSUPPORTED. */
int artgpu_gaussian_blur(artgpu_ctx *ctx, artgpu_plane *img, double sigma);

/* denoise::detail_mask(src, mask, scaling, threshold, ceiling, factor, BlurType::GAUSS, blur)
 * (rtengine/FTblockDN.cc:1408-1476).  mask: same size as src, written. */
int artgpu_detail_mask(artgpu_ctx *ctx, const artgpu_plane *src, artgpu_plane *mask, float scaling, float threshold,
                       float ceiling, float factor, float blur);

/* denoise::NLMeans(img, normcoeff, strength, detail_thresh, scale) (rtengine/nlmeans.cc:50-280) on one plane
 * (the Y plane after Imagefloat::setMode(YUV), ipdenoise.cc:1174-1177), in place. */
int artgpu_nlmeans(artgpu_ctx *ctx, artgpu_plane *img, float normcoeff, int strength, int detail_thresh, float scale);

/* NEUTRAL tone-curve mode (ToneCurveParams::TcMode::NEUTRAL, ART's default): replaces apply_tc(..., NEUTRAL, ...)
 * = NeutralToneCurve::ApplyState + BatchApply (rtengine/iptonecurve.cc:88-99, rtengine/curves.cc:854-1038) for
 * BcMode::LINEAR (basecurve == nullptr).  lut65536 = ToneCurve::lutToneCurve (host), whitecoeff = ToneCurve::whitecoeff.
 * ws / iws: ICCStore workingSpaceMatrix / workingSpaceInverseMatrix (doubles; cast to float inside, curves.cc:861-868);
 * to_out / to_work: the output-profile gamut matrices inverse(om)*work and iwork*om (curves.cc:870-878; identity when
 * the output profile has no matrix).  Pixels whose Jzazbz LMS response exceeds 1 evaluate powf per pixel (device powf:
 * <= 2 ulp of the host libm's), every other pixel is bit-exact. */
typedef struct {
    double ws[9], iws[9];
    float to_out[9], to_work[9];
} artgpu_neutral_state;
int artgpu_tone_curve_neutral(artgpu_ctx *ctx, artgpu_rgb *img, const float *lut65536, float whitecoeff, const artgpu_neutral_state *state);

/* NoiseCurve::Set(const std::vector<double>&) (rtengine/ipdenoise.cc:684-716): builds the 501-entry noise-curve LUT
 * from FlatCurve control points {kind, x, y, leftTangent, rightTangent, ...} on the host (pure table construction,
 * no context needed).  *sum receives NoiseCurve::getSum(); RGB_denoise uses the chroma curve only when sum > 5
 * (FTblockDN.cc:1672).  The curve ImProcFunctions::denoise always sets is
 * {1, 0.05,0.50,0.35,0.35, 0.35,0.05,0.35,0.35} (ipdenoise.cc:1139-1149). */
int artgpu_noise_curve_lut(const double *points, int npoints, float lut[501], float *sum);

/* The quarter-resolution chroma noise map `ccalc` of RGB_denoise: calclum = every second pixel of img
 * (ipdenoise.cc:1113-1129), converted by calclum_mat like RawImageSource::convertColorSpace does to it (L1131;
 * NULL = no conversion), then Color::rgbxyz(ws)/XYZ2Lab and the curve (FTblockDN.cc:1716-1777).
 * ccalc: (w+1)/2 x (h+1)/2, host or device. */
int artgpu_denoise_chroma_map(artgpu_ctx *ctx, const artgpu_rgb *img, const double *calclum_mat, const double ws[9],
                              const float noise_c_curve[501], artgpu_plane *ccalc);

/* Replaces ImProcFunctions::denoise (rtengine/ipdenoise.cc:1096-1189):
 *   calclum/ccalc map from the un-compensated image (only if noise_c_curve != NULL and its sum > 5);
 *   if (ecomp > 0) expcomp(+ecomp); RGB_denoise(kall 0, isRAW, expcomp 0); if (smoothing_enabled) {
 *   denoiseGuidedSmoothing; if (nl_strength) { setMode(YUV); NLMeans(Y, 65535, nl_strength, nl_detail, scale);
 *   setMode(RGB); } } if (ecomp > 0) expcomp(-ecomp).
 * ecomp = params->exposure.enabled ? params->exposure.expcomp : 0 (L1155).  ws: working-space matrix as doubles
 * (TMatrix); the float casts the reference makes (wpi, Imagefloat::ws_) are made inside.
 * calclum_mat / noise_c_curve as in artgpu_denoise_chroma_map; flags: ARTGPU_DN_* (0 = the reference's behaviour). */
typedef struct {
    artgpu_denoise_params dn;
    int32_t smoothing_enabled;
    int32_t guided_chroma_radius;
    int32_t nl_strength;
    int32_t nl_detail;
} artgpu_denoise_tool_params;
int artgpu_improc_denoise(artgpu_ctx *ctx, artgpu_rgb *img, const artgpu_denoise_tool_params *params, const double ws[9], const double *iws /* LAB mode only */,
                          double ecomp, double scale, const double *calclum_mat, const float *noise_c_curve, uint32_t flags);

/* The same tool with its neighbours in the processing order fused into its first and last pixel passes (simpleprocess.cc:259 getImage,
 * :311 convertColorSpace, :315 denoise, :389 process STAGE_1 -> ImProcFunctions::exposure, ipexposure.cc:28-79): per pixel the operations
 * and their order are those of the separate calls -- the image is simply not written and read again between them.
 *   demosaiced != NULL: `img` is an output only; its pixels are RawImageSource::getImage(demosaiced, sx1, sy1, skip 1, mul, do_clip) followed
 *                       by convertColorSpace(cam_to_work; NULL = none), evaluated where the denoise reads them (artgpu_get_image's arguments);
 *                       `img` must not overlap the demosaiced planes (as for artgpu_get_image: the crop shifts the pixels);
 *   exposure_enabled:   ImProcFunctions::exposure(exp_scale, black) (artgpu_exposure's arguments) is applied to the tool's result.
 * A part that cannot be fused for the given parameters (nothing to denoise, the guided smoothing / NL-means stages between the wavelet
 * denoise and the exposure, host planes) runs as the separate call it stands for: same result either way.  fusion == NULL: artgpu_improc_denoise. */
typedef struct {
    const artgpu_rgb *demosaiced;
    int32_t sx1, sy1;
    float mul[3];
    int32_t do_clip;
    const double *cam_to_work;     /* 9 doubles or NULL */
    int32_t exposure_enabled;
    float exp_scale, black;
} artgpu_denoise_fusion;
int artgpu_improc_denoise_fused(artgpu_ctx *ctx, artgpu_rgb *img, const artgpu_denoise_fusion *fusion, const artgpu_denoise_tool_params *params,
                                const double ws[9], const double *iws, double ecomp, double scale, const double *calclum_mat,
                                const float *noise_c_curve, uint32_t flags);

/* The step immediately before the path: RawImageSource::copyOriginalPixels without dark frame / flat
 * field (rawimagesource.cc:2325-2428: rawData = (float)src->data) followed by RawImageSource::scaleColors (L2677-2859):
 *   val = max(0, raw - cblacksom[c4]) * scale_mul[c4],  chmax[c] = max over the frame   (c4 = 3 for the second Bayer green).
 * src: the sensor data as uint16 (src_is_u16 != 0, row stride in BYTES as usual) or float; dst: float CFA plane.
 * cfa: Bayer `filters` word when xtrans == NULL, else the 6x6 X-Trans colour map.  The host keeps computing cblacksom /
 * scale_mul (calculate_scale_mul, L753-779: a handful of scalars).  chmax[4]: channel maxima (chmax[3] = chmax[1], L2855). */
int artgpu_scale_colors(artgpu_ctx *ctx, const void *src, int32_t w, int32_t h, int64_t src_row_stride_bytes, int32_t src_is_u16,
                        int32_t src_on_device, uint32_t filters, const int32_t *xtrans, const float cblacksom[4],
                        const float scale_mul[4], artgpu_plane *dst, float chmax[4]);

/* ImProcFunctions::denoiseComputeParams (rtengine/ipdenoise.cc:800-1093): the AUTOMATIC chrominance estimation (the default
 * chrominanceMethod, procparams.cc:1909).  Nine crops of (widIm/2) x (heiIm/2) of the white-balanced sensor planes go through
 * RGB_denoise_info (ipdenoise.cc:227-669): Lab hue/chroma/luminance statistics at half resolution, a 5-level wavelet
 * decomposition of the gamma-encoded chroma planes and the MADs of its 30 subbands (WaveletDenoiseAll_info / ShrinkAll_info,
 * FTblockDN.cc:1227-1362), calcautodn_info (ipdenoise.cc:66-206) per crop and the reduction of L960-1072.  Raw sources only
 * (imgsrc->isRAW()).
 *   planes        : RawImageSource red/green/blue after demosaic (what getImage reads)
 *   border        : RawImageSource::border (getFullSize subtracts it twice, transformRect adds it to every crop origin)
 *   mul, do_clip  : getImage's per-channel multipliers and clip flag, as for artgpu_get_image
 *   cam_to_work   : the convertColorSpace matrix, as for artgpu_convert_color_space
 *   ws            : ICCStore::workingSpaceMatrix(params->icm.workingProfile)
 *   store         : DenoiseInfoStore (improcfun.h:117-131).  valid != 0 on entry: only dn is refreshed from it (L802-809).
 *   dn            : in: gamma, aggressive, chrominance_method; out: chrominance, chrominance_red_green, chrominance_blue_yellow
 *                   = store value x chrominance_auto_factor.  chrominance_method != AUTOMATIC: nothing happens. */
typedef struct {
    int32_t valid;
    float   ch_M[9], max_r[9], max_b[9];
    double  chrominance, chrominance_red_green, chrominance_blue_yellow;
    /* diagnostics, not part of the reference's store: per crop (k = hcr*3 + wcr) {chaut, maxredaut, maxblueaut, minredaut,
     * minblueaut, chromina, lumema, redyel, skinc, nsknc, Nb, 0...} as RGB_denoise_info returns them */
    float   crop_info[9][16];
} artgpu_denoise_info_store;
int artgpu_denoise_compute_params(artgpu_ctx *ctx, const artgpu_rgb *planes, int border, const float mul[3], int do_clip,
                                  const double cam_to_work[9], const double ws[9], double chrominance_auto_factor,
                                  artgpu_denoise_info_store *store, artgpu_denoise_params *dn);

/* rtengine::guidedFilter(guide, src, dst, r, epsilon, multithread, subsampling) (guidedfilter.cc:78-241), the single-channel fast guided
 * filter behind several tools (hslEqualizer, guided smoothing, dehaze, local contrast masks): bilinear subsample by
 * calculate_subsampling (subsampling <= 0: L58-75), four box means, a / b, their means, bilinear upsample.  guide, src and dst are
 * planes of one size; dst may be src or guide.  The denoise tool's three-channel log variant is artgpu_denoise_guided_smoothing. */
int artgpu_guided_filter(artgpu_ctx *ctx, const artgpu_plane *guide, const artgpu_plane *src, artgpu_plane *dst, int r, float epsilon,
                         int subsampling);

/* ImProcFunctions::hslEqualizer (rtengine/iphsl.cc:29-221): hue / saturation / luminance adjustments as FlatCurves
 * over hue.  Each curve is the `std::vector<double>` of ProcParams (hsl.hCurve / sCurve / lCurve: {FCT_MinMaxCPoints, x, y, left
 * tangent, right tangent, ...}; NULL / n <= 4 or an identity curve = not applied), built into its polyline on the host exactly as
 * FlatCurve's constructor does (periodic, CURVES_MIN_POLY_POINTS / scale points) and evaluated per pixel on the device; the masks
 * are smoothed with rtengine::guidedFilter (radius from `smoothing` and scale, L118-123).  img is RGB on entry.  The reference leaves
 * the Imagefloat in YUV mode (g = Y, b = u, r = v); to_rgb != 0 applies Imagefloat::setMode(RGB) on top so that the planes are RGB again. */
int artgpu_hsl_equalizer(artgpu_ctx *ctx, artgpu_rgb *img, const double *hcurve, int nh, const double *scurve, int ns,
                         const double *lcurve, int nl, int smoothing, const double ws[9], double scale, int to_rgb);

/* RawImageSource::dual_demosaic_RT (rtengine/dual_demosaic_RT.cc:39-155), Bayer methods AMAZEBILINEAR / RCDBILINEAR / AMAZEVNG4 / RCDVNG4:
 * the first demosaicer (method = ARTGPU_BAYER_AMAZE or ARTGPU_BAYER_RCD, exactly artgpu_demosaic_bayer), then L* of its output
 * (Color::RGB2L, color.cc:1343-1379), the contrast blend mask (buildBlendMask, rt_algo.cc:315-498: sigmoid of the 8-neighbour contrast
 * against the threshold, 2-pixel frame, gaussian blur sigma 2) and the blend with a bilinear interpolation in flat regions
 * (bayer_bilinear_demosaic.cc:33-77).  *contrast is RAWParams::BayerSensor::dualDemosaicContrast in percent, in/out like the reference's
 * `double &contrast`: with auto_contrast != 0 the threshold is searched (flattest 80 / 40-pixel tile, calcContrastThreshold) and written
 * back.  contrast == 0 without auto_contrast runs only the first demosaicer.  second = ARTGPU_DUAL_VNG4 (AMAZEVNG4 / RCDVNG4): the flat
 * regions come from vng4_demosaic instead (all three channels of every pixel, dual_demosaic_RT.cc:128-148).  DCB as first demosaicer
 * and X-Trans (fast_xtrans_interpolate_blend) are not on the device path. */
int artgpu_dual_demosaic_bayer(artgpu_ctx *ctx, int method, int second, const artgpu_plane *raw, uint32_t filters, double initial_gain, int border,
                               double *contrast, int auto_contrast, artgpu_rgb *out);

/* ImProcFunctions::logEncoding (rtengine/iplogenc.cc:132-316,395-402): brightness-norm log tone mapping.
 * The struct holds the LogEncodingParams fields the function reads (procparams.h; defaults procparams.cc:2039-2051); `enabled == 0`
 * returns at once like the reference.  regularization > 0 smooths the posterised log-norm with rtengine::guidedFilter at radius
 * max(full_width, W, full_height, H) / 30 (full_width/height = ImProcFunctions::full_width/full_height, the uncropped image size).
 * highlight_compression > 0 (L148-170) evaluates std::pow(float, float) twice per pixel above 0.8; the C library's powf is not
 * specified bit for bit, the device uses double-precision pow rounded to float: identical to glibc's result except for a few
 * values per million (at most 1.2e-6 relative; tests/test_gpu_logenc.py holds the bound).  Every other branch is bit-exact. */
typedef struct artgpu_logenc_params {
    int32_t enabled;
    int32_t regularization;          /* 0..100 */
    int32_t satcontrol;
    int32_t highlight_compression;   /* 0..100 */
    double gain, target_gray, black_ev, white_ev;
} artgpu_logenc_params;
int artgpu_log_encoding(artgpu_ctx *ctx, artgpu_rgb *img, const artgpu_logenc_params *p, const double ws[9], int full_width, int full_height);

/* ImProcFunctions::labAdjustments (rtengine/iplabadjustments.cc:277-345) as the four device steps the function is
 * made of; the caller keeps building the three curves (get_L_curve / get_ab_curves, DiagonalCurve: host code), exactly where it does now:
 *   artgpu_rgb_to_lab      Imagefloat::setMode(LAB) from RGB = rgb_to_lab (imagefloat.cc:841-876): in place, afterwards g = L, r = a, b = b.
 *                          ws = the working-space TMatrix (narrowed to float like Imagefloat::get_ws).
 *   artgpu_lab_histogram   hist16[(int)L]++ over the L plane (L300-327; 65536 bins, index clamped like LUTu::operator[]) - only needed
 *                          when labCurve.contrast != 0.
 *   artgpu_lab_adjustments lab_adjustments' curve loop (L236-264): L = lcurve[L], a = (acurve[a + 32768] - 32768) * chroma, same for b.
 *                          lcurve has 32770 entries (LUTf(32770, 0)), acurve / bcurve 65536; chroma = (chromaticity + 100) / 100.
 *   artgpu_lab_to_rgb      the next setMode(RGB) = lab_to_rgb (imagefloat.cc:941-970), iws = the inverse working-space matrix.
 * The reference runs all of them four pixels at a time with a scalar tail and the two forms round differently; the device follows the
 * form of each pixel's column (and, for XYZ2Lab, of its group of four). */
int artgpu_rgb_to_lab(artgpu_ctx *ctx, artgpu_rgb *img, const double ws[9]);
int artgpu_lab_to_rgb(artgpu_ctx *ctx, artgpu_rgb *img, const double iws[9]);
int artgpu_lab_histogram(artgpu_ctx *ctx, const artgpu_rgb *img, uint32_t hist[65536]);
int artgpu_lab_adjustments(artgpu_ctx *ctx, artgpu_rgb *img, const float *lcurve, const float *acurve, const float *bcurve, float chroma);

/* SURVEY section 8f N1, the parts of the output stage that are plain arithmetic (everything lcms2 evaluates stays on the host):
 * artgpu_rgb2out_matrix : ARTOutputProfile::operator()(const Imagefloat*, Imagefloat*), the matrix + TRC fast path of
 *                         ImProcFunctions::rgb2out for matrix output profiles (iprgb2out.cc:94-172,452-461).  matrix = the host's
 *                         `matrix_` (inverse profile matrix x working space, L119); trc_linear != 0 for MODE_LINEAR; otherwise
 *                         lut/lutsz = the host's `lut_` (compute_lut, L207-215; preview pipelines use 65536 / 1024 / 256 entries).
 *                         Values > 1 in a non-linear mode need ARTOutputProfile::eval (lcms2 / libm): the call then fails with
 *                         ARTGPU_EUNSUPPORTED after counting them (dst is complete for all other pixels).
 * artgpu_get_scanlines  : Imagefloat::getScanline for all rows (imagefloat.cc:125-170): interleaved RGB as the TIFF/PNG/JPEG
 *                         writers consume it; (bps, is_float) = (8,0), (16,0), (16,1: half, DNG_FloatToHalf), (32,1).  Moves 3-6 B/px
 *                         to the host instead of 12. */
int artgpu_rgb2out_matrix(artgpu_ctx *ctx, const artgpu_rgb *src, artgpu_rgb *dst, const float matrix[9], int trc_linear,
                          const float *lut, int lutsz);
int artgpu_get_scanlines(artgpu_ctx *ctx, const artgpu_rgb *img, int bps, int is_float, void *dst, int64_t dst_row_stride_bytes,
                         int dst_on_device);

/* acc = 0; for (i = 0; i < n; ++i) acc += x[i];  in fp32 -- the order-defined sum the reference uses for its image statistics
 * (ShrinkAll_info, FTblockDN.cc:1237-1290), evaluated on the device by an exact parallel scan (values >= 0 take the fast path;
 * anything else is still exact, one value at a time).  Exported because it is the building block of
 * artgpu_denoise_compute_params that is worth testing on its own. */
int artgpu_ordered_sum_f32(artgpu_ctx *ctx, const float *x, int64_t n, int on_device, float *result);

/* The device-side math primitives of the path (art_amd/csrc/devmath.h, devsleef.h, paramcurve.h -- every kernel is built from them),
 * evaluated one value per lane on host arrays of n values: out0[i] = f(a[i] [, b[i] [, c[i]]]).  Diagnostic entry point: it lets a caller
 * (tests/test_gpu_primitives.py) compare the GPU's results bit for bit with fixtures generated from the reference's own headers
 * compiled in place -- rtengine/sleef.h:938-966,1198-1313 (xexpf, xlogf, pow_F, xlin2log, xlog2lin, xcbrtf, xatan2f, double xlog / xexp),
 * sleefsseavx.h:978-1000,1232-1345 (the 4-lane forms, whose last bits differ), sleefsseavx.h:1435-1442 + helpersse2.h:168-179 (vminf / vmaxf
 * operand order, vintpf), median.h (median3), LUT.h:349-377 / 436-459 (LUTf::operator[] for vfloat / float).  The product path never calls it.
 * XSINCOSF writes sin to out0 and cos to out1; XLIN2LOG / XLOG2LIN take the base in `param`; the LUTF_* forms look a[i] up in
 * table[table_size]; XLOG_D / XEXP_D read and write double arrays; FLOAT_TO_HALF (DNG_FloatToHalf, halffloat.h:9-46: the half-float scanlines)
 * writes 32-bit words with the half in the low 16 bits.  Returns ARTGPU_EINVAL for a missing operand. */
enum {
    ARTGPU_PRIM_XEXPF_S = 0, ARTGPU_PRIM_XEXPF_V, ARTGPU_PRIM_XEXPF_VN, ARTGPU_PRIM_XEXPF_V_LDEXP, ARTGPU_PRIM_XLOGF_S, ARTGPU_PRIM_XLOGF_V,
    ARTGPU_PRIM_XLOGF_VN, ARTGPU_PRIM_POW_F, ARTGPU_PRIM_XLIN2LOG, ARTGPU_PRIM_XLOG2LIN, ARTGPU_PRIM_XCBRTF, ARTGPU_PRIM_XATAN2F,
    ARTGPU_PRIM_XSINCOSF, ARTGPU_PRIM_LUTF_SCALAR, ARTGPU_PRIM_LUTF_VECTOR, ARTGPU_PRIM_MEDIAN3, ARTGPU_PRIM_VMINF, ARTGPU_PRIM_VMAXF,
    ARTGPU_PRIM_VINTPF, ARTGPU_PRIM_XDIV2F, ARTGPU_PRIM_XDIVF2, ARTGPU_PRIM_XLOG_D, ARTGPU_PRIM_XEXP_D, ARTGPU_PRIM_FLOAT_TO_HALF,
    ARTGPU_PRIM_COUNT
};
int artgpu_eval_primitive(artgpu_ctx *ctx, int prim, const void *a, const void *b, const void *c, void *out0, void *out1, int64_t n,
                          float param, const float *table, int table_size);

/* Two of the default-off pixelwise steps of ImProcFunctions::process, so that a frame with these
 * common edits stays on the device:
 * artgpu_channel_mixer : the pixel loop of ImProcFunctions::channelMixer (ipchmixer.cc:185-230); m = {RR,RG,RB, GR,GG,GB,
 *                        BR,BG,BB} as the function computes them (params/1000, or get_mixer_matrix for PRIMARIES_CHROMA).
 * artgpu_rgb_curves    : the pixel loop of ImProcFunctions::rgbCurves (iprgbcurves.cc:116-143); each LUT is the 65536-entry
 *                        `outCurve` RGBCurve() builds on the host (L41-53), NULL for an identity curve. */
int artgpu_channel_mixer(artgpu_ctx *ctx, artgpu_rgb *img, const float m[9]);
int artgpu_rgb_curves(artgpu_ctx *ctx, artgpu_rgb *img, const float *rcurve, const float *gcurve, const float *bcurve);
/* ImProcFunctions::saturationVibrance (ipsaturation.cc:43-83): saturation, vibrance = params->saturation.{saturation,vibrance}
 * (integers; both 0 = nothing to do), ws = working-space matrix (row 1 is the luminance). */
int artgpu_saturation_vibrance(artgpu_ctx *ctx, artgpu_rgb *img, int saturation, int vibrance, const double ws[9]);

/* The whole hot path for one frame in one call -- what ART's batch loop does per image between load and rgb2out
 * (simpleprocess.cc stage_init L215-259, stage_denoise L311-315, stage_finish L389-396):
 *   demosaic -> getImage (crop `border`, x mul, clip) + convertColorSpace matrix -> ImProcFunctions::denoise ->
 *   ImProcFunctions::exposure -> ImProcFunctions::toneCurve.
 * raw: CFA plane (host or device).  out: (W - 2*border) x (H - 2*border) planes (host or device; the frame stays on the device
 * between the stages either way).  Disabled stages are skipped exactly like their `enabled == false` early-outs. */
typedef struct {
    int32_t sensor;                 /* 0 = Bayer, 1 = X-Trans */
    int32_t bayer_method;           /* ARTGPU_BAYER_* */
    uint32_t filters;
    double initial_gain;
    int32_t xtrans_passes;          /* 1 (ONE_PASS) or 3 (THREE_PASS, CIELab) */
    int32_t xtrans[36];
    float rgb_cam[12];
    int32_t border;                 /* raw.bayersensor.border / raw.xtranssensor.border */
    float mul[3];                   /* rm, gm, bm of getImage */
    int32_t do_clip;
    int32_t has_cam_to_work;
    double cam_to_work[9];          /* matrix branch of convertColorSpace */
    double ws[9], iws[9];           /* working space <-> XYZ (TMatrix) */
    int32_t denoise_enabled;
    artgpu_denoise_tool_params denoise;
    int32_t exposure_enabled;
    double expcomp, black;
    int32_t tone_enabled;
    int32_t tone_mode;              /* ARTGPU_TONE_STD / ARTGPU_TONE_NEUTRAL */
    const float *tone_lut;          /* 65536 entries (host) */
    float white_point;
    float to_out[9], to_work[9];    /* NEUTRAL only */
    double scale;
    double chrominance_auto_factor; /* DenoiseParams::chrominanceAutoFactor; 0 means 1 (only read when denoise.dn.chrominance_method is AUTOMATIC:
                                     * artgpu_denoise_compute_params then runs on the demosaiced planes, as simpleprocess.cc:254-256 does) */
} artgpu_pipeline_params;
int artgpu_pipeline_run(artgpu_ctx *ctx, const artgpu_plane *raw, const artgpu_pipeline_params *params, artgpu_rgb *out);

/* This rank's share of a batch: frames are independent (batchProcessingThread handles them one after another,
 * simpleprocess.cc:586-612), so a multi-GPU batch is one context per GPU each running its own frames; the completion
 * barrier / gather lives in the host driver (bench.py, art_amd/batch.py), not in the data path. */
int artgpu_batch_run(artgpu_ctx *ctx, int nframes, const artgpu_plane *raws, const artgpu_pipeline_params *params, artgpu_rgb *outs);
/* Frames in flight per GPU for artgpu_batch_run (default 1 = one after another on the context's stream).  With lanes > 1 the
 * context keeps lanes-1 sibling contexts (own stream, work arenas and host thread); frame f runs on lane f % lanes, the results are
 * the same bits.  On return the secondary lanes have completed; lane 0 stays ordered on the context's stream as before. */
int artgpu_set_batch_lanes(artgpu_ctx *ctx, int lanes);

/* A batch whose frames arrive as the decoder delivers them and leave as the writers take them -- the data formats either side of the
 * path, so that 2 + 6 (or 2 + 3) bytes per pixel cross PCIe instead of 4 + 12:
 *   in : uint16 (or float) sensor data + the black levels / multipliers of RawImageSource::scaleColors (rawimagesource.cc:2677-2859,
 *        artgpu_scale_colors); the frame's channel maxima come back in `chmax`;
 *   out: ARTOutputProfile's matrix + TRC fast path of ImProcFunctions::rgb2out (iprgb2out.cc:94-172,452-461, artgpu_rgb2out_matrix;
 *        rgb2out_enabled = 0: the working-space image as it is) followed by Imagefloat::getScanline for every row
 *        (imagefloat.cc:125-170, artgpu_get_scanlines: (bps, is_float) = (8,0), (16,0), (16,1), (32,1)).
 * Between the two: artgpu_pipeline_run's stages with `params` (params->border trims the scanlines like getImage does).
 * The call is built for host buffers: every lane (artgpu_set_batch_lanes) keeps a frame's upload, its kernels and its download on three
 * streams with two staging slots each, so the upload of frame f + 1 and the download of frame f - 1 run beside the kernels of frame f and
 * the host thread never waits for a copy before the batch ends.  Pinned host memory (hipHostMalloc / hipHostRegister) is what makes the
 * copies asynchronous; pageable memory works and is staged by the runtime, one blocking copy at a time.  Device pointers (on_device != 0)
 * skip the corresponding copy.  `status` of a frame: ARTGPU_OK, or ARTGPU_EUNSUPPORTED when rgb2out met values above 1 with a
 * non-linear TRC (artgpu_rgb2out_matrix's rule; the scanlines are complete for every other pixel); the call returns the first
 * non-zero status.  Same bits as artgpu_scale_colors -> artgpu_pipeline_run -> artgpu_rgb2out_matrix -> artgpu_get_scanlines. */
typedef struct {
    const void *data;               /* sensor values, row-major */
    int32_t w, h;
    int64_t row_stride_bytes;
    int32_t is_u16;                 /* 1: uint16_t, 0: float */
    int32_t on_device;
    float cblacksom[4], scale_mul[4];
} artgpu_sensor_frame;
typedef struct {
    void *scanlines;                /* (h - 2 border) rows of (w - 2 border) interleaved RGB pixels */
    int64_t row_stride_bytes;
    int32_t bps, is_float;
    int32_t on_device;
    int32_t rgb2out_enabled;
    float out_matrix[9];
    int32_t trc_linear;
    const float *trc_lut;           /* host, trc_lutsz entries (NULL with trc_linear) */
    int32_t trc_lutsz;
    float chmax[4];                 /* out: scaleColors' channel maxima (chmax[3] = chmax[1]) */
    int32_t status;                 /* out */
} artgpu_scanline_frame;
int artgpu_batch_run_io(artgpu_ctx *ctx, int nframes, const artgpu_sensor_frame *in, const artgpu_pipeline_params *params,
                        artgpu_scanline_frame *out);

/* The completion step of a multi-GPU batch -- the only collective of the path (frames are independent; the reference's loop
 * simply finishes, simpleprocess.cc:591-611): every rank contributes one 64-byte record (by convention of art_amd/batch.py: rank,
 * frames done, status, checksum of the outputs, elapsed microseconds, 3 spare words) and receives all of them, rank-major, in
 * `all_records` (host memory, nranks * 8 words).  `rccl_comm` is the caller's ncclComm_t (RCCL) whose ranks each hold one
 * artgpu context on their own GPU; the all-gather runs on the context's stream after the frames queued there, and doubles as
 * the batch barrier.  NULL with nranks == 1 copies the record.  RCCL is loaded at run time (no link-time dependency). */
#define ARTGPU_BATCH_RECORD_WORDS 8
int artgpu_batch_complete(artgpu_ctx *ctx, void *rccl_comm, int nranks, const int64_t record[ARTGPU_BATCH_RECORD_WORDS], int64_t *all_records);

/* Bytes of device scratch the context currently holds (arena + staging + the denoise pool: for a 45 MP frame through
 * artgpu_improc_denoise about 5.0 GB -- two L band sets, the chroma band sets of both channels, the DCT block buffer; the hand-over ring of the
 * fused shrink pass is 24 MB -- per context and per batch lane; grown on demand, kept between frames, artgpu_trim_scratch gives it back). */
size_t artgpu_scratch_bytes(const artgpu_ctx *ctx);

/* Gives the context's device scratch back to the driver (work arenas, staging planes, the denoise pool, of the context and of its batch
 * lanes): waits for the context's queued work first, keeps settings, curves and streams; the next call grows what it needs again and rebuilds
 * the tables that lived in the pool.  The reference frees its scratch when each tool returns (FTblockDN.cc:2655-2689, amaze_demosaic_RT.cc:1583);
 * the device path keeps it between frames because allocation is what a steady-state batch must not pay for -- this is the call for the moment
 * a batch is over, a much smaller frame size follows, or several contexts / ranks have to share one GPU's memory. */
int artgpu_trim_scratch(artgpu_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* ARTGPU_H */
