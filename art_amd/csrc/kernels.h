// art_amd/csrc/kernels.h -- kernel argument blocks and launch geometry shared by the
// .hip kernels and the C-ABI host code (artgpu_api.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "paramcurve.h"
#include <vector>
#include <stddef.h>
#include <stdint.h>
#include <mutex>
#include <set>
#include <utility>

// Kernels that need more than 64 KB of dynamic LDS raise the limit once per (kernel, device): the attribute belongs to the device's copy
// of the function, and a host application may hold contexts on several devices and threads.
inline hipError_t dyn_lds_once(const void *fn, int bytes)
{
    static std::mutex m;
    static std::set<std::pair<const void *, int>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(m);
    if (done.count({fn, dev})) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.insert({fn, dev});
    return e;
}

// What a workgroup of the current device may have (asked once per device).  The kernels that keep a look-up table or a tile's state in LDS want
// 137 - 160 KB of dynamic LDS and 1024 threads; a device (or a build for another target) with less takes the plain form of the pass instead
// of failing at launch.
inline bool device_block_fits(int lds_bytes, int threads)
{
    static std::mutex m;
    static int lds_of[64], thr_of[64];
    static bool known[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    std::lock_guard<std::mutex> lk(m);
    if (!known[dev]) {
        int l = 0, t = 0;
        if (hipDeviceGetAttribute(&l, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) l = 0;
        if (hipDeviceGetAttribute(&t, hipDeviceAttributeMaxThreadsPerBlock, dev) != hipSuccess) t = 0;
        lds_of[dev] = l; thr_of[dev] = t; known[dev] = true;
    }
    return lds_of[dev] >= lds_bytes && thr_of[dev] >= threads;
}

namespace artgpu {

// ---- AMaZE (amaze.hip) ----
#ifndef ARTGPU_AMAZE_THREADS
#define ARTGPU_AMAZE_THREADS 512
#endif
constexpr int AMAZE_THREADS = ARTGPU_AMAZE_THREADS;
constexpr int AMAZE_ARENA_FLOATS = 362208; // 1 448 832 B per workgroup (reference: 1 448 767 B, amaze_demosaic_RT.cc:124)
constexpr int AMAZE_TS = 160;
constexpr int AMAZE_STEP = 128;

struct AmazeArgs {
    const float *raw;   // CFA plane
    size_t raw_stride;  // floats
    float *red, *green, *blue;
    size_t out_stride;  // floats
    float *arena;       // gridDim.x * AMAZE_ARENA_FLOATS
    int W, H;
    int ntx, ntiles;
    unsigned filters;
    float clip_pt, clip_pt8;
    int *bbox;          // 4 ints per arena: nyquist bounding box of the tile
    unsigned zero_mask;
    int zero_frame;     // > 0: regions 4-8 of full tiles are cleared only in a frame of this many rows / columns
    const int *tile_list;   // device: the tiles this launch processes (nullptr: tiles 0..ntiles-1)
    const int *tile_count;  // device: how many entries of tile_list are valid (read when the kernel starts)
    const int *queue_hdr;   // device (optional): redo-queue entries [hdr[1], hdr[0]) nobody streamed again are processed as well
    const unsigned long long *queue_words;
    int *queue_counters;    // device (optional): bookkeeping for artgpu_get_option
    int split;              // 1: one launch per phase over all tiles (profiling mode; needs one arena per tile)
};
hipError_t launch_amaze(const AmazeArgs &a, int grid, hipStream_t stream);

// ---- AMaZE v2, LDS row streaming (amaze_stream.hip): full 160x160 tiles only ----
struct AmazeStreamArgs {
    const float *raw;
    size_t raw_stride;  // floats
    float *red, *green, *blue;
    size_t out_stride;  // floats
    int W, H, ntx;
    unsigned filters;
    float clip_pt, clip_pt8;
    int g00, ey;        // (0,0) is green; row parity of the red sites
    const int *tiles;   // device: tile indices to stream
    int ntiles;
    int *fallback;      // device: [0] = count, [1..] = tiles the arena kernel has to (re)do
    // Redo queue: a streamed tile whose Nyquist sites did not all lie inside its bounding box (known only at the end of the tile)
    // is streamed a second time with the true box, by whichever workgroup runs out of tiles first.  hdr[0] = entries reserved,
    // hdr[1] = entries taken; words[k] = one 64-bit entry (0: not published yet), published / abandoned by compare-and-swap so that
    // every entry is either streamed again or lands in `fallback`.
    int *queue_hdr;
    unsigned long long *queue_words;
};
hipError_t launch_amaze_stream(const AmazeStreamArgs &s, int grid, hipStream_t stream);

// ---- device math primitives on caller arrays (selftest.hip; the ids are the ARTGPU_PRIM_* values of include/artgpu.h) ----
enum { PRIM_XEXPF_S = 0, PRIM_XEXPF_V, PRIM_XEXPF_VN, PRIM_XEXPF_V_LDEXP, PRIM_XLOGF_S, PRIM_XLOGF_V, PRIM_XLOGF_VN, PRIM_POW_F, PRIM_XLIN2LOG,
       PRIM_XLOG2LIN, PRIM_XCBRTF, PRIM_XATAN2F, PRIM_XSINCOSF, PRIM_LUTF_SCALAR, PRIM_LUTF_VECTOR, PRIM_MEDIAN3, PRIM_VMINF, PRIM_VMAXF,
       PRIM_VINTPF, PRIM_XDIV2F, PRIM_XDIVF2, PRIM_XLOG_D, PRIM_XEXP_D, PRIM_FLOAT_TO_HALF, PRIM_COUNT };
struct PrimArgs {
    int prim;
    long long n;
    const void *a, *b, *c;
    void *out0, *out1;
    float param;
    const float *table;
    int table_size;
};
hipError_t launch_prim_eval(const PrimArgs &p, hipStream_t s);

// ---- RCD (rcd_stream.hip: the tile state in LDS) ----
constexpr int RCD_TS = 194;
constexpr int RCD_BORDER = 9;
struct RcdStreamArgs {
    const float *raw;
    size_t raw_stride;
    float *red, *green, *blue;
    size_t out_stride;
    int W, H;
    int numTw, ntiles;
    unsigned filters;
    int vec2;           // out planes 8-byte aligned with an even stride: column pairs are stored as one 64-bit word
    int *counter;       // zeroed before the launch: next tile = gridDim.x + counter++
};
int rcd_stream_workgroups_per_cu(int rows_per_iter);
hipError_t launch_rcd_stream(const RcdStreamArgs &a, int rows_per_iter, int grid, hipStream_t stream);

// ---- X-Trans Markesteijn demosaic (xtrans.hip) ----
#define XTRANS_TS 114
#ifndef XTRANS_THREADS
#define XTRANS_THREADS 1024   // round 1, one load in flight per pixel: {256,512,1024} x min-waves {4,6,8}: 1024 x 8 the fastest; round 2, loads batched: 1024 x 4 (xtrans.hip)
#endif
struct XtransArgs {
    const float *raw; size_t raw_stride;
    float *red, *green, *blue; size_t out_stride;
    float *arena; size_t arena_floats;       // per workgroup: TS*TS*(ndir*4+3)+128 floats, reference layout
    const float *cbrt_lut;                   // cielab's 0x14000-entry LUT (device)
    int W, H, ntx, ntiles, passes, ndir, use_cielab;
    int sgrow, sgcol, right_shift[3];
    int xtrans[36];
    int allhex0[3][3][8];                    // offsets in the raw plane (h + v*raw_stride)
    int allhex1[3][3][8];                    // offsets in a tile plane (h + v*TS)
    float xyz_cam[9];
    int border;                              // xtrans_border_kernel only
    int *counter;                            // tile counter (device int, cleared before the launch); null: tiles blockIdx.x + k * gridDim.x
};
hipError_t launch_xtrans(const XtransArgs &a, int grid, hipStream_t s);

// ---- border_interpolate2 (border.hip) ----
struct BorderArgs {
    const float *raw;
    size_t raw_stride;
    float *red, *green, *blue;
    size_t out_stride;
    int W, H, bord;
    unsigned filters;
};
__global__ void border_interpolate2_kernel(BorderArgs a);
hipError_t launch_border_interpolate2(const BorderArgs &a, int grid, hipStream_t stream);

// ---- per-pixel stages (pixelops.hip) ----
struct PixArgs {
    const float *src[3];   // get_image: demosaiced planes
    size_t src_stride;
    int sx1, sy1;          // crop origin in src (RawImageSource::border)
    int skip, src_w, src_h; // get_image: PreviewProps::skip (0/1 = every pixel) and the plane size for its edge clamp
    float *dst[3];         // destination / in-place image
    size_t dst_stride;
    int w, h;
    float mul[3];          // rm, gm, bm
    int has_mul;           // get_image: apply the channel multipliers (0 = pure matrix conversion)
    int do_clip;           // get_image: CLIP; tone: filmlike_clip
    int has_mat;
    double mat[9];         // raw->working matrix, row-major
    float exp_scale, black;
    int chain_n;           // exposure / setMode(RGB): this many further ImProcFunctions::exposure steps on the value before it is stored
    float chain_scale[2], chain_black[2];
    const float *lut;      // tone: 65536-entry LUT on the device (nullable)
    float whitept;
    int tail_kind;         // curves::setLutVal above 65535 (artgpu_set_curve_tail): 0 LUT clip, 1 constant, 2 identity, 4 parametric
    double tail_y;
    ParamCurve tail_pc;    // kind 4 (artgpu_set_curve_tail_parametric)
    int no_lds_lut;        // artgpu_set_option "lut_lds" 0: the plain one-lane-per-pixel kernels (table lookups served by L2) on every frame size
    int cu_reserve;        // CUs a kernel running BESIDE this frame's holds (artgpu_batch_run_io's download): the persistent one-workgroup-per-CU shape, which shares
                           // no CU with anything, launches that many workgroups fewer -- a workgroup without a CU would start when that kernel ends, 5 ms later
};
hipError_t launch_get_image_convert(const PixArgs &a, hipStream_t s);
hipError_t launch_exposure(const PixArgs &a, hipStream_t s);
hipError_t launch_tone_std(const PixArgs &a, hipStream_t s);
hipError_t launch_yuv_mode(const PixArgs &a, hipStream_t s);
// copyOriginalPixels (no dark frame / flat field) + scaleColors (rawimagesource.cc:2325-2428,2677-2859)
struct ScaleArgs {
    const void *src; size_t src_stride;   // elements; uint16 when src_u16 else float
    int src_u16;
    float *dst; size_t dst_stride;
    int w, h;
    int cfa[36];                          // colour (0..2) at [row % 6][col % 6]; Bayer maps are tiled into it
    int bayer;                            // Bayer: the second green of each 2x2 uses black/scale index 3
    float cblacksom[4], scale_mul[4];
    int *chmax_bits;                      // [3] channel maxima as float bit patterns (values are >= 0)
};
hipError_t launch_scale_colors(const ScaleArgs &a, hipStream_t s);
// AUTOMATIC chrominance estimation (ipdenoise.cc:227-669,800-1093; FTblockDN.cc:1227-1362): the nine quarter-image crops
struct DnInfoArgs {
    const float *src[3];     // RawImageSource red/green/blue (full planes)
    size_t stride;
    float mul[3];
    int do_clip;
    int sx[9], sy[9];        // crop origins in plane coordinates (border included), crop k = hcr*3 + wcr
    int crW, crH, wid, hei;  // crop size and its half-resolution size
    double mat[9];           // camera -> working (convertColorSpace)
    float wp[9];             // working space -> XYZ
    const float *cachef, *cachefy, *gamcurve;
    float gain, gam, gamthresh, gamslope;
    float *maps;             // [9][3][wid*hei]: hue, chroma, luminance
    float *A, *B;            // labdn->a / labdn->b of ONE crop (crW*crH)
    int crop;                // which crop dninfo_ab fills
    float *stats;            // [9][8]: chro, lume, red_yel, skin_c sums, then nry, nsk as int bits
};
hipError_t launch_dninfo_maps(const DnInfoArgs &a, hipStream_t s);
hipError_t launch_dninfo_stats(const DnInfoArgs &a, hipStream_t s);
hipError_t launch_dninfo_ab(const DnInfoArgs &a, hipStream_t s);
// fp32 sum of x[0..n) in index order, rounding at every step like `for (...) acc += x[i]`, evaluated by an exact scan (orderedsum.hip)
hipError_t launch_ordered_sum(const float *x, long long n, float *out, hipStream_t s);
// channelMixer (ipchmixer.cc:185-230) and rgbCurves (iprgbcurves.cc:116-143) on a PixArgs image; mat[9] as floats in `mixf`
struct MixArgs { float *dst[3]; size_t stride; int w, h; float m[9]; const float *lut[3]; };
hipError_t launch_channel_mixer(const MixArgs &a, hipStream_t s);
hipError_t launch_rgb_curves(const MixArgs &a, hipStream_t s);
// ---- hslEqualizer (hsl.hip; iphsl.cc:29-221) ----
struct HslCurve { const double *x, *y, *slope; int n; };      // FlatCurve's polyline (poly_x, poly_y, dyByDx) on the device
struct HslArgs {
    float *img[3]; size_t stride; int w, h;
    float ws1[3];                 // working-space row 1 as floats (Imagefloat::ws_)
    HslCurve curve[4];            // 0 saturation, 1 luminance, 2 hue curve over hue; 3 the fixed `coeff` curve of the saturation step
    float *mask;                  // w*h
    int which, to_rgb;
};
hipError_t launch_hsl_prepare(const HslArgs &a, hipStream_t s);
hipError_t launch_hsl_mask(const HslArgs &a, hipStream_t s);
hipError_t launch_hsl_apply(const HslArgs &a, hipStream_t s);
hipError_t launch_hsl_finish(const HslArgs &a, hipStream_t s);
// ---- Imagefloat RGB <-> LAB and labAdjustments (lab.hip; imagefloat.cc:841-970, iplabadjustments.cc:236-345) ----
struct LabArgs {
    float *img[3]; size_t stride; int w, h;      // r = a, g = L, b = b in LAB mode
    float ws[9], iws[9];                         // Imagefloat::ws_ / iws_
    const float *cachef, *cachefy;               // Color::cachef / cachefy
    const float *lcurve, *acurve, *bcurve;       // 32770 / 65536 / 65536 entries
    float chroma;
    unsigned *hist;                              // 65536 bins
};
hipError_t launch_rgb_to_lab(const LabArgs &a, hipStream_t s);
hipError_t launch_lab_to_rgb(const LabArgs &a, hipStream_t s);
hipError_t launch_lab_hist(const LabArgs &a, hipStream_t s);
hipError_t launch_lab_adjust(const LabArgs &a, hipStream_t s);
// ---- dual demosaic blend (dualdemosaic.hip; dual_demosaic_RT.cc:73-152, rt_algo.cc:315-498) ----
struct DualArgs {
    float *rgb[3]; size_t stride;                // first demosaicer's output, blended in place
    const float *raw; size_t raw_stride;
    int w, h; unsigned filters;
    const float *cachefy;
    float *L, *blend;                            // w*h each
    float threshold;
};
hipError_t launch_rgb2l(const DualArgs &a, hipStream_t s);
hipError_t launch_blend_mask(const DualArgs &a, hipStream_t s);
hipError_t launch_tile_stats(const DualArgs &a, int nH, int nW, int y0, int x0, int step, int ts, float *var, hipStream_t s);
hipError_t launch_contrast_threshold(const DualArgs &a, int ty, int tx, int ts, float *result, hipStream_t s);
hipError_t launch_bilinear_blend(const DualArgs &a, hipStream_t s);
// ---- VNG4 (vng4.hip; vng4_demosaic_RT.cc:32-397) ----
constexpr int VNG4_CODE_INTS = 320;              // per (row & 7, col & 1) cell, like the reference's 1280-byte slots
struct Vng4Args {
    const float *raw; size_t raw_stride;
    float *red, *green, *blue; size_t out_stride;
    float *image;                                // w*h*4: the four-colour image
    const int *code;                             // 16 * VNG4_CODE_INTS
    int w, h; unsigned filters, prefilters;
};
unsigned vng4_prefilters(unsigned filters);
void vng4_build_code(unsigned pf, int width, int *codes);
hipError_t launch_vng4(const Vng4Args &a, hipStream_t s);
hipError_t launch_dual_blend_planes(const DualArgs &a, const float *r2, const float *g2, const float *b2, size_t s2, hipStream_t s);
// ---- logEncoding (logenc.hip; iplogenc.cc:132-316) ----
struct LogEncArgs {
    float *img[3]; size_t stride; int w, h;
    double ws1[3];                // working-space row 1 (Color::rgbLuminance<double>)
    float gray, shadows_range, dynamic_range, linbase, blend;
    int satcontrol;
    int hlcompr;                  // highlightCompression > 0 (iplogenc.cc:148-170)
    float hlcompr_factor, compr_p, compr_s;
    float *Y, *Y2;                // w*h: smoothed norm / its guide
};
hipError_t launch_logenc_direct(const LogEncArgs &a, hipStream_t s);
hipError_t launch_logenc_prepare(const LogEncArgs &a, hipStream_t s);
hipError_t launch_logenc_blend(const LogEncArgs &a, hipStream_t s);
float logenc_find_gray(float source_gray, float target_gray);
// FlatCurve(points, periodic, ppn): the polyline getVal searches (x, y, slope[n-1]); false = identity / empty curve
bool flat_curve_polyline(const double *pts, int npts, bool periodic, int ppn, double identity, std::vector<double> &x, std::vector<double> &y, std::vector<double> &slope);
// N1: ARTOutputProfile fast path (iprgb2out.cc:152-172) and Imagefloat::getScanline (imagefloat.cc:125-170)
struct OutArgs {
    const float *src[3]; size_t src_stride; float *dst[3]; size_t dst_stride; int w, h;
    float m[9]; int linear; const float *lut; int lutsz; int *unsupported;       // rgb2out
    unsigned char *out; size_t out_stride_bytes; int bps, is_float;             // scanlines
};
hipError_t launch_rgb2out_matrix(const OutArgs &a, hipStream_t s);
hipError_t launch_scanlines(const OutArgs &a, hipStream_t s);
// the scanlines written by a few persistent workgroups straight into mapped (pinned) host memory: `a.out` = its device address
bool scanlines_host_ok(const OutArgs &a);
hipError_t launch_scanlines_host(const OutArgs &a, int workgroups, hipStream_t s);
// saturationVibrance (ipsaturation.cc:43-83)
struct SatArgs { float *dst[3]; size_t stride; int w, h; float saturation, vibrance; int vib; double ws1[3]; };
hipError_t launch_saturation_vibrance(const SatArgs &a, hipStream_t s);
// NEUTRAL tone curve (curves.cc:854-1038)
struct NeutralArgs {
    float *img[3]; size_t stride; int w, h;
    const float *lut;            // 65536-entry tone LUT
    const float *pq, *pq_inv;    // 65536-entry PQ LUTs (host-built)
    float *hues;                 // [4] rhue, bhue, yhue, ohue (device; written by launch_neutral_hues)
    float ws[9], iws[9], to_out[9], to_work[9];
    float whitecoeff;
    int tail_kind;               // as in PixArgs
    double tail_y;
    ParamCurve tail_pc;
    int no_lds_lut;              // as in PixArgs
    int cu_reserve;              // as in PixArgs
};
hipError_t launch_neutral_hues(const NeutralArgs &a, hipStream_t s);
hipError_t launch_tone_neutral(const NeutralArgs &a, hipStream_t s);
void build_pq_luts(float *pq65536, float *pq_inv65536);
void dninfo_band_stats(const float *mad_a, const float *mad_b, int nbands, bool aggressive, float out[6]);
void dninfo_reduce(const float info[9][16], bool aggressive, float ch_M[9], float max_r[9], float max_b[9], float out3[3]);

// ---- wavelet_decomposition (wavelet.hip) ----
struct WaveArgs {
    const float *src;      // analysis: input plane / previous low-pass; synthesis: low-pass in
    size_t src_stride;     // floats (level-0 analysis input only)
    float *lo;             // analysis: low-pass out; Haar synthesis: low-pass out (ping-pong)
    float *b1, *b2, *b3;   // detail subbands (w2*h2, contiguous rows)
    float *dst;            // level-0 synthesis destination plane
    size_t dst_stride;
    int w, h, w2, h2;
    int skip;
    float blend;
};
hipError_t launch_wavelet_analysis0(const WaveArgs &a, hipStream_t s);
hipError_t launch_wavelet_haar_analysis(const WaveArgs &a, hipStream_t s);
hipError_t launch_wavelet_haar_synthesis(const WaveArgs &a, hipStream_t s);
hipError_t launch_wavelet_synthesis0(const WaveArgs &a, hipStream_t s);

// ---- wavelet denoise (denoise.hip) ----
// RawImageSource::getImage (skip 1) + the matrix branch of convertColorSpace, read by the first pixel passes of ImProcFunctions::denoise
// straight from the demosaiced planes (artgpu_improc_denoise_fused): pixel (y, x) of the image = planes at (sy1 + y, sx1 + x), * mul, CLIP,
// camera -> working matrix with double accumulation -- the arithmetic of get_image_convert_kernel (pixelops.hip)
struct GetImageFuse {
    int on;
    const float *src[3];
    size_t stride;
    int sx1, sy1;
    float mul[3];
    int do_clip, has_mat;
    double mat[9];
};
struct DnPixArgs {
    GetImageFuse gi;               // rgb2yuv: where the image comes from when it has not been materialised
    float exp_scale, exp_black;    // yuv2rgb: ImProcFunctions::exposure (process STAGE_1) behind the last pass, exp_on != 0
    int exp_on;
    float *rgb[3];          // Imagefloat planes (in for rgb2yuv, out for yuv2rgb)
    size_t stride;
    float *L, *A, *B;       // LabImage planes, contiguous w*h
    int w, h;
    float gain, newGain;
    float gam, gamthresh, gamslope, igam, igamthresh, igamslope;
    const float *gamcurve, *igamcurve;
    float ws1[3];           // working-space matrix row 1 (luminance)
    float realred, realblue, qhighFactor;
    float pre_scale, post_scale;   // != 0: exposure compensation fused in front of rgb2yuv / behind yuv2rgb
    int lab_mode;                  // DenoiseParams::colorSpace == LAB (FTblockDN.cc:1996)
    int igam_lds_lo;               // yuv2rgb_lds: entries [lo, lo + 40704) of the inverse gamma table live in LDS (multiple of 4)
    float wpi[9], iws[9];          // LAB: working space <-> XYZ, float casts
    const float *cachef, *cachefy, *dn_gamma, *dn_igamma;   // LAB: 65536-entry LUTs
    int no_lds_lut;                // artgpu_set_option "lut_lds" 0
    int cu_reserve;                // as in PixArgs
};
// chroma noise-curve map (ipdenoise.cc:1113-1131 + FTblockDN.cc:1716-1777)
struct ChromaMapArgs {
    GetImageFuse gi;                      // (see DnPixArgs)
    const float *src[3]; size_t stride;   // full-resolution image
    int wid, hei;                         // (w+1)/2 x (h+1)/2
    int has_mat; double mat[9];           // convertColorSpace matrix applied to the subsampled copy
    float wpi[9];                         // working space -> XYZ, float casts
    const float *cachef;                  // 65536-entry Lab f() LUT (device)
    const float *curve;                   // 501-entry NoiseCurve LUT (device)
    float *out;                           // wid x hei
    int no_lds_lut;                       // artgpu_set_option "lut_lds" 0
    int cu_reserve;                       // as in PixArgs
};
hipError_t launch_chroma_map(const ChromaMapArgs &a, hipStream_t s);
bool flat_curve_sample(const double *pts, int npts, bool periodic, int ppn, double identity, int nout, double *out);
float noise_curve_lut(const double *pts, int npts, float lut[501]);
void build_cachef(float *lut65536);
void build_cachefy(float *lut65536);
void build_denoise_gamma_tabs(float *gtab65536, float *igtab65536);

struct ShrinkArgs {
    float *coef;            // bands of the decomposition being shrunk: [nsub][n]
    const float *coefL;     // bands of the L decomposition (AB only)
    float *sfave;           // [nsub][n]
    size_t n;
    const float *madL;      // [nsub] SQR(MadRgb) of the L bands
    const float *madab;     // [nsub] SQR(MadRgb) of the ab bands (AB only)
    const float *noisevar;  // per-coefficient noise variance map (n floats) or nullptr
    float noisevar_const;   // L: noisevarL when no map
    float noisevar_scale;   // AB: maxNoiseVarab
    float noisevar_ab;
    int useNoiseCCurve;
};
struct BlurArgs {
    const float *src;       // hblur: sfave ; vblur: hblur output
    float *dst;             // hblur output
    const float *sfave;     // vblur: unblurred shrink factors
    float *coef;            // vblur: coefficients updated in place
    size_t n;
    int w, h;
    int rad[10];            // blur radius per level
    int steady_div;         // hblur: steady state divides by len (boxblur.h:318 variant) instead of multiplying by 1/len
    int plain;              // vblur: store the blurred value to dst (all columns take the vector form); no coefficient update
    int level0;             // level of the first band passed (blockIdx.y = 0)
};
constexpr int HBLUR_MAX_RADIUS = 900;   // window of 256 + 2 * radius columns x 16 rows in the CU's 160 KB LDS
hipError_t launch_gamma_lut(float *lut, float gamma, float start, float slope, float divisor, float factor, hipStream_t s);
hipError_t launch_rgb2yuv(const DnPixArgs &a, hipStream_t s);
hipError_t launch_yuv2rgb(const DnPixArgs &a, hipStream_t s);
hipError_t launch_bishrink_AB(const ShrinkArgs &a, int nsub, hipStream_t s);
constexpr int MAD_SCRATCH_INTS_PER_BAND = 8 + 1024 + 4096 + 8;      // launch_mad: `histo` holds nsub * (65536 + this) ints
hipError_t launch_mad(const float *bands, size_t n, int nsub, int *histo, float *out, hipStream_t s);
hipError_t launch_shrink_sf(const ShrinkArgs &a, int nsub, bool ab, hipStream_t s);
hipError_t launch_hblur(const BlurArgs &a, int nsub, hipStream_t s);
hipError_t launch_vblur_combine(const BlurArgs &a, int nsub, hipStream_t s);
// ShrinkAllL / ShrinkAllAB in one pass over the coefficients (shrinkblur.hip): shrink factor, both box-blur directions and the coefficient
// update; `coef_out` may be `coef` (in place) or a second band set
struct FusedShrinkArgs {
    // a launch holds nL bands of L (ShrinkAllL) followed by nsub - nL chroma bands (ShrinkAllAB), channel after channel
    const float *coef;      // L bands to shrink: [nL][n]
    float *coef_out;        // where their updated coefficients go (may be `coef`; has to be another band set when chroma bands are in the launch:
                            // their factors read the L coefficients as the decomposition left them)
    float *coefC;           // chroma bands, updated in place: [nsub - nL][n]
    const float *coefL;     // the L decomposition's bands as the chroma factors see them
    size_t n;
    int w, h;
    const float *madL;      // [bands per channel] SQR(MadRgb) of the L bands
    const float *madab;     // SQR(MadRgb) of the chroma bands: channel c's at madab + c * mad_ch_stride
    const float *noisevar;  // chroma: per-coefficient noise variance map (n floats) or nullptr
    int noisevar_nonneg;    // the map is known to hold no negative value (the library computed it): see shrinkblur.hip, FASTEXP
    float noisevar_const;   // L: noisevarL
    float noisevar_scale;   // chroma: maxNoiseVarab
    float noisevar_ab[2];   // chroma, per channel (chroma curve off)
    int useNoiseCCurve;
    int rad[10];            // blur radius per level
    int level0;             // level of the first band of a channel
    int nsub;               // bands in the launch
    int nL;                 // of which L bands
    int nsub_ch;            // chroma bands per channel (0: all of them one channel)
    int mad_ch_stride;
    // filled in by launch_shrink_blur
    int nstrips;
    float *hand;            // hand-over slots between the strips of a band
    int wpad;               // columns of a slot row (whole blocks)
    int *progress;          // [nsub][nstrips] blocks a strip has handed down
    int *ticket;
    int *diag;              // pinned host words (or nullptr): {magic, band, strip, block, counter} written when a bounded wait gives up
    long long wait_ticks;   // how long a strip waits for the strip above before it gives up (100 MHz ticks; 0: five seconds)
    int stall_band, stall_strip;   // test hook (option dn_debug_stall): this strip never publishes its progress (-1: none)
    long long *prof;        // -DFS_PROFILE: cycle counters (step time, busy time per role and per wave, time spent waiting for the strip above)
};
bool shrink_blur_supported(int w, int h, const int *rad, int level0, int nsub);
size_t shrink_blur_scratch_floats(int w, int h, int nsub, int maxr);
hipError_t launch_shrink_blur(FusedShrinkArgs a, float *scratch, hipStream_t s);

// ---- DCT detail recovery (detail.hip) ----
struct DetailArgs {
    float *L;               // denoised luminance, updated in place by the gather kernel
    const float *Lin;       // luminance before reconstruction
    const float *tm_in, *tm_out;    // 64x64 tile masks
    const float *costab, *costab_t; // C[k][j] = cos(pi (j+1/2) k / 64) and its transpose
    float *blocks;          // numblox_H * numblox_W * 64 * 64
    int w, h, numblox_W, numblox_H;
    float detail_hi, detail_lo;
    int blur_rad;
    const float *mask;      // luminanceDetailThreshold > 0: detail mask (w x h), else nullptr
    float params_Ldetail;
};
hipError_t launch_detail_blocks(const DetailArgs &a, hipStream_t s);
hipError_t launch_detail_gather(const DetailArgs &a, hipStream_t s);

// ---- guided chroma smoothing (guided.hip) ----
struct GuidedArgs {
    float *rgb[3];         // Imagefloat planes (0..65535), updated in place
    size_t stride;
    int W, H;              // full resolution
    int w, h;              // statistics grid (W/s, H/s)
    float *in[3];          // normalised input copy (iR,iG,iB), W*H each
    float *guide;          // W*H
    float *chan[3];        // log-encoded channels, W*H each
    float *low[8];         // w*h planes: I1/meanI, corrI, p1/meanp/a[3], corrIp/b[3]
    double ws1[3];         // working-space matrix row 1 (double, TMatrix)
    float epsilon;
    int nch;               // 0 / 3: the three log channels of the smoothing tool; 1: plain single-channel guidedFilter (chan[0] = src)
    float *q; size_t q_stride;   // plain mode: destination plane
};
hipError_t launch_gf_prepare(const GuidedArgs &a, hipStream_t s);
hipError_t launch_gf_subsample(const GuidedArgs &a, hipStream_t s);
hipError_t launch_gf_ab(const GuidedArgs &a, hipStream_t s);
hipError_t launch_gf_finish(const GuidedArgs &a, hipStream_t s);
hipError_t launch_gf_finish_plain(const GuidedArgs &a, hipStream_t s);   // q = bilinear(mean a) * I + bilinear(mean b) (guidedfilter.cc:225-240)

// ---- NL-means stage (nlmeans.hip) ----
struct MaskArgs {
    const float *src; size_t src_stride;   // W x H input (values ~ 0..scaling)
    float *L2, *m2;                        // W/4 x H/4 scratch
    float *mask;                           // W x H output (contiguous)
    int W, H, w4, h4;
    float scaling, threshold, ceiling, factor;
};
struct GaussArgs {
    float *img, *tmp;                      // W x H contiguous, blurred in place; tmp = forward-pass storage
    double *tmp64;                         // sigma >= 25: double forward buffer (selects the double kernels)
    int W, H;
    double B, b[3], M[9];
    float Bf, bf[3], Mf[9];
};
struct NlmArgs {
    float *img; size_t img_stride;         // Y plane, replaced by the filtered result
    float *src;                            // padded source WW x HH (already divided by factor)
    float *mask;                           // W x H: detail mask -> weight scale
    float *SW;                             // W x H weight sums
    float *explut;                         // 8192 entries
    int W, H, WW, HH, border, search_radius, patch_radius, ntiles_x, ntiles_y;
    float factor, h2;
};
hipError_t launch_detail_mask(const MaskArgs &a, hipStream_t s);
hipError_t launch_gaussian(const GaussArgs &a, hipStream_t s);
hipError_t launch_gaussian3(float *img, float *tmp, int W, int H, float c0, float c1, hipStream_t s);   // 0.25 <= sigma < 0.6, in place
hipError_t launch_nlm(const NlmArgs &a, hipStream_t s);
bool nlm_group_supported(const NlmArgs &a);
hipError_t launch_nlm_group(const NlmArgs &a, hipStream_t s);   // nlm_sweep.hip: workgroup per tile, a search row of offsets in flight

} // namespace artgpu
