// art_amd/csrc/denoise.hip -- wavelet part of denoise::RGB_denoise on gfx950
// (reference: rtengine/FTblockDN.cc:569-603 MadRgb, 638-839 ShrinkAllL/AB, 1781-1823 gamma LUTs,
//  2084-2128 RGB->YUV, 2502-2550 YUV->RGB; rtengine/boxblur.h:558-743; rtengine/color.cc:1128-1161).
//
// All 3 x levels subbands of one decomposition are processed by ONE launch per step (blockIdx.y =
// subband), and every data-dependent scalar (MAD) stays on the device, so the host never syncs:
//   mad_hist / mad_finish : int32 histogram of min(65535,|int(x)|) with LDS-privatised low bins,
//                           exact median walk -> SQR(MadRgb) per subband                 (integer, exact)
//   shrink_sf_L / _AB     : shrink factor per coefficient; bulk lanes use the vector sleef exp,
//                           the N%4 tail the scalar one (FTblockDN.cc:673-683,770-785)
//   hblur                 : sliding-sum box blur along rows.  The fp32 recurrence is sequential
//                           along the row (association order is part of the result), so
//                           parallelism comes from the rows: one lane per row, 64-row x 64-column
//                           tiles transposed through LDS so that HBM sees coalesced rows.
//   vblur_combine         : the vertical recurrence, one lane per column (naturally coalesced),
//                           fused with the coefficient update c *= (sfd^2+sf^2)/(sfd+sf+eps).
// Everything is HBM-bound streaming; bytes per coefficient: sf 8-16 in + 4 out, hblur 4+4,
// vblur 4+4+4 in + 4 out.
#include <hip/hip_runtime.h>
#include "devmath.h"
#include "devsleef.h"
#include "kernels.h"

namespace artgpu {

// ---------------------------------------------------------------- gamma LUT (color.cc:1128-1161, SSE form)
__global__ void __launch_bounds__(256) gamma_lut_kernel(float *lut, float gamma, float start, float slope, float divisor, float factor)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 65536) return;
    const float gammav = 1.f / gamma;
    const float slopev = (slope / divisor) * factor;
    const float divisorv = xlogf_s(divisor);
    const float comparev = start * divisor;
    const int border = (int)(start * divisor);
    const int border1 = border - (border & 3), border2 = border1 + 4;
    const float iv = (float)i;
    float r;
    if (i < border1) {
        r = iv * slopev;
    } else if (i < border2) {
        const float r0 = iv * slopev;
        const float r1 = xexpf_v((xlogf_v(iv) - divisorv) * gammav) * factor;
        r = iv <= comparev ? r0 : r1;
    } else {
        r = xexpf_v_nocheck((xlogf_v_nocheck(iv) - divisorv) * gammav) * factor;
    }
    lut[i] = r;
}

__device__ __forceinline__ float gammaf_s(float x, float gamma, float start, float slope)
{
    return x <= start ? x * slope : xexpf_s(xlogf_s(x) / gamma);
}

// ---------------------------------------------------------------- Lab helpers (LAB colour-space mode of RGB_denoise)
// LUTf::operator[](float) of a LUT constructed with flags 0: extrapolates on both sides (LUT.h:436-459)
__device__ __forceinline__ float lutf_noclip(const float *__restrict__ data, float index)
{
    int idx = (int)index;
    if (index < 0.f || !(index == index)) idx = 0;
    else if (index > 65534.f) idx = 65534;
    const float diff = index - (float)idx;
    const float p1 = data[idx], p2 = data[idx + 1] - p1;
    return p1 + p2 * diff;
}
// Color::computeXYZ2Lab (color.cc:1247-1259)
__device__ __forceinline__ float xyz2lab_f(const float *__restrict__ cachef, float f)
{
    if (f != f) return f;
    if (f < 0.f) return (float)(327.68 * (((24389.0 / 27.0) * (double)f / (double)65535.f + 16.0) / 116.0));
    if (f > 65535.f) return 327.68f * xcbrtf_s(f / 65535.f);
    return lutf_lookup<false>(cachef, 65536, f);
}
// Color::rgb2lab with a float matrix = rgbxyz + XYZ2Lab (color.h:630-636, color.cc:1262-1275,1382-1397)
__device__ __forceinline__ void rgb2lab_dev(const DnPixArgs &a, float R, float G, float B, float &l, float &la, float &lb)
{
    const float X = a.wpi[0] * R + a.wpi[1] * G + a.wpi[2] * B, Y = a.wpi[3] * R + a.wpi[4] * G + a.wpi[5] * B, Z = a.wpi[6] * R + a.wpi[7] * G + a.wpi[8] * B;
    const float x = X / 0.9642f, z = Z / 0.8249f, y = Y;
    const float fx = xyz2lab_f(a.cachef, x), fy = xyz2lab_f(a.cachef, y), fz = xyz2lab_f(a.cachef, z);
    if (y != y) l = y;
    else if (y < 0.f) l = (float)(327.68 * ((24389.0 / 27.0) * (double)y / (double)65535.f));
    else if (y > 65535.f) l = 327.68f * (116.f * xcbrtf_s(y / 65535.f) - 16.f);
    else l = lutf_lookup<false>(a.cachefy, 65536, y);
    la = 500.0f * (fx - fy);
    lb = 200.0f * (fy - fz);
}
// Color::lab2rgb = Lab2XYZ + xyz2rgb (color.h:638-644,767-770, color.cc:1203-1214)
__device__ __forceinline__ float f2xyz_f(float f)
{
    const float epsilonExpInv3f = (float)(6.0 / 29.0), kappaInvf = (float)(27.0 / 24389.0);
    return (f > epsilonExpInv3f) ? f * f * f : (116.f * f - 16.f) * kappaInvf;
}
__device__ __forceinline__ void lab2rgb_dev(const DnPixArgs &a, float l, float la, float lb, float &R, float &G, float &B)
{
    const float c1By116 = (float)(1.0 / 116.0), c16By116 = (float)(16.0 / 116.0);
    const float LL = l / 327.68f, aa = la / 327.68f, bb = lb / 327.68f;
    const float fy = (c1By116 * LL) + c16By116;
    const float fx = (0.002f * aa) + fy;
    const float fz = fy - (0.005f * bb);
    const float x = 65535.0f * f2xyz_f(fx) * 0.9642f;
    const float z = 65535.0f * f2xyz_f(fz) * 0.8249f;
    const float y = ((double)LL > 8.0) ? 65535.0f * fy * fy * fy : (float)((double)(65535.0f * LL) / (24389.0 / 27.0));
    R = a.iws[0] * x + a.iws[1] * y + a.iws[2] * z;
    G = a.iws[3] * x + a.iws[4] * y + a.iws[5] * z;
    B = a.iws[6] * x + a.iws[7] * y + a.iws[8] * z;
}

// ---------------------------------------------------------------- RGB -> gamma -> YUV (FTblockDN.cc:2084-2128)
// Two launch shapes of each pixel pass: the plain 2-D grid, and (large frames, gamma LUT in use) one persistent 1024-thread
// workgroup per CU with the lower part of the 65536-entry gamma LUT in LDS (lutf_lookup_lds): 0.61 -> ~0.35 ms at 45 MP.
// getImage + convertColorSpace of one pixel whose three raw plane values have been loaded (get_image_convert_kernel, pixelops.hip:55-88)
__device__ __forceinline__ void gi_convert(const GetImageFuse &g, float &r, float &gg, float &b)
{
    float r1 = 0.f, g1 = 0.f, b1 = 0.f;
    r1 += r; g1 += gg; b1 += b;
    r1 *= g.mul[0]; g1 *= g.mul[1]; b1 *= g.mul[2];
    if (g.do_clip) {
        r1 = std_max(0.f, std_min(r1, 65535.f)); g1 = std_max(0.f, std_min(g1, 65535.f)); b1 = std_max(0.f, std_min(b1, 65535.f));
    }
    if (g.has_mat) {
        const double dr = r1, dg = g1, db = b1;
        r1 = (float)(g.mat[0] * dr + g.mat[1] * dg + g.mat[2] * db);
        g1 = (float)(g.mat[3] * dr + g.mat[4] * dg + g.mat[5] * db);
        b1 = (float)(g.mat[6] * dr + g.mat[7] * dg + g.mat[8] * db);
    }
    r = r1; gg = g1; b = b1;
}
template <bool LDS>
__device__ __forceinline__ float gam_lookup(const float *lds, const float *__restrict__ lut, float v, int lo = 0)
{
    return LDS ? lutf_lookup_lds<false>(lds, lut, 65536, v, lo) : lutf_lookup<false>(lut, 65536, v);
}
template <bool LDS>
__device__ __forceinline__ void rgb2yuv_px(const DnPixArgs &a, const float *lds, int y, int x, float r0, float g0, float b0)
{
    const long long t = (long long)y * a.w + x;
    if (a.gi.on) gi_convert(a.gi, r0, g0, b0);
    if (a.pre_scale != 0.f) {   // fused ImProcFunctions::expcomp(+ecomp) (ipexposure.cc:56-70): 4-lane groups then scalar tail
        if (x < (a.w / 4) * 4) { r0 = sse_max(r0 * a.pre_scale - 0.f, 0.f); g0 = sse_max(g0 * a.pre_scale - 0.f, 0.f); b0 = sse_max(b0 * a.pre_scale - 0.f, 0.f); }
        else { r0 = std_max(r0 * a.pre_scale - 0.f, 0.f); g0 = std_max(g0 * a.pre_scale - 0.f, 0.f); b0 = std_max(b0 * a.pre_scale - 0.f, 0.f); }
    }
    float X = a.gain * r0, Y = a.gain * g0, Z = a.gain * b0;
    if (a.lab_mode) { X = lutf_noclip(a.dn_igamma, X); Y = lutf_noclip(a.dn_igamma, Y); Z = lutf_noclip(a.dn_igamma, Z); }   // L2094-2098
    if (a.gam > 1.f) {
        if (X > 0.f) X = X < 65535.f ? gam_lookup<LDS>(lds, a.gamcurve, X) : (gammaf_s(X / 65535.f, a.gam, a.gamthresh, a.gamslope) * 65535.f);
        if (Y > 0.f) Y = Y < 65535.f ? gam_lookup<LDS>(lds, a.gamcurve, Y) : (gammaf_s(Y / 65535.f, a.gam, a.gamthresh, a.gamslope) * 65535.f);
        if (Z > 0.f) Z = Z < 65535.f ? gam_lookup<LDS>(lds, a.gamcurve, Z) : (gammaf_s(Z / 65535.f, a.gam, a.gamthresh, a.gamslope) * 65535.f);
    }
    float l = X * a.ws1[0] + Y * a.ws1[1] + Z * a.ws1[2];
    float va = X - l, ub = l - Z;   // v -> labdn->a, u -> labdn->b
    if (a.lab_mode) rgb2lab_dev(a, X, Y, Z, l, va, ub);    // rgb2lab(X, Y, Z, l, v, u, wpi), L2114-2116
    a.L[t] = l;
    a.A[t] = va;
    a.B[t] = ub;
}
__global__ void __launch_bounds__(256) rgb2yuv_kernel(DnPixArgs a)
{
    FOR_IMAGE_XY(y, x, a.w, a.h) {
        if (a.gi.on) {
            const size_t si = (size_t)(a.gi.sy1 + y) * a.gi.stride + a.gi.sx1 + x;
            rgb2yuv_px<false>(a, nullptr, y, x, a.gi.src[0][si], a.gi.src[1][si], a.gi.src[2][si]);
        } else {
            const size_t si = (size_t)y * a.stride + x;
            rgb2yuv_px<false>(a, nullptr, y, x, a.rgb[0][si], a.rgb[1][si], a.rgb[2][si]);
        }
    }
}
__global__ void __launch_bounds__(1024) rgb2yuv_lds_kernel(DnPixArgs a)
{
    extern __shared__ float dn_lut_lds[];
    lut_lds_fill(dn_lut_lds, a.gamcurve, 1024);
    for (int yb = blockIdx.x; yb < a.h; yb += LDSK_ROWS * gridDim.x)
        for (int x0 = 0; x0 < a.w; x0 += LDSK_PX * 1024) {
            float r[LDSK_ROWS * LDSK_PX], g[LDSK_ROWS * LDSK_PX], b[LDSK_ROWS * LDSK_PX];
#pragma unroll
            for (int k = 0; k < LDSK_ROWS * LDSK_PX; ++k) {
                const int x = x0 + (k % LDSK_PX) * 1024 + (int)threadIdx.x, yr = yb + (k / LDSK_PX) * (int)gridDim.x, y = yr < a.h ? yr : a.h - 1;
                const int xc = x < a.w ? x : a.w - 1;
                const size_t si = a.gi.on ? (size_t)(a.gi.sy1 + y) * a.gi.stride + a.gi.sx1 + xc : (size_t)y * a.stride + xc;
                const float *const p0 = a.gi.on ? a.gi.src[0] : a.rgb[0], *const p1 = a.gi.on ? a.gi.src[1] : a.rgb[1], *const p2 = a.gi.on ? a.gi.src[2] : a.rgb[2];
                r[k] = p0[si]; g[k] = p1[si]; b[k] = p2[si];
            }
#pragma unroll
            for (int k = 0; k < LDSK_ROWS * LDSK_PX; ++k) {
                const int x = x0 + (k % LDSK_PX) * 1024 + (int)threadIdx.x, yr = yb + (k / LDSK_PX) * (int)gridDim.x, y = yr < a.h ? yr : a.h - 1;
                if (x < a.w && yr < a.h) rgb2yuv_px<true>(a, dn_lut_lds, y, x, r[k], g[k], b[k]);
            }
        }
}

// ---------------------------------------------------------------- YUV -> inverse gamma -> RGB (FTblockDN.cc:2502-2550)
template <bool LDS>
__device__ __forceinline__ void yuv2rgb_px(const DnPixArgs &a, const float *lds, int y, int x, float Lv, float av, float bv)
{
    const float c_h = sqrtf(sqr(av) + sqr(bv));
    if (c_h > 3000.f) {
        av *= 1.f + a.qhighFactor * a.realred / 100.f;
        bv *= 1.f + a.qhighFactor * a.realblue / 100.f;
    }
    float Z = Lv - bv;
    float X = av + Lv;
    float Y = (Lv - X * a.ws1[0] - Z * a.ws1[2]) / a.ws1[1];
    if (a.lab_mode) lab2rgb_dev(a, Lv, av, bv, X, Y, Z);    // L2522-2524
    if (a.gam > 1.f) {
        if (X > 0.f) X = X < 65536.f ? gam_lookup<LDS>(lds, a.igamcurve, X, a.igam_lds_lo) : (gammaf_s(X / 65535.f, a.igam, a.igamthresh, a.igamslope) * 65535.f);
        if (Y > 0.f) Y = Y < 65536.f ? gam_lookup<LDS>(lds, a.igamcurve, Y, a.igam_lds_lo) : (gammaf_s(Y / 65535.f, a.igam, a.igamthresh, a.igamslope) * 65535.f);
        if (Z > 0.f) Z = Z < 65536.f ? gam_lookup<LDS>(lds, a.igamcurve, Z, a.igam_lds_lo) : (gammaf_s(Z / 65535.f, a.igam, a.igamthresh, a.igamslope) * 65535.f);
    }
    const size_t di = (size_t)y * a.stride + x;
    if (a.lab_mode) { X = lutf_noclip(a.dn_gamma, X); Y = lutf_noclip(a.dn_gamma, Y); Z = lutf_noclip(a.dn_gamma, Z); }   // L2533-2537
    float ro = a.newGain * X, go = a.newGain * Y, bo = a.newGain * Z;
    if (a.post_scale != 0.f) {  // fused ImProcFunctions::expcomp(-ecomp)
        if (x < (a.w / 4) * 4) { ro = sse_max(ro * a.post_scale - 0.f, 0.f); go = sse_max(go * a.post_scale - 0.f, 0.f); bo = sse_max(bo * a.post_scale - 0.f, 0.f); }
        else { ro = std_max(ro * a.post_scale - 0.f, 0.f); go = std_max(go * a.post_scale - 0.f, 0.f); bo = std_max(bo * a.post_scale - 0.f, 0.f); }
    }
    if (a.exp_on) {             // fused ImProcFunctions::exposure (ipexposure.cc:28-79, exposure_kernel): v * exp_scale - black, floored at 0
        const float er = ro * a.exp_scale - a.exp_black, eg = go * a.exp_scale - a.exp_black, eb = bo * a.exp_scale - a.exp_black;
        if (x < (a.w / 4) * 4) { ro = sse_max(er, 0.f); go = sse_max(eg, 0.f); bo = sse_max(eb, 0.f); }
        else { ro = std_max(er, 0.f); go = std_max(eg, 0.f); bo = std_max(eb, 0.f); }
    }
    a.rgb[0][di] = ro;
    a.rgb[1][di] = go;
    a.rgb[2][di] = bo;
}
__global__ void __launch_bounds__(256) yuv2rgb_kernel(DnPixArgs a)
{
    FOR_IMAGE_XY(y, x, a.w, a.h) {
        const long long t = (long long)y * a.w + x;
        yuv2rgb_px<false>(a, nullptr, y, x, a.L[t], a.A[t], a.B[t]);
    }
}
__global__ void __launch_bounds__(1024) yuv2rgb_lds_kernel(DnPixArgs a)
{
    extern __shared__ float dn_lut_lds[];
    lut_lds_fill(dn_lut_lds, a.igamcurve, 1024, a.igam_lds_lo);
    for (int yb = blockIdx.x; yb < a.h; yb += LDSK_ROWS * gridDim.x)
        for (int x0 = 0; x0 < a.w; x0 += LDSK_PX * 1024) {
            float l[LDSK_ROWS * LDSK_PX], av[LDSK_ROWS * LDSK_PX], bv[LDSK_ROWS * LDSK_PX];
#pragma unroll
            for (int k = 0; k < LDSK_ROWS * LDSK_PX; ++k) {
                const int x = x0 + (k % LDSK_PX) * 1024 + (int)threadIdx.x, yr = yb + (k / LDSK_PX) * (int)gridDim.x, y = yr < a.h ? yr : a.h - 1;
                const long long t = (long long)y * a.w + (x < a.w ? x : a.w - 1);
                l[k] = a.L[t]; av[k] = a.A[t]; bv[k] = a.B[t];
            }
#pragma unroll
            for (int k = 0; k < LDSK_ROWS * LDSK_PX; ++k) {
                const int x = x0 + (k % LDSK_PX) * 1024 + (int)threadIdx.x, yr = yb + (k / LDSK_PX) * (int)gridDim.x, y = yr < a.h ? yr : a.h - 1;
                if (x < a.w && yr < a.h) yuv2rgb_px<true>(a, dn_lut_lds, y, x, l[k], av[k], bv[k]);
            }
        }
}

// ---------------------------------------------------------------- MadRgb (FTblockDN.cc:569-603)
#ifndef MAD_LDS_BINS_OVERRIDE
#define MAD_LDS_BINS_OVERRIDE 4096
#endif
constexpr int MAD_LDS_BINS = MAD_LDS_BINS_OVERRIDE;
#ifndef MAD_GRID
#define MAD_GRID 192
#endif
constexpr int MAD_SAMPLE_BINS = 4096;
constexpr int MAD_WIN_STRIDE = MAD_SCRATCH_INTS_PER_BAND;       // ints of scratch per band behind the full histograms (kernels.h)
__global__ void __launch_bounds__(256) mad_hist_kernel(const float *bands, size_t n, int *histo /*[nsub][65536]*/, const int *done /*[nsub] or null*/)
{
    if (done && done[blockIdx.y * MAD_WIN_STRIDE]) return;       // the windowed pass below already has this band's median
    // The first MAD_LDS_BINS bins live in LDS (16 KB: more LDS costs more in occupancy than it saves in global atomics); the
    // long tail goes straight to global memory.  The loop is bound by the latency of its loads, so eight are kept in flight.
    __shared__ int h[MAD_LDS_BINS];
    const int sub = blockIdx.y;
    const float *data = bands + (size_t)sub * n;
    int *gh = histo + (size_t)sub * 65536;
    for (int i = threadIdx.x; i < MAD_LDS_BINS; i += 256) h[i] = 0;
    __syncthreads();
    constexpr int U = 8;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i0 = blockIdx.x * (size_t)256 + threadIdx.x; i0 < n; i0 += stride * U) {
        float x[U];
#pragma unroll
        for (int k = 0; k < U; ++k) { const size_t i = i0 + k * stride; x[k] = data[i < n ? i : n - 1]; }     // (clamped, unconditional: see mad_window_kernel)
#pragma unroll
        for (int k = 0; k < U; ++k) {
            if (i0 + k * stride >= n) break;
            // min(65535, abs((int)x)) of the reference (FTblockDN.cc:587) for every x an int can hold; beyond that (|x| >= 2^31, Inf,
            // NaN) the reference's conversion is undefined and its index out of range -- here those coefficients land in the top bin
            // (clamping in float first; fminf returns the number when one operand is NaN)
            const int v = (int)fminf(fabsf(x[k]), 65535.f);
            if (v < MAD_LDS_BINS) atomicAdd(&h[v], 1);
            else atomicAdd(&gh[v], 1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < MAD_LDS_BINS; i += 256)
        if (h[i]) atomicAdd(&gh[i], h[i]);
}

// ---- MadRgb without the full histogram.  The median walk of the reference (`while (count < half) count += histo[median++]`) only
// needs (a) how many coefficients lie below some bin L that is itself below the median bin, and (b) the exact counts of the bins
// from L up to the median bin.  A 1/32 sub-sample brackets the median bin: L and the window width Wn come from the sub-sample's
// quantiles at half -+ 3 sqrt(N) (six standard deviations of the sampling error), 8 <= Wn <= 1024.  The exact pass then COUNTS
// what is below L with a per-thread add, the first 8 window bins in packed per-thread byte counters, and the bins 8 .. Wn with LDS
// atomics: a peaked distribution (most chroma bands: half of all coefficients within a few bins of the median) has a window of
// 8 and never reaches an atomic -- the full histogram is bound by LDS atomic throughput there, ~1 lane per clock --, a broad one has
// a wide window but then only a percent or so of the coefficients fall into it.
// If the true median bin is not inside the window after all (or the bracket is wider than 1024 bins / beyond the sub-sample's 4096)
// the band's `done` flag stays 0 and the full histogram above runs for it: the result is the reference's in every case.
// scratch per band (ints): [0] done, [1] L, [2] below, [3] sampled total, [4] Wn, [8 .. 8+1024) window, then the sub-sample histogram
constexpr int MAD_WMAX = 1024, MAD_REG_BINS = 8, MAD_SAMPLE_OFF = 8 + MAD_WMAX;
static_assert(MAD_SAMPLE_OFF + MAD_SAMPLE_BINS + 1 <= MAD_WIN_STRIDE, "MAD_SCRATCH_INTS_PER_BAND too small");
__global__ void __launch_bounds__(256) mad_sample_kernel(const float *bands, size_t n, int *scr)
{
    __shared__ int h[MAD_SAMPLE_BINS + 1];
    const int sub = blockIdx.y;
    const float *data = bands + (size_t)sub * n;
    int *s = scr + (size_t)sub * MAD_WIN_STRIDE;
    for (int i = threadIdx.x; i <= MAD_SAMPLE_BINS; i += 256) h[i] = 0;
    __syncthreads();
    // every 32nd chunk of 256 consecutive coefficients; eight chunks per iteration with their loads first (a load under `if (i < n)` is waited
    // for before the next one is issued: one load in flight per wave, 46 us for 63 MB)
    constexpr int U = 8;
    const size_t cstride = (size_t)gridDim.x * 32;
    for (size_t chunk0 = (size_t)blockIdx.x * 32; chunk0 * 256 < n; chunk0 += cstride * U) {
        float x[U];
#pragma unroll
        for (int k = 0; k < U; ++k) { const size_t i = (chunk0 + k * cstride) * 256 + threadIdx.x; x[k] = data[i < n ? i : n - 1]; }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const size_t i = (chunk0 + k * cstride) * 256 + threadIdx.x;
            if (i < n) {
                const int v = (int)fminf(fabsf(x[k]), 65535.f);
                atomicAdd(&h[v < MAD_SAMPLE_BINS ? v : MAD_SAMPLE_BINS], 1);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i <= MAD_SAMPLE_BINS; i += 256)
        if (h[i]) atomicAdd(&s[MAD_SAMPLE_OFF + i], h[i]);
}
__global__ void __launch_bounds__(256) mad_pick_kernel(int *scr)
{
    __shared__ int part[256];
    int *s = scr + (size_t)blockIdx.x * MAD_WIN_STRIDE;
    const int *h = s + MAD_SAMPLE_OFF;
    constexpr int PER = MAD_SAMPLE_BINS / 256;
    int p = 0;
    for (int i = 0; i < PER; ++i) p += h[threadIdx.x * PER + i];
    part[threadIdx.x] = p;
    __syncthreads();
    if (threadIdx.x == 0) {
        int total = h[MAD_SAMPLE_BINS];
        for (int i = 0; i < 256; ++i) total += part[i];
        const int half = total / 2, d = 3 * (int)sqrtf((float)total) + 1;
        // first bin whose cumulative count reaches `target` (whole 16-bin chunks first), -1 if none below the overflow bin
        auto first_reaching = [&](int target) {
            int count = 0, chunk = 0;
            while (chunk < 256 && count + part[chunk] < target) { count += part[chunk]; ++chunk; }
            if (chunk == 256) return -1;
            int b = chunk * PER;
            for (; b < (chunk + 1) * PER; ++b) { count += h[b]; if (count >= target) break; }
            return b;
        };
        int L = -1, Wn = 0;
        if (total > 0) {
            const int lo = half - d <= 0 ? 0 : first_reaching(half - d), hi = first_reaching(half + d);
            if (lo >= 0 && hi >= 0) {
                L = lo > 0 ? lo - 1 : 0;
                Wn = hi + 2 - L;
                Wn = Wn < MAD_REG_BINS ? MAD_REG_BINS : Wn;
                if (Wn > MAD_WMAX) L = -1;
            }
        }
        s[0] = 0; s[1] = L; s[2] = 0; s[3] = total; s[4] = Wn;
    }
}
#ifndef MAD_WINDOW_LOADS
#define MAD_WINDOW_LOADS 8
#endif
__global__ void __launch_bounds__(256) mad_window_kernel(const float *bands, size_t n, int *scr)
{
    __shared__ int win[MAD_WMAX];
    __shared__ int red[MAD_REG_BINS + 1];
    const int sub = blockIdx.y;
    int *s = scr + (size_t)sub * MAD_WIN_STRIDE;
    const int L = s[1], Wn = s[4];
    if (L < 0) return;
    const float *data = bands + (size_t)sub * n;
    for (int i = threadIdx.x; i < Wn; i += 256) win[i] = 0;
    if (threadIdx.x <= MAD_REG_BINS) red[threadIdx.x] = 0;
    __syncthreads();
    constexpr int U = MAD_WINDOW_LOADS;
    static_assert(U * 16 < 256, "the packed byte counters are flushed every 16 iterations");
    const size_t stride = (size_t)gridDim.x * 256;
    int below = 0, cnt[MAD_REG_BINS];
    unsigned long long packed = 0;          // eight 8-bit counters: window bins 0 .. 7
#pragma unroll
    for (int b = 0; b < MAD_REG_BINS; ++b) cnt[b] = 0;
    auto flush = [&]() {
#pragma unroll
        for (int b = 0; b < MAD_REG_BINS; ++b) cnt[b] += (int)((packed >> (8 * b)) & 0xffull);
        packed = 0;
    };
    int it = 0;
    for (size_t i0 = blockIdx.x * (size_t)256 + threadIdx.x; i0 < n; i0 += stride * U, ++it) {
        // unconditional loads from a clamped index (a load under `if (i < n)` is not hoisted above the previous element's
        // arithmetic: one load in flight per wave instead of U); elements past the end count in no bin
        float x[U];
#pragma unroll
        for (int k = 0; k < U; ++k) { const size_t i = i0 + k * stride; x[k] = data[i < n ? i : n - 1]; }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            int w = (int)fminf(fabsf(x[k]), 65535.f) - L;           // the bin of mad_hist_kernel, relative to the window
            w = i0 + k * stride < n ? w : 0x10000;
            below += w < 0;
            packed += (unsigned)w < (unsigned)MAD_REG_BINS ? 1ull << (8 * w) : 0ull;
            if (w >= MAD_REG_BINS && w < Wn) atomicAdd(&win[w], 1);
        }
        if ((it & 15) == 15) flush();
    }
    flush();
    // wave sums, one LDS atomic per wave and counter, one global atomic per workgroup and counter
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        below += __shfl_down(below, o);
#pragma unroll
        for (int b = 0; b < MAD_REG_BINS; ++b) cnt[b] += __shfl_down(cnt[b], o);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&red[MAD_REG_BINS], below);
#pragma unroll
        for (int b = 0; b < MAD_REG_BINS; ++b) atomicAdd(&red[b], cnt[b]);
    }
    __syncthreads();
    if (threadIdx.x < MAD_REG_BINS && red[threadIdx.x]) atomicAdd(&s[8 + threadIdx.x], red[threadIdx.x]);
    if (threadIdx.x == MAD_REG_BINS && red[MAD_REG_BINS]) atomicAdd(&s[2], red[MAD_REG_BINS]);
    for (int i = MAD_REG_BINS + threadIdx.x; i < Wn; i += 256)
        if (win[i]) atomicAdd(&s[8 + i], win[i]);
}
__global__ void __launch_bounds__(64) mad_window_finish_kernel(int *scr, int datalen, float *out)
{
    if (threadIdx.x) return;
    int *s = scr + (size_t)blockIdx.x * MAD_WIN_STRIDE;
    if (datalen <= 1) { out[blockIdx.x] = 0.f; s[0] = 1; return; }
    const int L = s[1], Wn = s[4], half = datalen / 2;
    int count = s[2];
    if (L < 0 || count >= half) return;              // no bracket, or the median bin lies below the window
    for (int b = 0; b < Wn; ++b) {
        const int hb = s[8 + b];
        count += hb;
        if (count >= half) {
            // the walk of mad_finish_kernel stopped with median = L + b + 1, count_ = count - h[median - 1]
            const int median = L + b + 1, count_ = count - hb;
            const float q = ((median - 1) + (half - count_) / ((float)(count - count_)));
            const float r = (float)((double)q / 0.6745);
            out[blockIdx.x] = r * r;
            s[0] = 1;
            return;
        }
    }
}

// one workgroup per subband: exact median walk; out[sub] = SQR(MadRgb)
__global__ void __launch_bounds__(256) mad_finish_kernel(const int *histo, int datalen, float *out, const int *done)
{
    __shared__ int part[256];
    const int sub = blockIdx.x;
    if (done && done[sub * MAD_WIN_STRIDE]) return;
    const int *h = histo + (size_t)sub * 65536;
    int s = 0;
    for (int i = 0; i < 256; ++i) s += h[threadIdx.x * 256 + i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float r = 0.f;
        if (datalen > 1) {
            const int half = datalen / 2;
            int count = 0, chunk = 0;
            // `while (count < half) { count += histo[median]; ++median; }` -- skip whole 256-bin chunks first
            while (chunk < 256 && count + part[chunk] < half) { count += part[chunk]; ++chunk; }
            int median = chunk * 256;
            while (count < half) { count += h[median]; ++median; }
            const int count_ = count - h[median - 1];
            const float q = ((median - 1) + (half - count_) / ((float)(count - count_)));
            r = (float)((double)q / 0.6745);
        }
        out[sub] = r * r;
    }
}

// ---------------------------------------------------------------- shrink factors
__global__ void __launch_bounds__(256) shrink_sf_L_kernel(ShrinkArgs a)
{
    const int sub = blockIdx.y, level = sub / 3;
    const float *c = a.coef + (size_t)sub * a.n;
    float *sf = a.sfave + (size_t)sub * a.n;
    const float mad_L = a.madL[sub];
    const float levelFactor = mad_L * 5.f / (float)(level + 1);
    const float eps = 0.01f;
    const size_t nv4 = (a.n / 4) * 4;
    // four coefficients per thread and iteration, their loads issued before any arithmetic (one per iteration left the kernel at
    // 3.8 TB/s waiting for its own loads)
    constexpr int U = 4;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i0 = blockIdx.x * (size_t)256 + threadIdx.x; i0 < a.n; i0 += stride * U) {
        float cv[U], nvv[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const size_t i = i0 + k * stride, ic = i < a.n ? i : a.n - 1;
            cv[k] = c[ic];
            nvv[k] = a.noisevar ? a.noisevar[ic] : a.noisevar_const;
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const size_t i = i0 + k * stride;
            const float nv = nvv[k];
            const float mag = sqr(cv[k]);
            float r;
            if (i < nv4) {
                const float madv = nv * levelFactor;
                r = mag / (mag + madv * xexpf_v(-mag / (9.0f * madv)) + eps);
            } else {
                r = mag / (mag + levelFactor * nv * xexpf_s(-mag / (9 * levelFactor * nv)) + eps);
            }
            if (i < a.n) sf[i] = r;
        }
    }
}

__global__ void __launch_bounds__(256) shrink_sf_AB_kernel(ShrinkArgs a)
{
    const int sub = blockIdx.y;
    const float *c = a.coef + (size_t)sub * a.n;
    const float *cL = a.coefL + (size_t)sub * a.n;
    float *sf = a.sfave + (size_t)sub * a.n;
    const float mad_L = a.madL[sub];
    float madab = a.madab[sub];
    madab = a.useNoiseCCurve ? madab : madab * a.noisevar_ab;
    const float rmadLm9 = 1.f / (mad_L * 9.f);
    const size_t nv4 = (a.n / 4) * 4;
    constexpr int U = 4;          // (see shrink_sf_L_kernel)
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i0 = blockIdx.x * (size_t)256 + threadIdx.x; i0 < a.n; i0 += stride * U) {
        float cv[U], clv[U], nvv[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const size_t i = i0 + k * stride, ic = i < a.n ? i : a.n - 1;
            cv[k] = c[ic];
            clv[k] = cL[ic];
            // noisevarchrom[i] = useNoiseCCurve ? maxNoiseVarab * ccalc[i] : 1 (FTblockDN.cc:2124)
            nvv[k] = a.noisevar ? a.noisevar[ic] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const size_t i = i0 + k * stride;
            const float nvc = a.noisevar ? a.noisevar_scale * nvv[k] : 1.f;
            const float mag_ab = sqr(cv[k]);
            float r;
            if (i < nv4) {
                const float mad_abv = nvc * madab;
                const float mag_L = sqr(clv[k]) * rmadLm9;
                r = 1.f - xexpf_v(-(mag_ab / mad_abv) - mag_L);
            } else {
                const float mag_L = sqr(clv[k]);
                r = 1.f - xexpf_s(-(mag_ab / (nvc * madab)) - (mag_L / (9.f * mad_L)));
            }
            if (i < a.n) sf[i] = r;
        }
    }
}

// WaveletDenoiseAll_BiShrinkAB, levels below the top one (FTblockDN.cc:1049-1090): point-wise shrink in place, no box blur.
// blockIdx.y = band; a.madab already holds SQR(MadRgb) of the untouched bands.
__global__ void __launch_bounds__(256) bishrink_AB_kernel(ShrinkArgs a)
{
    const int sub = blockIdx.y;
    float *c = a.coef + (size_t)sub * a.n;
    const float *cL = a.coefL + (size_t)sub * a.n;
    const float mad_Lr = a.madL[sub];
    const float mab = a.madab[sub];
    const float mad_abr = a.useNoiseCCurve ? a.noisevar_ab * mab : sqr(a.noisevar_ab) * mab;
    const float rmad_Lm9 = 1.f / (mad_Lr * 9.f);
    const size_t nv4 = (a.n / 4) * 4;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < a.n; i += (size_t)gridDim.x * 256) {
        const float nvc = a.noisevar ? a.noisevar_scale * a.noisevar[i] : 1.f;
        const float tempab = c[i];
        const float mag_ab = sqr(tempab);
        if (i < nv4) {
            const float mad_abv = nvc * mad_abr;
            const float mag_L = sqr(cL[i]) * rmad_Lm9;
            c[i] = tempab * sqr(1.f - xexpf_v(-(mag_ab / mad_abv) - (mag_L)));
        } else {
            const float mag_L = sqr(cL[i]);
            c[i] = tempab * sqr(1.f - xexpf_s(-(mag_ab / (nvc * mad_abr)) - (mag_L / (9.f * mad_Lr))));
        }
    }
}

// ---------------------------------------------------------------- horizontal box blur (boxblur.h:565-600)
// 16 rows per wave (not 64): the running sum is serial along the row, so rows are the only parallelism; 16-row groups give
// 10 waves per CU at 45 MP instead of 2.5 and keep ~24 loads per lane in flight (the stage is HBM-latency bound otherwise)
// HB_MAXR: 15 for the wavelet-level radii (window of two 64-column groups); 63 for the large radii of rtengine::guidedFilter
// callers (three groups).  Same arithmetic, only the LDS window differs.
constexpr int HB_ROWS = 16, HB_COLS = 64;
template <int HB_MAXR, bool SDIV>
__global__ void __launch_bounds__(64) hblur_kernel(BlurArgs a)
{
    constexpr int HB_TW = HB_COLS + 2 * HB_MAXR + 2; // source window held in LDS
    __shared__ float sT[HB_ROWS][HB_TW + 1];
    __shared__ float oT[HB_ROWS][HB_COLS + 1];
    const int sub = blockIdx.y, level = a.level0 + sub / 3;
    const int rad = a.rad[level];
    const int W = a.w, H = a.h;
    const float *src = a.src + (size_t)sub * a.n;
    float *dst = a.dst + (size_t)sub * a.n;
    const int r0 = blockIdx.x * HB_ROWS;
    const int lane = threadIdx.x;
    const int myrow = r0 + lane;
    const int nrows = min(HB_ROWS, H - r0);
    float tempval = 0.f;
    int len = rad + 1;
    float reclen = 0.f;
    // The workgroup is ONE wave: its LDS traffic is ordered by the hardware, so the three phases of a chunk -- window into LDS, running sums,
    // rows out -- need no s_barrier, only the compiler kept from moving LDS accesses across them.  (__syncthreads() also waited for the wave's
    // global stores and loads, vmcnt(0): the phases of a chunk ran strictly one after the other, a memory round trip each -- 5 us per chunk of
    // which the sums are 2; with 3.5 such waves per CU nothing else filled the gaps.)  Round 5: the next chunk's new columns are requested
    // BEFORE this chunk's sums and land while they run; the stores of a chunk are never waited for.
    auto wave_lds_sync = []() {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    // window of chunk 0: columns [-rad - 1, HB_COLS + rad): coalesced row segments -> LDS, 8 rows (16 independent loads) in flight per lane
    {
        const int wc0 = -rad - 1, wn = HB_COLS + 2 * rad + 1;
        const int colA = wc0 + lane, colB = wc0 + lane + 64, colC = wc0 + lane + 128;
        const bool okA = colA >= 0 && colA < W, okB = (lane + 64 < wn) && colB >= 0 && colB < W;
        const bool okC = HB_MAXR > 31 && (lane + 128 < wn) && colC >= 0 && colC < W;
        for (int k0 = 0; k0 < HB_ROWS; k0 += 8) {
            float va[8], vb[8], vc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = k0 + i;
                const size_t ro = (size_t)(r0 + k) * W;
                va[i] = (okA && k < nrows) ? src[ro + colA] : 0.f;
                vb[i] = (okB && k < nrows) ? src[ro + colB] : 0.f;
                if constexpr (HB_MAXR > 31) vc[i] = (okC && k < nrows) ? src[ro + colC] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                sT[k0 + i][lane] = va[i];
                if (lane + 64 < wn) sT[k0 + i][lane + 64] = vb[i];
                if constexpr (HB_MAXR > 31) { if (lane + 128 < wn) sT[k0 + i][lane + 128] = vc[i]; }
            }
        }
    }
    wave_lds_sync();
    for (int c0 = 0; c0 < W; c0 += HB_COLS) {
        // the next chunk's HB_COLS new columns (window index 2 * rad + 1 + lane of ITS window): requested now, stored into LDS behind this chunk's sums
        const bool more = c0 + HB_COLS < W;
        const int colN = c0 + HB_COLS + rad + lane;
        const bool okN = more && colN < W;
        // (unconditional loads from clamped addresses, the zero by a select: a load under its own condition sits in a basic block of its own,
        // and the wait-count pass then no longer knows at the loop's head which of them are still in flight -- it drained vmcnt there, i.e. waited
        // for the previous chunk's STORES every chunk)
        float vn[HB_ROWS];
        const int colNc = colN < W ? colN : W - 1;
#pragma unroll
        for (int k = 0; k < HB_ROWS; ++k) vn[k] = src[(size_t)(r0 + (k < nrows ? k : nrows - 1)) * W + colNc];
#pragma unroll
        for (int k = 0; k < HB_ROWS; ++k) vn[k] = (okN && k < nrows) ? vn[k] : 0.f;
        const int cend = min(HB_COLS, W - c0);
        if (lane < HB_ROWS && myrow < H) {
            const float *s = &sT[lane][rad + 1]; // s[j] = src[row][c0 + j]
            if (c0 > rad && c0 + HB_COLS <= W - rad) {
                // steady state for the whole chunk: tempval += (s[j+rad] - s[j-rad-1]) * reclen
                for (int j0 = 0; j0 < HB_COLS; j0 += 8) {
                    float hi[8], lo[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) { hi[i] = s[j0 + i + rad]; lo[i] = s[j0 + i - rad - 1]; }
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        if constexpr (SDIV) tempval = tempval + (hi[i] - lo[i]) / (float)len; else tempval = tempval + (hi[i] - lo[i]) * reclen;
                        oT[lane][j0 + i] = tempval;
                    }
                }
            } else {
                for (int j = 0; j < cend; ++j) {
                    const int col = c0 + j;
                    if (col == 0) {
                        tempval = s[0];
                        for (int q = 1; q <= rad; q++) tempval += s[q];
                        tempval = tempval / len;
                    } else if (col <= rad) {
                        tempval = (tempval * len + s[j + rad]) / (len + 1);
                        len++;
                        if (col == rad) reclen = 1.f / len;
                    } else if (col < W - rad) {
                        if constexpr (SDIV) tempval = tempval + (s[j + rad] - s[j - rad - 1]) / (float)len; else tempval = tempval + (s[j + rad] - s[j - rad - 1]) * reclen;
                    } else {
                        tempval = (tempval * len - s[j - rad - 1]) / (len - 1);
                        len--;
                    }
                    oT[lane][j] = tempval;
                }
            }
        }
        wave_lds_sync();
        {
            // (sixteen unconditional stores: a lane past the chunk's last column / a row past the workgroup's last row stores the neighbour's
            // value to the neighbour's place once more.  Stores under conditions of their own cannot be counted by the wait-count pass, and the
            // wait for the prefetched columns below then waits for them as well.)
            const int lc = lane < cend ? lane : cend - 1;
#pragma unroll
            for (int k0 = 0; k0 < HB_ROWS; k0 += 8) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = oT[k0 + i < nrows ? k0 + i : nrows - 1][lc];
#pragma unroll
                for (int i = 0; i < 8; ++i) dst[(size_t)(r0 + (k0 + i < nrows ? k0 + i : nrows - 1)) * W + c0 + lc] = v[i];
            }
        }
        {
            // the next window: the 2 * rad + 1 columns it shares with this one are moved inside LDS (their re-read was a third of the
            // kernel's traffic), the new ones come from the registers filled above.  (Also behind the last chunk, where nobody reads it:
            // were the registers only consumed under `if (more)`, the wait-count pass would again have to assume loads in flight at the
            // loop's head.)
            const int nov = 2 * rad + 1;                            // <= 2 * HB_MAXR + 1 columns per row
            constexpr int NMV = (HB_ROWS * (2 * HB_MAXR + 1) + 63) / 64;
            float mv[NMV];
#pragma unroll
            for (int q = 0; q < NMV; ++q) {
                const int e = lane + 64 * q, k = e / nov, x = e - k * nov;
                mv[q] = k < HB_ROWS ? sT[k][x + HB_COLS] : 0.f;
            }
            wave_lds_sync();
#pragma unroll
            for (int q = 0; q < NMV; ++q) {
                const int e = lane + 64 * q, k = e / nov, x = e - k * nov;
                if (k < HB_ROWS) sT[k][x] = mv[q];
            }
#pragma unroll
            for (int k = 0; k < HB_ROWS; ++k) sT[k][nov + lane] = vn[k];
        }
        wave_lds_sync();
    }
}

// Radii above 63 (rtengine::guidedFilter callers with image-sized radii: log encoding's regularisation, the hsl equaliser at
// scale 1): the same running sum over 256-column chunks with the window in dynamic LDS.  Only used on one plane at a time.
constexpr int HBB_COLS = 256;
__global__ void __launch_bounds__(64) hblur_big_kernel(BlurArgs a)
{
    extern __shared__ float hbb_lds[];
    const int sub = blockIdx.y, level = a.level0 + sub / 3;
    const int rad = a.rad[level];
    const int W = a.w, H = a.h;
    const int TS_ = HBB_COLS + 2 * rad + 3, OS_ = HBB_COLS + 1;   // odd strides
    float *const sT = hbb_lds, *const oT = hbb_lds + (size_t)HB_ROWS * TS_;
    const float *src = a.src + (size_t)sub * a.n;
    float *dst = a.dst + (size_t)sub * a.n;
    const int r0 = blockIdx.x * HB_ROWS;
    const int lane = threadIdx.x;
    const int myrow = r0 + lane;
    const int nrows = min(HB_ROWS, H - r0);
    float tempval = 0.f;
    int len = rad + 1;
    float reclen = 0.f;
    for (int c0 = 0; c0 < W; c0 += HBB_COLS) {
        const int wc0 = c0 - rad - 1, wn = HBB_COLS + 2 * rad + 1;
        for (int k = 0; k < nrows; ++k)
            for (int x = lane; x < wn; x += 64) {
                const int col = wc0 + x;
                sT[(size_t)k * TS_ + x] = (col >= 0 && col < W) ? src[(size_t)(r0 + k) * W + col] : 0.f;
            }
        __syncthreads();
        const int cend = min(HBB_COLS, W - c0);
        if (lane < HB_ROWS && myrow < H) {
            const float *s = sT + (size_t)lane * TS_ + rad + 1; // s[j] = src[row][c0 + j]
            float *o = oT + lane * OS_;
            for (int j = 0; j < cend; ++j) {
                const int col = c0 + j;
                if (col == 0) {
                    tempval = s[0];
                    for (int q = 1; q <= rad; q++) tempval += s[q];
                    tempval = tempval / len;
                } else if (col <= rad) {
                    tempval = (tempval * len + s[j + rad]) / (len + 1);
                    len++;
                    if (col == rad) reclen = 1.f / len;
                } else if (col < W - rad) {
                    tempval = a.steady_div ? tempval + (s[j + rad] - s[j - rad - 1]) / (float)len : tempval + (s[j + rad] - s[j - rad - 1]) * reclen;
                } else {
                    tempval = (tempval * len - s[j - rad - 1]) / (len - 1);
                    len--;
                }
                o[j] = tempval;
            }
        }
        __syncthreads();
        for (int k = 0; k < nrows; ++k)
            for (int x = lane; x < cend; x += 64) dst[(size_t)(r0 + k) * W + c0 + x] = oT[k * OS_ + x];
        __syncthreads();
    }
}

// ---------------------------------------------------------------- vertical box blur + coefficient update (boxblur.h:602-742)
// PLAIN (a.plain: the guided filter's box blurs -- the blurred value is stored, no factor / coefficient planes): the two loads per row leave
// room for 24 rows per batch (48 loads in flight per lane; 8 rows left a lone wave one memory round trip per 8 rows: 124 ns per row)
template <bool PLAIN>
__global__ void __launch_bounds__(64) vblur_combine_kernel(BlurArgs a)
{
    const int sub = blockIdx.y, level = a.level0 + sub / 3;
    const int rad = a.rad[level];
    const int W = a.w, H = a.h;
    const float *t = a.src + (size_t)sub * a.n;   // horizontally blurred
    const float *sfave = a.sfave + (size_t)sub * a.n;
    float *coef = a.coef + (size_t)sub * a.n;
    const int col = blockIdx.x * 64 + threadIdx.x;
    if (col >= W) return;
    const size_t nv4 = (a.n / 4) * 4;
    const float eps = 0.01f;
    const bool vec = PLAIN ? true : col < (W / 4) * 4;
    float *plain_dst = PLAIN ? a.dst + (size_t)sub * a.n : nullptr;
    float tv = 0.f;
    float lenf = (float)(rad + 1);
    int leni = rad + 1;
    float rlen = 0.f;
    // coefficient update (FTblockDN.cc:698-714,803-836): vector lanes (c*num)/den, tail c*(num/den)
    auto commit = [&](int row, float sfd, float sf, float c) {
        const size_t i = (size_t)row * W + col;
        if (plain_dst) { plain_dst[i] = sfd; return; }
        const float num = sqr(sfd) + sqr(sf), den = sfd + sf + eps;
        coef[i] = i < nv4 ? c * num / den : c * (num / den);
    };
    int row = 0;
    for (; row <= rad && row < H; ++row) {
        if (row == 0) {
            if (vec) {
                tv = t[col];
                for (int i = 1; i <= rad; i++) tv = tv + t[(size_t)i * W + col];
                tv = tv / lenf;
            } else {
                tv = t[col] / leni;
                for (int i = 1; i <= rad; i++) tv += t[(size_t)i * W + col] / leni;
            }
        } else if (vec) {
            const float lenp1 = lenf + 1.f;
            tv = (tv * lenf + t[(size_t)(row + rad) * W + col]) / lenp1;
            lenf = lenp1;
        } else {
            tv = (tv * leni + t[(size_t)(row + rad) * W + col]) / (leni + 1);
            leni++;
        }
        const size_t i = (size_t)row * W + col;
        commit(row, tv, PLAIN ? 0.f : sfave[i], PLAIN ? 0.f : coef[i]);
    }
    rlen = 1.f / lenf;
    // steady state, 8 rows of independent loads in flight
    const int steady_end = H - rad;
    // One wave owns 64 columns for the whole height, and there are only W / 64 x bands of them (fewer than four per CU), so a batch's
    // memory latency is not hidden by other waves: the next batch is loaded while this one is computed, and its loads are issued BEFORE
    // this batch's stores (vmcnt retires in order: a wait for loads issued after stores waits for the stores as well).
    constexpr int VB = PLAIN ? 24 : 8;       // rows per batch: 4 x 8 loads in flight per lane, two batches deep (16 rows per batch measured no better)
    float hi[VB], lo[VB], sf[VB], c[VB], nhi[VB], nlo[VB], nsf[VB], nc[VB];
    auto fetch = [&](int r, float *h_, float *l_, float *s_, float *c_) {
#pragma unroll
        for (int k = 0; k < VB; ++k) {
            h_[k] = t[(size_t)(r + k + rad) * W + col];
            l_[k] = t[(size_t)(r + k - rad - 1) * W + col];
            s_[k] = PLAIN ? 0.f : sfave[(size_t)(r + k) * W + col];
            c_[k] = PLAIN ? 0.f : coef[(size_t)(r + k) * W + col];
        }
    };
    if (row + VB <= steady_end) fetch(row, hi, lo, sf, c);
    for (; row + VB <= steady_end; row += VB) {
        const bool more = row + 2 * VB <= steady_end;
        fetch(more ? row + VB : row, nhi, nlo, nsf, nc);
#pragma unroll
        for (int k = 0; k < VB; ++k) {
            const float d = hi[k] - lo[k];
            tv = vec ? tv + d * rlen : tv + d / leni;
            commit(row + k, tv, sf[k], c[k]);
        }
#pragma unroll
        for (int k = 0; k < VB; ++k) { hi[k] = nhi[k]; lo[k] = nlo[k]; sf[k] = nsf[k]; c[k] = nc[k]; }
    }
    for (; row < steady_end; ++row) {
        const float d = t[(size_t)(row + rad) * W + col] - t[(size_t)(row - rad - 1) * W + col];
        tv = vec ? tv + d * rlen : tv + d / leni;
        const size_t i = (size_t)row * W + col;
        commit(row, tv, PLAIN ? 0.f : sfave[i], PLAIN ? 0.f : coef[i]);
    }
    for (; row < H; ++row) {
        if (vec) {
            const float lenm1 = lenf - 1.f;
            tv = (tv * lenf - t[(size_t)(row - rad - 1) * W + col]) / lenm1;
            lenf = lenm1;
        } else {
            tv = (tv * leni - t[(size_t)(row - rad - 1) * W + col]) / (leni - 1);
            leni--;
        }
        const size_t i = (size_t)row * W + col;
        commit(row, tv, PLAIN ? 0.f : sfave[i], PLAIN ? 0.f : coef[i]);
    }
}

// ---------------------------------------------------------------- launchers
static int flat_grid(long long n, int cap) { long long g = (n + 255) / 256; return (int)(g < cap ? g : cap); }

hipError_t launch_gamma_lut(float *lut, float gamma, float start, float slope, float divisor, float factor, hipStream_t s)
{
    hipLaunchKernelGGL(gamma_lut_kernel, dim3(256), dim3(256), 0, s, lut, gamma, start, slope, divisor, factor);
    return hipGetLastError();
}
// the LDS-table shape: large frames with the gamma LUTs in use (their pool memory is 16-byte aligned)
static bool dn_lds_shape(const DnPixArgs &a, const float *lut, int *cus)
{
    if (a.lab_mode || !(a.gam > 1.f) || (long long)a.w * a.h < (1 << 22) || (reinterpret_cast<uintptr_t>(lut) & 15) || a.no_lds_lut || !device_block_fits(LUT_LDS_N * (int)sizeof(float), 1024)) return false;
    int dev = 0;
    *cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (a.cu_reserve > 0) *cus = *cus - a.cu_reserve > 1 ? *cus - a.cu_reserve : 1;
    return true;
}
hipError_t launch_rgb2yuv(const DnPixArgs &a, hipStream_t s)
{
    int cus;
    if (dn_lds_shape(a, a.gamcurve, &cus)) {
        const size_t lds = (size_t)LUT_LDS_N * sizeof(float);
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(rgb2yuv_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(rgb2yuv_lds_kernel, dim3(cus < a.h ? cus : a.h), dim3(1024), lds, s, a);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(rgb2yuv_kernel, image_grid(a.w, a.h), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_yuv2rgb(const DnPixArgs &a, hipStream_t s)
{
    int cus;
    if (dn_lds_shape(a, a.igamcurve, &cus)) {
        const size_t lds = (size_t)LUT_LDS_N * sizeof(float);
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(yuv2rgb_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(yuv2rgb_lds_kernel, dim3(cus < a.h ? cus : a.h), dim3(1024), lds, s, a);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(yuv2rgb_kernel, image_grid(a.w, a.h), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_mad(const float *bands, size_t n, int nsub, int *histo, float *out, hipStream_t s)
{
    // histo: nsub * 65536 ints (the full histograms, only built for bands the windowed pass could not settle) followed by
    // nsub * MAD_SCRATCH_INTS_PER_BAND ints of scratch
    int *scr = histo + (size_t)nsub * 65536;
    hipError_t e = hipMemsetAsync(histo, 0, ((size_t)nsub * 65536 + (size_t)nsub * MAD_SCRATCH_INTS_PER_BAND) * sizeof(int), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(mad_sample_kernel, dim3(64, nsub), dim3(256), 0, s, bands, n, scr);
    hipLaunchKernelGGL(mad_pick_kernel, dim3(nsub), dim3(256), 0, s, scr);
    hipLaunchKernelGGL(mad_window_kernel, dim3(flat_grid((long long)n, MAD_GRID), nsub), dim3(256), 0, s, bands, n, scr);
    hipLaunchKernelGGL(mad_window_finish_kernel, dim3(nsub), dim3(64), 0, s, scr, (int)n, out);
    hipLaunchKernelGGL(mad_hist_kernel, dim3(flat_grid((long long)n, MAD_GRID), nsub), dim3(256), 0, s, bands, n, histo, (const int *)scr);
    hipLaunchKernelGGL(mad_finish_kernel, dim3(nsub), dim3(256), 0, s, (const int *)histo, (int)n, out, (const int *)scr);
    return hipGetLastError();
}
hipError_t launch_shrink_sf(const ShrinkArgs &a, int nsub, bool ab, hipStream_t s)
{
    dim3 grid(flat_grid((long long)a.n, 1024), nsub);
    if (ab) hipLaunchKernelGGL(shrink_sf_AB_kernel, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(shrink_sf_L_kernel, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_bishrink_AB(const ShrinkArgs &a, int nsub, hipStream_t s)
{
    hipLaunchKernelGGL(bishrink_AB_kernel, dim3(flat_grid((long long)a.n, 1024), nsub), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_hblur(const BlurArgs &a, int nsub, hipStream_t s)
{
    int maxr = 0;
    for (int l = 0; l < 10; ++l) maxr = a.rad[l] > maxr ? a.rad[l] : maxr;
    if (maxr > HBLUR_MAX_RADIUS) return hipErrorInvalidValue;
    if (maxr > 63) {
        const size_t lds = ((size_t)HB_ROWS * (HBB_COLS + 2 * maxr + 3) + (size_t)HB_ROWS * (HBB_COLS + 1)) * sizeof(float);
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(hblur_big_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(hblur_big_kernel, dim3((a.h + HB_ROWS - 1) / HB_ROWS, nsub), dim3(64), lds, s, a);
        return hipGetLastError();
    }
    // steady_div (the boxblur.h:318 variant of the guided filter) is a template parameter: as a run-time select both forms were evaluated
    const dim3 grid((a.h + HB_ROWS - 1) / HB_ROWS, nsub);
    if (maxr <= 15) {
        if (a.steady_div) hipLaunchKernelGGL((hblur_kernel<15, true>), grid, dim3(64), 0, s, a);
        else hipLaunchKernelGGL((hblur_kernel<15, false>), grid, dim3(64), 0, s, a);
    } else {
        if (a.steady_div) hipLaunchKernelGGL((hblur_kernel<63, true>), grid, dim3(64), 0, s, a);
        else hipLaunchKernelGGL((hblur_kernel<63, false>), grid, dim3(64), 0, s, a);
    }
    return hipGetLastError();
}
hipError_t launch_vblur_combine(const BlurArgs &a, int nsub, hipStream_t s)
{
    if (a.plain) hipLaunchKernelGGL(vblur_combine_kernel<true>, dim3((a.w + 63) / 64, nsub), dim3(64), 0, s, a);
    else hipLaunchKernelGGL(vblur_combine_kernel<false>, dim3((a.w + 63) / 64, nsub), dim3(64), 0, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Chroma noise-curve map.  calclum = every second pixel of the image (ipdenoise.cc:1113-1129), pushed through
// convertColorSpace's matrix branch again (L1131 -> rawimagesource.cc:3184-3213, double accumulation), then
// Color::rgbxyz(float wpi) + Color::XYZ2Lab (color.cc:833-838,1247-1259,1382-1397) and
// ccalc = SQR(1 + 4*noiseCCurve[cN/60]) for cN > 100, else the cN = 100 constant (FTblockDN.cc:1733-1771).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) chroma_map_kernel(ChromaMapArgs a)
{
    const float t0 = 1.f + 1.f * (4.f * lutf_lookup<true>(a.curve, 501, 100.f / 60.f));
    const float cn100 = t0 * t0;
    FOR_IMAGE_XY(ii, jj, a.wid, a.hei) {
        const long long t = (long long)ii * a.wid + jj;
        float RL, GL, BL;
        if (a.gi.on) {
            const size_t o = (size_t)(a.gi.sy1 + 2 * ii) * a.gi.stride + a.gi.sx1 + 2 * jj;
            RL = a.gi.src[0][o]; GL = a.gi.src[1][o]; BL = a.gi.src[2][o];
            gi_convert(a.gi, RL, GL, BL);
        } else {
            const size_t o = (size_t)(2 * ii) * a.stride + 2 * jj;
            RL = a.src[0][o]; GL = a.src[1][o]; BL = a.src[2][o];
        }
        if (a.has_mat) {
            const double dr = RL, dg = GL, db = BL;
            RL = (float)(a.mat[0] * dr + a.mat[1] * dg + a.mat[2] * db);
            GL = (float)(a.mat[3] * dr + a.mat[4] * dg + a.mat[5] * db);
            BL = (float)(a.mat[6] * dr + a.mat[7] * dg + a.mat[8] * db);
        }
        const float XL = a.wpi[0] * RL + a.wpi[1] * GL + a.wpi[2] * BL;
        const float YL = a.wpi[3] * RL + a.wpi[4] * GL + a.wpi[5] * BL;
        const float ZL = a.wpi[6] * RL + a.wpi[7] * GL + a.wpi[8] * BL;
        const float fx = xyz2lab_f(a.cachef, XL / 0.9642f), fy = xyz2lab_f(a.cachef, YL), fz = xyz2lab_f(a.cachef, ZL / 0.8249f);
        const float A = 500.0f * (fx - fy), B = 200.0f * (fy - fz);
        const float cN = sqrtf(A * A + B * B);   // sqrtf is the correctly rounded one (__fsqrt_rn maps to the native approximation)
        float r = cn100;
        if (cN > 100) {
            const float u = 1.f + 1.f * (4.f * lutf_lookup<true>(a.curve, 501, cN / 60.f));
            r = u * u;
        }
        a.out[t] = r;
    }
}
// The same per map pixel with the lower LUT_LDS_N entries of the Lab f() table in LDS (round 5): the three lookups of a pixel were L2 line
// gathers from a 256 KB table -- the kernel moved 0.33 GB in 160 us, 2 TB/s, its waves waiting for the texture path four cycles in five.  One
// persistent 1024-thread workgroup per CU, two map rows of four pixels per thread and batch, the loads first.
__device__ __forceinline__ float xyz2lab_f_lds(const float *lds, const float *__restrict__ cachef, float f)
{
    if (f != f) return f;
    if (f < 0.f) return (float)(327.68 * (((24389.0 / 27.0) * (double)f / (double)65535.f + 16.0) / 116.0));
    if (f > 65535.f) return 327.68f * xcbrtf_s(f / 65535.f);
    return lutf_lookup_lds<false>(lds, cachef, 65536, f);
}
__global__ void __launch_bounds__(1024) chroma_map_lds_kernel(ChromaMapArgs a)
{
    extern __shared__ float dn_lut_lds[];
    lut_lds_fill(dn_lut_lds, a.cachef, 1024);
    const float t0 = 1.f + 1.f * (4.f * lutf_lookup<true>(a.curve, 501, 100.f / 60.f));
    const float cn100 = t0 * t0;
    constexpr int NR = 2, NPX = 4;
    const float *const p0 = a.gi.on ? a.gi.src[0] : a.src[0], *const p1 = a.gi.on ? a.gi.src[1] : a.src[1], *const p2 = a.gi.on ? a.gi.src[2] : a.src[2];
    for (int yb = blockIdx.x; yb < a.hei; yb += NR * gridDim.x)
        for (int x0 = 0; x0 < a.wid; x0 += NPX * 1024) {
            float r[NR * NPX], g[NR * NPX], b[NR * NPX];
#pragma unroll
            for (int k = 0; k < NR * NPX; ++k) {
                const int jj = x0 + (k % NPX) * 1024 + (int)threadIdx.x, ir = yb + (k / NPX) * (int)gridDim.x, ii = ir < a.hei ? ir : a.hei - 1;
                const int jc = jj < a.wid ? jj : a.wid - 1;
                const size_t o = a.gi.on ? (size_t)(a.gi.sy1 + 2 * ii) * a.gi.stride + a.gi.sx1 + 2 * jc : (size_t)(2 * ii) * a.stride + 2 * jc;
                r[k] = p0[o]; g[k] = p1[o]; b[k] = p2[o];
            }
#pragma unroll
            for (int k = 0; k < NR * NPX; ++k) {
                const int jj = x0 + (k % NPX) * 1024 + (int)threadIdx.x, ii = yb + (k / NPX) * (int)gridDim.x;
                if (jj >= a.wid || ii >= a.hei) continue;
                float RL = r[k], GL = g[k], BL = b[k];
                if (a.gi.on) gi_convert(a.gi, RL, GL, BL);
                if (a.has_mat) {
                    const double dr = RL, dg = GL, db = BL;
                    RL = (float)(a.mat[0] * dr + a.mat[1] * dg + a.mat[2] * db);
                    GL = (float)(a.mat[3] * dr + a.mat[4] * dg + a.mat[5] * db);
                    BL = (float)(a.mat[6] * dr + a.mat[7] * dg + a.mat[8] * db);
                }
                const float XL = a.wpi[0] * RL + a.wpi[1] * GL + a.wpi[2] * BL;
                const float YL = a.wpi[3] * RL + a.wpi[4] * GL + a.wpi[5] * BL;
                const float ZL = a.wpi[6] * RL + a.wpi[7] * GL + a.wpi[8] * BL;
                const float fx = xyz2lab_f_lds(dn_lut_lds, a.cachef, XL / 0.9642f), fy = xyz2lab_f_lds(dn_lut_lds, a.cachef, YL), fz = xyz2lab_f_lds(dn_lut_lds, a.cachef, ZL / 0.8249f);
                const float A = 500.0f * (fx - fy), B = 200.0f * (fy - fz);
                const float cN = sqrtf(A * A + B * B);
                float res = cn100;
                if (cN > 100) {
                    const float u = 1.f + 1.f * (4.f * lutf_lookup<true>(a.curve, 501, cN / 60.f));
                    res = u * u;
                }
                a.out[(long long)ii * a.wid + jj] = res;
            }
        }
}
hipError_t launch_chroma_map(const ChromaMapArgs &a, hipStream_t s)
{
    if ((long long)a.wid * a.hei >= (1 << 20) && (reinterpret_cast<uintptr_t>(a.cachef) & 15) == 0 && !a.no_lds_lut && device_block_fits(LUT_LDS_N * (int)sizeof(float), 1024)) {
        const size_t lds = (size_t)LUT_LDS_N * sizeof(float);
        hipError_t e = dyn_lds_once(reinterpret_cast<const void *>(chroma_map_lds_kernel), (int)lds);
        if (e != hipSuccess) return e;
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (a.cu_reserve > 0) cus = cus - a.cu_reserve > 1 ? cus - a.cu_reserve : 1;
        hipLaunchKernelGGL(chroma_map_lds_kernel, dim3(cus < a.hei ? cus : a.hei), dim3(1024), lds, s, a);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(chroma_map_kernel, image_grid(a.wid, a.hei), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------- AUTOMATIC chrominance estimation (RGB_denoise_info)
// getImage for one pixel of a crop (rawimagesource.cc:940-1025, skip 1): 0 + v, * mul, CLIP
__device__ __forceinline__ float dninfo_fetch(const DnInfoArgs &a, int c, size_t si)
{
    float t = 0.f;
    t += a.src[c][si];
    t *= a.mul[c];
    if (a.do_clip) t = std_max(0.f, std_min(t, 65535.f));
    return t;
}
// hue / chroma / luminance maps of all nine crops: provicalc (every second pixel) -> convertColorSpace -> rgbxyz -> XYZ2Lab
// (ipdenoise.cc:902-911,268-283), then L384-458.  The 4-lane xatan2f equals the scalar one bit for bit (tests/golden/sleef3.npz);
// the chroma floor is vmaxf in the 4-lane columns and a compare in the tail.
__global__ void __launch_bounds__(256) dninfo_maps_kernel(DnInfoArgs a)
{
    const int k = blockIdx.z;
    const long long n2 = (long long)a.wid * a.hei;
    const int nvec = 4 * (a.crW / 8);
    float *hue = a.maps + (size_t)k * 3 * n2, *chrom = hue + n2, *lum = chrom + n2;
    FOR_IMAGE_XY(ii, jj, a.wid, a.hei) {
        const long long t = (long long)ii * a.wid + jj;
        const size_t si = (size_t)(a.sy[k] + 2 * ii) * a.stride + a.sx[k] + 2 * jj;
        const double dr = dninfo_fetch(a, 0, si), dg = dninfo_fetch(a, 1, si), db = dninfo_fetch(a, 2, si);
        const float RL = (float)(a.mat[0] * dr + a.mat[1] * dg + a.mat[2] * db);
        const float GL = (float)(a.mat[3] * dr + a.mat[4] * dg + a.mat[5] * db);
        const float BL = (float)(a.mat[6] * dr + a.mat[7] * dg + a.mat[8] * db);
        const float X = a.wp[0] * RL + a.wp[1] * GL + a.wp[2] * BL, Y = a.wp[3] * RL + a.wp[4] * GL + a.wp[5] * BL, Z = a.wp[6] * RL + a.wp[7] * GL + a.wp[8] * BL;
        const float x = X / 0.9642f, z = Z / 0.8249f, y = Y;
        const float fx = xyz2lab_f(a.cachef, x), fy = xyz2lab_f(a.cachef, y), fz = xyz2lab_f(a.cachef, z);
        float l;
        if (y != y) l = y;
        else if (y < 0.f) l = (float)(327.68 * ((24389.0 / 27.0) * (double)y / (double)65535.f));
        else if (y > 65535.f) l = 327.68f * (116.f * xcbrtf_s(y / 65535.f) - 16.f);
        else l = lutf_lookup<false>(a.cachefy, 65536, y);
        const float la = 500.0f * (fx - fy), lb = 200.0f * (fy - fz);
        hue[t] = xatan2f_s(lb, la);
        float cN = sqrtf(la * la + lb * lb);
        if (jj < nvec) cN = sse_max(cN, 100.f); else if (cN < 100.f) cN = 100.f;
        chrom[t] = cN;
        float Ll = l < 2.f ? 2.f : l;
        Ll = Ll > 32768.f ? 32768.f : Ll;
        lum[t] = Ll;
    }
}
// (ShrinkAll_info's statistics over these maps: orderedsum.hip)
// labdn->a / labdn->b of one crop: gain, gamma (LUT flags 0 below 65535, the analytic curve above; factor 32768), rgb2yuv
// (ipdenoise.cc:460-482)
__global__ void __launch_bounds__(256) dninfo_ab_kernel(DnInfoArgs a)
{
    const int k = a.crop;
    FOR_IMAGE_XY(i, j, a.crW, a.crH) {
        const long long t = (long long)i * a.crW + j;
        const size_t si = (size_t)(a.sy[k] + i) * a.stride + a.sx[k] + j;
        float X = a.gain * dninfo_fetch(a, 0, si), Y = a.gain * dninfo_fetch(a, 1, si), Z = a.gain * dninfo_fetch(a, 2, si);
        X = X < 65535.f ? lutf_noclip(a.gamcurve, X) : (gammaf_s(X / 65535.f, a.gam, a.gamthresh, a.gamslope) * 32768.f);
        Y = Y < 65535.f ? lutf_noclip(a.gamcurve, Y) : (gammaf_s(Y / 65535.f, a.gam, a.gamthresh, a.gamslope) * 32768.f);
        Z = Z < 65535.f ? lutf_noclip(a.gamcurve, Z) : (gammaf_s(Z / 65535.f, a.gam, a.gamthresh, a.gamslope) * 32768.f);
        const float l = X * a.wp[3] + Y * a.wp[4] + Z * a.wp[5];
        a.A[t] = X - l;
        a.B[t] = l - Z;
    }
}
hipError_t launch_dninfo_maps(const DnInfoArgs &a, hipStream_t s)
{
    dim3 g = image_grid(a.wid, a.hei);
    g.z = 9;
    hipLaunchKernelGGL(dninfo_maps_kernel, g, dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_dninfo_ab(const DnInfoArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(dninfo_ab_kernel, image_grid(a.crW, a.crH), dim3(256), 0, s, a);
    return hipGetLastError();
}

} // namespace artgpu
