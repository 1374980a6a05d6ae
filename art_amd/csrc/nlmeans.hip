// art_amd/csrc/nlmeans.hip -- NL-means stage of the denoise tool on gfx950
// (reference: rtengine/nlmeans.cc:50-280 denoise::NLMeans; rtengine/FTblockDN.cc:1366-1476 detail_mask +
//  laplacian; rtengine/gauss.cc:94-126,554-665,716-856 Young-van Vliet recursive gaussian, x86-64 path).
//
// v1, correctness first.  Everything keeps the reference's fp32 association order:
//   gauss_h / gauss_v : 3rd-order IIR forward+backward per line; one lane per line (the recurrence is
//                       sequential along the line).  Lines in the reference's SSE groups (rows < H-H%4,
//                       columns < W-W%8) use float coefficients, the tail lines double ones.
//   nlm_tile          : one workgroup per REFERENCE tile (150x150, stride 150-2*border: the per-tile
//                       fp32 integral image is part of the result).  For each of the (2r+1)^2 offsets the
//                       integral image St is swept along anti-diagonals (one lane per row, a barrier per
//                       diagonal, 16 diagonals kept in LDS); patch distances come from the four corners,
//                       weights from the 8192-entry exp LUT (LDS), accumulation in offset order.
//                       MXCSR flush-to-zero (nlmeans.cc:157-160) is reproduced with explicit ftz().
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdlib.h>
#include "devmath.h"
#include "devsleef.h"
#include "kernels.h"

namespace artgpu {

namespace {
__device__ __forceinline__ float ftz(float x) { return fabsf(x) < FLT_MIN ? copysignf(0.f, x) : x; }
__device__ __forceinline__ float bilinear(const float *__restrict__ src, int W, int H, float x, float y)
{
    const int xi = min((int)x, W - 1), yi = min((int)y, H - 1);
    const float xf = x - xi, yf = y - yi;
    const int xi1 = min(xi + 1, W - 1), yi1 = min(yi + 1, H - 1);
    const float bl = src[(size_t)yi * W + xi], br = src[(size_t)yi * W + xi1];
    const float tl = src[(size_t)yi1 * W + xi], tr = src[(size_t)yi1 * W + xi1];
    const float b = xf * br + (1.f - xf) * bl;
    const float t = xf * tr + (1.f - xf) * tl;
    return yf * t + (1.f - yf) * b;
}
} // namespace

// ---------------------------------------------------------------- detail_mask (FTblockDN.cc:1408-1476)
// L2 = xlin2log(rescaleBilinear(src -> W/4 x H/4) / scaling, 50)
__global__ void __launch_bounds__(256) dm_down_log_kernel(MaskArgs a)
{
    const long long n = (long long)a.w4 * a.h4;
    const float col_scale = (float)a.W / (float)a.w4, row_scale = (float)a.H / (float)a.h4;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(t / a.w4), x = (int)(t - (long long)y * a.w4);
        // src may have a row stride: bilinear on a strided plane
        const float fx = x * col_scale, fy = y * row_scale;
        const int xi = min((int)fx, a.W - 1), yi = min((int)fy, a.H - 1);
        const float xf = fx - xi, yf = fy - yi;
        const int xi1 = min(xi + 1, a.W - 1), yi1 = min(yi + 1, a.H - 1);
        const float bl = a.src[(size_t)yi * a.src_stride + xi], br = a.src[(size_t)yi * a.src_stride + xi1];
        const float tl = a.src[(size_t)yi1 * a.src_stride + xi], tr = a.src[(size_t)yi1 * a.src_stride + xi1];
        const float b = xf * br + (1.f - xf) * bl;
        const float tt = xf * tr + (1.f - xf) * tl;
        const float v = yf * tt + (1.f - yf) * b;
        a.L2[t] = xlin2log(v / a.scaling, 50.f);
    }
}
// laplacian (FTblockDN.cc:1366-1403)
__global__ void __launch_bounds__(256) dm_laplacian_kernel(MaskArgs a)
{
    const int w = a.w4, h = a.h4;
    const long long n = (long long)w * h;
    const float thr = a.threshold / a.scaling, ceil_ = a.ceiling / a.scaling;
    const float f = a.factor / ceil_;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(t / w), x = (int)(t - (long long)y * w);
        const int nn = (y - 1 < 0) ? y + 1 : y - 1, s = (y + 1 >= h) ? y - 1 : y + 1;
        const int ww = (x - 1 < 0) ? x + 1 : x - 1, e = (x + 1 >= w) ? x - 1 : x + 1;
#define GETL(yy, xx) std_max(a.L2[(size_t)(yy) * w + (xx)], 0.f)
        const float v = -8.f * GETL(y, x) + GETL(nn, x) + GETL(s, x) + GETL(y, ww) + GETL(y, e) + GETL(nn, ww) + GETL(nn, e) + GETL(s, ww) + GETL(s, e);
#undef GETL
        float tt = fabsf(v) - thr;
        tt = std_max(0.f, std_min(tt, ceil_));
        a.m2[t] = tt * f;
    }
}
// mask = scurve(LIM01(rescaleBilinear(m2 -> W x H) + 1 - factor))
__global__ void __launch_bounds__(256) dm_up_scurve_kernel(MaskArgs a)
{
    const long long n = (long long)a.W * a.H;
    const float col_scale = (float)a.w4 / (float)a.W, row_scale = (float)a.h4 / (float)a.H;
    const float thr1 = 1.f - a.factor;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(t / a.W), x = (int)(t - (long long)y * a.W);
        const float v = lim01(bilinear(a.m2, a.w4, a.h4, x * col_scale, y * row_scale) + thr1);
        a.mask[t] = xlin2log(pow_F(v, 2.23f), 101.f);
    }
}

// ---------------------------------------------------------------- Young-van Vliet gaussian (gauss.cc:554-665,716-856)
// A line (row or column) is one serial recurrence: forward into tmp, Triggs-Sdika boundary, backward into the image.  One lane
// per line is the only parallelism there is, and a wave that does its own load and store per step is limited by the 64 memory
// operations it may have outstanding (vmcnt) and by their latency, whatever the prefetch depth or the number of waves (measured:
// 0.52 ms vertical, 0.74 ms horizontal per 45 MP plane).  So the recurrence wave of a workgroup touches LDS only: four loader
// waves bring 64-step chunks of the group's 64 lines into an LDS ring by LDS-DMA, three chunks ahead (no registers in between,
// each wave with its own window of outstanding operations), three storer waves write the chunk the recurrence wave has filtered
// in place back, one LDS-only barrier per chunk.  Loaders and storers are different waves because vmcnt counts loads and stores
// together: a wave waiting for its prefetch would wait for its younger stores as well.
// Lines the reference filters with double coefficients (the rows / columns behind its 4- / 8-wide vector loops) form a
// workgroup of their own in the same launch.
namespace {
typedef __attribute__((address_space(3))) float *gs_lf;
constexpr int GS_CH = 64;             // steps per chunk
constexpr int GS_PITCH = 65;          // LDS row pitch (65: the recurrence wave's column walk through horizontal chunks is conflict-free)
constexpr int GS_AHEAD = 3;           // chunks in flight
constexpr int GS_NB = GS_AHEAD + 2;   // ring: in flight, being filtered, being written back
constexpr int GS_LOADERS = 4, GS_STORERS = 3, GS_NT = 64 * (1 + GS_LOADERS + GS_STORERS);
constexpr int GS_LPER = GS_CH / GS_LOADERS, GS_SPER = (GS_CH + GS_STORERS - 1) / GS_STORERS;
constexpr int GS_LDS_BYTES = GS_NB * GS_CH * GS_PITCH * 4;
static_assert(GS_AHEAD * GS_LPER < 64, "vmcnt window");

// workgroup barrier that orders LDS traffic only: a __syncthreads() would also drain the DMA in flight and the stores
__device__ __forceinline__ void gs_lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// s_waitcnt immediate of gfx9 that waits for vmcnt <= n only (vmcnt is bits [3:0] and [15:14]; expcnt [6:4] and lgkmcnt [11:8] left at their maxima)
constexpr int gs_vmcnt(int n) { return 0x0f70 | (n & 15) | ((n >> 4) << 14); }
// position of step s of chunk k along the line
__device__ __forceinline__ int gs_pos(bool fwd, int n, int k, int s) { return fwd ? k * GS_CH + s : n - 1 - k * GS_CH - s; }

template <bool HORIZ, typename C>
__device__ __forceinline__ void gauss_stream_group(const GaussArgs &a, int line0, int nl, C B, C b1, C b2, C b3, const C *M, gs_lf lds)
{
    const int n = HORIZ ? a.W : a.H, nlines = line0 + nl;
    const size_t W = (size_t)a.W;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nchunks = (n + GS_CH - 1) / GS_CH;
    const bool compute = wave == 0, loader = wave >= 1 && wave <= GS_LOADERS, storer = wave > GS_LOADERS;
    const int m = loader ? wave - 1 : wave - 1 - GS_LOADERS;
    // chunk row j = step j of the chunk (vertical: lanes are the lines) or line j of the group (horizontal: lanes are the steps)
    float m1 = 0.f, m2 = 0.f, m3 = 0.f, sl = 0.f;
    if (compute) {
        const int l = min(line0 + lane, nlines - 1);
        sl = a.img[HORIZ ? (size_t)l * W + (n - 1) : (size_t)(n - 1) * W + l];
    }

    for (int pass = 0; pass < 2; ++pass) {
        const bool fwd = pass == 0;
        const float *__restrict__ src = fwd ? a.img : a.tmp;
        float *__restrict__ dst = fwd ? a.tmp : a.img;
        // chunk k -> ring slot k % GS_NB; every lane fetches (addresses clamped into the plane) so that the number of loads in flight is known
        auto fetch = [&](int k) {
            const gs_lf slot = lds + (k % GS_NB) * (GS_CH * GS_PITCH);
#pragma unroll
            for (int q = 0; q < GS_LPER; ++q) {
                const int j = m + GS_LOADERS * q;
                const int line = min(line0 + (HORIZ ? j : lane), nlines - 1);
                const int pos = min(max(gs_pos(fwd, n, k, HORIZ ? lane : j), 0), n - 1);
                const float *g = HORIZ ? src + (size_t)line * W + pos : src + (size_t)pos * W + line;
                __builtin_amdgcn_global_load_lds(g, slot + j * GS_PITCH, 4, 0, 0);      // to (wave-uniform LDS base) + 4 * lane
            }
        };
        auto drain = [&](int k) {
            const gs_lf slot = lds + (k % GS_NB) * (GS_CH * GS_PITCH);
            float v[GS_SPER];
#pragma unroll
            for (int q = 0; q < GS_SPER; ++q) v[q] = slot[min(m + GS_STORERS * q, GS_CH - 1) * GS_PITCH + lane];
#pragma unroll
            for (int q = 0; q < GS_SPER; ++q) {
                const int j = m + GS_STORERS * q;
                const int line = line0 + (HORIZ ? j : lane), pos = gs_pos(fwd, n, k, HORIZ ? lane : j);
                if (j < GS_CH && line < nlines && pos >= 0 && pos < n) dst[HORIZ ? (size_t)line * W + pos : (size_t)pos * W + line] = v[q];
            }
        };
        if (loader) {
#pragma unroll
            for (int k = 0; k < GS_AHEAD; ++k) fetch(k);
            __builtin_amdgcn_s_waitcnt(gs_vmcnt((GS_AHEAD - 1) * GS_LPER));   // chunk 0 has landed
        }
        gs_lds_barrier();
        for (int i = 0; i <= nchunks; ++i) {
            if (loader) {
                // chunk i + 1 has landed when at most the GS_AHEAD - 2 younger chunks are outstanding; then chunk i + GS_AHEAD goes out
                __builtin_amdgcn_s_waitcnt(gs_vmcnt((GS_AHEAD - 2) * GS_LPER));
                fetch(i + GS_AHEAD);
            } else if (storer) {
                if (i >= 1) drain(i - 1);
            } else if (i < nchunks) {
                // chunk i is filtered in place
                const gs_lf in = lds + (i % GS_NB) * (GS_CH * GS_PITCH) + (HORIZ ? lane * GS_PITCH : lane);
                constexpr int ST = HORIZ ? 1 : GS_PITCH;
                const int cnt = min(GS_CH, n - i * GS_CH);
                int s = 0;
                if (i == 0) {
                    if (fwd) {
                        // the first three outputs (gauss.cc:573-576 and the vector forms)
                        const float s0 = in[0];
                        const float t0 = s0 * (B + b1 + b2 + b3);
                        const float t1 = sizeof(C) == 4 ? (float)(in[ST] * B + t0 * b1 + s0 * (b2 + b3)) : (float)(B * in[ST] + b1 * t0 + s0 * (b2 + b3));
                        const float t2 = sizeof(C) == 4 ? (float)(in[2 * ST] * B + t1 * b1 + t0 * b2 + s0 * b3) : (float)(B * in[2 * ST] + b1 * t1 + b2 * t0 + b3 * s0);
                        in[0] = t0; in[ST] = t1; in[2 * ST] = t2;
                        m3 = t0; m2 = t1; m1 = t2;
                    } else {
                        // Triggs-Sdika boundary (m1 = tmp[n-1], m2 = tmp[n-2], m3 = tmp[n-3]); outputs n-1, n-2, n-3
                        const float t2Wp1 = (float)(sl + M[6] * (m1 - sl) + M[7] * (m2 - sl) + M[8] * (m3 - sl));
                        const float t2W = (float)(sl + M[3] * (m1 - sl) + M[4] * (m2 - sl) + M[5] * (m3 - sl));
                        const float r1 = (float)(sl + M[0] * (m1 - sl) + M[1] * (m2 - sl) + M[2] * (m3 - sl));
                        const float r2 = (float)(B * m2 + b1 * r1 + b2 * t2W + b3 * t2Wp1);
                        const float r3 = (float)(B * m3 + b1 * r2 + b2 * r1 + b3 * t2W);
                        in[0] = r1; in[ST] = r2; in[2 * ST] = r3;
                        m1 = r3; m2 = r2; m3 = r1;
                    }
                    s = 3;
                }
                // the loads do not depend on the recurrence: the next batch is read before this batch's results are stored over their inputs
                float x[16], xn[16];
                if (s + 16 <= cnt) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) x[k] = in[(s + k) * ST];
                }
                for (; s + 16 <= cnt; s += 16) {
                    const int sn = s + 32 <= cnt ? s + 16 : s;
#pragma unroll
                    for (int k = 0; k < 16; ++k) xn[k] = in[(sn + k) * ST];
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const float v = sizeof(C) == 4 ? (float)(x[k] * B + m1 * b1 + m2 * b2 + m3 * b3) : (float)(B * x[k] + b1 * m1 + b2 * m2 + b3 * m3);
                        in[(s + k) * ST] = v;
                        m3 = m2; m2 = m1; m1 = v;
                    }
#pragma unroll
                    for (int k = 0; k < 16; ++k) x[k] = xn[k];
                }
                for (; s < cnt; ++s) {
                    const float xv = in[s * ST];
                    const float v = sizeof(C) == 4 ? (float)(xv * B + m1 * b1 + m2 * b2 + m3 * b3) : (float)(B * xv + b1 * m1 + b2 * m2 + b3 * m3);
                    in[s * ST] = v;
                    m3 = m2; m2 = m1; m1 = v;
                }
            }
            gs_lds_barrier();
        }
        // the DMA that ran ahead of the last chunk has to land before the ring is reused; the backward pass reads what other waves of this group stored
        __threadfence();
        __syncthreads();
    }
}

template <bool HORIZ>
__global__ void __launch_bounds__(GS_NT) gauss_stream_kernel(GaussArgs a, int nfloat)
{
    extern __shared__ float gs_dyn_lds[];
    const gs_lf lds = (gs_lf)gs_dyn_lds;
    const int nlines = HORIZ ? a.H : a.W;
    const int line0 = blockIdx.x * 64;
    if (line0 < nfloat) gauss_stream_group<HORIZ, float>(a, line0, min(64, nfloat - line0), a.Bf, a.bf[0], a.bf[1], a.bf[2], a.Mf, lds);
    else gauss_stream_group<HORIZ, double>(a, nfloat, nlines - nfloat, a.B, a.b[0], a.b[1], a.b[2], a.M, lds);
}
} // namespace

// sigma >= 25 (GAUSS_DOUBLE): gaussHorizontal<T> / gaussVertical<T> (gauss.cc:669-713,1148-1225): every line in double, with
// a double forward buffer; one lane per line.
__device__ __forceinline__ void yvv_line64(float *p, size_t st, double *tmp, size_t tst, int n, double B, double b1, double b2, double b3, const double *M)
{
    const double s0 = p[0];
    tmp[0] = B * s0 + b1 * s0 + b2 * s0 + b3 * s0;
    tmp[tst] = B * (double)p[st] + b1 * tmp[0] + b2 * s0 + b3 * s0;
    tmp[2 * tst] = B * (double)p[2 * st] + b1 * tmp[tst] + b2 * tmp[0] + b3 * s0;
    double m3 = tmp[0], m2 = tmp[tst], m1 = tmp[2 * tst];
    for (int j = 3; j < n; j++) {
        const double v = B * (double)p[(size_t)j * st] + b1 * m1 + b2 * m2 + b3 * m3;
        tmp[(size_t)j * tst] = v;
        m3 = m2; m2 = m1; m1 = v;
    }
    const double sl = p[(size_t)(n - 1) * st];
    const double t2Wm1 = sl + M[0] * (m1 - sl) + M[1] * (m2 - sl) + M[2] * (m3 - sl);
    const double t2W = sl + M[3] * (m1 - sl) + M[4] * (m2 - sl) + M[5] * (m3 - sl);
    const double t2Wp1 = sl + M[6] * (m1 - sl) + M[7] * (m2 - sl) + M[8] * (m3 - sl);
    const double r1 = t2Wm1;
    const double r2 = B * m2 + b1 * r1 + b2 * t2W + b3 * t2Wp1;
    const double r3 = B * m3 + b1 * r2 + b2 * r1 + b3 * t2W;
    p[(size_t)(n - 1) * st] = (float)r1; p[(size_t)(n - 2) * st] = (float)r2; p[(size_t)(n - 3) * st] = (float)r3;
    double a1 = r3, a2 = r2, a3 = r1;
    for (int j = n - 4; j >= 0; j--) {
        const double v = B * tmp[(size_t)j * tst] + b1 * a1 + b2 * a2 + b3 * a3;
        p[(size_t)j * st] = (float)v;
        a3 = a2; a2 = a1; a1 = v;
    }
}
__global__ void __launch_bounds__(64) gauss_h64_kernel(GaussArgs a)
{
    const int row = blockIdx.x * 64 + threadIdx.x;
    if (row >= a.H) return;
    yvv_line64(a.img + (size_t)row * a.W, 1, a.tmp64 + (size_t)row * a.W, 1, a.W, a.B, a.b[0], a.b[1], a.b[2], a.M);
}
__global__ void __launch_bounds__(64) gauss_v64_kernel(GaussArgs a)
{
    const int col = blockIdx.x * 64 + threadIdx.x;
    if (col >= a.W) return;
    yvv_line64(a.img + col, (size_t)a.W, a.tmp64 + col, (size_t)a.W, a.H, a.B, a.b[0], a.b[1], a.b[2], a.M);
}

// ---------------------------------------------------------------- NL-means
// padded source (nlmeans.cc:98-109), dst = 0 (L111-119), mask -> (1/(mask*h2))/lutfactor (L129-136)
__global__ void __launch_bounds__(256) nlm_prepare_kernel(NlmArgs a)
{
    const long long npad = (long long)a.WW * a.HH, n = (long long)a.W * a.H;
    const float lutfactor = 100.f / 8191.f;
    // (the walk covers the 8192 table entries as well: a padded frame of fewer pixels -- below about 76 x 76 -- used to leave the table's tail unwritten)
    const long long nwalk = npad > 8192 ? npad : 8192;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < nwalk; t += (long long)gridDim.x * blockDim.x) {
        if (t < npad) {
            const int y = (int)(t / a.WW), x = (int)(t - (long long)y * a.WW);
            const int yy = y <= a.border ? 0 : y >= a.H ? a.H - 1 : y - a.border;
            const int xx = x <= a.border ? 0 : x >= a.W ? a.W - 1 : x - a.border;
            a.src[t] = a.img[(size_t)yy * a.img_stride + xx] / a.factor;
        }
        if (t < n) a.mask[t] = (1.f / (a.mask[t] * a.h2)) / lutfactor;
        if (t < 8192) a.explut[t] = xexpf_s(-((float)t * lutfactor));
    }
}
__global__ void __launch_bounds__(256) nlm_zero_kernel(NlmArgs a)
{
    const long long n = (long long)a.W * a.H;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(t / a.W), x = (int)(t - (long long)y * a.W);
        a.img[(size_t)y * a.img_stride + x] = 0.f;
        a.SW[t] = 0.f;
    }
}

static int fgrid(long long n) { long long g = (n + 255) / 256; return (int)(g < 16384 ? g : 16384); }
hipError_t launch_detail_mask(const MaskArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(dm_down_log_kernel, dim3(fgrid((long long)a.w4 * a.h4)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(dm_laplacian_kernel, dim3(fgrid((long long)a.w4 * a.h4)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(dm_up_scurve_kernel, dim3(fgrid((long long)a.W * a.H)), dim3(256), 0, s, a);
    return hipGetLastError();
}
// gaussianBlur with 0.25 <= sigma < 0.6 and src == dst (gauss.cc:1474-1483): separated 3-tap filter, gaussHorizontal3 (L446-465) into
// a line buffer, then gaussVertical3 (L467-526; its 8-column vector loop and its scalar tail evaluate the same expression up to the
// order of one commutative addition); the first / last column (row) of each pass is copied.
__global__ void gauss3_h_kernel(const float *img, float *tmp, int W, int H, float c0, float c1)
{
    FOR_IMAGE_XY(y, x, W, H) {
        const float *r = img + (size_t)y * W;
        tmp[(size_t)y * W + x] = (x == 0 || x == W - 1) ? r[x] : c1 * (r[x - 1] + r[x + 1]) + c0 * r[x];
    }
}
__global__ void gauss3_v_kernel(const float *tmp, float *img, int W, int H, float c0, float c1)
{
    FOR_IMAGE_XY(y, x, W, H) {
        const size_t i = (size_t)y * W + x;
        img[i] = (y == 0 || y == H - 1) ? tmp[i] : c1 * (tmp[i + W] + tmp[i - W]) + tmp[i] * c0;
    }
}
hipError_t launch_gaussian3(float *img, float *tmp, int W, int H, float c0, float c1, hipStream_t s)
{
    hipLaunchKernelGGL(gauss3_h_kernel, image_grid(W, H), dim3(256), 0, s, img, tmp, W, H, c0, c1);
    hipLaunchKernelGGL(gauss3_v_kernel, image_grid(W, H), dim3(256), 0, s, tmp, img, W, H, c0, c1);
    return hipGetLastError();
}

hipError_t launch_gaussian(const GaussArgs &a, hipStream_t s)
{
    if (a.tmp64) {
        hipLaunchKernelGGL(gauss_h64_kernel, dim3((a.H + 63) / 64), dim3(64), 0, s, a);
        hipLaunchKernelGGL(gauss_v64_kernel, dim3((a.W + 63) / 64), dim3(64), 0, s, a);
        return hipGetLastError();
    }
    // gaussHorizontalSse filters H - H % 4 rows with float coefficients and the rest with double ones (gauss.cc:1163-1225);
    // gaussVerticalSse W - W % 8 columns (gauss.cc:716-856)
    const int fh = a.H - a.H % 4, fv = a.W - a.W % 8;
    if (hipError_t e = dyn_lds_once(reinterpret_cast<const void *>(&gauss_stream_kernel<true>), GS_LDS_BYTES); e != hipSuccess) return e;
    if (hipError_t e = dyn_lds_once(reinterpret_cast<const void *>(&gauss_stream_kernel<false>), GS_LDS_BYTES); e != hipSuccess) return e;
    hipLaunchKernelGGL(gauss_stream_kernel<true>, dim3((fh + 63) / 64 + (fh < a.H ? 1 : 0)), dim3(GS_NT), GS_LDS_BYTES, s, a, fh);
    hipLaunchKernelGGL(gauss_stream_kernel<false>, dim3((fv + 63) / 64 + (fv < a.W ? 1 : 0)), dim3(GS_NT), GS_LDS_BYTES, s, a, fv);
    return hipGetLastError();
}
hipError_t launch_nlm(const NlmArgs &a, hipStream_t s)
{
    const long long npad = (long long)a.WW * a.HH;
    hipLaunchKernelGGL(nlm_prepare_kernel, dim3(fgrid(npad > 8192 ? npad : 8192)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(nlm_zero_kernel, dim3(fgrid((long long)a.W * a.H)), dim3(256), 0, s, a);
    // one workgroup per reference tile, a search row of offsets in flight (nlm_sweep.hip); covers every radius the reference
    // can ask for at scale >= 1 (search <= 5, patch 1..2, nlmeans.cc:62-66) -- anything else is a caller error, not a slow path
    if (!nlm_group_supported(a)) return hipErrorInvalidValue;
    return launch_nlm_group(a, s);
}

} // namespace artgpu
