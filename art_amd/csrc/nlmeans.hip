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
__device__ __forceinline__ float pow_F(float a, float b) { return xexpf_s(b * xlogf_s(a)); }
__device__ __forceinline__ float xlin2log(float x, float base) { return xlogf_s(x * (base - 1.f) + 1.f) / xlogf_s(base); }
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
template <typename C>
__device__ __forceinline__ void yvv_line(float *__restrict__ p, size_t st, float *__restrict__ tmp, size_t tst, int n, C B, C b1, C b2, C b3, const C *M)
{
    // forward (tmp may alias nothing; float storage as in the reference's AlignedMatrix<float>)
    const float s0 = p[0], sl = p[(size_t)(n - 1) * st];
    float t0 = s0 * (B + b1 + b2 + b3);
    float t1 = sizeof(C) == 4 ? (float)(p[st] * B + t0 * b1 + s0 * (b2 + b3)) : (float)(B * p[st] + b1 * t0 + s0 * (b2 + b3));
    float t2 = sizeof(C) == 4 ? (float)(p[2 * st] * B + t1 * b1 + t0 * b2 + s0 * b3) : (float)(B * p[2 * st] + b1 * t1 + b2 * t0 + b3 * s0);
    tmp[0] = t0; tmp[tst] = t1; tmp[2 * tst] = t2;
    float m3 = t0, m2 = t1, m1 = t2;
    // the loads do not depend on the recurrence: sixteen of them are issued before the sixteen steps that consume them -- and one batch
    // AHEAD of the stores of the current batch: vmcnt retires in order, so a wait for loads issued after stores would wait for the stores too
    int j = 3;
    float x[16], xn[16];
    if (j + 16 <= n) {
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = p[(size_t)(j + k) * st];
    }
    for (; j + 16 <= n; j += 16) {
        const bool more = j + 32 <= n;
#pragma unroll
        for (int k = 0; k < 16; ++k) xn[k] = p[(size_t)(more ? j + 16 + k : j + k) * st];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float v = sizeof(C) == 4 ? (float)(x[k] * B + m1 * b1 + m2 * b2 + m3 * b3) : (float)(B * x[k] + b1 * m1 + b2 * m2 + b3 * m3);
            tmp[(size_t)(j + k) * tst] = v;
            m3 = m2; m2 = m1; m1 = v;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = xn[k];
    }
    for (; j < n; j++) {
        const float v = sizeof(C) == 4 ? (float)(p[(size_t)j * st] * B + m1 * b1 + m2 * b2 + m3 * b3) : (float)(B * p[(size_t)j * st] + b1 * m1 + b2 * m2 + b3 * m3);
        tmp[(size_t)j * tst] = v;
        m3 = m2; m2 = m1; m1 = v;
    }
    // Triggs-Sdika boundary (m1 = tmp[n-1], m2 = tmp[n-2], m3 = tmp[n-3])
    const float t2Wp1 = (float)(sl + M[6] * (m1 - sl) + M[7] * (m2 - sl) + M[8] * (m3 - sl));
    const float t2W = (float)(sl + M[3] * (m1 - sl) + M[4] * (m2 - sl) + M[5] * (m3 - sl));
    const float r1 = (float)(sl + M[0] * (m1 - sl) + M[1] * (m2 - sl) + M[2] * (m3 - sl));
    const float r2 = (float)(B * m2 + b1 * r1 + b2 * t2W + b3 * t2Wp1);
    const float r3 = (float)(B * m3 + b1 * r2 + b2 * r1 + b3 * t2W);
    p[(size_t)(n - 1) * st] = r1; p[(size_t)(n - 2) * st] = r2; p[(size_t)(n - 3) * st] = r3;
    float a1 = r3, a2 = r2, a3 = r1; // outputs at j+1, j+2, j+3
    int jb = n - 4;
    if (jb - 15 >= 0) {
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = tmp[(size_t)(jb - k) * tst];
    }
    for (; jb - 15 >= 0; jb -= 16) {
        const bool more = jb - 31 >= 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) xn[k] = tmp[(size_t)(more ? jb - 16 - k : jb - k) * tst];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float v = sizeof(C) == 4 ? (float)(x[k] * B + a1 * b1 + a2 * b2 + a3 * b3) : (float)(B * x[k] + b1 * a1 + b2 * a2 + b3 * a3);
            p[(size_t)(jb - k) * st] = v;
            a3 = a2; a2 = a1; a1 = v;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = xn[k];
    }
    for (; jb >= 0; jb--) {
        const float tj = tmp[(size_t)jb * tst];
        const float v = sizeof(C) == 4 ? (float)(tj * B + a1 * b1 + a2 * b2 + a3 * b3) : (float)(B * tj + b1 * a1 + b2 * a2 + b3 * a3);
        p[(size_t)jb * st] = v;
        a3 = a2; a2 = a1; a1 = v;
    }
}

__global__ void __launch_bounds__(64) gauss_h_kernel(GaussArgs a)
{
    const int row = blockIdx.x * 64 + threadIdx.x;
    if (row >= a.H) return;
    float *p = a.img + (size_t)row * a.W, *tmp = a.tmp + (size_t)row * a.W;
    if (row < a.H - (a.H % 4)) yvv_line<float>(p, 1, tmp, 1, a.W, a.Bf, a.bf[0], a.bf[1], a.bf[2], a.Mf);
    else yvv_line<double>(p, 1, tmp, 1, a.W, a.B, a.b[0], a.b[1], a.b[2], a.M);
}
__global__ void __launch_bounds__(64) gauss_v_kernel(GaussArgs a)
{
    const int col = blockIdx.x * 64 + threadIdx.x;
    if (col >= a.W) return;
    float *p = a.img + col, *tmp = a.tmp + col;
    if (col < a.W - (a.W % 8)) yvv_line<float>(p, (size_t)a.W, tmp, (size_t)a.W, a.H, a.Bf, a.bf[0], a.bf[1], a.bf[2], a.Mf);
    else yvv_line<double>(p, (size_t)a.W, tmp, (size_t)a.W, a.H, a.B, a.b[0], a.b[1], a.b[2], a.M);
}

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
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < npad; t += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(t / a.WW), x = (int)(t - (long long)y * a.WW);
        const int yy = y <= a.border ? 0 : y >= a.H ? a.H - 1 : y - a.border;
        const int xx = x <= a.border ? 0 : x >= a.W ? a.W - 1 : x - a.border;
        a.src[t] = a.img[(size_t)yy * a.img_stride + xx] / a.factor;
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
    hipLaunchKernelGGL(gauss_h_kernel, dim3((a.H + 63) / 64), dim3(64), 0, s, a);
    hipLaunchKernelGGL(gauss_v_kernel, dim3((a.W + 63) / 64), dim3(64), 0, s, a);
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
