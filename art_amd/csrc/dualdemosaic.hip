// art_amd/csrc/dualdemosaic.hip -- the blend half of RawImageSource::dual_demosaic_RT on gfx950 (Bayer, second demosaicer = bilinear;
// reference: rtengine/dual_demosaic_RT.cc:73-152, rtengine/color.cc:1343-1379 RGB2L, rtengine/rt_algo.cc:40-176,315-498
// buildBlendMask + automatic contrast threshold, rtengine/bayer_bilinear_demosaic.cc:33-77).
//
//   rgb2l            L* of the first demosaicer's output through Color::cachefy: groups of four columns take the vector LUT form unless
//                    one of their lanes leaves [0, 65535] (then all four take the scalar one), the W % 4 tail is scalar
//   blend_contrast   the 8-neighbour contrast -> sigmoid blend factor (vector / scalar exp by column), then the 2-pixel frame
//   tile_stats       tileAverage / tileVariance of the automatic threshold search: the reference sums a tile in four SSE lanes down the
//                    rows and adds them as (v0+v2)+(v1+v3); four GPU lanes per tile do exactly that (one shuffle pair), thousands of tiles
//                    in parallel; the first-minimum search over the resulting table runs on the host like the reference's serial loop
//   contrast_threshold  calcContrastThreshold: 99 candidate thresholds x 4 lanes in one workgroup over the flattest tile
//   bilinear_blend   red/green/blue = intp(blend, first demosaicer, bilinear) per Bayer pair
// All streaming (L 12 B/px in + 4 out, mask 4 + 4, blend 20 + 12 B/px); the gaussian blur of the mask is the shared YvV kernel.
#include <hip/hip_runtime.h>
#include "devmath.h"
#include "devsleef.h"
#include "kernels.h"

namespace artgpu {
namespace {

constexpr float MAXVALF = 65535.f;
__device__ __forceinline__ float xyz2laby_s(const float *__restrict__ cachefy, float f)
{
    if (f != f) return f;
    if (f < 0.f) return (float)(327.68 * ((24389.0 / 27.0) * (double)f / (double)MAXVALF));
    if (f > 65535.f) return 327.68f * (116.f * xcbrtf_s(f / MAXVALF) - 16.f);
    return lutf_lookup<false>(cachefy, 65536, f);
}

__global__ void __launch_bounds__(256) rgb2l_kernel(DualArgs a)
{
    const int W = a.w, W4 = W & ~3;
    const float w0 = 0.212671f, w1 = 0.715160f, w2 = 0.072169f;
    for (int y = blockIdx.y; y < a.h; y += gridDim.y)
        for (int x0 = blockIdx.x * 256; x0 < W; x0 += gridDim.x * 256) {
            const int x = x0 + (int)threadIdx.x;
            const bool in = x < W;
            const size_t i = (size_t)y * a.stride + (in ? x : W - 1);
            const float yv = w0 * a.rgb[0][i] + w1 * a.rgb[1][i] + w2 * a.rgb[2][i];
            int slow = (yv > MAXVALF) || (yv < 0.f);
            slow |= __shfl_xor(slow, 1);
            slow |= __shfl_xor(slow, 2);
            const float L = (x >= W4 || slow) ? xyz2laby_s(a.cachefy, yv) : lutf_vlookup(a.cachefy, 65536, yv);
            if (in) a.L[(size_t)y * W + x] = L;
        }
}

__device__ __forceinline__ float contrast_at(const float *__restrict__ p, int W, float scale)
{
    return sqrtf(sqr(p[1] - p[-1]) + sqr(p[W] - p[-W]) + sqr(p[2] - p[-2]) + sqr(p[2 * W] - p[-2 * W])) * scale;
}
__device__ __forceinline__ float blend_factor(float val, float thr, bool vec)
{
    const float e = 16.f - 16.f * val / thr;
    return 1.f / (1.f + (vec ? xexpf_v(e) : xexpf_s(e)));
}
// rt_algo.cc:436-461: interior of the mask
__global__ void __launch_bounds__(256) blend_contrast_kernel(DualArgs a)
{
    const int W = a.w, H = a.h;
    const float scale = 0.0625f / 327.68f * 1.f;
    const int nvec = W - 5 > 2 ? 4 * ((W - 7 + 3) / 4) : 0;      // columns [2, 2 + nvec) are covered by the 4-wide loop
    FOR_IMAGE_XY(j, i, W, H) {
        if (j < 2 || j >= H - 2 || i < 2 || i >= W - 2) continue;
        const float c = contrast_at(a.L + (size_t)j * W + i, W, scale);
        a.blend[(size_t)j * W + i] = 1.f * blend_factor(c, a.threshold, i < 2 + nvec);
    }
}
// rt_algo.cc:466-483: step 0 = the two top / bottom rows (columns 2..W-3), step 1 = the two left / right columns of every row
__global__ void __launch_bounds__(256) blend_frame_kernel(DualArgs a, int step)
{
    const int W = a.w, H = a.h;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (step == 0) {
        if (t >= 2 && t < W - 2) {
            a.blend[t] = a.blend[(size_t)W + t] = a.blend[(size_t)2 * W + t];
            a.blend[(size_t)(H - 2) * W + t] = a.blend[(size_t)(H - 1) * W + t] = a.blend[(size_t)(H - 3) * W + t];
        }
    } else if (t < H) {
        float *b = a.blend + (size_t)t * W;
        b[0] = b[1] = b[2];
        b[W - 2] = b[W - 1] = b[W - 3];
    }
}
__global__ void __launch_bounds__(256) blend_fill_kernel(DualArgs a, float v)
{
    FOR_IMAGE_XY(j, i, a.w, a.h) a.blend[(size_t)j * a.w + i] = v;
}

// tileAverage + tileVariance (rt_algo.cc:58-110) for tiles (y0 + ti * step, x0 + tj * step), tile size ts (a multiple of four: no scalar
// tail).  Four lanes per tile = the four SSE lanes; var[ti * nW + tj] with the reference's infinity rules (rt_algo.cc:336-347).
__global__ void __launch_bounds__(256) tile_stats_kernel(DualArgs a, int nH, int nW, int y0, int x0, int step, int ts, float *var)
{
    const long long ntiles = (long long)nH * nW;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    long long tile = t >> 2;
    const int k = (int)(t & 3);
    const bool valid = tile < ntiles;
    if (!valid) tile = ntiles - 1;
    const int ti = (int)(tile / nW), tj = (int)(tile - (long long)ti * nW);
    const float *p = a.L + (size_t)(y0 + ti * step) * a.w + x0 + tj * step + k;
    float v = 0.f;
    for (int y = 0; y < ts; ++y)
        for (int x = 0; x < ts; x += 4) v += p[(size_t)y * a.w + x];
    float s = v + __shfl_xor(v, 2);
    s = s + __shfl_xor(s, 1);
    float avg = 0.f;
    avg += s;
    avg = avg / (float)(ts * ts);
    v = 0.f;
    for (int y = 0; y < ts; ++y)
        for (int x = 0; x < ts; x += 4) v += sqr(p[(size_t)y * a.w + x] - avg);
    s = v + __shfl_xor(v, 2);
    s = s + __shfl_xor(s, 1);
    float vr = 0.f;
    vr += s;
    vr = vr / ((float)(ts * ts) * avg);
    float r = __builtin_inff();
    if (!(avg < 2000.f || avg > 20000.f)) r = vr < 0.5f ? __builtin_inff() : vr;
    if (valid && k == 0) var[tile] = r;
}

// calcContrastThreshold (rt_algo.cc:112-176) of the tile at (ty, tx); ts = 80 or 40 (ts - 4 is a multiple of four: all sums are 4-lane)
__global__ void __launch_bounds__(448) contrast_threshold_kernel(DualArgs a, int ty, int tx, int ts, float *result)
{
    __shared__ float bl[76 * 76];
    __shared__ int ok[100];
    const int n = ts - 4;
    const float scale = 0.0625f / 327.68f * 1.f;
    for (int q = threadIdx.x; q < n * n; q += 448) {
        const int j = q / n, i = q - j * n;
        bl[q] = contrast_at(a.L + (size_t)(ty + 2 + j) * a.w + tx + 2 + i, a.w, scale);
    }
    __syncthreads();
    const int tid = threadIdx.x;
    const int c = min(1 + tid / 4, 99), k = tid & 3;
    const float thr = c / 100.f;
    float sv = 0.f;
    for (int j = 0; j < n; ++j)
        for (int i = k; i < n; i += 4) sv += blend_factor(bl[j * n + i], thr, true);
    float s = sv + __shfl_xor(sv, 2);
    s = s + __shfl_xor(s, 1);
    float sum = 0.f;
    sum += s;
    const float limit = (float)(n * n) / 100.f;
    if (k == 0 && 1 + tid / 4 <= 99) ok[c] = sum <= limit;
    __syncthreads();
    if (tid == 0) {
        int cc = 1;
        for (; cc < 100; ++cc) if (ok[cc]) break;
        *result = cc / 100.f;
    }
}

// bayer_bilinear_demosaic(blend, ...) (bayer_bilinear_demosaic.cc:44-62): one thread per pair (green site + its right neighbour)
__global__ void __launch_bounds__(256) bilinear_blend_kernel(DualArgs a)
{
    const int W = a.w, H = a.h;
    for (int i = 1 + blockIdx.y; i < H - 1; i += gridDim.y) {
        const bool swap = fc(a.filters, i, 0) == 2 || fc(a.filters, i, 1) == 2;
        float *ng1 = swap ? a.rgb[2] : a.rgb[0], *ng2 = swap ? a.rgb[0] : a.rgb[2], *green = a.rgb[1];
        const int j0 = 2 - (fc(a.filters, i, 1) & 1);
        for (int p = blockIdx.x * 256 + threadIdx.x; j0 + 2 * p < W - 2; p += gridDim.x * 256) {
            const int j = j0 + 2 * p;
            const float *r = a.raw + (size_t)i * a.raw_stride + j;
            const long long rs = (long long)a.raw_stride;
            const size_t o = (size_t)i * a.stride + j;
            const float b0 = a.blend[(size_t)i * W + j], b1 = a.blend[(size_t)i * W + j + 1];
            green[o] = intp(b0, green[o], r[0]);
            ng1[o] = intp(b0, ng1[o], (r[-1] + r[1]) * 0.5f);
            ng2[o] = intp(b0, ng2[o], (r[-rs] + r[rs]) * 0.5f);
            green[o + 1] = intp(b1, green[o + 1], ((r[-rs + 1] + r[0]) + (r[2] + r[rs + 1])) * 0.25f);
            ng1[o + 1] = intp(b1, ng1[o + 1], r[1]);
            ng2[o + 1] = intp(b1, ng2[o + 1], ((r[-rs] + r[-rs + 2]) + (r[rs] + r[rs + 2])) * 0.25f);
        }
    }
}

} // namespace

hipError_t launch_rgb2l(const DualArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(rgb2l_kernel, image_grid(a.w, a.h), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_blend_mask(const DualArgs &a, hipStream_t s)
{
    if (a.threshold == 0.f) {
        hipLaunchKernelGGL(blend_fill_kernel, image_grid(a.w, a.h), dim3(256), 0, s, a, 1.f);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(blend_contrast_kernel, image_grid(a.w, a.h), dim3(256), 0, s, a);
    hipLaunchKernelGGL(blend_frame_kernel, dim3((a.w + 255) / 256), dim3(256), 0, s, a, 0);
    hipLaunchKernelGGL(blend_frame_kernel, dim3((a.h + 255) / 256), dim3(256), 0, s, a, 1);
    return hipGetLastError();
}
hipError_t launch_tile_stats(const DualArgs &a, int nH, int nW, int y0, int x0, int step, int ts, float *var, hipStream_t s)
{
    const long long threads = (long long)nH * nW * 4;
    hipLaunchKernelGGL(tile_stats_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, a, nH, nW, y0, x0, step, ts, var);
    return hipGetLastError();
}
hipError_t launch_contrast_threshold(const DualArgs &a, int ty, int tx, int ts, float *result, hipStream_t s)
{
    hipLaunchKernelGGL(contrast_threshold_kernel, dim3(1), dim3(448), 0, s, a, ty, tx, ts, result);
    return hipGetLastError();
}
hipError_t launch_bilinear_blend(const DualArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(bilinear_blend_kernel, image_grid((a.w + 1) / 2, a.h), dim3(256), 0, s, a);
    return hipGetLastError();
}

} // namespace artgpu
