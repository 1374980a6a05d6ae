// art_amd/csrc/guided.hip -- guided chroma smoothing of the denoise stage on gfx950
// (reference: rtengine/ipsmoothing.cc:334-409,875-897 denoiseGuidedSmoothing -> guided_smoothing(Channel::C);
//  rtengine/guidedfilter.cc:58-265 guidedFilter/guidedFilterLog; rtengine/rescale.h:27-74;
//  box blurs: rtengine/boxblur.h:318-556, run by the shared hblur/vblur kernels of denoise.hip).
//
// Full-resolution work is three streaming kernels (normalise+log, upsample+combine+exp, chroma
// recombination); the guided-filter statistics live on a 1/s-resolution grid (s = 3 for the default
// radius), 14 small planes.  HBM-bound: ~60 B/px.
#include <hip/hip_runtime.h>
#include "devmath.h"
#include "devsleef.h"
#include "kernels.h"

namespace artgpu {

namespace {

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

// 1. normalizeFloatTo1 (x * (1/65535)), keep the normalised input (iR,iG,iB), guide = xlin2log(max(lum,0),10),
//    chan = xlin2log(max(chan,0),10)                                      (ipsmoothing.cc:353-369, guidedfilter.cc:246-253)
__global__ void __launch_bounds__(256) gf_prepare_kernel(GuidedArgs a)
{
    const long long n = (long long)a.W * a.H;
    const float f1 = 1.f / 65535.f;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(t / a.W), x = (int)(t - (long long)y * a.W);
        const size_t si = (size_t)y * a.stride + x;
        const float r = a.rgb[0][si] * f1, g = a.rgb[1][si] * f1, b = a.rgb[2][si] * f1;
        a.in[0][t] = r; a.in[1][t] = g; a.in[2][t] = b;
        const float l = (float)(r * a.ws1[0] + g * a.ws1[1] + b * a.ws1[2]); // double matrix (TMatrix)
        a.guide[t] = xlin2log(std_max(l, 0.f), 10.f);
        a.chan[0][t] = xlin2log(std_max(r, 0.f), 10.f);
        a.chan[1][t] = xlin2log(std_max(g, 0.f), 10.f);
        a.chan[2][t] = xlin2log(std_max(b, 0.f), 10.f);
    }
}

// 2. rescaleBilinear of guide and the three channels to w x h, and the products (guidedfilter.cc:186-204):
//    low[0]=I1 -> meanI, low[1]=I1*I1 -> corrI, low[2+c]=p1 -> meanp, low[5+c]=I1*p1 -> corrIp
__global__ void __launch_bounds__(256) gf_subsample_kernel(GuidedArgs a)
{
    const long long n = (long long)a.w * a.h;
    const float col_scale = (float)a.W / (float)a.w, row_scale = (float)a.H / (float)a.h;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(t / a.w), x = (int)(t - (long long)y * a.w);
        const float fx = x * col_scale, fy = y * row_scale;
        const bool same = a.w == a.W && a.h == a.H;
        const float I1 = same ? a.guide[t] : bilinear(a.guide, a.W, a.H, fx, fy);
        a.low[0][t] = I1;
        a.low[1][t] = I1 * I1;
        const int nch = a.nch == 1 ? 1 : 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (c >= nch) break;
            const float p1 = same ? a.chan[c][t] : bilinear(a.chan[c], a.W, a.H, fx, fy);
            a.low[2 + c][t] = p1;
            a.low[5 + c][t] = I1 * p1;
        }
    }
}

// 3. after the four means: a = covIp / (varI + eps), b = meanp - a * meanI (guidedfilter.cc:206-220);
//    results overwrite low[2+c] (a) and low[5+c] (b), which are then blurred (mean a, mean b)
__global__ void __launch_bounds__(256) gf_ab_kernel(GuidedArgs a)
{
    const long long n = (long long)a.w * a.h;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
        const float meanI = a.low[0][t], corrI = a.low[1][t];
        const float varI = corrI - (meanI * meanI);
        const int nch = a.nch == 1 ? 1 : 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (c >= nch) break;
            const float meanp = a.low[2 + c][t], corrIp = a.low[5 + c][t];
            const float covIp = corrIp - (meanI * meanp);
            const float av = covIp / (varI + a.epsilon);
            const float bv = meanp - (av * meanI);
            a.low[2 + c][t] = av;
            a.low[5 + c][t] = bv;
        }
    }
}

// 4. q = bilinear(meana) * I + bilinear(meanb); chan = xlog2lin(max(q,0),10); chroma recombination with the
//    original luminance (ipsmoothing.cc:382-406); x 65535                      (guidedfilter.cc:225-240,258-264)
__global__ void __launch_bounds__(256) gf_finish_kernel(GuidedArgs a)
{
    const long long n = (long long)a.W * a.H;
    const float col_scale = (float)a.w / (float)a.W, row_scale = (float)a.h / (float)a.H;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(t / a.W), x = (int)(t - (long long)y * a.W);
        const float fx = x * col_scale, fy = y * row_scale;
        const float I = a.guide[t];
        float o[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float q = bilinear(a.low[2 + c], a.w, a.h, fx, fy) * I + bilinear(a.low[5 + c], a.w, a.h, fx, fy);
            o[c] = xlog2lin(std_max(q, 0.f), 10.f);
        }
        const float ir = a.in[0][t], ig = a.in[1][t], ib = a.in[2][t];
        const float iY = (float)(ir * a.ws1[0] + ig * a.ws1[1] + ib * a.ws1[2]);
        float oY = (float)(o[0] * a.ws1[0] + o[1] * a.ws1[1] + o[2] * a.ws1[2]);
        float ou = oY - o[2], ov = o[0] - oY;
        const float bump = oY > 1e-5f ? iY / oY : 1.f;
        ou *= bump;
        ov *= bump;
        oY = iY;
        const float B = oY - ou;
        const float R = ov + oY;
        const float G = (float)((oY - R * a.ws1[0] - B * a.ws1[2]) / a.ws1[1]);
        const size_t di = (size_t)y * a.stride + x;
        a.rgb[0][di] = R * 65535.f;
        a.rgb[1][di] = G * 65535.f;
        a.rgb[2][di] = B * 65535.f;
    }
}

// 4'. plain guidedFilter: q = bilinear(mean a) * I + bilinear(mean b) (guidedfilter.cc:225-240); q may be the src or guide plane
__global__ void __launch_bounds__(256) gf_finish_plain_kernel(GuidedArgs a)
{
    const float col_scale = (float)a.w / (float)a.W, row_scale = (float)a.h / (float)a.H;
    FOR_IMAGE_XY(y, x, a.W, a.H) {
        const float ymrs = y * row_scale;
        const float I = a.guide[(size_t)y * a.W + x];
        a.q[(size_t)y * a.q_stride + x] = bilinear(a.low[2], a.w, a.h, x * col_scale, ymrs) * I + bilinear(a.low[5], a.w, a.h, x * col_scale, ymrs);
    }
}

static int fgrid(long long n) { long long g = (n + 255) / 256; return (int)(g < 16384 ? g : 16384); }
hipError_t launch_gf_prepare(const GuidedArgs &a, hipStream_t s) { hipLaunchKernelGGL(gf_prepare_kernel, dim3(fgrid((long long)a.W * a.H)), dim3(256), 0, s, a); return hipGetLastError(); }
hipError_t launch_gf_subsample(const GuidedArgs &a, hipStream_t s) { hipLaunchKernelGGL(gf_subsample_kernel, dim3(fgrid((long long)a.w * a.h)), dim3(256), 0, s, a); return hipGetLastError(); }
hipError_t launch_gf_ab(const GuidedArgs &a, hipStream_t s) { hipLaunchKernelGGL(gf_ab_kernel, dim3(fgrid((long long)a.w * a.h)), dim3(256), 0, s, a); return hipGetLastError(); }
hipError_t launch_gf_finish_plain(const GuidedArgs &a, hipStream_t s) { hipLaunchKernelGGL(gf_finish_plain_kernel, image_grid(a.W, a.H), dim3(256), 0, s, a); return hipGetLastError(); }
hipError_t launch_gf_finish(const GuidedArgs &a, hipStream_t s) { hipLaunchKernelGGL(gf_finish_kernel, dim3(fgrid((long long)a.W * a.H)), dim3(256), 0, s, a); return hipGetLastError(); }

} // namespace artgpu
