// art_amd/csrc/lab.hip -- Imagefloat's RGB <-> LAB mode switch and ImProcFunctions::labAdjustments' pixel work on gfx950
// (reference: rtengine/imagefloat.cc:841-876,941-970; rtengine/color.cc:826-894,1203-1275,1382-1437; rtengine/LUT.h:349-377,436-459;
//  rtengine/iplabadjustments.cc:236-345).
//
// The reference runs these loops four pixels at a time (SSE2) with a scalar tail, and the two forms differ in their arithmetic:
//   * LUTf::operator[](vfloat) interpolates as diff*upper + (1-diff)*lower with clamped indices, operator[](float) as p1 + (p2-p1)*diff
//     with the LUT's clip flags;
//   * Color::XYZ2Lab(vfloat) takes the per-lane scalar path for the WHOLE group of four as soon as one lane is outside [0, 65535];
//   * Color::Lab2XYZ(vfloat) computes Y as select(fy^3, L/kappa) * 65535, the scalar one as 65535*fy*fy*fy or 65535*L/kappa in double.
// A pixel's form is decided by its column (x < W - W%4: vector) and, for XYZ2Lab, by its group's lanes (one shuffle-or over the
// aligned group of four lanes).  Planes follow Imagefloat's LAB convention: g = L, r = a, b = b.  All HBM-bound streaming: 24 B/px
// per conversion, 24 B/px for the curves (the three LUTs, 0.6 MB, stay in L2), 4 B/px for the histogram.
#include <hip/hip_runtime.h>
#include "devmath.h"
#include "devsleef.h"
#include "kernels.h"

namespace artgpu {
namespace {

constexpr float D50x = 0.9642f, D50z = 0.8249f, MAXVALF = 65535.f;

// Color::computeXYZ2Lab / computeXYZ2LabY (color.cc:1247-1275)
__device__ __forceinline__ float xyz2lab_s(const float *__restrict__ cachef, float f)
{
    if (f != f) return f;
    if (f < 0.f) return (float)(327.68 * (((24389.0 / 27.0) * (double)f / (double)MAXVALF + 16.0) / 116.0));
    if (f > 65535.f) return 327.68f * xcbrtf_s(f / MAXVALF);
    return lutf_lookup<false>(cachef, 65536, f);
}
__device__ __forceinline__ float xyz2laby_s(const float *__restrict__ cachefy, float f)
{
    if (f != f) return f;
    if (f < 0.f) return (float)(327.68 * ((24389.0 / 27.0) * (double)f / (double)MAXVALF));
    if (f > 65535.f) return 327.68f * (116.f * xcbrtf_s(f / MAXVALF) - 16.f);
    return lutf_lookup<false>(cachefy, 65536, f);
}
__device__ __forceinline__ float f2xyz(float f)
{
    const float epsilonExpInv3f = (float)(6.0 / 29.0), kappaInvf = (float)(27.0 / 24389.0);
    const float res1 = f * f * f, res2 = (116.f * f - 16.f) * kappaInvf;
    return f > epsilonExpInv3f ? res1 : res2;
}

// Imagefloat::rgb_to_lab (imagefloat.cc:841-876).  All 256 threads stay in the column loop so that the group shuffle is defined.
__global__ void __launch_bounds__(256) rgb_to_lab_kernel(LabArgs a)
{
    const int W = a.w, W4 = W & ~3;
    for (int y = blockIdx.y; y < a.h; y += gridDim.y)
        for (int x0 = blockIdx.x * 256; x0 < W; x0 += gridDim.x * 256) {
            const int x = x0 + (int)threadIdx.x;
            const bool in = x < W;
            const size_t i = (size_t)y * a.stride + (in ? x : W - 1);
            const float R = a.img[0][i], G = a.img[1][i], B = a.img[2][i];
            const float X = a.ws[0] * R + a.ws[1] * G + a.ws[2] * B, Y = a.ws[3] * R + a.ws[4] * G + a.ws[5] * B, Z = a.ws[6] * R + a.ws[7] * G + a.ws[8] * B;
            const float xs = X / D50x, zs = Z / D50z;
            int slow = (sse_max(xs, sse_max(Y, zs)) > MAXVALF) || (sse_min(xs, sse_min(Y, zs)) < 0.f);
            slow |= __shfl_xor(slow, 1);
            slow |= __shfl_xor(slow, 2);
            float L, A, Bv;
            if (x >= W4 || slow) {                  // scalar tail, or a group with a lane outside [0, 65535] (color.cc:1415-1427)
                const float fx = xyz2lab_s(a.cachef, xs), fy = xyz2lab_s(a.cachef, Y), fz = xyz2lab_s(a.cachef, zs);
                L = xyz2laby_s(a.cachefy, Y);
                A = 500.0f * (fx - fy);
                Bv = 200.0f * (fy - fz);
            } else {
                const float fx = lutf_vlookup(a.cachef, 65536, xs), fy = lutf_vlookup(a.cachef, 65536, Y), fz = lutf_vlookup(a.cachef, 65536, zs);
                L = lutf_vlookup(a.cachefy, 65536, Y);
                A = 500.f * (fx - fy);
                Bv = 200.f * (fy - fz);
            }
            if (in) { a.img[1][i] = L; a.img[0][i] = A; a.img[2][i] = Bv; }
        }
}

// Imagefloat::lab_to_rgb (imagefloat.cc:941-970)
__global__ void __launch_bounds__(256) lab_to_rgb_kernel(LabArgs a)
{
    const int W4 = a.w & ~3;
    const float c1By116 = (float)(1.0 / 116.0), c16By116 = (float)(16.0 / 116.0);
    FOR_IMAGE_XY(y, x, a.w, a.h) {
        const size_t i = (size_t)y * a.stride + x;
        const float LL = a.img[1][i] / 327.68f, aa = a.img[0][i] / 327.68f, bb = a.img[2][i] / 327.68f;
        const float fy = (c1By116 * LL) + c16By116;
        const float fx = (0.002f * aa) + fy;
        const float fz = fy - (0.005f * bb);
        const float xx = 65535.0f * f2xyz(fx) * D50x;
        const float zz = 65535.0f * f2xyz(fz) * D50z;
        float yy;
        if (x < W4) {
            const float res1 = fy * fy * fy, res2 = LL / (float)(24389.0 / 27.0);
            yy = (LL > 8.f ? res1 : res2) * 65535.f;
        } else {
            yy = ((double)LL > 8.0) ? 65535.0f * fy * fy * fy : (float)((double)(65535.0f * LL) / (24389.0 / 27.0));
        }
        a.img[0][i] = a.iws[0] * xx + a.iws[1] * yy + a.iws[2] * zz;
        a.img[1][i] = a.iws[3] * xx + a.iws[4] * yy + a.iws[5] * zz;
        a.img[2][i] = a.iws[6] * xx + a.iws[7] * yy + a.iws[8] * zz;
    }
}

// hist16[(int)L]++ (iplabadjustments.cc:300-327); LUTu::operator[](int) clamps the index, an out-of-range float -> int is x86's INT_MIN
__global__ void __launch_bounds__(256) lab_hist_kernel(LabArgs a)
{
    FOR_IMAGE_XY(y, x, a.w, a.h) {
        const float L = a.img[1][(size_t)y * a.stride + x];
        int idx = (L >= -2147483648.f && L < 2147483648.f) ? (int)L : (int)0x80000000;
        idx = idx < 0 ? 0 : (idx > 65535 ? 65535 : idx);
        atomicAdd(&a.hist[idx], 1u);
    }
}

// lab_adjustments' curve loop (iplabadjustments.cc:236-264)
__global__ void __launch_bounds__(256) lab_adjust_kernel(LabArgs a)
{
    const int W4 = a.w & ~3;
    FOR_IMAGE_XY(y, x, a.w, a.h) {
        const size_t i = (size_t)y * a.stride + x;
        float L = a.img[1][i], A = a.img[0][i], B = a.img[2][i];
        if (x < W4) {
            L = lutf_vlookup(a.lcurve, 32770, L);
            A = (lutf_vlookup(a.acurve, 65536, A + 32768.f) - 32768.f) * a.chroma;
            B = (lutf_vlookup(a.bcurve, 65536, B + 32768.f) - 32768.f) * a.chroma;
        } else {
            {   // LUTf(32770, 0): no clipping on either side
                int idx = (int)L;
                if (L < 0.f || !(L == L)) idx = 0; else if (L > 32768.f) idx = 32768;
                const float diff = L - (float)idx, p1 = a.lcurve[idx], p2 = a.lcurve[idx + 1] - p1;
                L = p1 + p2 * diff;
            }
            A = (lutf_lookup<true>(a.acurve, 65536, A + 32768.f) - 32768.f) * a.chroma;
            B = (lutf_lookup<true>(a.bcurve, 65536, B + 32768.f) - 32768.f) * a.chroma;
        }
        a.img[1][i] = L; a.img[0][i] = A; a.img[2][i] = B;
    }
}

} // namespace

hipError_t launch_rgb_to_lab(const LabArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(rgb_to_lab_kernel, image_grid(a.w, a.h), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_lab_to_rgb(const LabArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(lab_to_rgb_kernel, image_grid(a.w, a.h), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_lab_hist(const LabArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(lab_hist_kernel, image_grid(a.w, a.h), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_lab_adjust(const LabArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(lab_adjust_kernel, image_grid(a.w, a.h), dim3(256), 0, s, a);
    return hipGetLastError();
}

} // namespace artgpu
