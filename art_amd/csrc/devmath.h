// art_amd/csrc/devmath.h -- device-side fp32 primitives with the reference's x86-64 semantics.
//
// Everything here is compiled with -ffp-contract=off and hipcc's default correctly rounded
// fp32 division, so an expression written in the reference's operation order gives the
// reference's bits (rtengine is built with -ffp-contract=off, CMakeLists.txt:35-38).
//
//   sse_min/sse_max : _mm_min_ps/_mm_max_ps operand order (rtengine/helpersse2.h:168-179)
//   intp            : a*b + (1-a)*c (rtengine/rt_math.h:109-118, sleefsseavx.h:1435-1442)
//   median3         : rtengine/median.h:52-64
//   xdiv2f/xdivf    : exponent-field arithmetic (rtengine/sleef.h:1267-1301)
//   fc              : RawImage::FC (rtengine/rawimage.h:186-189)
#pragma once
#include <hip/hip_runtime.h>
#include "paramcurve.h"
#include <stdint.h>

namespace artgpu {

__device__ __forceinline__ unsigned fc(unsigned filters, unsigned row, unsigned col)
{
    return (filters >> (((((row) << 1) & 14u) + ((col) & 1u)) << 1)) & 3u;
}
__device__ __forceinline__ float sse_min(float x, float y) { return x < y ? x : y; }
__device__ __forceinline__ float sse_max(float x, float y) { return x > y ? x : y; }
__device__ __forceinline__ float std_min(float a, float b) { return b < a ? b : a; }
__device__ __forceinline__ float std_max(float a, float b) { return a < b ? b : a; }
__device__ __forceinline__ float sqr(float x) { return x * x; }
__device__ __forceinline__ float intp(float a, float b, float c) { return a * b + (1.f - a) * c; }
__device__ __forceinline__ float median3(float a, float b, float c)
{
    return sse_max(sse_min(a, b), sse_min(c, sse_max(a, b)));
}
__device__ __forceinline__ float lim01(float a) { return std_max(0.f, std_min(a, 1.f)); }
__device__ __forceinline__ float xdiv2f(float d)
{
    int i = __float_as_int(d);
    if (i & 0x7FFFFFFF) i -= 1 << 23;
    return __int_as_float(i);
}
__device__ __forceinline__ float xdivf(float d, int n)
{
    int i = __float_as_int(d);
    if (i & 0x7FFFFFFF) i -= n << 23;
    return __int_as_float(i);
}
// curves::setLutVal above the LUT (curves.h:228-230): curve->getVal(val / 65535.f) * 65535.f with getVal = the last point's y
// (kind 1: DCT_Linear / DCT_Spline / DCT_CatmullRom), t (kind 2: DCT_Empty, DCT_NURBS beyond its hash) or the analytic form of a
// DCT_Parametric curve (kind 4, paramcurve.h), diagonalcurves.cc:443-561
// PC: the kernel was instantiated for a parametric tail.  The pixel kernels exist twice: the double-precision chain of the parametric form
// costs them a third of their scalar registers (and scratch in one case) when it is merely PRESENT, so the instantiation every other
// curve kind uses does not contain it.
template <bool PC>
__device__ __forceinline__ float curve_tail(int kind, double y_last, const ParamCurve &pc, float val)
{
    const double t = (double)(val / 65535.f);
    if constexpr (PC) { if (kind == 4) return (float)(pc_getval(pc, t) * (double)65535.f); }
    return (float)((kind == 1 ? y_last : t) * (double)65535.f);
}
__device__ __forceinline__ int ngroups(int start, int bound, int step)
{
    return bound > start ? (bound - start + step - 1) / step : 0;
}

// DNG_FloatToHalf (halffloat.h:9-46)
__device__ __forceinline__ unsigned short float_to_half_dng(float f)
{
    const unsigned u = __float_as_uint(f);
    const int sign = (u >> 16) & 0x8000;
    int exponent = (int)((u >> 23) & 0xff) - (127 - 15);
    int mantissa = u & 0x007fffff;
    if (exponent <= 0) {
        if (exponent < -10) return (unsigned short)sign;
        mantissa = (mantissa | 0x00800000) >> (1 - exponent);
        if (mantissa & 0x00001000) mantissa += 0x00002000;
        return (unsigned short)(sign | (mantissa >> 13));
    } else if (exponent == 0xff - (127 - 15)) {
        return (unsigned short)(mantissa == 0 ? (sign | 0x7c00) : (sign | 0x7c00 | (mantissa >> 13)));
    }
    if (mantissa & 0x00001000) {
        mantissa += 0x00002000;
        if (mantissa & 0x00800000) { mantissa = 0; exponent += 1; }
    }
    if (exponent > 30) return (unsigned short)(sign | 0x7c00);
    return (unsigned short)(sign | (exponent << 10) | (mantissa >> 13));
}

} // namespace artgpu

// pixels per thread and batch of the persistent 1024-thread LUT-in-LDS pixel kernels (rgb2yuv_lds, yuv2rgb_lds, tone_std_lds): with one
// such workgroup per CU the loads of a batch are all that is in flight, so the batch is what hides the memory latency
#ifndef LDSK_PX
#define LDSK_PX 8
#endif
#ifndef LDSK_ROWS
#define LDSK_ROWS 2      // rows per batch (the rows a workgroup owns are gridDim.x apart)
#endif

// Pixel loops without integer division: blockIdx.y strides over rows, blockIdx.x * blockDim.x + threadIdx.x over columns
// (a flat index costs a 64-bit division per pixel, which showed up as ~1.5 TB/s on kernels that should stream at 4 TB/s).
#define FOR_IMAGE_XY(yv, xv, W, H)                              \
    for (int yv = blockIdx.y; yv < (H); yv += gridDim.y)        \
        for (int xv = blockIdx.x * blockDim.x + threadIdx.x; xv < (W); xv += gridDim.x * blockDim.x)
static inline dim3 image_grid(int w, int h)
{
    const int gx = (w + 255) / 256, gy = h;
    return dim3(gx < 1 ? 1 : (gx > 64 ? 64 : gx), gy < 1 ? 1 : (gy > 32768 ? 32768 : gy));
}
