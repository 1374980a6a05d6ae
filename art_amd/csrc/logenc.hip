// art_amd/csrc/logenc.hip -- ImProcFunctions::logEncoding on gfx950 (reference: rtengine/iplogenc.cc:95-316,395-402).
//
// Per-pixel brightness-norm tone mapping (ACES-style log2 encode + optional log2lin toe) with ART's "regularisation": the
// norm is posterised in log space, smoothed by rtengine::guidedFilter with an image-sized radius (max(W,H)/30) and blended
// with the per-pixel factor.  Three streaming kernels around the shared guided filter; everything is scalar sleef
// (xlogf/xexpf/pow_F) in the reference's operation order, ws in double like Color::rgbLuminance<double>.
// HBM-bound: 24 B/px direct; prepare 12+8, blend 16+12 B/px (+ the guided filter on the W x H norm planes).
#include <hip/hip_runtime.h>
#include <float.h>
#include "devmath.h"
#include "devsleef.h"
#include "kernels.h"

namespace artgpu {
namespace {

__device__ __forceinline__ float le_pow_F(float a, float b) { return xexpf_s(b * xlogf_s(a)); }

// power_norm / norm (iplogenc.cc:95-116)
__device__ __forceinline__ float le_norm(const LogEncArgs &a, float r, float g, float b)
{
    const float hi = FLT_MAX / 100.f;
    const float lum = (float)((double)r * a.ws1[0] + (double)g * a.ws1[1] + (double)b * a.ws1[2]);
    r = fabsf(r); g = fabsf(g); b = fabsf(b);
    const float r2 = r * r, g2 = g * g, b2 = b * b;
    const float d = r2 + g2 + b2;
    const float n = r * r2 + g * g2 + b * b2;
    const float pn = n / std_max(d, 1e-12f);
    return std_min(hi, pn / 2.f + lum / 2.f);
}
// The highlight-compression curve (L152-170).  The reference evaluates it with the C library's powf, which is not specified bit
// for bit (glibc documents < 1 ULP); here powf(a, b) is the double-precision pow rounded to float -- the correctly rounded value
// except for results within ~1e-16 relative of a rounding boundary, which is also what glibc's powf returns in all but about one
// case in 10^5.  This is the ONE place of the tool that is tolerance-checked instead of bit-checked (tests/test_gpu_logenc.py).
__device__ __forceinline__ float le_powf(float x, float y) { return (float)pow((double)x, (double)y); }
__device__ __forceinline__ float le_compr(const LogEncArgs &a, float x)
{
    constexpr float compr_t = 0.8f;
    if (x < compr_t) return x;
    const float n = (x - compr_t) / a.compr_s;
    const float d = le_powf(1.f + le_powf((x - compr_t) / a.compr_s, a.compr_p), 1.f / a.compr_p);
    float res = compr_t + a.compr_s * n / d;
    if (a.hlcompr_factor < 0.1f) res = intp(a.hlcompr_factor * 10.f, res, x);
    return res;
}
// the `apply` lambda (L172-188)
__device__ __forceinline__ float le_apply(const LogEncArgs &a, float noise, float log2, float x)
{
    x = std_max(x, noise);
    x = std_max(x / a.gray, noise);
    if (a.hlcompr) x = le_compr(a, x);
    x = std_max((xlogf_s(x) / log2 - a.shadows_range) / a.dynamic_range, noise);
    if (a.linbase > 0.f) x = (le_pow_F(a.linbase, x) - 1.f) / (a.linbase - 1.f);   // xlog2lin (sleef.h:1309-1313)
    return x;
}
__device__ __forceinline__ float le_sf(float noise, float s, float c) { return c > noise ? 1.f - std_min(fabsf(s) / c, 1.f) : 0.f; }
// apply_sat (L200-211)
__device__ __forceinline__ void le_apply_sat(const LogEncArgs &a, float noise, float &r, float &g, float &b, float f)
{
    const float ll = (float)((double)r * a.ws1[0] + (double)g * a.ws1[1] + (double)b * a.ws1[2]);
    const float rl = r - ll, gl = g - ll, bl = b - ll;
    const float m = std_max(std_max(le_sf(noise, rl, r), le_sf(noise, gl, g)), le_sf(noise, bl, b));
    const float s = intp(m, le_pow_F(f, 0.3f) * 0.6f + 0.4f, 1.f);
    r = ll + s * rl; g = ll + s * gl; b = ll + s * bl;
}

// regularization == 0 (L215-244)
__global__ void __launch_bounds__(256) logenc_direct_kernel(LogEncArgs a)
{
    const float noise = le_pow_F(2.f, -16.f), log2 = xlogf_s(2.f);
    FOR_IMAGE_XY(y, x, a.w, a.h) {
        const size_t i = (size_t)y * a.stride + x;
        float r = a.img[0][i], g = a.img[1][i], b = a.img[2][i];
        const float m = le_norm(a, r / 65535.f, g / 65535.f, b / 65535.f);
        if (m > noise) {
            const float f = le_apply(a, noise, log2, m) / m;
            r *= f; b *= f; g *= f;
            if (a.satcontrol && f < 1.f) le_apply_sat(a, noise, r, g, b, f);
        }
        a.img[0][i] = r; a.img[1][i] = g; a.img[2][i] = b;
    }
}
// the posterised log-norm and its guide (L246-270)
__global__ void __launch_bounds__(256) logenc_prepare_kernel(LogEncArgs a)
{
    FOR_IMAGE_XY(y, x, a.w, a.h) {
        const size_t i = (size_t)y * a.stride + x, o = (size_t)y * a.w + x;
        float v = le_norm(a, a.img[0][i], a.img[1][i], a.img[2][i]) / 65535.f;
        v = std_max(1e-5f, std_min(v, 128.f));
        a.Y2[o] = v;
        const float l = xlogf_s(v);
        const float ll = roundf(l * 20.f) / 20.f;
        a.Y[o] = xexpf_s(ll);
    }
}
// blend of the smoothed and per-pixel factors (L271-314)
__global__ void __launch_bounds__(256) logenc_blend_kernel(LogEncArgs a)
{
    const float noise = le_pow_F(2.f, -16.f), log2 = xlogf_s(2.f);
    FOR_IMAGE_XY(y, x, a.w, a.h) {
        const size_t i = (size_t)y * a.stride + x;
        const float t = a.Y[(size_t)y * a.w + x];
        if (!(t > noise)) continue;
        float r = a.img[0][i], g = a.img[1][i], b = a.img[2][i];
        const float t2 = le_norm(a, r / 65535.f, g / 65535.f, b / 65535.f);
        if (!(t2 > noise)) continue;
        float f = le_apply(a, noise, log2, t) / t;
        const float f2 = le_apply(a, noise, log2, t2) / t2;
        f = intp(a.blend, f, f2);
        r *= f; g *= f; b *= f;
        if (a.satcontrol && f < 1.f) le_apply_sat(a, noise, r, g, b, f);
        a.img[0][i] = r; a.img[1][i] = g; a.img[2][i] = b;
    }
}

} // namespace

hipError_t launch_logenc_direct(const LogEncArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(logenc_direct_kernel, image_grid(a.w, a.h), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_logenc_prepare(const LogEncArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(logenc_prepare_kernel, image_grid(a.w, a.h), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_logenc_blend(const LogEncArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(logenc_blend_kernel, image_grid(a.w, a.h), dim3(256), 0, s, a);
    return hipGetLastError();
}

// find_gray (iplogenc.cc:38-91): host bisection with the C library's powf, as the reference runs it
float logenc_find_gray(float source_gray, float target_gray)
{
    if (source_gray <= 0.f) return 0.f;
    const auto f = [=](float x) -> float { return std::pow(x, source_gray) - 1 - target_gray * x + target_gray; };
    float lo = 1.f;
    while (f(lo) <= 0.f) lo *= 2.f;
    float hi = lo * 2.f;
    while (f(hi) >= 0.f) hi *= 2.f;
    if (std::isinf(hi)) return 0.f;
    for (int iter = 0; iter < 100; ++iter) {
        const float mid = lo + (hi - lo) / 2.f;
        const float v = f(mid);
        if (std::abs(v) < 1e-4f || (hi - lo) / lo <= 1e-4f) return mid;
        if (v > 0.f) lo = mid; else hi = mid;
    }
    return 0.f;
}

} // namespace artgpu
