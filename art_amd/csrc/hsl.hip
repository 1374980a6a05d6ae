// art_amd/csrc/hsl.hip -- ImProcFunctions::hslEqualizer (rtengine/iphsl.cc:29-221) on gfx950.
//
// The tool works in YUV normalised to 1 with (v, u) turned into (hue, saturation) (Color::yuv2hsl / hsl2yuv, color.cc:6691-6703);
// each of its three FlatCurves (saturation, luminance, hue over hue) becomes a per-pixel mask = curve(hue), optionally smoothed by
// rtengine::guidedFilter with the luminance as guide (artgpu_guided_filter's kernels), then applied.  FlatCurve::getVal is the
// reference's: a binary search in the curve's polyline, evaluated in double (flatcurves.cc:339-365); the polylines are built on
// the host exactly as FlatCurve's constructor does (hostluts.hip) and uploaded.
#include <hip/hip_runtime.h>
#include "devmath.h"
#include "devsleef.h"
#include "kernels.h"

namespace artgpu {

namespace {
__device__ __forceinline__ float sgnf(float v) { return (float)((0.f < v) - (v < 0.f)); }

// FlatCurve::getVal, FCT_MinMaxCPoints (flatcurves.cc:344-365)
__device__ __forceinline__ double flat_curve_val(const HslCurve &c, double t)
{
    if (t < c.x[0]) t += 1.0;
    unsigned lo = 0, hi = (unsigned)c.n - 1;
    while (hi > 1 + lo) {
        const unsigned mid = (hi + lo) / 2;
        if (c.x[mid] > t) hi = mid; else lo = mid;
    }
    return c.y[lo] + (t - c.x[lo]) * c.slope[lo];
}
__device__ __forceinline__ float hue01(float h)
{
    const float pi2 = 2.f * 3.14159265358979323846f;
    const float v = h / pi2;
    if (v < 0.f) return 1.f + v;
    if (v > 1.f) return v - 1.f;
    return v;
}
__device__ __forceinline__ float tolin(float y, float base)
{
    const float v = (y - 0.5f) * 2.f;
    return sgnf(v) * lim01(xlog2lin(fabsf(v), base));
}
} // namespace

// setMode(YUV) (imagefloat.cc:700-725) + normalizeFloatTo1 (L429-432) + yuv2hsl: r <- hue, g <- Y, b <- saturation
__global__ void __launch_bounds__(256) hsl_prepare_kernel(HslArgs a)
{
    const float f1 = 1.f / 65535.f;
    FOR_IMAGE_XY(y, x, a.w, a.h) {
        const size_t di = (size_t)y * a.stride + x;
        const float r = a.img[0][di], g = a.img[1][di], b = a.img[2][di];
        float Y = r * a.ws1[0] + g * a.ws1[1] + b * a.ws1[2];
        float u = Y - b, v = r - Y;
        Y *= f1; u *= f1; v *= f1;
        a.img[1][di] = Y;
        a.img[2][di] = sqrtf(sqr(u) + sqr(v));
        a.img[0][di] = xatan2f_s(u, v);
    }
}
// mask = curve(hue01(hue)) (iphsl.cc:108-116,150-158,177-185)
__global__ void __launch_bounds__(256) hsl_mask_kernel(HslArgs a)
{
    const HslCurve c = a.curve[a.which];
    FOR_IMAGE_XY(y, x, a.w, a.h) {
        const float h = a.img[0][(size_t)y * a.stride + x];
        a.mask[(size_t)y * a.w + x] = (float)flat_curve_val(c, (double)hue01(h));
    }
}
// the three applications (L133-146, 169-174, 196-204); which: 0 saturation, 1 luminance, 2 hue
__global__ void __launch_bounds__(256) hsl_apply_kernel(HslArgs a)
{
    const HslCurve coeff = a.curve[3];
    FOR_IMAGE_XY(y, x, a.w, a.h) {
        const size_t di = (size_t)y * a.stride + x;
        const float mk = a.mask[(size_t)y * a.w + x];
        if (a.which == 0) {
            const float f = tolin(mk, 2.f);
            const float sv = a.img[2][di];
            const double cv = flat_curve_val(coeff, (double)sv);
            const float s = (float)(1.f + (f < 0 ? cv : 1.f - cv));
            a.img[2][di] = sv * (1.f + sgnf(f) * pow_F(lim01(fabsf(f)), s));
        } else if (a.which == 1) {
            const float f = 1.f + tolin(mk, 10.f);
            a.img[1][di] *= f;
        } else {
            const float f = tolin(mk, 32.f) * 3.14159265358979323846f;
            a.img[0][di] += f;
        }
    }
}
// hsl2yuv + normalizeFloatTo65535 (L207-220): b <- u, r <- v; the image is left in YUV mode like the reference's, unless to_rgb
// asks for Imagefloat::setMode(RGB) (imagefloat.cc:779-804) on top
__global__ void __launch_bounds__(256) hsl_finish_kernel(HslArgs a)
{
    FOR_IMAGE_XY(y, x, a.w, a.h) {
        const size_t di = (size_t)y * a.stride + x;
        const float h = a.img[0][di], s = a.img[2][di];
        float sn, cs;
        xsincosf_v(h, sn, cs);
        float u = s * sn, v = s * cs, Y = a.img[1][di];
        Y *= 65535.f; u *= 65535.f; v *= 65535.f;
        if (a.to_rgb) {
            const float b = Y - u, r = v + Y;
            const float g = (Y - r * a.ws1[0] - b * a.ws1[2]) / a.ws1[1];
            a.img[0][di] = r; a.img[1][di] = g; a.img[2][di] = b;
        } else {
            a.img[1][di] = Y; a.img[2][di] = u; a.img[0][di] = v;
        }
    }
}

hipError_t launch_hsl_prepare(const HslArgs &a, hipStream_t s) { hipLaunchKernelGGL(hsl_prepare_kernel, image_grid(a.w, a.h), dim3(256), 0, s, a); return hipGetLastError(); }
hipError_t launch_hsl_mask(const HslArgs &a, hipStream_t s) { hipLaunchKernelGGL(hsl_mask_kernel, image_grid(a.w, a.h), dim3(256), 0, s, a); return hipGetLastError(); }
hipError_t launch_hsl_apply(const HslArgs &a, hipStream_t s) { hipLaunchKernelGGL(hsl_apply_kernel, image_grid(a.w, a.h), dim3(256), 0, s, a); return hipGetLastError(); }
hipError_t launch_hsl_finish(const HslArgs &a, hipStream_t s) { hipLaunchKernelGGL(hsl_finish_kernel, image_grid(a.w, a.h), dim3(256), 0, s, a); return hipGetLastError(); }

} // namespace artgpu
