// oracle/ref_driver.cc -- thin extern "C" shim over the parts of the REFERENCE that compile in
// this image from the sources where they lie (no stand-in headers): the sleef scalar and SSE
// math (rtengine/sleef.h, sleefsseavx.h, helpersse2.h), LUTf (rtengine/LUT.h, with -DNDEBUG),
// median.h, rt_math.h, rescale.h (+ array2D.h), iccmatrices.h.  Built by oracle/Makefile.ref into oracle/_ref/libartref.so (git-ignored)
// and used ONLY by tests to pin the oracle's restatement of these primitives.
// Everything that needs rtengine.h / rawimagesource.h / StopWatch.h (-> glibmm, lcms2) is
// unbuildable here: amaze_demosaic_RT.cc, rcd_demosaic.cc, demosaic_algos.cc, boxblur.h,
// gauss.cc, guidedfilter.cc, FTblockDN.cc, nlmeans.cc, ip*.cc.
#include <cstddef>
#include "sleef.h"
#include "LUT.h"
#include "median.h"
#include "rt_math.h"
#include "cplx_wavelet_dec.h"
#include "rescale.h"
#include "iccmatrices.h"
#include "linalgebra.h"
#include "halffloat.h"

extern "C" {

// ---- rtengine::wavelet_decomposition (cplx_wavelet_dec.h, subsampling = 1, Daub4 6 taps) ----
void *ref_wavelet_new(float *src, int w, int h, int maxlvl) { return new rtengine::wavelet_decomposition(src, w, h, maxlvl, 1, 1, 1); }
int ref_wavelet_maxlevel(void *p) { return static_cast<rtengine::wavelet_decomposition *>(p)->maxlevel(); }
int ref_wavelet_W(void *p, int l) { return static_cast<rtengine::wavelet_decomposition *>(p)->level_W(l); }
int ref_wavelet_H(void *p, int l) { return static_cast<rtengine::wavelet_decomposition *>(p)->level_H(l); }
int ref_wavelet_stride(void *p, int l) { return static_cast<rtengine::wavelet_decomposition *>(p)->level_stride(l); }
float *ref_wavelet_band(void *p, int l, int dir) { return static_cast<rtengine::wavelet_decomposition *>(p)->level_coeffs(l)[dir]; }
float *ref_wavelet_coeff0(void *p) { return static_cast<rtengine::wavelet_decomposition *>(p)->coeff0; }
void ref_wavelet_reconstruct(void *p, float *dst, float blend) { static_cast<rtengine::wavelet_decomposition *>(p)->reconstruct(dst, blend); }
void ref_wavelet_delete(void *p) { delete static_cast<rtengine::wavelet_decomposition *>(p); }

// ---- rescaleBilinear / getBilinearValue (rescale.h:27-74), used by guidedFilter and detail_mask ----
void ref_rescale_bilinear(const float *src, int Ws, int Hs, float *dst, int Wd, int Hd)
{
    rtengine::array2D<float> s(Ws, Hs), d(Wd, Hd);
    for (int y = 0; y < Hs; ++y) for (int x = 0; x < Ws; ++x) s[y][x] = src[(size_t)y * Ws + x];
    rtengine::rescaleBilinear(s, d, false);
    for (int y = 0; y < Hd; ++y) for (int x = 0; x < Wd; ++x) dst[(size_t)y * Wd + x] = d[y][x];
}
// ---- the Rec2020 working-space matrices (iccmatrices.h:151-161) ----
void ref_rec2020_matrices(float *xyz_rec2020_9, float *rec2020_xyz_9)
{
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { xyz_rec2020_9[3 * i + j] = rtengine::xyz_rec2020[i][j]; rec2020_xyz_9[3 * i + j] = rtengine::rec2020_xyz[i][j]; }
}

// linalgebra.h:141-239 (NeutralToneCurve builds its matrices with these and applies them per pixel, curves.cc:880-1038)
void ref_mat33_dot_vec3(const float *m9, const float *v3, float *r3)
{
    float a[3][3];
    for (int i = 0; i < 9; ++i) a[i / 3][i % 3] = m9[i];
    rtengine::Vec3<float> r = rtengine::dot_product(a, v3);
    for (int i = 0; i < 3; ++i) r3[i] = r[i];
}
void ref_mat33_dot_mat33(const float *a9, const float *b9, float *r9)
{
    float a[3][3], b[3][3];
    for (int i = 0; i < 9; ++i) { a[i / 3][i % 3] = a9[i]; b[i / 3][i % 3] = b9[i]; }
    rtengine::Mat33<float> r = rtengine::dot_product(a, b);
    for (int i = 0; i < 9; ++i) r9[i] = r[i / 3][i % 3];
}
int ref_mat33_inverse(const float *m9, float *r9)
{
    float a[3][3];
    for (int i = 0; i < 9; ++i) a[i / 3][i % 3] = m9[i];
    rtengine::Mat33<float> r;
    const bool ok = rtengine::inverse(a, r);
    for (int i = 0; i < 9; ++i) r9[i] = r[i / 3][i % 3];
    return ok ? 1 : 0;
}
void ref_xlog(const double *x, double *y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = xlog(x[i]); }
void ref_xexp(const double *x, double *y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = xexp(x[i]); }
void ref_xexpf(const float *x, float *y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = xexpf(x[i]); }
void ref_xcbrtf(const float *x, float *y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = xcbrtf(x[i]); }
void ref_xsincosf(const float *d, float *sn, float *cs, size_t n) { for (size_t i = 0; i < n; ++i) { float2 v = xsincosf(d[i]); sn[i] = v.x; cs[i] = v.y; } }
void ref_xlogf(const float *x, float *y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = xlogf(x[i]); }
void ref_xsinf(const float *x, float *y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = xsinf(x[i]); }
void ref_xcosf(const float *x, float *y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = xcosf(x[i]); }
void ref_xatan2f(const float *a, const float *b, float *y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = xatan2f(a[i], b[i]); }
void ref_xdiv2f(const float *x, float *y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = xdiv2f(x[i]); }
void ref_xdivf2(const float *x, float *y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = xdivf(x[i], 2); }
void ref_xlin2log(const float *x, float base, float *y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = xlin2log(x[i], base); }
void ref_xlog2lin(const float *x, float base, float *y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = xlog2lin(x[i], base); }
void ref_pow_F(const float *a, const float *b, float *y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = pow_F(a[i], b[i]); }

#ifdef __SSE2__
// 4-lane SSE variants (n must be a multiple of 4)
void ref_vexpf(const float *x, float *y, size_t n) { for (size_t i = 0; i < n; i += 4) _mm_storeu_ps(y + i, xexpf(_mm_loadu_ps(x + i))); }
void ref_vlogf(const float *x, float *y, size_t n) { for (size_t i = 0; i < n; i += 4) _mm_storeu_ps(y + i, xlogf(_mm_loadu_ps(x + i))); }
void ref_vexpf_nocheck(const float *x, float *y, size_t n) { for (size_t i = 0; i < n; i += 4) _mm_storeu_ps(y + i, xexpfNoCheck(_mm_loadu_ps(x + i))); }
void ref_vlogf_nocheck(const float *x, float *y, size_t n) { for (size_t i = 0; i < n; i += 4) _mm_storeu_ps(y + i, xlogfNoCheck(_mm_loadu_ps(x + i))); }
void ref_vatan2f(const float *a, const float *b, float *y, size_t n) { for (size_t i = 0; i < n; i += 4) _mm_storeu_ps(y + i, xatan2f(_mm_loadu_ps(a + i), _mm_loadu_ps(b + i))); }
void ref_vmedian3(const float *a, const float *b, const float *c, float *y, size_t n)
{
    for (size_t i = 0; i < n; i += 4) _mm_storeu_ps(y + i, median(_mm_loadu_ps(a + i), _mm_loadu_ps(b + i), _mm_loadu_ps(c + i)));
}
void ref_vintpf(const float *a, const float *b, const float *c, float *y, size_t n)
{
    for (size_t i = 0; i < n; i += 4) _mm_storeu_ps(y + i, vintpf(_mm_loadu_ps(a + i), _mm_loadu_ps(b + i), _mm_loadu_ps(c + i)));
}
void ref_vminmax(const float *a, const float *b, float *mn, float *mx, size_t n)
{
    for (size_t i = 0; i < n; i += 4) {
        _mm_storeu_ps(mn + i, vminf(_mm_loadu_ps(a + i), _mm_loadu_ps(b + i)));
        _mm_storeu_ps(mx + i, vmaxf(_mm_loadu_ps(a + i), _mm_loadu_ps(b + i)));
    }
}
// LUTf lookups: scalar operator[](float) and the SSE operator[](vfloat) (LUT.h:349-459)
void ref_lutf(const float *table, size_t tsize, const float *x, float *y_scalar, float *y_vec, size_t n)
{
    LUTf lut(tsize);
    for (size_t i = 0; i < tsize; ++i) lut[(int)i] = table[i];
    for (size_t i = 0; i < n; ++i) y_scalar[i] = lut[x[i]];
    for (size_t i = 0; i + 3 < n; i += 4) _mm_storeu_ps(y_vec + i, lut[_mm_loadu_ps(x + i)]);
}
#endif

// halffloat.h:9-46 (the half-float scanlines of Imagefloat::getScanline)
void ref_float_to_half(const float *x, unsigned short *y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = rtengine::DNG_FloatToHalf(x[i]); }
} // extern "C"
