/*
 * oracle/hsl.c -- CPU restatement of ImProcFunctions::hslEqualizer (rtengine/iphsl.cc:29-221), pipette buffer aside.
 * TEST INFRASTRUCTURE ONLY.  PARITY: xatan2f, xsincosf, pow_F/xlog2lin (sleef) pinned; FlatCurve and guidedFilter unpinned
 * (curves.h / guidedfilter.cc need glibmm).
 */
#include "oracle.h"
#include "oracle_common.h"
#include <stdlib.h>

void *oracle_flat_curve_new(const double *pts, int npts, int periodic, int ppn, double identity);
int oracle_flat_curve_is_identity(const void *h);
double oracle_flat_curve_get(const void *h, double t);
void oracle_flat_curve_free(void *h);

static float hue01(float h)
{
    const float pi2 = 2.f * (float)3.14159265358979323846;
    const float v = h / pi2;
    if (v < 0.f) return 1.f + v;
    if (v > 1.f) return v - 1.f;
    return v;
}
static float sgnf(float v) { return (float)((0.f < v) - (v < 0.f)); }
static float tolin(float y, float base)
{
    const float v = (y - 0.5f) * 2.f;
    return sgnf(v) * lim01f(oracle_xlog2lin(fabsf(v), base));
}

/* img: RGB planes (contiguous W x H) in, YUV mode out (g = Y, b = u, r = v) like the reference; to_rgb: setMode(RGB) on top */
void oracle_hsl_equalizer(float *const img[3], int W, int H, const double *hcurve, int nh, const double *scurve, int ns, const double *lcurve, int nl,
                          int smoothing, const double ws[9], double scale, int to_rgb)
{
    const size_t n = (size_t)W * H;
    const float ws1[3] = {(float)ws[3], (float)ws[4], (float)ws[5]};
    float *r = img[0], *g = img[1], *b = img[2];
    /* setMode(YUV) (imagefloat.cc:700-725), normalizeFloatTo1, yuv2hsl */
    const float f1 = 1.f / 65535.f;
#pragma omp parallel for
    for (size_t k = 0; k < n; ++k) {
        float Y = r[k] * ws1[0] + g[k] * ws1[1] + b[k] * ws1[2];
        float u = Y - b[k], v = r[k] - Y;
        Y *= f1; u *= f1; v *= f1;
        g[k] = Y;
        b[k] = sqrtf(u * u + v * v);
        r[k] = oracle_xatan2f(u, v);
    }
    const int ppn = (int)(1000 / scale);
    void *hc = oracle_flat_curve_new(hcurve, nh, 1, ppn, 0.5), *sc = oracle_flat_curve_new(scurve, ns, 1, ppn, 0.5), *lc = oracle_flat_curve_new(lcurve, nl, 1, ppn, 0.5);
    static const double coeff_pts[9] = {1, 0.25, 0.0, 0.5, 0.18, 1, 1, 0, 0.35};
    void *coeff = oracle_flat_curve_new(coeff_pts, 9, 1, 1000, 0.5);
    const float sm = smoothing / 10.f;
    const float smooth = powf(10.f, lim01f(sm)) - 1.f;
    float *mask = (float *)malloc(sizeof(float) * n);
    if (!oracle_flat_curve_is_identity(sc)) {
#pragma omp parallel for
        for (size_t k = 0; k < n; ++k) mask[k] = (float)oracle_flat_curve_get(sc, hue01(r[k]));
        const int radius = (int)(4 / scale * smooth + 0.5);
        if (radius > 0) oracle_guided_filter(g, mask, mask, W, H, radius, 0.001f, 0);
#pragma omp parallel for
        for (size_t k = 0; k < n; ++k) {
            const float f = tolin(mask[k], 2.f);
            const double cv = oracle_flat_curve_get(coeff, b[k]);
            const float s = 1.f + (f < 0 ? cv : 1.f - cv);
            b[k] *= 1.f + sgnf(f) * oracle_pow_F(lim01f(fabsf(f)), s);
        }
    }
    if (!oracle_flat_curve_is_identity(lc)) {
#pragma omp parallel for
        for (size_t k = 0; k < n; ++k) mask[k] = (float)oracle_flat_curve_get(lc, hue01(r[k]));
        const int radius = (int)(25 / scale * smooth + 0.5);
        if (radius > 0) oracle_guided_filter(g, mask, mask, W, H, radius, 0.0001f, 0);
#pragma omp parallel for
        for (size_t k = 0; k < n; ++k) g[k] *= 1.f + tolin(mask[k], 10.f);
    }
    if (!oracle_flat_curve_is_identity(hc)) {
#pragma omp parallel for
        for (size_t k = 0; k < n; ++k) mask[k] = (float)oracle_flat_curve_get(hc, hue01(r[k]));
        const int radius = (int)(4 / scale * smooth + 0.5);
        if (radius > 0) oracle_guided_filter(g, mask, mask, W, H, radius, 0.001f, 0);
#pragma omp parallel for
        for (size_t k = 0; k < n; ++k) r[k] += tolin(mask[k], 32.f) * (float)3.14159265358979323846;
    }
    free(mask);
    oracle_flat_curve_free(hc); oracle_flat_curve_free(sc); oracle_flat_curve_free(lc); oracle_flat_curve_free(coeff);
    /* hsl2yuv, normalizeFloatTo65535 */
#pragma omp parallel for
    for (size_t k = 0; k < n; ++k) {
        float sn, cs;
        oracle_xsincosf(r[k], &sn, &cs);
        float u = b[k] * sn, v = b[k] * cs, Y = g[k];
        Y *= 65535.f; u *= 65535.f; v *= 65535.f;
        if (to_rgb) {
            const float B = Y - u, R = v + Y;
            const float G = (Y - R * ws1[0] - B * ws1[2]) / ws1[1];
            r[k] = R; g[k] = G; b[k] = B;
        } else {
            g[k] = Y; b[k] = u; r[k] = v;
        }
    }
}
