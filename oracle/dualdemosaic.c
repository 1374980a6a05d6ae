/*
 * oracle/dualdemosaic.c -- CPU restatement of the blend half of RawImageSource::dual_demosaic_RT (rtengine/dual_demosaic_RT.cc:73-152,
 * Bayer, second demosaicer = bilinear): Color::RGB2L (color.cc:1343-1379), buildBlendMask with its automatic contrast threshold
 * (rt_algo.cc:40-176,315-498) and bayer_bilinear_demosaic's blend form (bayer_bilinear_demosaic.cc:33-77).  The first demosaicer
 * (AMaZE / RCD) is oracle_amaze_demosaic / oracle_rcd.  TEST INFRASTRUCTURE ONLY.  PARITY: xexpf, the gaussian blur and the LUT forms
 * pinned elsewhere (tests/golden); these functions unpinned (rt_algo.cc needs fftw3/glibmm headers).  The reference blurs the mask
 * with flush-to-zero on; the mask is >= 1e-7, no denormals arise, and the oracle does not switch modes.
 */
#include "oracle.h"
#include "oracle_common.h"
#include <stdio.h>
#include <stdlib.h>

#define MAXVALF 65535.f
#define KAPPA (24389.0 / 27.0)
static float *g_cfy;
static float xyz2laby(float y)      /* Color::computeXYZ2LabY (color.cc:1262-1275) */
{
    if (y != y) return y;
    if (y < 0.f) return (float)(327.68 * (KAPPA * y / MAXVALF));
    if (y > 65535.f) return 327.68f * (116.f * oracle_xcbrtf(y / MAXVALF) - 16.f);
    int idx = (int)y;
    if (y > 65534.f) idx = 65534;
    const float diff = y - (float)idx, p1 = g_cfy[idx], p2 = g_cfy[idx + 1] - p1;
    return p1 + p2 * diff;
}
/* Color::RGB2L over an image of contiguous rows with wp = sRGB's XYZ matrix (dual_demosaic_RT.cc:99-112) */
void oracle_rgb2l(const float *R, const float *G, const float *B, float *L, int W, int H)
{
#pragma omp critical(dualdemosaic_luts)
    if (!g_cfy) { float *t = (float *)malloc(sizeof(float) * 65536); oracle_cachefy(t); g_cfy = t; }
    const float w0 = 0.212671f, w1 = 0.715160f, w2 = 0.072169f;
#pragma omp parallel for
    for (int y = 0; y < H; ++y) {
        const float *r = R + (size_t)y * W, *g = G + (size_t)y * W, *b = B + (size_t)y * W;
        float *l = L + (size_t)y * W;
        int i = 0;
        for (; i < W - 3; i += 4) {
            float yv[4];
            int slow = 0;
            for (int k = 0; k < 4; ++k) {
                yv[k] = w0 * r[i + k] + w1 * g[i + k] + w2 * b[i + k];
                if (yv[k] > MAXVALF || yv[k] < 0.f) slow = 1;
            }
            for (int k = 0; k < 4; ++k) l[i + k] = slow ? xyz2laby(yv[k]) : oracle_lutf_vec(g_cfy, 65536, yv[k]);
        }
        for (; i < W; ++i) l[i] = xyz2laby(w0 * r[i] + w1 * g[i] + w2 * b[i]);
    }
}

static float vhadd(const float v[4]) { return (v[0] + v[2]) + (v[1] + v[3]); }
static float blend_factor_v(float val, float thr) { return 1.f / (1.f + oracle_xexpf_v(16.f - 16.f * val / thr)); }
static float blend_factor_s(float val, float thr) { return 1.f / (1.f + oracle_xexpf_s(16.f - 16.f * val / thr)); }
static float contrast_at(const float *L, int W, int j, int i, float scale)
{
    const float *p = L + (size_t)j * W + i;
    return sqrtf(sqrf(p[1] - p[-1]) + sqrf(p[W] - p[-W]) + sqrf(p[2] - p[-2]) + sqrf(p[2 * W] - p[-2 * W])) * scale;
}
/* tileAverage / tileVariance (rt_algo.cc:58-110) */
static float tile_average(const float *L, int W, int ty, int tx, int ts)
{
    float avg = 0.f, v[4] = {0, 0, 0, 0};
    for (int y = ty; y < ty + ts; ++y) {
        int x = tx;
        for (; x < tx + ts - 3; x += 4) for (int k = 0; k < 4; ++k) v[k] += L[(size_t)y * W + x + k];
        for (; x < tx + ts; ++x) avg += L[(size_t)y * W + x];
    }
    avg += vhadd(v);
    return avg / (float)(ts * ts);
}
static float tile_variance(const float *L, int W, int ty, int tx, int ts, float avg)
{
    float var = 0.f, v[4] = {0, 0, 0, 0};
    for (int y = ty; y < ty + ts; ++y) {
        int x = tx;
        for (; x < tx + ts - 3; x += 4) for (int k = 0; k < 4; ++k) v[k] += sqrf(L[(size_t)y * W + x + k] - avg);
        for (; x < tx + ts; ++x) var += sqrf(L[(size_t)y * W + x] - avg);
    }
    var += vhadd(v);
    return var / ((float)(ts * ts) * avg);
}
/* calcContrastThreshold (rt_algo.cc:112-176) */
float oracle_calc_contrast_threshold(const float *L, int W, int tileY, int tileX, int ts, float factor)
{
    const float scale = 0.0625f / 327.68f * factor;
    const int n = ts - 4;
    float *bl = (float *)malloc(sizeof(float) * n * n);
    for (int j = tileY + 2; j < tileY + ts - 2; ++j)
        for (int i = tileX + 2; i < tileX + ts - 2; ++i) bl[(j - tileY - 2) * n + i - tileX - 2] = contrast_at(L, W, j, i, scale);
    const float limit = (float)(n * n) / 100.f;
    int c;
    for (c = 1; c < 100; ++c) {
        const float thr = c / 100.f;
        float sum = 0.f, sv[4] = {0, 0, 0, 0};
        for (int j = 0; j < n; ++j) {
            int i = 0;
            for (; i < ts - 7; i += 4) for (int k = 0; k < 4; ++k) sv[k] += blend_factor_v(bl[j * n + i + k], thr);
            for (; i < n; ++i) sum += blend_factor_s(bl[j * n + i], thr);
        }
        sum += vhadd(sv);
        if (sum <= limit) break;
    }
    free(bl);
    return c / 100.f;
}

static void variances_of(const float *L, int W, int nH, int nW, int y0, int x0, int step, int ts, float minLum, float maxLum, float *var)
{
#pragma omp parallel for schedule(dynamic)
    for (int i = 0; i < nH; ++i)
        for (int j = 0; j < nW; ++j) {
            const int ty = y0 + i * step, tx = x0 + j * step;
            const float avg = tile_average(L, W, ty, tx, ts);
            float v = INFINITY;
            if (!(avg < minLum || avg > maxLum)) {
                v = tile_variance(L, W, ty, tx, ts, avg);
                if (v < 0.5f) v = INFINITY;
            }
            var[(size_t)i * nW + j] = v;
        }
}
static float argmin_first(const float *var, int nH, int nW, int *mi, int *mj)
{
    float minvar = INFINITY;
    *mi = *mj = 0;
    for (int i = 0; i < nH; ++i)
        for (int j = 0; j < nW; ++j)
            if (var[(size_t)i * nW + j] < minvar) { minvar = var[(size_t)i * nW + j]; *mi = i; *mj = j; }
    return minvar;
}

/* buildBlendMask (rt_algo.cc:315-498), amount = 1, blur_radius = 2, luminance_factor = 1.  Returns the contrast threshold in use. */
float oracle_build_blend_mask(const float *L, float *blend, int W, int H, float contrastThreshold, int autoContrast)
{
    if (autoContrast) {
        for (int pass = 0; pass < 2; ++pass) {
            const int ts = 80 / (pass + 1);
            const int skip = pass == 0 ? ts : ts / 4;
            const int nW = W / skip - 3 * pass, nH = H / skip - 3 * pass;
            float *var = (float *)malloc(sizeof(float) * (size_t)(nW > 0 ? nW : 1) * (nH > 0 ? nH : 1));
            variances_of(L, W, nH > 0 ? nH : 0, nW > 0 ? nW : 0, 0, 0, skip, ts, 2000.f, 20000.f, var);
            int mi, mj;
            const float minvar = argmin_first(var, nH > 0 ? nH : 0, nW > 0 ? nW : 0, &mi, &mj);
            free(var);
            if (getenv("ORACLE_DUAL_DEBUG")) fprintf(stderr, "buildBlendMask pass %d: minvar %g at tile (%d, %d)\n", pass, (double)minvar, mi, mj);
            if (minvar <= 1.f || pass == 1) {
                const int minY = skip * mi, minX = skip * mj;
                if (pass == 0) {
                    contrastThreshold = oracle_calc_contrast_threshold(L, W, minY, minX, ts, 1.f);
                    break;
                }
                const int y0 = minY - skip > 0 ? minY - skip : 0, x0 = minX - skip > 0 ? minX - skip : 0;
                const int y1 = minY + skip < H - ts ? minY + skip : H - ts, x1 = minX + skip < W - ts ? minX + skip : W - ts;
                const int nH2 = y1 - y0 + 1, nW2 = x1 - x0 + 1;
                float *var2 = (float *)malloc(sizeof(float) * (size_t)nH2 * nW2);
                variances_of(L, W, nH2, nW2, y0, x0, 1, ts, 2000.f, 20000.f, var2);
                int mi2, mj2;
                const float minvar2 = argmin_first(var2, nH2, nW2, &mi2, &mj2);
                free(var2);
                contrastThreshold = minvar2 <= 8.f ? oracle_calc_contrast_threshold(L, W, y0 + mi2, x0 + mj2, ts, 1.f) : 0.f;
            }
        }
    }
    const size_t n = (size_t)W * H;
    if (contrastThreshold == 0.f) {
        for (size_t k = 0; k < n; ++k) blend[k] = 1.f;
        return contrastThreshold;
    }
    const float scale = 0.0625f / 327.68f * 1.f;
#pragma omp parallel for
    for (int j = 2; j < H - 2; ++j) {
        int i = 2;
        for (; i < W - 5; i += 4)
            for (int k = 0; k < 4; ++k) blend[(size_t)j * W + i + k] = 1.f * blend_factor_v(contrast_at(L, W, j, i + k, scale), contrastThreshold);
        for (; i < W - 2; ++i) blend[(size_t)j * W + i] = 1.f * blend_factor_s(contrast_at(L, W, j, i, scale), contrastThreshold);
    }
    for (int j = 0; j < 2; ++j) for (int i = 2; i < W - 2; ++i) blend[(size_t)j * W + i] = blend[(size_t)2 * W + i];
    for (int j = H - 2; j < H; ++j) for (int i = 2; i < W - 2; ++i) blend[(size_t)j * W + i] = blend[(size_t)(H - 3) * W + i];
    for (int j = 0; j < H; ++j) {
        float *b = blend + (size_t)j * W;
        b[0] = b[1] = b[2];
        b[W - 2] = b[W - 1] = b[W - 3];
    }
    oracle_gaussian_blur(blend, W, H, 2.0);
    return contrastThreshold;
}

/* bayer_bilinear_demosaic(blend, ...) (bayer_bilinear_demosaic.cc:44-62): rows 1..H-2, pairs starting at a green site */
void oracle_bayer_bilinear_blend(const float *blend, const float *raw, float *red, float *green, float *blue, int W, int H, unsigned filters)
{
#pragma omp parallel for
    for (int i = 1; i < H - 1; ++i) {
        float *ng1 = red, *ng2 = blue;
        if (fc(filters, i, 0) == 2 || fc(filters, i, 1) == 2) { ng1 = blue; ng2 = red; }
        const size_t o = (size_t)i * W;
        for (int j = 2 - (fc(filters, i, 1) & 1); j < W - 2; j += 2) {
            const float *r = raw + o + j;
            const float b0 = blend[o + j], b1 = blend[o + j + 1];
            green[o + j] = intpf(b0, green[o + j], r[0]);
            ng1[o + j] = intpf(b0, ng1[o + j], (r[-1] + r[1]) * 0.5f);
            ng2[o + j] = intpf(b0, ng2[o + j], (r[-W] + r[W]) * 0.5f);
            green[o + j + 1] = intpf(b1, green[o + j + 1], ((r[-W + 1] + r[0]) + (r[2] + r[W + 1])) * 0.25f);
            ng1[o + j + 1] = intpf(b1, ng1[o + j + 1], r[1]);
            ng2[o + j + 1] = intpf(b1, ng2[o + j + 1], ((r[-W] + r[-W + 2]) + (r[W] + r[W + 2])) * 0.25f);
        }
    }
}

/* the blend half of dual_demosaic_RT on already demosaiced planes; *contrast in percent in/out like the reference's `double &contrast` */
void oracle_dual_demosaic_blend2(const float *raw, float *red, float *green, float *blue, int W, int H, unsigned filters, double *contrast, int autoContrast, int vng4);
void oracle_dual_demosaic_blend(const float *raw, float *red, float *green, float *blue, int W, int H, unsigned filters, double *contrast, int autoContrast)
{
    oracle_dual_demosaic_blend2(raw, red, green, blue, W, H, filters, contrast, autoContrast, 0);
}
/* vng4 != 0: the second demosaicer is vng4_demosaic, blended over all three channels of every pixel (dual_demosaic_RT.cc:128-148) */
void oracle_dual_demosaic_blend2(const float *raw, float *red, float *green, float *blue, int W, int H, unsigned filters, double *contrast, int autoContrast, int vng4)
{
    if (*contrast == 0.0 && !autoContrast) return;
    const size_t n = (size_t)W * H;
    float *L = (float *)malloc(sizeof(float) * n), *blend = (float *)malloc(sizeof(float) * n);
    oracle_rgb2l(red, green, blue, L, W, H);
    float cf = (float)(*contrast / 100.0);
    cf = oracle_build_blend_mask(L, blend, W, H, cf, autoContrast);
    *contrast = cf * 100.f;
    if (vng4) {
        float *t = (float *)malloc(sizeof(float) * 3 * n);
        oracle_vng4_demosaic(raw, W, H, filters, 0, t, t + n, t + 2 * n);
#pragma omp parallel for
        for (size_t k = 0; k < n; ++k) {
            red[k] = intpf(blend[k], red[k], t[k]);
            green[k] = intpf(blend[k], green[k], t[n + k]);
            blue[k] = intpf(blend[k], blue[k], t[2 * n + k]);
        }
        free(t);
    } else
        oracle_bayer_bilinear_blend(blend, raw, red, green, blue, W, H, filters);
    free(L); free(blend);
}
