/*
 * oracle/nlmeans.c -- CPU restatement of the NL-means stage of the denoise tool and its helpers:
 *   oracle_yvv_factors      calculateYvVFactors<double> + the M rescaling      rtengine/gauss.cc:94-126,556-562
 *   oracle_gaussian_blur    gaussianBlur, 0.6 <= sigma < 25, GAUSS_STANDARD, x86-64 path:
 *                           gaussHorizontalSse (gauss.cc:554-665: rows in groups of 4 with float
 *                           coefficients, the H%4 tail rows with double coefficients) then
 *                           gaussVerticalSse (L716-856: columns in groups of 8 / W%8 tail), in place
 *   oracle_detail_mask      detail_mask + laplacian, BlurType::GAUSS            rtengine/FTblockDN.cc:1366-1476
 *   oracle_lutf_vec         LUTf::operator[](vfloat)                            rtengine/LUT.h:349-377
 *   oracle_nlmeans          denoise::NLMeans                                    rtengine/nlmeans.cc:50-280
 *
 * TEST INFRASTRUCTURE ONLY.  PARITY: sleef math, scalar and vector LUT lookups and bilinear rescale
 * are pinned against the reference headers (tests/golden); the rest is UNPINNED (gauss.cc includes
 * boxblur.h -> StopWatch.h -> glibmm; nlmeans.cc and FTblockDN.cc need improcfun.h / fftw3.h).
 *
 * NL-means quirks kept: the padded source maps rows/cols >= H / >= W to the last pixel
 * (nlmeans.cc:102-108), tiles of 150 with stride 150-2*border, per-tile fp32 integral image with
 * the association (a+b)-(c-s) (L192-204), 4-lane bulk / scalar tail split (L213,230), and the
 * tile loop runs with MXCSR flush-to-zero on (L157-160): ftz() is applied to every result there.
 */
#include "oracle.h"
#include "oracle_common.h"
#include <stdlib.h>
#include <float.h>

static void oracle_yvv_factors_raw(double sigma, double *b1, double *b2, double *b3, double *B, double M[9])
{
    double q;
    if (sigma < 2.5) q = 3.97156 - 4.14554 * sqrt(1.0 - 0.26891 * sigma);
    else q = 0.98711 * sigma - 0.96330;
    double b0 = 1.57825 + 2.44413 * q + 1.4281 * q * q + 0.422205 * q * q * q;
    *b1 = 2.44413 * q + 2.85619 * q * q + 1.26661 * q * q * q;
    *b2 = -1.4281 * q * q - 1.26661 * q * q * q;
    *b3 = 0.422205 * q * q * q;
    *B = 1.0 - (*b1 + *b2 + *b3) / b0;
    *b1 /= b0; *b2 /= b0; *b3 /= b0;
    const double c1 = *b1, c2 = *b2, c3 = *b3;
    M[0] = -c3 * c1 + 1.0 - c3 * c3 - c2;
    M[1] = (c3 + c1) * (c2 + c3 * c1);
    M[2] = c3 * (c1 + c3 * c2);
    M[3] = c1 + c3 * c2;
    M[4] = -(c2 - 1.0) * (c2 + c3 * c1);
    M[5] = -(c3 * c1 + c3 * c3 + c2 - 1.0) * c3;
    M[6] = c3 * c1 + c2 + c1 * c1 - c2 * c2;
    M[7] = c1 * c2 + c3 * c2 * c2 - c1 * c3 * c3 - c3 * c3 * c3 - c3 * c2 + c3;
    M[8] = c3 * (c1 + c3 * c2);
}

/* calculateYvVFactors + the normalisation of the Sse functions (gauss.cc:559-563) */
void oracle_yvv_factors(double sigma, double *b1, double *b2, double *b3, double *B, double M[9])
{
    oracle_yvv_factors_raw(sigma, b1, b2, b3, B, M);
    const double c1 = *b1, c2 = *b2, c3 = *b3;
    for (int i = 0; i < 9; ++i) {
        M[i] *= (1.0 + c2 + (c1 - c3) * c3);
        M[i] /= (1.0 + c1 - c2 + c3) * (1.0 - c1 - c2 - c3);
    }
}

/* one line (stride `st`, length n), float-coefficient form (the SSE lanes) */
static void yvv_line_f(float *p, size_t st, int n, float *tmp, float B, float b1, float b2, float b3, const float M[9])
{
    const float s0 = p[0];
    tmp[0] = s0 * (B + b1 + b2 + b3);
    tmp[1] = p[st] * B + tmp[0] * b1 + s0 * (b2 + b3);
    tmp[2] = p[2 * st] * B + tmp[1] * b1 + tmp[0] * b2 + s0 * b3;
    for (int j = 3; j < n; j++) tmp[j] = p[(size_t)j * st] * B + tmp[j - 1] * b1 + tmp[j - 2] * b2 + tmp[j - 3] * b3;
    const float Tv = p[(size_t)(n - 1) * st];
    float Rv = tmp[n - 1], Tm2 = tmp[n - 2], Tm3 = tmp[n - 3];
    const float t2Wp1 = Tv + M[6] * (Rv - Tv) + M[7] * (Tm2 - Tv) + M[8] * (Tm3 - Tv);
    const float t2W = Tv + M[3] * (Rv - Tv) + M[4] * (Tm2 - Tv) + M[5] * (Tm3 - Tv);
    Rv = Tv + M[0] * (Rv - Tv) + M[1] * (Tm2 - Tv) + M[2] * (Tm3 - Tv);
    tmp[n - 1] = Rv;
    Tm2 = B * Tm2 + b1 * Rv + b2 * t2W + b3 * t2Wp1;
    tmp[n - 2] = Tm2;
    Tm3 = B * Tm3 + b1 * Tm2 + b2 * Rv + b3 * t2W;
    tmp[n - 3] = Tm3;
    float T = Rv;
    Rv = Tm3;
    Tm3 = T;
    for (int j = n - 4; j >= 0; j--) {
        T = Rv;
        Rv = tmp[j] * B + T * b1 + Tm2 * b2 + Tm3 * b3;
        tmp[j] = Rv;
        Tm3 = Tm2;
        Tm2 = T;
    }
    for (int j = 0; j < n; ++j) p[(size_t)j * st] = tmp[j];
}

/* double-coefficient form (the "borders are done without SSE" lines); tmp is float storage */
static void yvv_line_d(float *p, size_t st, int n, float *tmp, double B, double b1, double b2, double b3, const double M[9])
{
    const float s0 = p[0], sl = p[(size_t)(n - 1) * st];
    tmp[0] = s0 * (B + b1 + b2 + b3);
    tmp[1] = B * p[st] + b1 * tmp[0] + s0 * (b2 + b3);
    tmp[2] = B * p[2 * st] + b1 * tmp[1] + b2 * tmp[0] + b3 * s0;
    for (int j = 3; j < n; j++) tmp[j] = B * p[(size_t)j * st] + b1 * tmp[j - 1] + b2 * tmp[j - 2] + b3 * tmp[j - 3];
    float t2Wm1 = sl + M[0] * (tmp[n - 1] - sl) + M[1] * (tmp[n - 2] - sl) + M[2] * (tmp[n - 3] - sl);
    float t2W = sl + M[3] * (tmp[n - 1] - sl) + M[4] * (tmp[n - 2] - sl) + M[5] * (tmp[n - 3] - sl);
    float t2Wp1 = sl + M[6] * (tmp[n - 1] - sl) + M[7] * (tmp[n - 2] - sl) + M[8] * (tmp[n - 3] - sl);
    tmp[n - 1] = t2Wm1;
    tmp[n - 2] = B * tmp[n - 2] + b1 * tmp[n - 1] + b2 * t2W + b3 * t2Wp1;
    tmp[n - 3] = B * tmp[n - 3] + b1 * tmp[n - 2] + b2 * tmp[n - 1] + b3 * t2W;
    for (int j = n - 4; j >= 0; j--) tmp[j] = B * tmp[j] + b1 * tmp[j + 1] + b2 * tmp[j + 2] + b3 * tmp[j + 3];
    for (int j = 0; j < n; ++j) p[(size_t)j * st] = tmp[j];
}

/* sigma >= 25: gaussHorizontal<T> / gaussVertical<T> (gauss.cc:669-713,1148-1225), all double, M normalised differently */
static void yvv_line_64(float *p, size_t st, int n, double *tmp, double B, double b1, double b2, double b3, const double M[9])
{
    const double s0 = p[0];
    tmp[0] = B * s0 + b1 * s0 + b2 * s0 + b3 * s0;
    tmp[1] = B * p[st] + b1 * tmp[0] + b2 * s0 + b3 * s0;
    tmp[2] = B * p[2 * st] + b1 * tmp[1] + b2 * tmp[0] + b3 * s0;
    for (int j = 3; j < n; j++) tmp[j] = B * p[(size_t)j * st] + b1 * tmp[j - 1] + b2 * tmp[j - 2] + b3 * tmp[j - 3];
    const double sl = p[(size_t)(n - 1) * st];
    const double t2Wm1 = sl + M[0] * (tmp[n - 1] - sl) + M[1] * (tmp[n - 2] - sl) + M[2] * (tmp[n - 3] - sl);
    const double t2W = sl + M[3] * (tmp[n - 1] - sl) + M[4] * (tmp[n - 2] - sl) + M[5] * (tmp[n - 3] - sl);
    const double t2Wp1 = sl + M[6] * (tmp[n - 1] - sl) + M[7] * (tmp[n - 2] - sl) + M[8] * (tmp[n - 3] - sl);
    tmp[n - 1] = t2Wm1;
    tmp[n - 2] = B * tmp[n - 2] + b1 * tmp[n - 1] + b2 * t2W + b3 * t2Wp1;
    tmp[n - 3] = B * tmp[n - 3] + b1 * tmp[n - 2] + b2 * tmp[n - 1] + b3 * t2W;
    for (int j = n - 4; j >= 0; j--) tmp[j] = B * tmp[j] + b1 * tmp[j + 1] + b2 * tmp[j + 2] + b3 * tmp[j + 3];
    for (int j = 0; j < n; ++j) p[(size_t)j * st] = (float)tmp[j];
}

void oracle_gaussian_blur(float *img, int W, int H, double sigma_d)
{
    if (sigma_d < 0.25) return;             /* GAUSS_SKIP (gauss.cc:1389,1437-1443): src == dst, nothing to do */
    if (sigma_d < 0.6) {                    /* GAUSS_3X3_LIMIT, src == dst: gaussHorizontal3 + gaussVertical3 (gauss.cc:1474-1483,446-526) */
        double c1d = exp(-1.0 / (2.0 * sigma_d * sigma_d));
        const double csum = 2.0 * c1d + 1.0;
        c1d /= csum;
        const float c0 = (float)(1.0 / csum), c1 = (float)c1d;
        float *temp = (float *)malloc(sizeof(float) * (size_t)(W > H ? W : H));
        for (int i = 0; i < H; ++i) {
            float *r = img + (size_t)i * W;
            for (int j = 1; j < W - 1; ++j) temp[j] = c1 * (r[j - 1] + r[j + 1]) + c0 * r[j];
            for (int j = 1; j < W - 1; ++j) r[j] = temp[j];
        }
        const int wv = W - (W % 8);
        for (int i = 0; i < W; ++i) {
            for (int j = 1; j < H - 1; ++j) {
                const float up = img[(size_t)(j - 1) * W + i], dn = img[(size_t)(j + 1) * W + i], c = img[(size_t)j * W + i];
                temp[j] = i < wv ? c1 * (dn + up) + c * c0 : c1 * (up + dn) + c0 * c;   /* vector loop (L479-505) / scalar tail (L508-525) */
            }
            for (int j = 1; j < H - 1; ++j) img[(size_t)j * W + i] = temp[j];
        }
        free(temp);
        return;
    }
    if (sigma_d >= 25.0) {
        double b1, b2, b3, B, M[9];
        oracle_yvv_factors_raw(sigma_d, &b1, &b2, &b3, &B, M);
        for (int i = 0; i < 9; ++i) M[i] /= (1.0 + b1 - b2 + b3) * (1.0 + b2 + (b1 - b3) * b3);
#pragma omp parallel
        {
            double *tmp = (double *)malloc(sizeof(double) * (size_t)(W > H ? W : H));
#pragma omp for
            for (int i = 0; i < H; ++i) yvv_line_64(img + (size_t)i * W, 1, W, tmp, B, b1, b2, b3, M);
#pragma omp for
            for (int i = 0; i < W; ++i) yvv_line_64(img + i, (size_t)W, H, tmp, B, b1, b2, b3, M);
            free(tmp);
        }
        return;
    }
    const float sigma = (float)sigma_d; /* the Sse functions take `const float sigma` */
    double b1, b2, b3, B, M[9];
    oracle_yvv_factors(sigma, &b1, &b2, &b3, &B, M);
    const float Bf = (float)B, b1f = (float)b1, b2f = (float)b2, b3f = (float)b3;
    float Mf[9];
    for (int i = 0; i < 9; ++i) Mf[i] = (float)M[i];
    const int hv = H - (H % 4), wv = W - (W % 8);
#pragma omp parallel
    {
        float *tmp = (float *)malloc(sizeof(float) * (size_t)(W > H ? W : H));
#pragma omp for
        for (int i = 0; i < H; ++i) {
            if (i < hv) yvv_line_f(img + (size_t)i * W, 1, W, tmp, Bf, b1f, b2f, b3f, Mf);
            else yvv_line_d(img + (size_t)i * W, 1, W, tmp, B, b1, b2, b3, M);
        }
#pragma omp for
        for (int i = 0; i < W; ++i) {
            if (i < wv) yvv_line_f(img + i, (size_t)W, H, tmp, Bf, b1f, b2f, b3f, Mf);
            else yvv_line_d(img + i, (size_t)W, H, tmp, B, b1, b2, b3, M);
        }
        free(tmp);
    }
}

void oracle_detail_mask(const float *src, float *mask, int W, int H, float scaling, float threshold, float ceiling, float factor, float blur)
{
    if (W < 8 || H < 8) {
        for (size_t k = 0; k < (size_t)W * H; ++k) mask[k] = 1.f;
        return;
    }
    const int w4 = W / 4, h4 = H / 4;
    float *L2 = (float *)malloc(sizeof(float) * 2 * (size_t)w4 * h4), *m2 = L2 + (size_t)w4 * h4;
    oracle_rescale_bilinear(src, W, H, L2, w4, h4);
#pragma omp parallel for
    for (int k = 0; k < w4 * h4; ++k) L2[k] = oracle_xlin2log(L2[k] / scaling, 50.f);
    {
        const float thr = threshold / scaling, ceil_ = ceiling / scaling;
        const float f = factor / ceil_;
#pragma omp parallel for
        for (int y = 0; y < h4; ++y) {
            const int n = (y - 1 < 0) ? y + 1 : y - 1, s = (y + 1 >= h4) ? y - 1 : y + 1;
            for (int x = 0; x < w4; ++x) {
                const int w = (x - 1 < 0) ? x + 1 : x - 1, e = (x + 1 >= w4) ? x - 1 : x + 1;
#define GETL(yy, xx) std_maxf(L2[(size_t)(yy) * w4 + (xx)], 0.f)
                float v = -8.f * GETL(y, x) + GETL(n, x) + GETL(s, x) + GETL(y, w) + GETL(y, e) + GETL(n, w) + GETL(n, e) + GETL(s, w) + GETL(s, e);
#undef GETL
                float t = fabsf(v) - thr;
                t = rt_maxf(0.f, rt_minf(t, ceil_));
                m2[(size_t)y * w4 + x] = t * f;
            }
        }
    }
    oracle_rescale_bilinear(m2, w4, h4, mask, W, H);
    const float thr1 = 1.f - factor;
#pragma omp parallel for
    for (long long k = 0; k < (long long)W * H; ++k) {
        float x = lim01f(mask[k] + thr1);
        mask[k] = oracle_xlin2log(oracle_pow_F(x, 2.23f), 101.f);
    }
    oracle_gaussian_blur(mask, W, H, blur);
    free(L2);
}

/* LUTf::operator[](vfloat) per lane (LUT.h:349-377), default clip flags */
float oracle_lutf_vec(const float *data, int size, float index)
{
    const float maxs = (float)(size - 2), sizev = (float)(size - 1);
    float clamped = sse_maxf(sse_minf(maxs, index), 0.f);  /* vclampf(value, low, high) = vmaxf(vminf(high, value), low) */
    int idx = (int)clamped;
    float lower = data[idx], upper = data[idx + 1];
    float diff = sse_maxf(sse_minf(sizev, index), 0.f) - (float)idx;
    return intpf(diff, upper, lower);
}

static inline float ftz(float x) { return fabsf(x) < FLT_MIN ? copysignf(0.f, x) : x; }

void oracle_nlmeans(float *img, int W, int H, float normcoeff, int strength, int detail_thresh, float scale)
{
    if (!strength) return;
    const int search_radius = (int)ceilf(5.f / scale), patch_radius = (int)ceilf(2.f / scale);
    const float h2 = sqrf(powf((float)strength / 100.f, 0.9f) / 10.f / scale);
    float amount = (float)detail_thresh / 100.f;
    amount = rt_maxf(0.f, rt_minf(amount, 0.99f));
    const size_t n = (size_t)W * H;
    float *mask = (float *)malloc(sizeof(float) * n);
    oracle_detail_mask(img, mask, W, H, normcoeff, 1e-3f * normcoeff, normcoeff, amount, 2.f / scale);
    const int border = search_radius + patch_radius;
    const int WW = W + border * 2, HH = H + border * 2;
    const float factor = normcoeff;
    float *src = (float *)malloc(sizeof(float) * (size_t)WW * HH);
#pragma omp parallel for
    for (int y = 0; y < HH; ++y) {
        int yy = y <= border ? 0 : y >= H ? H - 1 : y - border;
        for (int x = 0; x < WW; ++x) {
            int xx = x <= border ? 0 : x >= W ? W - 1 : x - border;
            src[(size_t)y * WW + x] = img[(size_t)yy * W + xx] / factor;
        }
    }
    memset(img, 0, sizeof(float) * n);
    enum { lutsz = 8192 };
    const float lutfactor = 100.f / (float)(lutsz - 1);
    float *explut = (float *)malloc(sizeof(float) * lutsz);
    for (int i = 0; i < lutsz; ++i) explut[i] = oracle_xexpf_s(-((float)i * lutfactor));
#pragma omp parallel for
    for (long long k = 0; k < (long long)n; ++k) mask[k] = (1.f / (mask[k] * h2)) / lutfactor;

    const int tile_size = 150;
    const int ntiles_x = (int)ceilf((float)WW / (tile_size - 2 * border));
    const int ntiles_y = (int)ceilf((float)HH / (tile_size - 2 * border));
#pragma omp parallel for schedule(dynamic, 2)
    for (int tile = 0; tile < ntiles_x * ntiles_y; ++tile) {
        const int tile_y = tile / ntiles_x, tile_x = tile % ntiles_x;
        const int start_y = tile_y * (tile_size - 2 * border), end_y = start_y + tile_size < HH ? start_y + tile_size : HH;
        const int TH = end_y - start_y;
        const int start_x = tile_x * (tile_size - 2 * border), end_x = start_x + tile_size < WW ? start_x + tile_size : WW;
        const int TW = end_x - start_x;
        if (TH <= 0 || TW <= 0) continue;
        float *St = (float *)malloc(sizeof(float) * 2 * (size_t)TW * TH), *SW = St + (size_t)TW * TH;
        memset(SW, 0, sizeof(float) * (size_t)TW * TH);
#define YC(y) ((y) + start_y < 0 ? 0 : ((y) + start_y > HH - 1 ? HH - 1 : (y) + start_y))
#define XC(x) ((x) + start_x < 0 ? 0 : ((x) + start_x > WW - 1 ? WW - 1 : (x) + start_x))
#define SCORE(tx, ty, zx, zy) ftz(sqrf(ftz(src[(size_t)YC(zy) * WW + XC(zx)] - src[(size_t)YC((zy) + (ty)) * WW + XC((zx) + (tx))])))
        for (int ty = -search_radius; ty <= search_radius; ++ty)
            for (int tx = -search_radius; tx <= search_radius; ++tx) {
                St[0] = 0.f;
                for (int xx = 1; xx < TW; ++xx) St[xx] = ftz(St[xx - 1] + SCORE(tx, ty, xx, 0));
                for (int yy = 1; yy < TH; ++yy) St[(size_t)yy * TW] = ftz(St[(size_t)(yy - 1) * TW] + SCORE(tx, ty, 0, yy));
                for (int yy = 1; yy < TH; ++yy)
                    for (int xx = 1; xx < TW; ++xx)
                        St[(size_t)yy * TW + xx] = ftz(ftz(St[(size_t)yy * TW + xx - 1] + St[(size_t)(yy - 1) * TW + xx]) - ftz(St[(size_t)(yy - 1) * TW + xx - 1] - SCORE(tx, ty, xx, yy)));
                for (int yy = start_y + border; yy < end_y - border; ++yy) {
                    const int y = yy - border;
                    const int xvec_end = end_x - border - 3; /* 4-lane groups while xx < end_x-border-3 */
                    int xx = start_x + border;
                    int nvec = xvec_end > xx ? (xvec_end - xx + 3) / 4 * 4 : 0;
                    for (; xx < end_x - border; ++xx) {
                        const int vec = (xx - (start_x + border)) < nvec;
                        const int x = xx - border, sx = xx + tx, sy = yy + ty, sty = yy - start_y, stx = xx - start_x;
                        float dist2 = ftz(ftz(ftz(St[(size_t)(sty + patch_radius) * TW + stx + patch_radius] + St[(size_t)(sty - patch_radius) * TW + stx - patch_radius]) -
                                              St[(size_t)(sty + patch_radius) * TW + stx - patch_radius]) - St[(size_t)(sty - patch_radius) * TW + stx + patch_radius]);
                        dist2 = vec ? sse_maxf(dist2, 0.f) : std_maxf(dist2, 0.f);
                        const float d = ftz(dist2 * mask[(size_t)y * W + x]);
                        float weight;
                        if (vec) {
                            /* LUTf::operator[](vfloat) with FTZ on the interpolation products/sum */
                            const float maxs = (float)(lutsz - 2), sizev = (float)(lutsz - 1);
                            float clamped = sse_maxf(sse_minf(maxs, d), 0.f);
                            int idx = (int)clamped;
                            float diff = ftz(sse_maxf(sse_minf(sizev, d), 0.f) - (float)idx);
                            weight = ftz(ftz(diff * explut[idx + 1]) + ftz(ftz(1.f - diff) * explut[idx]));
                        } else {
                            /* LUTf::operator[](float), clip below and above */
                            if (d < 0.f || !(d == d)) weight = explut[0];
                            else if (d > (float)(lutsz - 2)) weight = explut[lutsz - 1];
                            else {
                                int idx = (int)d;
                                float diff = ftz(d - (float)idx);
                                float p1 = explut[idx], p2 = ftz(explut[idx + 1] - p1);
                                weight = ftz(p1 + ftz(p2 * diff));
                            }
                        }
                        float *sw = &SW[(size_t)(y - start_y) * TW + (x - start_x)];
                        *sw = ftz(*sw + weight);
                        const float Yv = ftz(weight * src[(size_t)sy * WW + sx]);
                        img[(size_t)y * W + x] = ftz(img[(size_t)y * W + x] + Yv);
                    }
                }
            }
        for (int yy = start_y + border; yy < end_y - border; ++yy) {
            const int y = yy - border;
            for (int xx = start_x + border; xx < end_x - border; ++xx) {
                const int x = xx - border;
                const float Yv = img[(size_t)y * W + x];
                const float f = ftz(1e-5f + SW[(size_t)(y - start_y) * TW + (x - start_x)]);
                img[(size_t)y * W + x] = ftz(ftz(Yv / f) * factor);
            }
        }
#undef YC
#undef XC
#undef SCORE
        free(St);
    }
    free(mask); free(src); free(explut);
}
