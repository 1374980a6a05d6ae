/*
 * oracle/guided.c -- CPU restatement of the guided chroma smoothing of the denoise stage:
 *   oracle_boxblur_ring       boxblur(float**,float**,int radius,W,H,mt)   rtengine/boxblur.h:318-556
 *   oracle_bilinear / rescale getBilinearValue, rescaleBilinear            rtengine/rescale.h:27-74
 *   oracle_guided_filter      guidedFilter                                 rtengine/guidedfilter.cc:58-241
 *   oracle_guided_filter_log  guidedFilterLog                              guidedfilter.cc:244-265
 *   oracle_denoise_guided_smoothing  denoise::denoiseGuidedSmoothing -> guided_smoothing(Channel::C)
 *                                    rtengine/ipsmoothing.cc:334-409,875-897; Imagefloat::multiply imagefloat.cc:396-424
 *
 * TEST INFRASTRUCTURE ONLY.  PARITY: xlin2log/xlog2lin (sleef) pinned; the rest UNPINNED
 * (guidedfilter.cc includes boxblur.h -> StopWatch.h -> glibmm; ipsmoothing.cc needs improcfun.h).
 */
#include "oracle.h"
#include "oracle_common.h"
#include <stdlib.h>

/* in place; horizontal steady state divides by len, vertical multiplies by rlen (all columns) */
void oracle_boxblur_ring(float *img, int radius, int W, int H)
{
    if (radius == 0) return;
#pragma omp parallel
    {
        float *lineBuffer = (float *)malloc(sizeof(float) * (radius + 1));
#pragma omp for
        for (int row = 0; row < H; row++) {
            float *s = img + (size_t)row * W;
            float len = radius + 1;
            float tempval = s[0];
            lineBuffer[0] = tempval;
            for (int j = 1; j <= radius; j++) tempval += s[j];
            tempval /= len;
            s[0] = tempval;
            for (int col = 1; col <= radius; col++) {
                lineBuffer[col] = s[col];
                tempval = (tempval * len + s[col + radius]) / (len + 1);
                s[col] = tempval;
                ++len;
            }
            int pos = 0;
            for (int col = radius + 1; col < W - radius; col++) {
                const float oldVal = lineBuffer[pos];
                lineBuffer[pos] = s[col];
                s[col] = tempval = tempval + (s[col + radius] - oldVal) / len;
                ++pos;
                pos = pos <= radius ? pos : 0;
            }
            for (int col = W - radius; col < W; col++) {
                s[col] = tempval = (tempval * len - lineBuffer[pos]) / (len - 1);
                --len;
                ++pos;
                pos = pos <= radius ? pos : 0;
            }
        }
        free(lineBuffer);
        float *rowBuffer = (float *)malloc(sizeof(float) * (radius + 1));
#pragma omp for
        for (int col = 0; col < W; ++col) {
            float len = radius + 1;
            float tv = img[col];
            rowBuffer[0] = tv;
            for (int i = 1; i <= radius; i++) tv = tv + img[(size_t)i * W + col];
            tv = tv / len;
            img[col] = tv;
            for (int row = 1; row <= radius; row++) {
                rowBuffer[row] = img[(size_t)row * W + col];
                tv = (tv * len + img[(size_t)(row + radius) * W + col]) / (len + 1.f);
                img[(size_t)row * W + col] = tv;
                len = len + 1.f;
            }
            const float rlen = 1.f / len;
            int pos = 0;
            for (int row = radius + 1; row < H - radius; row++) {
                const float oldVal = rowBuffer[pos];
                rowBuffer[pos] = img[(size_t)row * W + col];
                tv = tv + (img[(size_t)(row + radius) * W + col] - oldVal) * rlen;
                img[(size_t)row * W + col] = tv;
                ++pos;
                pos = pos <= radius ? pos : 0;
            }
            for (int row = H - radius; row < H; row++) {
                tv = (tv * len - rowBuffer[pos]) / (len - 1.f);
                img[(size_t)row * W + col] = tv;
                len = len - 1.f;
                ++pos;
                pos = pos <= radius ? pos : 0;
            }
        }
        free(rowBuffer);
    }
}

float oracle_bilinear(const float *src, int W, int H, float x, float y)
{
    int xi = (int)x < W - 1 ? (int)x : W - 1;
    int yi = (int)y < H - 1 ? (int)y : H - 1;
    float xf = x - xi, yf = y - yi;
    int xi1 = xi + 1 < W - 1 ? xi + 1 : W - 1;
    int yi1 = yi + 1 < H - 1 ? yi + 1 : H - 1;
    float bl = src[(size_t)yi * W + xi], br = src[(size_t)yi * W + xi1];
    float tl = src[(size_t)yi1 * W + xi], tr = src[(size_t)yi1 * W + xi1];
    float b = xf * br + (1.f - xf) * bl;
    float t = xf * tr + (1.f - xf) * tl;
    return yf * t + (1.f - yf) * b;
}

void oracle_rescale_bilinear(const float *src, int Ws, int Hs, float *dst, int Wd, int Hd)
{
    float col_scale = (float)Ws / (float)Wd, row_scale = (float)Hs / (float)Hd;
#pragma omp parallel for
    for (int y = 0; y < Hd; ++y) {
        float ymrs = y * row_scale;
        for (int x = 0; x < Wd; ++x) dst[(size_t)y * Wd + x] = oracle_bilinear(src, Ws, Hs, x * col_scale, ymrs);
    }
}

static int calculate_subsampling(int w, int h, int r)
{
    if (r == 1) return 1;
    if ((w > h ? w : h) <= 600) return 1;
    for (int s = 5; s > 0; --s)
        if (r % s == 0) return s;
    int t = r / 2;
    return t < 2 ? 2 : (t > 4 ? 4 : t);
}

static void f_mean(float *d, int w, int h, float radf)
{
    int rad = (int)radf; /* LIM(rad, 0, (min(w,h)-1)/2 - 1) with rad converted from float r1 */
    int hi = ((w < h ? w : h) - 1) / 2 - 1;
    rad = rad < 0 ? 0 : rad;          /* LIM = max(low, min(val, high)) */
    rad = rad < hi ? rad : hi;
    rad = rad > 0 ? rad : 0;
    oracle_boxblur_ring(d, rad, w, h);
}

/* dst may alias src */
void oracle_guided_filter(const float *guide, const float *src, float *dst, int W, int H, int r, float epsilon, int subsampling)
{
    if (subsampling <= 0) subsampling = calculate_subsampling(W, H, r);
    const int w = W / subsampling, h = H / subsampling;
    const size_t n = (size_t)w * h;
    float *I1 = (float *)malloc(sizeof(float) * 4 * n), *p1 = I1 + n, *meanI = p1 + n, *meanp = meanI + n;
    if (w == W && h == H) {
        memcpy(I1, guide, sizeof(float) * n);
        memcpy(p1, src, sizeof(float) * n);
    } else {
        oracle_rescale_bilinear(guide, W, H, I1, w, h);
        oracle_rescale_bilinear(src, W, H, p1, w, h);
    }
    const float r1 = (float)r / subsampling;
    memcpy(meanI, I1, sizeof(float) * n); f_mean(meanI, w, h, r1);
    memcpy(meanp, p1, sizeof(float) * n); f_mean(meanp, w, h, r1);
    float *corrIp = p1, *corrI = I1;
    for (size_t k = 0; k < n; ++k) corrIp[k] = I1[k] * p1[k];
    f_mean(corrIp, w, h, r1);
    for (size_t k = 0; k < n; ++k) corrI[k] = I1[k] * I1[k];
    f_mean(corrI, w, h, r1);
    float *varI = corrI, *covIp = corrIp;
    for (size_t k = 0; k < n; ++k) varI[k] = corrI[k] - (meanI[k] * meanI[k]);      /* SUBMUL: c - a*b */
    for (size_t k = 0; k < n; ++k) covIp[k] = corrIp[k] - (meanI[k] * meanp[k]);
    float *a = varI, *b = covIp;
    for (size_t k = 0; k < n; ++k) a[k] = covIp[k] / (varI[k] + epsilon);          /* DIVEPSILON */
    for (size_t k = 0; k < n; ++k) b[k] = meanp[k] - (a[k] * meanI[k]);            /* SUBMUL */
    f_mean(a, w, h, r1);
    f_mean(b, w, h, r1);
    const float col_scale = (float)w / (float)W, row_scale = (float)h / (float)H;
#pragma omp parallel for
    for (int y = 0; y < H; ++y) {
        float ymrs = y * row_scale;
        for (int x = 0; x < W; ++x)
            dst[(size_t)y * W + x] = oracle_bilinear(a, w, h, x * col_scale, ymrs) * guide[(size_t)y * W + x] + oracle_bilinear(b, w, h, x * col_scale, ymrs);
    }
    free(I1);
}

void oracle_guided_filter_log(const float *guide, float base, float *chan, int W, int H, int r, float eps, int subsampling)
{
    const size_t n = (size_t)W * H;
#pragma omp parallel for
    for (long long k = 0; k < (long long)n; ++k) chan[k] = oracle_xlin2log(rt_maxf(chan[k], 0.f), base);
    oracle_guided_filter(guide, chan, chan, W, H, r, eps, subsampling);
#pragma omp parallel for
    for (long long k = 0; k < (long long)n; ++k) chan[k] = oracle_xlog2lin(rt_maxf(chan[k], 0.f), base);
}

/* img planes contiguous (stride == W), values 0..65535; ws = working-space matrix in DOUBLE (TMatrix) */
void oracle_denoise_guided_smoothing(float *const img[3], int W, int H, const double ws[9], int guidedChromaRadius, double scale)
{
    if (guidedChromaRadius == 0) return;
    const size_t n = (size_t)W * H;
    const float f1 = 1.f / 65535.f;
    for (int c = 0; c < 3; ++c)
        for (size_t k = 0; k < n; ++k) img[c][k] *= f1;
    int r = (int)round(guidedChromaRadius / scale);
    r = r > 0 ? r : 0;
    if (r > 0) {
        float *iR = (float *)malloc(sizeof(float) * 4 * n), *iG = iR + n, *iB = iG + n, *guide = iB + n;
        memcpy(iR, img[0], sizeof(float) * n); memcpy(iG, img[1], sizeof(float) * n); memcpy(iB, img[2], sizeof(float) * n);
#pragma omp parallel for
        for (long long k = 0; k < (long long)n; ++k) {
            float l = img[0][k] * ws[3] + img[1][k] * ws[4] + img[2][k] * ws[5]; /* rgbLuminance with double ws */
            guide[k] = oracle_xlin2log(rt_maxf(l, 0.f), 10.f);
        }
        for (int c = 0; c < 3; ++c) oracle_guided_filter_log(guide, 10.f, img[c], W, H, r, 0.001f, 0);
#pragma omp parallel for
        for (long long k = 0; k < (long long)n; ++k) {
            float rr = img[0][k], gg = img[1][k], bb = img[2][k], ir = iR[k], ig = iG[k], ib = iB[k];
            /* Color::rgb2yuv with the double matrix (color.h:783-788,204-207) */
            float iY = ir * ws[3] + ig * ws[4] + ib * ws[5];
            float oY = rr * ws[3] + gg * ws[4] + bb * ws[5];
            float ou = oY - bb, ov = rr - oY;
            float bump = oY > 1e-5f ? iY / oY : 1.f;
            ou *= bump;
            ov *= bump;
            oY = iY;
            /* Color::yuv2rgb (color.h:791-796) */
            float B = oY - ou;
            float R = ov + oY;
            float G = (oY - R * ws[3] - B * ws[5]) / ws[4];
            img[0][k] = R; img[1][k] = G; img[2][k] = B;
        }
        free(iR);
    }
    for (int c = 0; c < 3; ++c)
        for (size_t k = 0; k < n; ++k) img[c][k] *= 65535.f;
}
