/*
 * oracle/vng4.c -- CPU restatement of RawImageSource::vng4_demosaic (rtengine/vng4_demosaic_RT.cc:32-397): dcraw's VNG with the 4-colour
 * CFA description (`prefilters`: the second green is colour 3), green from the gradient-thresholded neighbourhood, red / blue by linear
 * colour-difference interpolation against that green, 3-pixel border by border_interpolate2.  TEST INFRASTRUCTURE ONLY.
 * PARITY: unpinned (needs rtengine.h -> glibmm / lcms2).  Plain scalar fp32 in the reference (the `__SSE2__` block only stores the
 * weight as a float), so there is one form.
 */
#include "oracle.h"
#include "oracle_common.h"
#include <limits.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* dcraw identify(): mark the second green of a three-colour Bayer pattern as colour 3 (RawImage::set_prefilters keeps this value) */
unsigned oracle_prefilters(unsigned filters)
{
    return filters | ((((filters >> 2) & 0x22222222u) | ((filters << 2) & 0x88888888u)) & (filters << 1));
}

static const signed short vng_terms[] = {
    -2, -2, +0, -1, 0, 0x01, -2, -2, +0, +0, 1, 0x01, -2, -1, -1, +0, 0, 0x01,
    -2, -1, +0, -1, 0, 0x02, -2, -1, +0, +0, 0, 0x03, -2, -1, +0, +1, 1, 0x01,
    -2, +0, +0, -1, 0, 0x06, -2, +0, +0, +0, 1, 0x02, -2, +0, +0, +1, 0, 0x03,
    -2, +1, -1, +0, 0, 0x04, -2, +1, +0, -1, 1, 0x04, -2, +1, +0, +0, 0, 0x06,
    -2, +1, +0, +1, 0, 0x02, -2, +2, +0, +0, 1, 0x04, -2, +2, +0, +1, 0, 0x04,
    -1, -2, -1, +0, 0, 0x80, -1, -2, +0, -1, 0, 0x01, -1, -2, +1, -1, 0, 0x01,
    -1, -2, +1, +0, 1, 0x01, -1, -1, -1, +1, 0, 0x88, -1, -1, +1, -2, 0, 0x40,
    -1, -1, +1, -1, 0, 0x22, -1, -1, +1, +0, 0, 0x33, -1, -1, +1, +1, 1, 0x11,
    -1, +0, -1, +2, 0, 0x08, -1, +0, +0, -1, 0, 0x44, -1, +0, +0, +1, 0, 0x11,
    -1, +0, +1, -2, 1, 0x40, -1, +0, +1, -1, 0, 0x66, -1, +0, +1, +0, 1, 0x22,
    -1, +0, +1, +1, 0, 0x33, -1, +0, +1, +2, 1, 0x10, -1, +1, +1, -1, 1, 0x44,
    -1, +1, +1, +0, 0, 0x66, -1, +1, +1, +1, 0, 0x22, -1, +1, +1, +2, 0, 0x10,
    -1, +2, +0, +1, 0, 0x04, -1, +2, +1, +0, 1, 0x04, -1, +2, +1, +1, 0, 0x04,
    +0, -2, +0, +0, 1, 0x80, +0, -1, +0, +1, 1, 0x88, +0, -1, +1, -2, 0, 0x40,
    +0, -1, +1, +0, 0, 0x11, +0, -1, +2, -2, 0, 0x40, +0, -1, +2, -1, 0, 0x20,
    +0, -1, +2, +0, 0, 0x30, +0, -1, +2, +1, 1, 0x10, +0, +0, +0, +2, 1, 0x08,
    +0, +0, +2, -2, 1, 0x40, +0, +0, +2, -1, 0, 0x60, +0, +0, +2, +0, 1, 0x20,
    +0, +0, +2, +1, 0, 0x30, +0, +0, +2, +2, 1, 0x10, +0, +1, +1, +0, 0, 0x44,
    +0, +1, +1, +2, 0, 0x10, +0, +1, +2, -1, 1, 0x40, +0, +1, +2, +0, 0, 0x60,
    +0, +1, +2, +1, 0, 0x20, +0, +1, +2, +2, 0, 0x10, +1, -2, +1, +0, 0, 0x80,
    +1, -1, +1, +1, 0, 0x88, +1, +0, +1, +2, 0, 0x08, +1, +0, +2, -1, 0, 0x40,
    +1, +0, +2, +1, 0, 0x10
};
static const signed short vng_chood[] = {-1, -1, -1, 0, -1, +1, 0, +1, +1, +1, +1, 0, +1, -1, 0, -1};

/* The VNG code table of one (row & 7, col & 1) cell (vng4_demosaic_RT.cc:224-282), offsets in units of pixels * 4 + colour.  The weight
 * is stored as its float bit pattern like the reference's SSE2 build.  Returns the number of ints written (<= 320). */
int oracle_vng4_code(unsigned pf, int width, int row, int col, int32_t *ip0)
{
    int32_t *ip = ip0;
    const signed short *cp = vng_terms;
    for (int t = 0; t < 64; t++) {
        const int y1 = *cp++, x1 = *cp++, y2 = *cp++, x2 = *cp++, weight = *cp++, grads = *cp++;
        const unsigned color = fc(pf, row + y1, col + x1);
        if (fc(pf, row + y2, col + x2) != color) continue;
        const int diag = (fc(pf, row, col + 1) == color && fc(pf, row + 1, col) == color) ? 2 : 1;
        if (abs(y1 - y2) == diag && abs(x1 - x2) == diag) continue;
        *ip++ = (y1 * width + x1) * 4 + (int)color;
        *ip++ = (y2 * width + x2) * 4 + (int)color;
        { const float w = (float)(1 << weight); int32_t b; memcpy(&b, &w, 4); *ip++ = b; }
        for (int g = 0; g < 8; g++) if (grads & (1 << g)) *ip++ = g;
        *ip++ = -1;
    }
    *ip++ = INT_MAX;
    cp = vng_chood;
    for (int g = 0; g < 8; g++) {
        const int y = *cp++, x = *cp++;
        *ip++ = (y * width + x) * 4;
        const unsigned color = fc(pf, row, col);
        if (fc(pf, row + y, col + x) != color && fc(pf, row + y * 2, col + x * 2) == color) *ip++ = (y * width + x) * 8 + (int)color;
        else *ip++ = 0;
    }
    return (int)(ip - ip0);
}

static float min8(const float *g) { float m = g[0]; for (int k = 1; k < 8; ++k) m = rt_minf(m, g[k]); return m; }
static float max8(const float *g) { float m = g[0]; for (int k = 1; k < 8; ++k) m = rt_maxf(m, g[k]); return m; }

/* filters: the three-colour pattern (RawImage::filters), prefilters: the four-colour one (0 = derive it like dcraw) */
void oracle_vng4_demosaic(const float *raw, int W, int H, unsigned filters, unsigned prefilters, float *red, float *green, float *blue)
{
    const unsigned pf = prefilters ? prefilters : oracle_prefilters(filters);
    const int width = W, height = H;
    float (*image)[4] = (float (*)[4])calloc((size_t)H * W, sizeof *image);
    for (int i = 0; i < H; ++i)
        for (int j = 0; j < W; ++j) image[(size_t)i * W + j][fc(pf, i, j)] = raw[(size_t)i * W + j];
    /* first linear interpolation (L117-217): every colour the pixel does not have = weighted mean of its 3x3 neighbours of that colour */
#pragma omp parallel for
    for (int row = 1; row < H - 1; ++row)
        for (int col = 1; col < W - 1; ++col) {
            float *pix = image[(size_t)row * W + col];
            float sum[4] = {0, 0, 0, 0}, wsum[4] = {0, 0, 0, 0};
            for (int y = -1; y <= 1; y++)
                for (int x = -1; x <= 1; x++) {
                    const int shift = (y == 0) + (x == 0);
                    if (shift == 2) continue;
                    const unsigned color = fc(pf, row + y, col + x);
                    sum[color] += pix[(width * y + x) * 4 + (int)color] * (float)(1 << shift);
                    wsum[color] += (float)(1 << shift);
                }
            for (unsigned c = 0; c < 4; c++)
                if (c != fc(pf, row, col)) pix[c] = sum[c] * (1.f / wsum[c]);
        }
    /* the 8 x 2 code tables */
    int32_t *code[8][2];
    int32_t *buf = (int32_t *)calloc(16, 1280);
    for (int row = 0; row < 8; row++)
        for (int col = 0; col < 2; col++) { code[row][col] = buf + (row * 2 + col) * 320; oracle_vng4_code(pf, width, row, col, code[row][col]); }
    /* VNG green (L302-357) */
#pragma omp parallel for
    for (int row = 2; row < height - 2; row++)
        for (int col = 2; col < width - 2; col++) {
            const float *pix = image[(size_t)row * width + col];
            int color = (int)fc(pf, row, col);
            const int32_t *ip = code[row & 7][col & 1];
            float gval[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            while (ip[0] != INT_MAX) {
                float w; memcpy(&w, &ip[2], 4);
                const float diff = fabsf(pix[ip[0]] - pix[ip[1]]) * w;
                gval[ip[3]] += diff;
                ip += 5;
                if (ip[-1] != -1) { gval[ip[-1]] += diff; ip++; }
            }
            ip++;
            const float thold = min8(gval) + max8(gval) * 0.5f;
            float sum0 = 0.f, sum1 = 0.f;
            const float greenval = pix[color];
            int num = 0;
            if (color & 1) {
                color ^= 2;
                for (int g = 0; g < 8; g++, ip += 2)
                    if (gval[g] <= thold) {
                        if (ip[1]) sum0 += greenval + pix[ip[1]];
                        sum1 += pix[ip[0] + color];
                        num++;
                    }
                sum0 *= 0.5f;
            } else {
                for (int g = 0; g < 8; g++, ip += 2)
                    if (gval[g] <= thold) {
                        if (ip[1]) sum0 += greenval + pix[ip[1]];
                        sum1 += pix[ip[0] + 1] + pix[ip[0] + 3];
                        num++;
                    }
            }
            green[(size_t)row * width + col] = std_maxf(0.f, greenval + (sum1 - sum0) / (2 * num));
        }
    free(buf);
    free(image);
    /* vng4interpolate_row_redblue (L32-57) on rows 3 .. H-4 */
#pragma omp parallel for
    for (int i = 3; i < H - 3; ++i) {
        float *ar = red + (size_t)i * W, *ab = blue + (size_t)i * W;
        if (fc(filters, i, 0) == 2 || fc(filters, i, 1) == 2) { float *t = ar; ar = ab; ab = t; }
        const float *pg = green + (size_t)(i - 1) * W, *cg = green + (size_t)i * W, *ng = green + (size_t)(i + 1) * W;
        const float *r = raw + (size_t)i * W;
        for (int j = 3; j < W - 3; ++j) {
            if (fc(filters, i, j) != 1) {
                ar[j] = r[j];
                float rb = (r[-W + j - 1] - pg[j - 1] + r[W + j - 1] - ng[j - 1]);
                rb += (r[-W + j + 1] - pg[j + 1] + r[W + j + 1] - ng[j + 1]);
                ab[j] = std_maxf(0.f, cg[j] + rb * 0.25f);
            } else {
                ar[j] = std_maxf(0.f, cg[j] + (r[j - 1] - cg[j - 1] + r[j + 1] - cg[j + 1]) / 2);
                ab[j] = std_maxf(0.f, cg[j] + (r[-W + j] - pg[j] + r[W + j] - ng[j]) / 2);
            }
        }
    }
    oracle_border_interpolate2(W, H, 3, raw, (size_t)W, filters, red, green, blue, (size_t)W);
}
