/*
 * oracle/labadj.c -- CPU restatement of Imagefloat::rgb_to_lab / lab_to_rgb (rtengine/imagefloat.cc:841-876,941-970, SSE2 groups of
 * four + scalar tail; rtengine/color.cc:826-894,1203-1275,1382-1437) and of labAdjustments' histogram and curve loop
 * (rtengine/iplabadjustments.cc:236-264,300-327).  TEST INFRASTRUCTURE ONLY.  PARITY: xcbrtf (sleef) pinned by tests/golden; the rest
 * unpinned (imagefloat.cc / color.cc need glibmm + lcms2 headers).
 */
#include "oracle.h"
#include "oracle_common.h"
#include <stdlib.h>

#define D50X 0.9642f
#define D50Z 0.8249f
#define MAXVALF 65535.f
#define KAPPA (24389.0 / 27.0)

static float *g_cf, *g_cfy;
static void luts(void)
{
#pragma omp critical(labadj_luts)
    if (!g_cf) {
        float *a = (float *)malloc(sizeof(float) * 65536), *b = (float *)malloc(sizeof(float) * 65536);
        oracle_cachef(a); oracle_cachefy(b);
        g_cfy = b; g_cf = a;
    }
}
float oracle_xyz2lab_f(const float *cachef, float f);
static float xyz2laby(float y)
{
    if (y != y) return y;
    if (y < 0.f) return (float)(327.68 * (KAPPA * y / MAXVALF));
    if (y > 65535.f) return 327.68f * (116.f * oracle_xcbrtf(y / MAXVALF) - 16.f);
    int idx = (int)y;
    if (y > 65534.f) idx = 65534;
    const float diff = y - (float)idx, p1 = g_cfy[idx], p2 = g_cfy[idx + 1] - p1;
    return p1 + p2 * diff;
}

/* img: planes r, g, b (contiguous W x H); after the call g = L, r = a, b = b.  ws as double[9] (the TMatrix), narrowed like get_ws() */
void oracle_image_rgb_to_lab(float *const img[3], int W, int H, const double wsd[9])
{
    luts();
    float ws[9];
    for (int k = 0; k < 9; ++k) ws[k] = (float)wsd[k];
#pragma omp parallel for
    for (int y = 0; y < H; ++y) {
        float *r = img[0] + (size_t)y * W, *g = img[1] + (size_t)y * W, *b = img[2] + (size_t)y * W;
        int x = 0;
        for (; x < W - 3; x += 4) {
            float X[4], Y[4], Z[4];
            int slow = 0;
            for (int k = 0; k < 4; ++k) {
                const float R = r[x + k], G = g[x + k], B = b[x + k];
                X[k] = (ws[0] * R + ws[1] * G + ws[2] * B) / D50X;
                Y[k] = ws[3] * R + ws[4] * G + ws[5] * B;
                Z[k] = (ws[6] * R + ws[7] * G + ws[8] * B) / D50Z;
                if (sse_maxf(X[k], sse_maxf(Y[k], Z[k])) > MAXVALF) slow = 1;
                if (sse_minf(X[k], sse_minf(Y[k], Z[k])) < 0.f) slow = 1;
            }
            for (int k = 0; k < 4; ++k) {
                float fx, fy, fz, L;
                if (slow) {
                    fx = oracle_xyz2lab_f(g_cf, X[k]); fy = oracle_xyz2lab_f(g_cf, Y[k]); fz = oracle_xyz2lab_f(g_cf, Z[k]);
                    L = xyz2laby(Y[k]);
                } else {
                    fx = oracle_lutf_vec(g_cf, 65536, X[k]); fy = oracle_lutf_vec(g_cf, 65536, Y[k]); fz = oracle_lutf_vec(g_cf, 65536, Z[k]);
                    L = oracle_lutf_vec(g_cfy, 65536, Y[k]);
                }
                g[x + k] = L; r[x + k] = 500.f * (fx - fy); b[x + k] = 200.f * (fy - fz);
            }
        }
        for (; x < W; ++x) {
            float L, A, B;
            oracle_rgb2lab(r[x], g[x], b[x], &L, &A, &B, ws);
            g[x] = L; r[x] = A; b[x] = B;
        }
    }
}

static float f2xyz(float f)
{
    const float epsilonExpInv3f = (float)(6.0 / 29.0), kappaInvf = (float)(27.0 / 24389.0);
    return f > epsilonExpInv3f ? f * f * f : (116.f * f - 16.f) * kappaInvf;
}
void oracle_image_lab_to_rgb(float *const img[3], int W, int H, const double iwsd[9])
{
    float iws[9];
    for (int k = 0; k < 9; ++k) iws[k] = (float)iwsd[k];
    const float c1By116 = (float)(1.0 / 116.0), c16By116 = (float)(16.0 / 116.0);
#pragma omp parallel for
    for (int y = 0; y < H; ++y) {
        float *r = img[0] + (size_t)y * W, *g = img[1] + (size_t)y * W, *b = img[2] + (size_t)y * W;
        int x = 0;
        for (; x < W - 3; x += 4)
            for (int k = 0; k < 4; ++k) {                   /* Color::Lab2XYZ(vfloat) + xyz2rgb(vfloat) */
                const float L = g[x + k] / 327.68f, aa = r[x + k] / 327.68f, bb = b[x + k] / 327.68f;
                const float fy = c1By116 * L + c16By116;
                const float fx = 0.002f * aa + fy;
                const float fz = fy - (0.005f * bb);
                const float xx = 65535.f * f2xyz(fx) * D50X, zz = 65535.f * f2xyz(fz) * D50Z;
                const float res1 = fy * fy * fy, res2 = L / (float)KAPPA;
                const float yy = (L > 8.f ? res1 : res2) * 65535.f;
                r[x + k] = iws[0] * xx + iws[1] * yy + iws[2] * zz;
                g[x + k] = iws[3] * xx + iws[4] * yy + iws[5] * zz;
                b[x + k] = iws[6] * xx + iws[7] * yy + iws[8] * zz;
            }
        for (; x < W; ++x) {
            float R, G, B;
            oracle_lab2rgb(g[x], r[x], b[x], &R, &G, &B, iws);
            r[x] = R; g[x] = G; b[x] = B;
        }
    }
}

/* hist16[(int)L]++ with LUTu's index clamp (iplabadjustments.cc:300-327) */
void oracle_lab_histogram(const float *L, int W, int H, unsigned hist[65536])
{
    for (int i = 0; i < 65536; ++i) hist[i] = 0;
    for (size_t k = 0; k < (size_t)W * H; ++k) {
        const float v = L[k];
        int idx = (v >= -2147483648.f && v < 2147483648.f) ? (int)v : (int)0x80000000;
        idx = idx < 0 ? 0 : (idx > 65535 ? 65535 : idx);
        hist[idx]++;
    }
}

/* the curve loop (iplabadjustments.cc:236-264): lcurve = LUTf(32770, 0), acurve / bcurve = LUTf(65536) */
void oracle_lab_adjustments(float *const img[3], int W, int H, const float *lcurve, const float *acurve, const float *bcurve, float chroma)
{
#pragma omp parallel for
    for (int y = 0; y < H; ++y) {
        float *a = img[0] + (size_t)y * W, *L = img[1] + (size_t)y * W, *b = img[2] + (size_t)y * W;
        int x = 0;
        for (; x < W - 3; x += 4)
            for (int k = 0; k < 4; ++k) {
                L[x + k] = oracle_lutf_vec(lcurve, 32770, L[x + k]);
                a[x + k] = (oracle_lutf_vec(acurve, 65536, a[x + k] + 32768.f) - 32768.f) * chroma;
                b[x + k] = (oracle_lutf_vec(bcurve, 65536, b[x + k] + 32768.f) - 32768.f) * chroma;
            }
        for (; x < W; ++x) {
            L[x] = oracle_lutf_noclip(lcurve, 32770, L[x]);
            a[x] = (oracle_lutf(acurve, 65536, a[x] + 32768.f) - 32768.f) * chroma;
            b[x] = (oracle_lutf(bcurve, 65536, b[x] + 32768.f) - 32768.f) * chroma;
        }
    }
}
