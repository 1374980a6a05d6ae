/*
 * oracle/detail.c -- CPU restatement of detail_recovery (reference: rtengine/FTblockDN.cc:1479-1635)
 * with RGBtile_denoise (L494-525), RGBoutput_tile_row (L531-558), the tile masks (L1828-1846) and
 * boxabsblur (rtengine/boxblur.h:745-886), for luminanceDetailThreshold == 0 (no detail_mask).
 *
 * TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED, and not pinnable at the bit level: the reference
 * calls FFTW3 (fftwf_plan_many_r2r, FFTW_REDFT10 / FFTW_REDFT01, FFTW_MEASURE; L1604,1614,1924-1931),
 * a system library that is not part of /root/reference and not installed here; its round-off
 * depends on the planner's codelet choice, so the reference itself is not reproducible at the ULP
 * level across machines.  This restatement evaluates FFTW's documented definitions
 *     REDFT10: Y_k = 2 sum_j X_j cos(pi (j+1/2) k / n)
 *     REDFT01: Y_k = X_0 + 2 sum_{j>=1} X_j cos(pi j (k+1/2) / n)
 * directly with double accumulation (rounded to float per 1-D pass, like FFTW's float plans), and
 * tests compare the device result with a tolerance.  Block accumulation order: vblk ascending,
 * hblk ascending (the reference's serial order; its OpenMP loop over vblk also races on totwt).
 */
#include "oracle.h"
#include "oracle_common.h"
#include <stdlib.h>

#define DTS 64
#define DOFF 25
#define DBLKRAD 1

void oracle_detail_tilemasks(float *tilemask_in, float *tilemask_out)
{
    const float epsilon = 0.001f / (DTS * DTS);
    const int border = 4; /* MAX(2, TS/16) */
    for (int i = 0; i < DTS; ++i) {
        float i1 = abs((i > DTS / 2 ? i - DTS + 1 : i));
        float vmask = (i1 < border ? sqr_d(sin((M_PI * i1) / (2 * border))) : 1.0f);
        float vmask2 = (i1 < 2 * border ? sqr_d(sin((M_PI * i1) / (2 * border))) : 1.0f);
        for (int j = 0; j < DTS; ++j) {
            float j1 = abs((j > DTS / 2 ? j - DTS + 1 : j));
            tilemask_in[i * DTS + j] = (vmask * (j1 < border ? sqr_d(sin((M_PI * j1) / (2 * border))) : 1.0f)) + epsilon;
            tilemask_out[i * DTS + j] = (vmask2 * (j1 < 2 * border ? sqr_d(sin((M_PI * j1) / (2 * border))) : 1.0f)) + epsilon;
        }
    }
}

float oracle_detail_factor(float d)
{
    /* compute_detail (L1482-1486) */
    float t = (float)(((100. - d) * (100. - d)) + 50. * (100. - d)) * DTS * 0.5f;
    return t * t;
}

/* test hook: 0 = double accumulation (the checker's default: the mathematically exact DCT rounded once per 1-D pass);
   1 = plain fp32 direct form (float products and float running sums in index order): what a straightforward single-precision
   implementation gives -- the yardstick tests/test_gpu_denoise.py measures the device's fp32 fast DCT against */
int oracle_detail_dct_f32 = 0;

/* TIMING variant, never the checker (bench.py's cpu_baseline sets oracle_fast = 1): what a CPU implementation written for speed does
   where the checker is written for exactness -- the 64-point DCTs as Lee's O(n log n) recursion in fp32 (the reference calls FFTW's
   r2r plans, FTblockDN.cc:1604-1614), and the per-band loops of the shrink passes handing all cores to one band at a time instead of
   one core per band (the checker has at most 15 bands to spread over the host's cores).  Results then differ from the checker's by
   fp32 rounding of the DCT; nothing compares them. */
int oracle_fast = 0;

static const float LEE_TW[6][32] = {
    {0.707106781f},
    {0.5411961f, 1.30656296f},
    {0.509795579f, 0.601344887f, 0.899976223f, 2.56291545f},
    {0.502419286f, 0.522498615f, 0.566944035f, 0.646821783f, 0.788154623f, 1.06067769f, 1.7224471f, 5.10114862f},
    {0.500602998f, 0.50547096f, 0.51544731f, 0.531042591f, 0.553103896f, 0.582934968f, 0.622504123f, 0.674808341f, 0.744536271f, 0.839349645f, 0.972568238f, 1.16943993f, 1.48416462f, 2.05778101f, 3.40760842f, 10.1900081f},
    {0.500150636f, 0.501358452f, 0.503788726f, 0.507471172f, 0.512451479f, 0.518792713f, 0.526577315f, 0.535909817f, 0.546920438f, 0.559769813f, 0.574655184f, 0.591818536f, 0.611557348f, 0.634238937f, 0.660319808f, 0.690372128f, 0.725120522f, 0.765494165f, 0.812702091f, 0.868344715f, 0.934583597f, 1.01440826f, 1.11207162f, 1.23383274f, 1.38929396f, 1.59397228f, 1.87467598f, 2.28205007f, 2.92462843f, 4.08461108f, 6.79675071f, 20.3738782f}};

/* x <- F, F[k] = sum_n x[n] cos(pi (2n+1) k / (2N)); lg = log2(N) */
static void lee_fwd(float *x, int N, int lg)
{
    if (N == 1) return;
    float a[32], b[32];
    const float *tw = LEE_TW[lg - 1];
    const int h = N / 2;
    for (int n = 0; n < h; ++n) { a[n] = x[n] + x[N - 1 - n]; b[n] = (x[n] - x[N - 1 - n]) * tw[n]; }
    lee_fwd(a, h, lg - 1);
    lee_fwd(b, h, lg - 1);
    for (int k = 0; k < h; ++k) { x[2 * k] = a[k]; x[2 * k + 1] = k + 1 < h ? b[k] + b[k + 1] : b[k]; }
}
/* z <- y, y[n] = sum_k z[k] cos(pi (2n+1) k / (2N)) */
static void lee_inv(float *z, int N, int lg)
{
    if (N == 1) return;
    float a[32], b[32];
    const float *tw = LEE_TW[lg - 1];
    const int h = N / 2;
    for (int k = 0; k < h; ++k) { a[k] = z[2 * k]; b[k] = k > 0 ? z[2 * k + 1] + z[2 * k - 1] : z[1]; }
    lee_inv(a, h, lg - 1);
    lee_inv(b, h, lg - 1);
    for (int n = 0; n < h; ++n) { const float t = b[n] * tw[n]; z[n] = a[n] + t; z[N - 1 - n] = a[n] - t; }
}
/* REDFT10 = 2 lee_fwd; REDFT01(X) = lee_inv(X[0], 2 X[1], 2 X[2], ...) -- rows, then columns */
static void dct2d_fast(float *blk, int inverse)
{
    float line[DTS];
    for (int pass = 0; pass < 2; ++pass)
        for (int r = 0; r < DTS; ++r) {
            for (int j = 0; j < DTS; ++j) line[j] = pass == 0 ? blk[r * DTS + j] : blk[j * DTS + r];
            if (!inverse) {
                lee_fwd(line, DTS, 6);
                for (int j = 0; j < DTS; ++j) line[j] *= 2.f;
            } else {
                for (int j = 1; j < DTS; ++j) line[j] *= 2.f;
                lee_inv(line, DTS, 6);
            }
            for (int j = 0; j < DTS; ++j) { if (pass == 0) blk[r * DTS + j] = line[j]; else blk[j * DTS + r] = line[j]; }
        }
}

static void dct2d_f32(float *blk, const double *costab, int inverse)
{
    float tmp[DTS * DTS];
    for (int pass = 0; pass < 2; ++pass) {
        const float *src = pass == 0 ? blk : tmp;
        float *dst = pass == 0 ? tmp : blk;
        for (int r = 0; r < DTS; ++r)
            for (int k = 0; k < DTS; ++k) {
                float acc;
                if (!inverse) {
                    acc = 0.f;
                    for (int j = 0; j < DTS; ++j) acc += (pass == 0 ? src[r * DTS + j] : src[j * DTS + r]) * (float)costab[k * DTS + j];
                    acc *= 2.f;
                } else {
                    acc = pass == 0 ? src[r * DTS] : src[r];
                    for (int j = 1; j < DTS; ++j) acc += 2.f * (pass == 0 ? src[r * DTS + j] : src[j * DTS + r]) * (float)costab[j * DTS + k];
                }
                if (pass == 0) dst[r * DTS + k] = acc; else dst[k * DTS + r] = acc;
            }
    }
}

static void dct2d(float *blk, const double *costab, int inverse)
{
    if (oracle_fast) { dct2d_fast(blk, inverse); return; }
    if (oracle_detail_dct_f32) { dct2d_f32(blk, costab, inverse); return; }
    /* costab[k*64+j] = cos(pi*(j+0.5)*k/64).  Rows then columns; float storage between passes. */
    float tmp[DTS * DTS];
    for (int pass = 0; pass < 2; ++pass) {
        const float *src = pass == 0 ? blk : tmp;
        float *dst = pass == 0 ? tmp : blk;
        for (int r = 0; r < DTS; ++r)      /* line index in the non-transformed dimension */
            for (int k = 0; k < DTS; ++k) { /* output index */
                double acc;
                if (!inverse) {
                    acc = 0.0;
                    for (int j = 0; j < DTS; ++j) acc += (double)(pass == 0 ? src[r * DTS + j] : src[j * DTS + r]) * costab[k * DTS + j];
                    acc *= 2.0;
                } else {
                    acc = (double)(pass == 0 ? src[r * DTS] : src[r]);
                    for (int j = 1; j < DTS; ++j) acc += 2.0 * (double)(pass == 0 ? src[r * DTS + j] : src[j * DTS + r]) * costab[j * DTS + k];
                }
                if (pass == 0) dst[r * DTS + k] = (float)acc; else dst[k * DTS + r] = (float)acc;
            }
    }
}

static void boxabsblur64(const float *src, float *dst, int rad)
{
    float temp[DTS * DTS];
    for (int row = 0; row < DTS; row++) {
        const float *s = src + row * DTS;
        int len = rad + 1;
        float tempval = fabsf(s[0]);
        for (int j = 1; j <= rad; j++) tempval += fabsf(s[j]);
        tempval /= len;
        temp[row * DTS] = tempval;
        for (int col = 1; col <= rad; col++) {
            tempval = (tempval * len + fabsf(s[col + rad])) / (len + 1);
            temp[row * DTS + col] = tempval;
            len++;
        }
        float rlen = 1.f / (float)len;
        for (int col = rad + 1; col < DTS - rad; col++) {
            tempval = tempval + ((float)(fabsf(s[col + rad]) - fabsf(s[col - rad - 1]))) * rlen;
            temp[row * DTS + col] = tempval;
        }
        for (int col = DTS - rad; col < DTS; col++) {
            tempval = (tempval * len - fabsf(s[col - rad - 1])) / (len - 1);
            temp[row * DTS + col] = tempval;
            len--;
        }
    }
    for (int col = 0; col < DTS; ++col) { /* W % 4 == 0: every column takes the SSE form */
        float len = (float)(rad + 1);
        float tv = temp[col];
        for (int i = 1; i <= rad; i++) tv = tv + temp[i * DTS + col];
        tv = tv / len;
        dst[col] = tv;
        for (int row = 1; row <= rad; row++) {
            float lenp1 = len + 1.f;
            tv = (tv * len + temp[(row + rad) * DTS + col]) / lenp1;
            dst[row * DTS + col] = tv;
            len = lenp1;
        }
        float rlen = 1.f / len;
        for (int row = rad + 1; row < DTS - rad; row++) {
            tv = tv + (temp[(row + rad) * DTS + col] - temp[(row - rad - 1) * DTS + col]) * rlen;
            dst[row * DTS + col] = tv;
        }
        for (int row = DTS - rad; row < DTS; row++) {
            float lenm1 = len - 1.f;
            tv = (tv * len - temp[(row - rad - 1) * DTS + col]) / lenm1;
            dst[row * DTS + col] = tv;
            len = lenm1;
        }
    }
}

void oracle_detail_recovery_ex(int width, int height, float *L, const float *Lin, float params_Ldetail, double scale, int detail_thresh);
void oracle_detail_recovery(int width, int height, float *L, const float *Lin, float params_Ldetail, double scale)
{
    oracle_detail_recovery_ex(width, height, L, Lin, params_Ldetail, scale, 0);
}

/* detail_thresh = DenoiseParams::luminanceDetailThreshold (FTblockDN.cc:1502-1507,1583) */
void oracle_detail_recovery_ex(int width, int height, float *L, const float *Lin, float params_Ldetail, double scale, int detail_thresh)
{
    float *mask = NULL;
    if (detail_thresh > 0) {
        float amount = (float)detail_thresh / 100.f;
        amount = amount < 0.f ? 0.f : (amount > 1.f ? 1.f : amount);
        mask = (float *)malloc(sizeof(float) * (size_t)width * height);
        oracle_detail_mask(L, mask, width, height, 65535.f, 25.f, 10000.f, amount, (float)(25.f / scale));
    }
    const float detail_hi = oracle_detail_factor(params_Ldetail), detail_lo = oracle_detail_factor(0.f);
    const int numblox_W = (int)ceil(((float)width) / DOFF) + 2 * DBLKRAD;
    const int numblox_H = (int)ceil(((float)height) / DOFF) + 2 * DBLKRAD;
    float tm_in[DTS * DTS], tm_out[DTS * DTS];
    oracle_detail_tilemasks(tm_in, tm_out);
    double *costab = (double *)malloc(sizeof(double) * DTS * DTS);
    for (int k = 0; k < DTS; ++k)
        for (int j = 0; j < DTS; ++j) costab[k * DTS + j] = cos(M_PI * (j + 0.5) * k / DTS);
    const float DCTnorm = 1.0f / (4 * DTS * DTS);
    const int blur_rad = (int)(3 / scale) > 1 ? (int)(3 / scale) : 1;
    const size_t n = (size_t)width * height;
    float *Ldetail = (float *)calloc(n, sizeof(float)), *totwt = (float *)calloc(n, sizeof(float));
    /* block results first (parallel), accumulation afterwards in the reference's serial order */
    float *blocks = (float *)malloc(sizeof(float) * (size_t)numblox_W * numblox_H * DTS * DTS);
#pragma omp parallel for collapse(2) schedule(dynamic)
    for (int vblk = 0; vblk < numblox_H; ++vblk)
        for (int hblk = 0; hblk < numblox_W; ++hblk) {
            float *blk = blocks + ((size_t)vblk * numblox_W + hblk) * DTS * DTS;
            float factor[DTS * DTS], nbrwt[DTS * DTS];
            const int top = (vblk - DBLKRAD) * DOFF, left = (hblk - DBLKRAD) * DOFF;
            for (int i = 0; i < DTS; ++i) {
                int row = top + i, rr = row;
                if (row < 0) rr = -row < height - 1 ? -row : height - 1;
                else if (row >= height) rr = 2 * height - 2 - row > 0 ? 2 * height - 2 - row : 0;
                for (int j = 0; j < DTS; ++j) {
                    int col = left + j, cc = col;
                    /* datarow padding (L1556-1562) */
                    if (col < 0) cc = -col < width - 1 ? -col : width - 1;
                    else if (col >= width) cc = 2 * width - 2 - col > 0 ? 2 * width - 2 - col : 0;
                    const float v = Lin[(size_t)rr * width + cc] - L[(size_t)rr * width + cc];
                    blk[i * DTS + j] = tm_in[i * DTS + j] * v;
                    factor[i * DTS + j] = (row >= 0 && row < height && col >= 0 && col < width)
                                              ? (mask ? oracle_detail_factor(params_Ldetail * mask[(size_t)row * width + col]) : detail_hi) : detail_lo;
                }
            }
            dct2d(blk, costab, 0);
            boxabsblur64(blk, nbrwt, blur_rad);
            for (int k = 0; k < DTS * DTS; ++k) blk[k] = blk[k] * (1.0f - oracle_xexpf_v(-sqrf(nbrwt[k]) / factor[k]));
            dct2d(blk, costab, 1);
        }
    /* every pixel takes its contributions in the reference's order (vblk ascending, then hblk ascending); rows are independent */
#pragma omp parallel for schedule(dynamic, 8)
    for (int y = 0; y < height; ++y)
    for (int vblk = 0; vblk < numblox_H; ++vblk) {
        const int top = (vblk - DBLKRAD) * DOFF;
        {
            const int i = y - top;
            if (i < 0 || i >= DTS) continue;
            for (int hblk = 0; hblk < numblox_W; ++hblk) {
                const int left = (hblk - DBLKRAD) * DOFF;
                const float *blk = blocks + ((size_t)vblk * numblox_W + hblk) * DTS * DTS;
                for (int j = 0; j < DTS; ++j) {
                    const int x = left + j;
                    if (x < 0 || x >= width) continue;
                    Ldetail[(size_t)y * width + x] += tm_out[i * DTS + j] * blk[i * DTS + j] * DCTnorm;
                    totwt[(size_t)y * width + x] += tm_in[i * DTS + j] * tm_out[i * DTS + j];
                }
            }
        }
    }
#pragma omp parallel for
    for (long long k = 0; k < (long long)n; ++k) L[k] += Ldetail[k] / totwt[k];
    free(blocks); free(Ldetail); free(totwt); free(costab); free(mask);
}
