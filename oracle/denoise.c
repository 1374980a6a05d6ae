/*
 * oracle/denoise.c -- CPU restatement of the wavelet part of denoise::RGB_denoise (FTblockDN):
 *   oracle_madrgb          MadRgb                         rtengine/FTblockDN.cc:569-603
 *   oracle_boxblur_flat    boxblur(T*,A*,A*,radx,rady,W,H) rtengine/boxblur.h:558-743 (SSE column groups)
 *   oracle_shrink_all_L    ShrinkAllL                     FTblockDN.cc:638-726
 *   oracle_shrink_all_AB   ShrinkAllAB                    FTblockDN.cc:729-839
 *   oracle_gamma_lut       Color::gammaf2lut (SSE form)   rtengine/color.cc:1128-1161
 *   oracle_rgb_denoise     RGB_denoise, isRAW, colorSpace RGB, QUALITY_STANDARD, one tile
 *                          (Tile_calc always returns one tile, L442-480): L1781-1823 (gamma LUTs),
 *                          L2084-2128 (RGB->YUV), L2246-2438 (decompose / MAD / shrink / reconstruct),
 *                          L2502-2550 (chroma boost, YUV->RGB, inverse gamma).  The DCT detail recovery
 *                          (L2453, detail_recovery) is a separate stage (oracle/detail.c, when present).
 *
 * TEST INFRASTRUCTURE ONLY.  PARITY: the building blocks this file composes are pinned (sleef
 * exp/log, LUTf, wavelet); the orchestration, MadRgb, boxblur and the shrink formulas are
 * UNPINNED (FTblockDN.cc needs fftw3.h/glibmm headers; boxblur.h includes StopWatch.h -> glibmm).
 */
#include "oracle.h"
#include "oracle_common.h"
#include <stdlib.h>

extern int oracle_fast;       /* detail.c: timing variant */
#include <omp.h>
/* timing variant: the band loops stay one task per band (at most 15), and each task's inner loops get their own team -- cores / 15
   threads -- instead of running on the task's single thread (nested regions are inactive in the checker: max-active-levels 1) */
static void fast_inner_team(void)
{
    if (!oracle_fast) return;
    int t = omp_get_num_procs() / 15;
    omp_set_num_threads(t < 1 ? 1 : (t > 16 ? 16 : t));
}

float oracle_madrgb(const float *data, int datalen)
{
    if (datalen <= 1) return 0;
    int *histo = (int *)calloc(65536, sizeof(int));
    if (oracle_fast) {
        /* timing variant: partial histograms of sixteen slices (integer counts: the same histogram) */
        enum { NS = 16 };
        int *part = (int *)calloc((size_t)NS * 65536, sizeof(int));
#pragma omp parallel for num_threads(NS) schedule(static, 1)
        for (int s = 0; s < NS; ++s) {
            int *h = part + (size_t)s * 65536;
            const long long i0 = (long long)datalen * s / NS, i1 = (long long)datalen * (s + 1) / NS;
            for (long long i = i0; i < i1; ++i) h[(int)fminf(fabsf(data[i]), 65535.f)]++;
        }
        for (int s = 0; s < NS; ++s)
            for (int b = 0; b < 65536; ++b) histo[b] += part[(size_t)s * 65536 + b];
        free(part);
    } else
    for (int i = 0; i < datalen; ++i) {
        /* FTblockDN.cc:587: histo[min(65535, abs(static_cast<int>(x)))]; for |x| >= 2^31, Inf and NaN that conversion is undefined
           (x86 gives INT_MIN and an out-of-range index): clamped in float first, which is the same bin for every x an int can hold */
        histo[(int)fminf(fabsf(data[i]), 65535.f)]++;
    }
    int median = 0, count = 0;
    while (count < datalen / 2) {
        count += histo[median];
        ++median;
    }
    int count_ = count - histo[median - 1];
    free(histo);
    return (((median - 1) + (datalen / 2 - count_) / ((float)(count - count_))) / 0.6745);
}

void oracle_boxblur_flat(const float *src, float *dst, float *temp, int radx, int rady, int W, int H)
{
    if (radx == 0) {
        memcpy(temp, src, sizeof(float) * (size_t)W * H);
    } else {
#pragma omp parallel for
        for (int row = H - 1; row >= 0; row--) {
            const float *s = src + (size_t)row * W;
            float *t = temp + (size_t)row * W;
            int len = radx + 1;
            float tempval = s[0];
            for (int j = 1; j <= radx; j++) tempval += s[j];
            tempval = tempval / len;
            t[0] = tempval;
            for (int col = 1; col <= radx; col++) {
                tempval = (tempval * len + s[col + radx]) / (len + 1);
                t[col] = tempval;
                len++;
            }
            float reclen = 1.f / len;
            for (int col = radx + 1; col < W - radx; col++) {
                tempval = tempval + ((float)(s[col + radx] - s[col - radx - 1])) * reclen;
                t[col] = tempval;
            }
            for (int col = W - radx; col < W; col++) {
                tempval = (tempval * len - s[col - radx - 1]) / (len - 1);
                t[col] = tempval;
                len--;
            }
        }
    }
    if (rady == 0) {
        memcpy(dst, temp, sizeof(float) * (size_t)W * H);
        return;
    }
    const int wv = (W / 4) * 4; /* 8- and 4-column SSE groups cover [0, 4*floor(W/4)) */
#pragma omp parallel for
    for (int col = 0; col < W; ++col) {
        if (col < wv) {
            float len = (float)(rady + 1);
            float tv = temp[col];
            for (int i = 1; i <= rady; i++) tv = tv + temp[(size_t)i * W + col];
            tv = tv / len;
            dst[col] = tv;
            for (int row = 1; row <= rady; row++) {
                float lenp1 = len + 1.f;
                tv = (tv * len + temp[(size_t)(row + rady) * W + col]) / lenp1;
                dst[(size_t)row * W + col] = tv;
                len = lenp1;
            }
            float rlen = 1.f / len;
            for (int row = rady + 1; row < H - rady; row++) {
                tv = tv + (temp[(size_t)(row + rady) * W + col] - temp[(size_t)(row - rady - 1) * W + col]) * rlen;
                dst[(size_t)row * W + col] = tv;
            }
            for (int row = H - rady; row < H; row++) {
                float lenm1 = len - 1.f;
                tv = (tv * len - temp[(size_t)(row - rady - 1) * W + col]) / lenm1;
                dst[(size_t)row * W + col] = tv;
                len = lenm1;
            }
        } else {
            int len = rady + 1;
            dst[col] = temp[col] / len;
            for (int i = 1; i <= rady; i++) dst[col] += temp[(size_t)i * W + col] / len;
            for (int row = 1; row <= rady; row++) {
                dst[(size_t)row * W + col] = (dst[(size_t)(row - 1) * W + col] * len + temp[(size_t)(row + rady) * W + col]) / (len + 1);
                len++;
            }
            for (int row = rady + 1; row < H - rady; row++)
                dst[(size_t)row * W + col] = dst[(size_t)(row - 1) * W + col] + (temp[(size_t)(row + rady) * W + col] - temp[(size_t)(row - rady - 1) * W + col]) / len;
            for (int row = H - rady; row < H; row++) {
                dst[(size_t)row * W + col] = (dst[(size_t)(row - 1) * W + col] * len - temp[(size_t)(row - rady - 1) * W + col]) / (len - 1);
                len--;
            }
        }
    }
}

static int blur_radius(int level, double scale)
{
    int r = (int)((level + 2) / scale);
    return r > 1 ? r : 1;
}

void oracle_shrink_all_L(oracle_wavelet *L, int level, int dir, const float *noisevarlum, const float *madL3, double scale)
{
    fast_inner_team();
    const float eps = 0.01f;
    const int N = L->w2 * L->h2, nv4 = (N / 4) * 4;
    float *c = L->band[level][dir];
    float *sfave = (float *)malloc(sizeof(float) * 3 * (size_t)N), *sfaved = sfave + N, *blurBuffer = sfaved + N;
    const float mad_L = madL3[dir - 1];
    const float levelFactor = mad_L * 5.f / (float)(level + 1);
#pragma omp parallel for
    for (int i = 0; i < N; ++i) {
        if (i < nv4) {
            float madv = noisevarlum[i] * levelFactor;
            float mag = sqrf(c[i]);
            sfave[i] = mag / (mag + madv * oracle_xexpf_v(-mag / (9.0f * madv)) + eps);
        } else {
            float mag = sqrf(c[i]);
            sfave[i] = mag / (mag + levelFactor * noisevarlum[i] * oracle_xexpf_s(-mag / (9 * levelFactor * noisevarlum[i])) + eps);
        }
    }
    const int r = blur_radius(level, scale);
    oracle_boxblur_flat(sfave, sfaved, blurBuffer, r, r, L->w2, L->h2);
#pragma omp parallel for
    for (int i = 0; i < N; ++i) {
        float sf = sfave[i];
        /* vector: c * (sfd^2 + sf^2) / (sfd + sf + eps) = (c*num)/den ; scalar: c *= num/den */
        if (i < nv4) c[i] = c[i] * (sqrf(sfaved[i]) + sqrf(sf)) / (sfaved[i] + sf + eps);
        else c[i] *= (sqrf(sfaved[i]) + sqrf(sf)) / (sfaved[i] + sf + eps);
    }
    free(sfave);
}

void oracle_shrink_all_AB(const oracle_wavelet *L, oracle_wavelet *ab, int level, int dir, const float *noisevarchrom,
                          float noisevar_ab, int useNoiseCCurve, int autoch, const float *madL3, double scale)
{
    fast_inner_team();
    const float eps = 0.01f;
    if (autoch && noisevar_ab <= 0.001f) noisevar_ab = 0.02f;
    const int N = ab->w2 * ab->h2, nv4 = (N / 4) * 4;
    const float *cL = L->band[level][dir];
    float *c = ab->band[level][dir];
    const float mad_L = madL3[dir - 1];
    float madab = sqrf(oracle_madrgb(c, N));
    if (!(noisevar_ab > 0.001f)) return;
    madab = useNoiseCCurve ? madab : madab * noisevar_ab;
    float *sfave = (float *)malloc(sizeof(float) * 3 * (size_t)N), *sfaved = sfave + N, *blurBuffer = sfaved + N;
    const float rmadLm9 = 1.f / (mad_L * 9.f);
#pragma omp parallel for
    for (int i = 0; i < N; ++i) {
        if (i < nv4) {
            float mad_abv = noisevarchrom[i] * madab;
            float mag_L = cL[i];
            float mag_ab = sqrf(c[i]);
            mag_L = sqrf(mag_L) * rmadLm9;
            sfave[i] = 1.f - oracle_xexpf_v(-(mag_ab / mad_abv) - mag_L);
        } else {
            float mag_L = sqrf(cL[i]);
            float mag_ab = sqrf(c[i]);
            sfave[i] = 1.f - oracle_xexpf_s(-(mag_ab / (noisevarchrom[i] * madab)) - (mag_L / (9.f * mad_L)));
        }
    }
    const int r = blur_radius(level, scale);
    oracle_boxblur_flat(sfave, sfaved, blurBuffer, r, r, ab->w2, ab->h2);
#pragma omp parallel for
    for (int i = 0; i < N; ++i) {
        float sf = sfave[i];
        if (i < nv4) c[i] = c[i] * (sqrf(sfaved[i]) + sqrf(sf)) / (sfaved[i] + sf + eps);
        else c[i] *= (sqrf(sfaved[i]) + sqrf(sf)) / (sfaved[i] + sf + eps);
    }
    free(sfave);
}

/* WaveletDenoiseAll_BiShrinkAB (FTblockDN.cc:976-1108): MAD of every band first; the top level goes through ShrinkAllAB with
 * that (identical) MAD, the lower levels get a point-wise shrink without the box blur */
void oracle_bishrink_AB(const oracle_wavelet *L, oracle_wavelet *ab, const float *noisevarchrom, float noisevar_ab, int useNoiseCCurve,
                        int autoch, float madL[8][3], double scale)
{
    const int maxlvl = L->nlevels;
    if (autoch && noisevar_ab <= 0.001f) noisevar_ab = 0.02f;
    float madab[8][3];
    const int N = ab->w2 * ab->h2, nv4 = (N / 4) * 4;
    for (int lvl = 0; lvl < maxlvl; ++lvl)
        for (int dir = 1; dir < 4; ++dir) madab[lvl][dir - 1] = sqrf(oracle_madrgb(ab->band[lvl][dir], N));
    for (int lvl = maxlvl - 1; lvl >= 0; lvl--)
        for (int dir = 1; dir < 4; ++dir) {
            if (lvl == maxlvl - 1) {
                oracle_shrink_all_AB(L, ab, lvl, dir, noisevarchrom, noisevar_ab, useNoiseCCurve, autoch, madL[lvl], scale);
            } else {
                const float mad_Lr = madL[lvl][dir - 1];
                const float mad_abr = useNoiseCCurve ? noisevar_ab * madab[lvl][dir - 1] : sqrf(noisevar_ab) * madab[lvl][dir - 1];
                if (noisevar_ab > 0.001f) {
                    const float *cL = L->band[lvl][dir];
                    float *c = ab->band[lvl][dir];
                    const float rmad_Lm9 = 1.f / (mad_Lr * 9.f);
#pragma omp parallel for
                    for (int i = 0; i < N; ++i) {
                        if (i < nv4) {
                            const float mad_abv = noisevarchrom[i] * mad_abr;
                            const float tempab = c[i];
                            float mag_L = cL[i];
                            const float mag_ab = sqrf(tempab);
                            mag_L = sqrf(mag_L) * rmad_Lm9;
                            c[i] = tempab * sqrf(1.f - oracle_xexpf_v(-(mag_ab / mad_abv) - (mag_L)));
                        } else {
                            const float mag_L = sqrf(cL[i]), mag_ab = sqrf(c[i]);
                            c[i] *= sqrf(1.f - oracle_xexpf_s(-(mag_ab / (noisevarchrom[i] * mad_abr)) - (mag_L / (9.f * mad_Lr))));
                        }
                    }
                }
            }
        }
}

/* Color::gammaf2lut, SSE form: lut[65536] */
void oracle_gamma_lut(float *lut, float gamma, float start, float slope, float divisor, float factor)
{
    const float gammav = 1.f / gamma;
    const float slopev = (slope / divisor) * factor;
    const float divisorv = oracle_xlogf_s(divisor);
    const float comparev = start * divisor;
    const int border = (int)(start * divisor);
    const int border1 = border - (border & 3), border2 = border1 + 4;
    int i = 0;
    for (; i < border1; ++i) lut[i] = (float)i * slopev;
    for (; i < border2 && i < 65536; ++i) {
        float iv = (float)i;
        float r0 = iv * slopev;
        float r1 = oracle_xexpf_v((oracle_xlogf_v(iv) - divisorv) * gammav) * factor;
        lut[i] = iv <= comparev ? r0 : r1;
    }
    for (; i < 65536; ++i) lut[i] = oracle_xexpf_v_nocheck((oracle_xlogf_v_nocheck((float)i) - divisorv) * gammav) * factor;
}

/* LUTf::operator[](float) for a LUT constructed with LUT_CLIP_BELOW only (LUT.h:436-459) */
static inline float lutf_clip_below(const float *data, int size, float index)
{
    const int maxs = size - 2;
    int idx = (int)index;
    if (index < 0.f || !(index == index)) return data[0];
    if (index > (float)maxs) idx = maxs;
    float diff = index - (float)idx;
    float p1 = data[idx];
    float p2 = data[idx + 1] - p1;
    return p1 + p2 * diff;
}
static inline float gammaf_s(float x, float gamma, float start, float slope)
{
    return x <= start ? x * slope : oracle_xexpf_s(oracle_xlogf_s(x) / gamma);
}

int oracle_rgb_denoise_ex(float *const img[3], size_t stride, int w, int h, const oracle_denoise_params *p,
                          const float wpi[9], const float *noisevarchrom_in, float *Lin_out, float *Lden_out, int detail_recovery, float *resid_out);
int oracle_rgb_denoise(float *const img[3], size_t stride, int w, int h, const oracle_denoise_params *p,
                       const float wpi[9], const float *noisevarchrom_in, float *Lin_out, float *Lden_out, int detail_recovery)
{
    return oracle_rgb_denoise_ex(img, stride, w, h, p, wpi, noisevarchrom_in, Lin_out, Lden_out, detail_recovery, NULL);
}

/* resid_out (nullable): {nresi, highresi} of Noise_residualAB (FTblockDN.cc:605-635, 2389-2396) */
int oracle_rgb_denoise_ex(float *const img[3], size_t stride, int w, int h, const oracle_denoise_params *p,
                          const float wpi[9], const float *noisevarchrom_in, float *Lin_out, float *Lden_out, int detail_recovery, float *resid_out)
{
    omp_set_max_active_levels(oracle_fast ? 2 : 1);      /* (timing variant: see fast_inner_team) */
    const double scale = p->scale > 0 ? p->scale : 1.0;
    const float noiseluma = (float)p->luminance;
    /* the luminance noise curve is never set in ART (ipdenoise.cc:1108 leaves noiseLCurve empty) */
    const double nl_t = (noiseluma / 125.0) * (1.0 + noiseluma / 25.0);
    const float noisevarL = (float)(nl_t * nl_t); /* static_cast<float>(SQR(...)) in double, L1687 */
    const int denoiseLuminance = noisevarL > 0.00001f;
    const int useNoiseCCurve = noisevarchrom_in != NULL;
    if (p->luminance == 0 && p->chrominance == 0 && !useNoiseCCurve) return 0;

    /* gamma LUTs (L1781-1823) */
    float gam = (float)p->gamma;
    const float gamthresh = 0.001f;
    float *gamcurve = (float *)malloc(sizeof(float) * 65536 * 2), *igamcurve = gamcurve + 65536;
    const float gamslope = exp(log((double)gamthresh) / gam) / gamthresh;
    oracle_gamma_lut(gamcurve, gam, gamthresh, gamslope, 65535.f, 65535.f);
    const float igam = 1.f / gam, igamthresh = gamthresh * gamslope, igamslope = 1.f / gamslope;
    oracle_gamma_lut(igamcurve, igam, igamthresh, igamslope, 65535.f, 65535.f);
    const float gain = powf(2.0f, (float)p->expcomp);
    float *dn_gtab = NULL, *dn_igtab = NULL;
    if (p->lab_mode) { dn_gtab = (float *)malloc(sizeof(float) * 65536 * 2); dn_igtab = dn_gtab + 65536; oracle_denoise_gamma_tabs(dn_gtab, dn_igtab); }

    const int w2 = (w + 1) / 2, h2 = (h + 1) / 2;
    const size_t n = (size_t)w * h, n2 = (size_t)w2 * h2;
    float *lab = (float *)malloc(sizeof(float) * 3 * n), *labL = lab, *laba = lab + n, *labb = lab + 2 * n;
    float *noisevarlum = (float *)malloc(sizeof(float) * 2 * n2), *noisevarchrom = noisevarlum + n2;

    const float interm_med = (float)p->chrominance / 10.0;
    float intermred = p->chrominanceRedGreen > 0. ? (p->chrominanceRedGreen / 10.) : (float)p->chrominanceRedGreen / 7.0;
    float intermblue = p->chrominanceBlueYellow > 0. ? (p->chrominanceBlueYellow / 10.) : (float)p->chrominanceBlueYellow / 7.0;
    float realred = interm_med + intermred;
    if (realred <= 0.f) realred = 0.001f;
    float realblue = interm_med + intermblue;
    if (realblue <= 0.f) realblue = 0.001f;
    const float noisevarab_r = sqrf(realred), noisevarab_b = sqrf(realblue);
    const float maxNoiseVarab = rt_maxf(noisevarab_b, noisevarab_r);

    /* RGB -> gamma -> YUV (L2084-2128) */
#pragma omp parallel for
    for (int i = 0; i < h; ++i)
        for (int j = 0; j < w; ++j) {
            float X = gain * img[0][(size_t)i * stride + j];
            float Y = gain * img[1][(size_t)i * stride + j];
            float Z = gain * img[2][(size_t)i * stride + j];
            if (p->lab_mode) {      /* L2094-2098 */
                X = oracle_lutf_noclip(dn_igtab, 65536, X);
                Y = oracle_lutf_noclip(dn_igtab, 65536, Y);
                Z = oracle_lutf_noclip(dn_igtab, 65536, Z);
            }
#define APPLY_GAMMA(v) if (gam > 1.f && v > 0.f) v = v < 65535.f ? lutf_clip_below(gamcurve, 65536, v) : (gammaf_s(v / 65535.f, gam, gamthresh, gamslope) * 65535.f)
            APPLY_GAMMA(X); APPLY_GAMMA(Y); APPLY_GAMMA(Z);
#undef APPLY_GAMMA
            /* Color::rgb2yuv with the float working-space matrix (color.h:783-788,204-207) */
            float l = X * wpi[3] + Y * wpi[4] + Z * wpi[5];
            float u = l - Z, v = X - l;
            if (p->lab_mode) oracle_rgb2lab(X, Y, Z, &l, &v, &u, wpi);     /* L2114-2116: a -> labdn->a, b -> labdn->b */
            labL[(size_t)i * w + j] = l;
            laba[(size_t)i * w + j] = v;
            labb[(size_t)i * w + j] = u;
            if (((i | j) & 1) == 0) {
                noisevarlum[(size_t)(i >> 1) * w2 + (j >> 1)] = noisevarL;
                noisevarchrom[(size_t)(i >> 1) * w2 + (j >> 1)] = useNoiseCCurve ? maxNoiseVarab * noisevarchrom_in[(size_t)(i >> 1) * w2 + (j >> 1)] : 1.f;
            }
        }

    /* wavelet levels (L2246-2293) */
    int levwav = 5;
    float maxreal = rt_maxf(realred, realblue);
    if (maxreal < 8.f) levwav = 5; else if (maxreal < 10.f) levwav = 6; else if (maxreal < 15.f) levwav = 7; else levwav = 8;
    if (p->aggressive) levwav += 2;      /* QUALITY_HIGH (L2260-2262) */
    if (levwav > 8) levwav = 8;
    { int t = (int)(levwav - ceil(log(scale))); levwav = t > 5 ? t : 5; }
    int minsizetile = w < h ? w : h, maxlev2 = 8;
    if (minsizetile < 256) maxlev2 = 7;
    if (minsizetile < 128) maxlev2 = 6;
    if (minsizetile < 64) maxlev2 = 5;
    if (minsizetile < 32) maxlev2 = 4;
    if (minsizetile < 16) maxlev2 = 3;
    levwav = levwav < maxlev2 ? levwav : maxlev2;

    oracle_wavelet *Ldecomp = oracle_wavelet_decompose(labL, w, h, levwav);
    float madL[8][3];
#pragma omp parallel for collapse(2) schedule(dynamic)
    for (int lvl = 0; lvl < levwav; ++lvl)
        for (int dir = 1; dir < 4; ++dir) madL[lvl][dir - 1] = sqrf(oracle_madrgb(Ldecomp->band[lvl][dir], (int)n2));

    float chresidtemp = 0.f, chmaxresidtemp = 0.f;
    for (int ch = 0; ch < 2; ++ch) {
        float *plane = ch == 0 ? laba : labb;
        oracle_wavelet *d = oracle_wavelet_decompose(plane, w, h, levwav);
        if (p->aggressive) oracle_bishrink_AB(Ldecomp, d, noisevarchrom, ch == 0 ? noisevarab_r : noisevarab_b, useNoiseCCurve, p->autoch, madL, scale);
#pragma omp parallel for collapse(2) schedule(dynamic)
        for (int lvl = 0; lvl < levwav; ++lvl)
            for (int dir = 1; dir < 4; ++dir)
                oracle_shrink_all_AB(Ldecomp, d, lvl, dir, noisevarchrom, ch == 0 ? noisevarab_r : noisevarab_b, useNoiseCCurve, p->autoch, madL[lvl], scale);
        if (resid_out) {
            float resid = 0.f, maxresid = 0.f;
            for (int lvl = 0; lvl < levwav; ++lvl)
                for (int dir = 1; dir < 4; ++dir) {
                    const float madC = sqrf(oracle_madrgb(d->band[lvl][dir], (int)n2));
                    resid += madC;
                    if (madC > maxresid) maxresid = madC;
                }
            if (ch == 0) { chresidtemp = resid; chmaxresidtemp = maxresid; }
            else {
                float chresid = resid + chresidtemp, chmaxresid = maxresid + chmaxresidtemp;
                chresid = sqrtf(chresid / (6 * (levwav)));
                resid_out[1] = chresid + 0.66f * (sqrtf(chmaxresid) - chresid);
                resid_out[0] = chresid;
            }
        }
        oracle_wavelet_reconstruct(d, plane, 1.f);
        oracle_wavelet_free(d);
    }
    if (denoiseLuminance) {
        const int maxlvl = levwav < 5 ? levwav : 5;
        /* QUALITY_HIGH: WaveletDenoiseAll_BiShrinkL first (L842-973) -- its per-band body is ShrinkAllL's, top level included --
         * then the standard pass; madL is not recomputed in between (L2408-2421) */
        for (int rep = p->aggressive ? 0 : 1; rep < 2; ++rep) {
#pragma omp parallel for collapse(2) schedule(dynamic)
            for (int lvl = 0; lvl < maxlvl; ++lvl)
                for (int dir = 1; dir < 4; ++dir) oracle_shrink_all_L(Ldecomp, lvl, dir, noisevarlum, madL[lvl], scale);
        }
        float *Lin = (float *)malloc(sizeof(float) * n);
        memcpy(Lin, labL, sizeof(float) * n);
        if (Lin_out) memcpy(Lin_out, labL, sizeof(float) * n);
        oracle_wavelet_reconstruct(Ldecomp, labL, 1.f);
        if (Lden_out) memcpy(Lden_out, labL, sizeof(float) * n);
        if (detail_recovery) {
            float params_Ldetail = rt_minf((float)p->luminanceDetail, 99.9f);
            oracle_detail_recovery_ex(w, h, labL, Lin, params_Ldetail, scale, p->detail_thresh);
        }
        free(Lin);
    }
    oracle_wavelet_free(Ldecomp);

    /* chroma boost, YUV -> RGB, inverse gamma (L2502-2550); numtiles == 1 */
    const float qhighFactor = p->aggressive ? 1.f / (float)0.9 : 1.0f;   /* L1672 */
    const float newGain = 1.f / gain;
#pragma omp parallel for
    for (int i = 0; i < h; ++i)
        for (int j = 0; j < w; ++j) {
            float a = laba[(size_t)i * w + j], b = labb[(size_t)i * w + j], Lv = labL[(size_t)i * w + j];
            float c_h = sqrtf(sqrf(a) + sqrf(b));
            if (c_h > 3000.f) {
                a *= 1.f + qhighFactor * realred / 100.f;
                b *= 1.f + qhighFactor * realblue / 100.f;
            }
            /* Color::yuv2rgb(L, u=b, v=a) (color.h:791-796) */
            float Z = Lv - b;
            float X = a + Lv;
            float Y = (Lv - X * wpi[3] - Z * wpi[5]) / wpi[4];
            if (p->lab_mode) oracle_lab2rgb(Lv, a, b, &X, &Y, &Z, p->iws);       /* L2522-2524 */
#define APPLY_IGAMMA(v) if (gam > 1.f && v > 0.f) v = v < 65536.f ? lutf_clip_below(igamcurve, 65536, v) : (gammaf_s(v / 65535.f, igam, igamthresh, igamslope) * 65535.f)
            APPLY_IGAMMA(X); APPLY_IGAMMA(Y); APPLY_IGAMMA(Z);
#undef APPLY_IGAMMA
            if (p->lab_mode) {      /* L2533-2537 */
                X = oracle_lutf_noclip(dn_gtab, 65536, X);
                Y = oracle_lutf_noclip(dn_gtab, 65536, Y);
                Z = oracle_lutf_noclip(dn_gtab, 65536, Z);
            }
            img[0][(size_t)i * stride + j] = newGain * X;
            img[1][(size_t)i * stride + j] = newGain * Y;
            img[2][(size_t)i * stride + j] = newGain * Z;
        }
    free(lab);
    free(noisevarlum);
    free(gamcurve);
    free(dn_gtab);
    return 0;
}
