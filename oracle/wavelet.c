/*
 * oracle/wavelet.c -- restatement of rtengine::wavelet_decomposition for subsampling == 1
 * (reference: rtengine/cplx_wavelet_dec.h:97-270, cplx_wavelet_level.h:206-744,
 *  cplx_wavelet_filter_coeffs.h:25-35): level 0 = decimated Daub4 (6 taps, offset 2), levels >= 1
 * = undecimated Haar with skip = 2^(level-1).
 *
 * TEST INFRASTRUCTURE ONLY.  PARITY PINNED: checked bit-for-bit against the reference's own
 * cplx_wavelet_dec.{h,cc}/cplx_wavelet_level.h compiled in place (oracle/_ref,
 * tests/golden/wavelet.npz).
 *
 * Layout: levels[l].band[1..3] are the three detail subbands (w2 x h2 each), coeff0 the final
 * low-pass.  Accumulation order of every tap sum follows the reference (`lo += f[j]*src`,
 * j = 0..taps-1, zero taps included; synthesis `tot += fLo[j]*lo + fHi[j]*hi`, j = begin,begin+2,..).
 */
#include "oracle.h"
#include "oracle_common.h"
#include <stdlib.h>

static const float DAUB4[2][6] = {
    {0.f, 0.f, 0.34150635f, 0.59150635f, 0.15849365f, -0.091506351f},
    {-0.091506351f, -0.15849365f, 0.59150635f, -0.34150635f, 0.f, 0.f}};
enum { TAPS = 6, OFFSET = 2 };

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

int oracle_wavelet_skip(int level) { return level <= 1 ? 1 : 1 << (level - 1); }

/* level 0: src (w x h) -> lo, b1, b2, b3 (each w2 x h2) */
static void analyse_subsamp(const float *src, int w, int h, float *lo, float *b1, float *b2, float *b3)
{
    const int w2 = (w + 1) / 2;
#pragma omp parallel
    {
        float *tmpLo = (float *)malloc(sizeof(float) * 2 * (size_t)w), *tmpHi = tmpLo + w;
#pragma omp for
        for (int row = 0; row < h; row += 2) {
            for (int k = 0; k < w; ++k) {
                float l = 0.f, hh = 0.f;
                for (int j = 0; j < TAPS; ++j) {
                    const float s = src[(size_t)clampi(row + (OFFSET - j), 0, h - 1) * w + k];
                    l += DAUB4[0][j] * s;
                    hh += DAUB4[1][j] * s;
                }
                tmpLo[k] = l;
                tmpHi[k] = hh;
            }
            for (int pass = 0; pass < 2; ++pass) {
                const float *t = pass ? tmpHi : tmpLo;
                float *dLo = pass ? b2 : lo, *dHi = pass ? b3 : b1;
                for (int i = 0; i < w; i += 2) {
                    float l = 0.f, hh = 0.f;
                    for (int j = 0; j < TAPS; ++j) {
                        const float s = t[clampi(i + (OFFSET - j), 0, w - 1)];
                        l += DAUB4[0][j] * s;
                        hh += DAUB4[1][j] * s;
                    }
                    dLo[(size_t)(row / 2) * w2 + i / 2] = l;
                    dHi[(size_t)(row / 2) * w2 + i / 2] = hh;
                }
            }
        }
        free(tmpLo);
    }
}

/* levels >= 1: undecimated Haar (cplx_wavelet_level.h:206-238) */
static void analyse_haar(const float *src, int w, int h, int skip, float *lo, float *b1, float *b2, float *b3)
{
#pragma omp parallel
    {
        float *tmpLo = (float *)calloc(2 * (size_t)w, sizeof(float)), *tmpHi = tmpLo + w;
#pragma omp for
        for (int row = 0; row < h; ++row) {
            if (row < h - skip) {
                for (int j = 0; j < w; ++j) {
                    tmpLo[j] = 0.25f * (src[(size_t)row * w + j] + src[(size_t)(row + skip) * w + j]);
                    tmpHi[j] = 0.25f * (src[(size_t)row * w + j] - src[(size_t)(row + skip) * w + j]);
                }
            } else if (row >= (h - skip > skip ? h - skip : skip)) {
                for (int j = 0; j < w; ++j) {
                    tmpLo[j] = 0.25f * (src[(size_t)row * w + j] + src[(size_t)(row - skip) * w + j]);
                    tmpHi[j] = 0.25f * (src[(size_t)row * w + j] - src[(size_t)(row - skip) * w + j]);
                }
            }
            for (int pass = 0; pass < 2; ++pass) {
                const float *t = pass ? tmpHi : tmpLo;
                float *dLo = pass ? b2 : lo, *dHi = pass ? b3 : b1;
                for (int i = 0; i < w - skip; ++i) {
                    dLo[(size_t)row * w + i] = t[i] + t[i + skip];
                    dHi[(size_t)row * w + i] = t[i] - t[i + skip];
                }
                for (int i = (w - skip > skip ? w - skip : skip); i < w; ++i) {
                    dLo[(size_t)row * w + i] = t[i] + t[i - skip];
                    dHi[(size_t)row * w + i] = t[i] - t[i - skip];
                }
            }
        }
        free(tmpLo);
    }
}

oracle_wavelet *oracle_wavelet_decompose(const float *src, int w, int h, int maxlvl)
{
    oracle_wavelet *d = (oracle_wavelet *)calloc(1, sizeof *d);
    d->w = w; d->h = h; d->nlevels = maxlvl;
    d->w2 = (w + 1) / 2; d->h2 = (h + 1) / 2;
    const size_t n = (size_t)d->w2 * d->h2;
    float *cur = (float *)malloc(n * sizeof(float)), *nxt = (float *)malloc(n * sizeof(float));
    for (int l = 0; l < maxlvl; ++l) {
        float *blk = (float *)malloc(3 * n * sizeof(float));
        d->band[l][0] = NULL; d->band[l][1] = blk; d->band[l][2] = blk + n; d->band[l][3] = blk + 2 * n;
        if (l == 0) {
            analyse_subsamp(src, w, h, cur, blk, blk + n, blk + 2 * n);
        } else {
            analyse_haar(cur, d->w2, d->h2, oracle_wavelet_skip(l), nxt, blk, blk + n, blk + 2 * n);
            float *t = cur; cur = nxt; nxt = t;
        }
    }
    d->coeff0 = cur;
    free(nxt);
    return d;
}

void oracle_wavelet_free(oracle_wavelet *d)
{
    if (!d) return;
    for (int l = 0; l < d->nlevels; ++l) free(d->band[l][1]);
    free(d->coeff0);
    free(d);
}

static void synth_haar_h(const float *lo, const float *hi, float *dst, int w, int h, int skip)
{
#pragma omp parallel for
    for (int k = 0; k < h; ++k) {
        for (int i = 0; i < skip && i < w; ++i) dst[(size_t)k * w + i] = lo[(size_t)k * w + i] + hi[(size_t)k * w + i];
        for (int i = skip; i < w; ++i)
            dst[(size_t)k * w + i] = 0.5f * (lo[(size_t)k * w + i] + hi[(size_t)k * w + i] + lo[(size_t)k * w + i - skip] - hi[(size_t)k * w + i - skip]);
    }
}
static void synth_haar_v(const float *lo, const float *hi, float *dst, int w, int h, int skip)
{
#pragma omp parallel for
    for (int i = 0; i < h; ++i)
        for (int j = 0; j < w; ++j)
            dst[(size_t)i * w + j] = i < skip ? lo[(size_t)i * w + j] + hi[(size_t)i * w + j]
                                              : 0.5f * (lo[(size_t)i * w + j] + hi[(size_t)i * w + j] + lo[(size_t)(i - skip) * w + j] - hi[(size_t)(i - skip) * w + j]);
}

/* level 0 synthesis: horizontal w2 -> w (rows h2), vertical h2 -> h with the x4 and blend */
static void synth_subsamp_h(const float *lo, const float *hi, float *dst, int srcw, int dstw, int rows)
{
    const int shift = TAPS - OFFSET - 1; /* skip == 1 */
    float fLo[TAPS], fHi[TAPS];
    for (int i = 0; i < TAPS; ++i) { fLo[i] = DAUB4[0][TAPS - 1 - i]; fHi[i] = DAUB4[1][TAPS - 1 - i]; }
#pragma omp parallel for
    for (int k = 0; k < rows; ++k)
        for (int i = 0; i < dstw; ++i) {
            float tot = 0.f;
            const int i_src = (i + shift) / 2, begin = (i + shift) % 2;
            for (int j = begin, l = 0; j < TAPS; j += 2, l += 1) {
                const int arg = clampi(i_src - l, 0, srcw - 1);
                tot += (fLo[j] * lo[(size_t)k * srcw + arg] + fHi[j] * hi[(size_t)k * srcw + arg]);
            }
            dst[(size_t)k * dstw + i] = tot;
        }
}
static void synth_subsamp_v(const float *lo, const float *hi, float *dst, int w, int srch, int dsth, float blend)
{
    const int shift = TAPS - OFFSET - 1;
    const float srcFactor = 1.f - blend;
    float fLo[TAPS], fHi[TAPS];
    for (int i = 0; i < TAPS; ++i) { fLo[i] = DAUB4[0][TAPS - 1 - i]; fHi[i] = DAUB4[1][TAPS - 1 - i]; }
#pragma omp parallel for
    for (int i = 0; i < dsth; ++i) {
        const int i_src = (i + shift) / 2, begin = (i + shift) % 2;
        for (int k = 0; k < w; ++k) {
            float tot = 0.f;
            for (int j = begin, l = 0; j < TAPS; j += 2, l += 1) {
                const size_t arg = (size_t)clampi(i_src - l, 0, srch - 1) * w + k;
                tot += (fLo[j] * lo[arg] + fHi[j] * hi[arg]);
            }
            dst[(size_t)i * w + k] = dst[(size_t)i * w + k] * srcFactor + blend * 4.f * tot;
        }
    }
}

void oracle_wavelet_reconstruct(oracle_wavelet *d, float *dst, float blend)
{
    const int w2 = d->w2, h2 = d->h2;
    const size_t n = (size_t)w2 * h2;
    float *tmpHi = (float *)malloc(sizeof(float) * (size_t)d->w * h2);
    float *tmpLo = (float *)malloc(sizeof(float) * (size_t)d->w * h2);
    for (int l = d->nlevels - 1; l > 0; --l) {
        const int skip = oracle_wavelet_skip(l);
        synth_haar_h(d->band[l][2], d->band[l][3], tmpHi, w2, h2, skip);
        synth_haar_h(d->coeff0, d->band[l][1], tmpLo, w2, h2, skip);
        synth_haar_v(tmpLo, tmpHi, d->coeff0, w2, h2, skip);
    }
    (void)n;
    synth_subsamp_h(d->band[0][2], d->band[0][3], tmpHi, w2, d->w, h2);
    synth_subsamp_h(d->coeff0, d->band[0][1], tmpLo, w2, d->w, h2);
    synth_subsamp_v(tmpLo, tmpHi, dst, d->w, h2, d->h, blend);
    free(tmpHi);
    free(tmpLo);
}
