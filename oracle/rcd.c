/*
 * oracle/rcd.c -- CPU restatement of RawImageSource::rcd_demosaic
 * (reference: rtengine/rcd_demosaic.cc:51-347, RCD 2.3, tiled).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle_common.h).  PARITY UNPINNED: the reference
 * has no tests or golden vectors for this function and its translation unit cannot
 * be compiled in this image without stand-in headers (rawimagesource.h -> glibmm,
 * lcms2 ...), so this restatement is checked by reading only.
 *
 * Semantics kept from the reference:
 *   - tile grid: 194x194 tiles, stride 176, origin (0,0); write-back margin 9
 *     (rcd_demosaic.cc:82-87,112-125,304-316).
 *   - input LIM01(raw/65536) (L130), output max(0, v*65536) (L312-314).
 *   - VH_Dir is only defined on [4,rows-4)x[4,cols-4); step 3 at row/col 4 reads
 *     row/col 3 of it, which the reference leaves at its calloc value 0 for full
 *     tiles (L101,137-166,199).  PQ_Dir aliases lpf (L103), so PQ_Dir positions
 *     outside [4,rows-4) hold lpf values.
 *   - partial (right/bottom edge) tiles: the reference re-uses per-thread buffers
 *     without clearing them, so positions it does not recompute hold values of
 *     whatever tile the thread processed before (schedule-dependent).  This
 *     restatement defines them as a freshly calloc'ed buffer (the state a thread has
 *     on its first tile), i.e. every tile starts from all-zero work planes.
 */
#include "oracle.h"
#include "oracle_common.h"
#include <stdlib.h>

#define RCD_TS 194
#define RCD_BORDER 9

/* test hook: 1 = what ONE reference thread does -- the work buffer is calloc'ed once and never cleared again (rcd_demosaic.cc:99-105),
   tiles in raster order -- instead of this restatement's "every tile starts from zeroed planes" */
int oracle_rcd_stale = 0;

void oracle_rcd_tile(const float *raw, size_t rs, int W, int H, unsigned filters,
                     int tr, int tc, int numTh, int numTw,
                     float *red, float *green, float *blue, size_t os, float *work)
{
    enum { ts = RCD_TS, w1 = ts, w2 = 2 * ts, w3 = 3 * ts, w4 = 4 * ts };
    const int tileSizeN = ts - 2 * RCD_BORDER;
    const float eps = 1e-5f, epssq = 1e-10f, scale = 65536.f;
    unsigned cf[2][2] = {{fc(filters, 0, 0), fc(filters, 0, 1)}, {fc(filters, 1, 0), fc(filters, 1, 1)}};
#define FCT(r, c) (cf[(r) & 1][(c) & 1])

    const int rowStart = tr * tileSizeN;
    const int rowEnd = rowStart + ts < H ? rowStart + ts : H;
    if (rowStart + RCD_BORDER == rowEnd - RCD_BORDER) return;
    const int colStart = tc * tileSizeN;
    const int colEnd = colStart + ts < W ? colStart + ts : W;
    if (colStart + RCD_BORDER == colEnd - RCD_BORDER) return;
    const int tileRows = rowEnd - rowStart < ts ? rowEnd - rowStart : ts;
    const int tilecols = colEnd - colStart < ts ? colEnd - colStart : ts;

    /* work planes: cfa, rgb[3], VH_Dir (full) ; PQ_Dir(=lpf), P, Q (half) -- all zeroed */
    const size_t full = (size_t)ts * ts, half = full / 2;
    if (!oracle_rcd_stale) memset(work, 0, sizeof(float) * (5 * full + 3 * half));
    float *cfa = work;
    float *rgb[3] = {work + full, work + 2 * full, work + 3 * full};
    float *VH_Dir = work + 4 * full;
    float *PQ_Dir = work + 5 * full;
    float *lpf = PQ_Dir;
    float *P_CDiff_Hpf = PQ_Dir + half;
    float *Q_CDiff_Hpf = P_CDiff_Hpf + half;

    for (int row = rowStart; row < rowEnd; row++) {
        const int c0 = FCT(row, colStart), c1 = FCT(row, colStart + 1);
        for (int col = colStart, indx = (row - rowStart) * ts; col < colEnd; ++col, ++indx) {
            float v = lim01f(raw[(size_t)row * rs + col] / scale);
            cfa[indx] = rgb[c0][indx] = rgb[c1][indx] = v;
        }
    }

    /* Step 1: V/H colour-difference high-pass, squared; VH_Dir on [4,rows-4)x[4,cols-4) */
    for (int row = 4; row < tileRows - 4; ++row) {
        for (int col = 4, indx = row * ts + col; col < tilecols - 4; ++col, ++indx) {
            float V[3], Hh[3];
            for (int k = -1; k <= 1; ++k) {
                int i = indx + k * w1;
                V[k + 1] = sqrf((cfa[i - w3] - cfa[i - w1] - cfa[i + w1] + cfa[i + w3]) - 3.f * (cfa[i - w2] + cfa[i + w2]) + 6.f * cfa[i]);
                int j = indx + k;
                Hh[k + 1] = sqrf((cfa[j - 3] - cfa[j - 1] - cfa[j + 1] + cfa[j + 3]) - 3.f * (cfa[j - 2] + cfa[j + 2]) + 6.f * cfa[j]);
            }
            float V_Stat = std_maxf(epssq, V[0] + V[1] + V[2]);
            float H_Stat = std_maxf(epssq, Hh[0] + Hh[1] + Hh[2]);
            VH_Dir[indx] = V_Stat / (V_Stat + H_Stat);
        }
    }

    /* Step 2: low-pass filter at non-green sites (half-res index) */
    for (int row = 2; row < tileRows - 2; ++row) {
        for (int col = 2 + (FCT(row, 0) & 1), indx = row * ts + col, lp = indx / 2; col < tilecols - 2; col += 2, indx += 2, ++lp) {
            lpf[lp] = cfa[indx] +
                      0.5f * (cfa[indx - w1] + cfa[indx + w1] + cfa[indx - 1] + cfa[indx + 1]) +
                      0.25f * (cfa[indx - w1 - 1] + cfa[indx - w1 + 1] + cfa[indx + w1 - 1] + cfa[indx + w1 + 1]);
        }
    }

    /* Step 3: green at red/blue sites */
    for (int row = 4; row < tileRows - 4; ++row) {
        for (int col = 4 + (FCT(row, 0) & 1), indx = row * ts + col, lp = indx / 2; col < tilecols - 4; col += 2, indx += 2, ++lp) {
            const float cfai = cfa[indx];
            const float N_Grad = eps + (fabsf(cfa[indx - w1] - cfa[indx + w1]) + fabsf(cfai - cfa[indx - w2])) + (fabsf(cfa[indx - w1] - cfa[indx - w3]) + fabsf(cfa[indx - w2] - cfa[indx - w4]));
            const float S_Grad = eps + (fabsf(cfa[indx - w1] - cfa[indx + w1]) + fabsf(cfai - cfa[indx + w2])) + (fabsf(cfa[indx + w1] - cfa[indx + w3]) + fabsf(cfa[indx + w2] - cfa[indx + w4]));
            const float W_Grad = eps + (fabsf(cfa[indx - 1] - cfa[indx + 1]) + fabsf(cfai - cfa[indx - 2])) + (fabsf(cfa[indx - 1] - cfa[indx - 3]) + fabsf(cfa[indx - 2] - cfa[indx - 4]));
            const float E_Grad = eps + (fabsf(cfa[indx - 1] - cfa[indx + 1]) + fabsf(cfai - cfa[indx + 2])) + (fabsf(cfa[indx + 1] - cfa[indx + 3]) + fabsf(cfa[indx + 2] - cfa[indx + 4]));

            const float lpfi = lpf[lp];
            const float N_Est = cfa[indx - w1] * (lpfi + lpfi) / (eps + lpfi + lpf[lp - w1]);
            const float S_Est = cfa[indx + w1] * (lpfi + lpfi) / (eps + lpfi + lpf[lp + w1]);
            const float W_Est = cfa[indx - 1] * (lpfi + lpfi) / (eps + lpfi + lpf[lp - 1]);
            const float E_Est = cfa[indx + 1] * (lpfi + lpfi) / (eps + lpfi + lpf[lp + 1]);

            const float V_Est = (S_Grad * N_Est + N_Grad * S_Est) / (N_Grad + S_Grad);
            const float H_Est = (W_Grad * E_Est + E_Grad * W_Est) / (E_Grad + W_Grad);

            const float VH_C = VH_Dir[indx];
            const float VH_N = 0.25f * ((VH_Dir[indx - w1 - 1] + VH_Dir[indx - w1 + 1]) + (VH_Dir[indx + w1 - 1] + VH_Dir[indx + w1 + 1]));
            const float VH_Disc = fabsf(0.5f - VH_C) < fabsf(0.5f - VH_N) ? VH_N : VH_C;
            rgb[1][indx] = intpf(VH_Disc, H_Est, V_Est);
        }
    }

    /* Step 4.0: P/Q diagonal high-pass, squared (half-res index) */
    for (int row = 3; row < tileRows - 3; ++row) {
        for (int col = 3, indx = row * ts + col, i2 = indx / 2; col < tilecols - 3; col += 2, indx += 2, i2++) {
            P_CDiff_Hpf[i2] = sqrf((cfa[indx - w3 - 3] - cfa[indx - w1 - 1] - cfa[indx + w1 + 1] + cfa[indx + w3 + 3]) - 3.f * (cfa[indx - w2 - 2] + cfa[indx + w2 + 2]) + 6.f * cfa[indx]);
            Q_CDiff_Hpf[i2] = sqrf((cfa[indx - w3 + 3] - cfa[indx - w1 + 1] - cfa[indx + w1 - 1] + cfa[indx + w3 - 3]) - 3.f * (cfa[indx - w2 + 2] + cfa[indx + w2 - 2]) + 6.f * cfa[indx]);
        }
    }

    /* Step 4.1: P/Q discrimination strength.  PQ_Dir aliases lpf: it is written after
       the last read of lpf (step 3), positions it does not cover keep lpf values. */
    for (int row = 4; row < tileRows - 4; ++row) {
        for (int col = 4 + (FCT(row, 0) & 1), indx = row * ts + col, i2 = indx / 2, i3 = (indx - w1 - 1) / 2, i4 = (indx + w1 - 1) / 2; col < tilecols - 4; col += 2, indx += 2, i2++, i3++, i4++) {
            float P_Stat = std_maxf(epssq, P_CDiff_Hpf[i3] + P_CDiff_Hpf[i2] + P_CDiff_Hpf[i4 + 1]);
            float Q_Stat = std_maxf(epssq, Q_CDiff_Hpf[i3 + 1] + Q_CDiff_Hpf[i2] + Q_CDiff_Hpf[i4]);
            PQ_Dir[i2] = P_Stat / (P_Stat + Q_Stat);
        }
    }

    /* Step 4.2: red/blue at blue/red sites */
    for (int row = 4; row < tileRows - 4; ++row) {
        for (int col = 4 + (FCT(row, 0) & 1), indx = row * ts + col, c = 2 - FCT(row, col), pq = indx / 2, pq2 = (indx - w1 - 1) / 2, pq3 = (indx + w1 - 1) / 2; col < tilecols - 4; col += 2, indx += 2, ++pq, ++pq2, ++pq3) {
            float PQ_C = PQ_Dir[pq];
            float PQ_N = 0.25f * (PQ_Dir[pq2] + PQ_Dir[pq2 + 1] + PQ_Dir[pq3] + PQ_Dir[pq3 + 1]);
            float PQ_Disc = (fabsf(0.5f - PQ_C) < fabsf(0.5f - PQ_N)) ? PQ_N : PQ_C;
            const float *rc = rgb[c], *r1 = rgb[1];
            float NW_Grad = eps + fabsf(rc[indx - w1 - 1] - rc[indx + w1 + 1]) + fabsf(rc[indx - w1 - 1] - rc[indx - w3 - 3]) + fabsf(r1[indx] - r1[indx - w2 - 2]);
            float NE_Grad = eps + fabsf(rc[indx - w1 + 1] - rc[indx + w1 - 1]) + fabsf(rc[indx - w1 + 1] - rc[indx - w3 + 3]) + fabsf(r1[indx] - r1[indx - w2 + 2]);
            float SW_Grad = eps + fabsf(rc[indx - w1 + 1] - rc[indx + w1 - 1]) + fabsf(rc[indx + w1 - 1] - rc[indx + w3 - 3]) + fabsf(r1[indx] - r1[indx + w2 - 2]);
            float SE_Grad = eps + fabsf(rc[indx - w1 - 1] - rc[indx + w1 + 1]) + fabsf(rc[indx + w1 + 1] - rc[indx + w3 + 3]) + fabsf(r1[indx] - r1[indx + w2 + 2]);
            float NW_Est = rc[indx - w1 - 1] - r1[indx - w1 - 1];
            float NE_Est = rc[indx - w1 + 1] - r1[indx - w1 + 1];
            float SW_Est = rc[indx + w1 - 1] - r1[indx + w1 - 1];
            float SE_Est = rc[indx + w1 + 1] - r1[indx + w1 + 1];
            float P_Est = (NW_Grad * SE_Est + SE_Grad * NW_Est) / (NW_Grad + SE_Grad);
            float Q_Est = (NE_Grad * SW_Est + SW_Grad * NE_Est) / (NE_Grad + SW_Grad);
            rgb[c][indx] = r1[indx] + intpf(PQ_Disc, Q_Est, P_Est);
        }
    }

    /* Step 4.3: red/blue at green sites */
    for (int row = 4; row < tileRows - 4; ++row) {
        for (int col = 4 + (FCT(row, 1) & 1), indx = row * ts + col; col < tilecols - 4; col += 2, indx += 2) {
            float VH_C = VH_Dir[indx];
            float VH_N = 0.25f * ((VH_Dir[indx - w1 - 1] + VH_Dir[indx - w1 + 1]) + (VH_Dir[indx + w1 - 1] + VH_Dir[indx + w1 + 1]));
            float VH_Disc = (fabsf(0.5f - VH_C) < fabsf(0.5f - VH_N)) ? VH_N : VH_C;
            const float *r1 = rgb[1];
            float rgb1 = r1[indx];
            float N1 = eps + fabsf(rgb1 - r1[indx - w2]);
            float S1 = eps + fabsf(rgb1 - r1[indx + w2]);
            float W1 = eps + fabsf(rgb1 - r1[indx - 2]);
            float E1 = eps + fabsf(rgb1 - r1[indx + 2]);
            float rgb1mw1 = r1[indx - w1], rgb1pw1 = r1[indx + w1], rgb1m1 = r1[indx - 1], rgb1p1 = r1[indx + 1];
            for (int c = 0; c <= 2; c += 2) {
                float *rc = rgb[c];
                float SNabs = fabsf(rc[indx - w1] - rc[indx + w1]);
                float EWabs = fabsf(rc[indx - 1] - rc[indx + 1]);
                float N_Grad = N1 + SNabs + fabsf(rc[indx - w1] - rc[indx - w3]);
                float S_Grad = S1 + SNabs + fabsf(rc[indx + w1] - rc[indx + w3]);
                float W_Grad = W1 + EWabs + fabsf(rc[indx - 1] - rc[indx - 3]);
                float E_Grad = E1 + EWabs + fabsf(rc[indx + 1] - rc[indx + 3]);
                float N_Est = rc[indx - w1] - rgb1mw1;
                float S_Est = rc[indx + w1] - rgb1pw1;
                float W_Est = rc[indx - 1] - rgb1m1;
                float E_Est = rc[indx + 1] - rgb1p1;
                float V_Est = (N_Grad * S_Est + S_Grad * N_Est) / (N_Grad + S_Grad);
                float H_Est = (E_Grad * W_Est + W_Grad * E_Est) / (E_Grad + W_Grad);
                rc[indx] = rgb1 + intpf(VH_Disc, H_Est, V_Est);
            }
        }
    }

    const int firstVertical = rowStart + RCD_BORDER;
    const int lastVertical = rowEnd - RCD_BORDER;
    const int firstHorizontal = colStart + RCD_BORDER;
    const int lastHorizontal = colEnd - RCD_BORDER;
    (void)numTh; (void)numTw; /* rcdBorder == tileBorder: outermost tiles use the same margin */
    for (int row = firstVertical; row < lastVertical; ++row) {
        for (int col = firstHorizontal; col < lastHorizontal; ++col) {
            int idx = (row - rowStart) * ts + col - colStart;
            red[(size_t)row * os + col] = std_maxf(0.f, rgb[0][idx] * scale);
            green[(size_t)row * os + col] = std_maxf(0.f, rgb[1][idx] * scale);
            blue[(size_t)row * os + col] = std_maxf(0.f, rgb[2][idx] * scale);
        }
    }
#undef FCT
}

int oracle_rcd_demosaic(const float *raw, size_t raw_stride, int W, int H, unsigned filters,
                        float *red, float *green, float *blue, size_t out_stride)
{
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++)
            if (fc(filters, i, j) == 3) return -1; /* reference falls back to igv_interpolate: out of scope */
    const int tileSizeN = RCD_TS - 2 * RCD_BORDER;
    const int numTh = H / tileSizeN + ((H % tileSizeN) ? 1 : 0);
    const int numTw = W / tileSizeN + ((W % tileSizeN) ? 1 : 0);
    int fail = 0;
#pragma omp parallel if (!oracle_rcd_stale)
    {
        float *work = (float *)calloc((size_t)RCD_TS * RCD_TS * 13 / 2, sizeof(float));
        if (!work) {
#pragma omp atomic write
            fail = 1;
        }
#pragma omp for schedule(dynamic, 2) collapse(2)
        for (int tr = 0; tr < numTh; ++tr)
            for (int tc = 0; tc < numTw; ++tc)
                if (work) oracle_rcd_tile(raw, raw_stride, W, H, filters, tr, tc, numTh, numTw, red, green, blue, out_stride, work);
        free(work);
    }
    if (fail) return -2;
    oracle_border_interpolate2(W, H, RCD_BORDER, raw, raw_stride, filters, red, green, blue, out_stride);
    return 0;
}
