/*
 * oracle/dninfo.c -- CPU restatement of the AUTOMATIC chrominance estimation:
 *   oracle_autodn_adjust          calcautodn_info with the constants the only caller passes
 *                                 (levaut 0, mode 1, lissage 0)       rtengine/ipdenoise.cc:66-206,1006-1013
 *   oracle_denoise_info_crop      RGB_denoise_info for one crop (isRAW) + WaveletDenoiseAll_info / ShrinkAll_info
 *                                                                     ipdenoise.cc:227-669, FTblockDN.cc:1227-1362
 *   oracle_denoise_compute_params ImProcFunctions::denoiseComputeParams: the nine crops and the reduction
 *                                                                     ipdenoise.cc:800-1093
 *
 * TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED as a whole: ipdenoise.cc / FTblockDN.cc do not compile here (glibmm,
 * fftw3); the leaves this file composes are pinned (xatan2f 4-lane and scalar: tests/golden/sleef2.npz, sleef3.npz;
 * LUTf, xexpf/xlogf, the wavelet decomposition: tests/golden/ npz files).
 *
 * Facts of the reference this file relies on:
 *  - Tile_calc ignores its arguments and returns one tile (FTblockDN.cc:442-480), so crW = widIm/2, crH = heiIm/2
 *    (ipdenoise.cc:875-876) and the inner tile loop of RGB_denoise_info runs once over the whole crop.
 *  - `sigma` and `sigma_L` (the running-mean deviations of ShrinkAll_info) are written into locals nobody reads
 *    (ipdenoise.cc:935,937); they are not restated.
 */
#include "oracle.h"
#include "oracle_common.h"
#include <stdlib.h>
#include <string.h>

float oracle_xyz2lab_f(const float *cachef, float f);

/* calcautodn_info, levaut = 0, mode = 1, lissage = 0.  *chaut in/out, returns delta. */
float oracle_autodn_adjust(float *chaut_io, int Nb, float maxmax, float lumema, float chromina, float redyel, float skinc,
                           float nsknc, int aggressive)
{
    const float reducdelta = aggressive ? (float)0.9 : 1.f;
    float chaut = *chaut_io;
    chaut = (chaut * Nb - maxmax) / (Nb - 1);
    if ((redyel > 5000.f || skinc > 1000.f) && nsknc < 0.4f && chromina > 3000.f) chaut *= 0.45f;
    else if ((redyel > 12000.f || skinc > 1200.f) && nsknc < 0.3f && chromina > 3000.f) chaut *= 0.3f;
    /* mode 1 */
    if (chromina > 10000.f) chaut *= 0.8f;
    else if (chromina > 6000.f) chaut *= 0.9f;
    else if (chromina < 3000.f) chaut *= 1.5f;      /* the `< 2000` arm after it can never be taken */
    if (lumema < 2500.f) chaut *= 1.2f;
    else if (lumema < 5000.f) chaut *= 1.1f;
    else if (lumema > 20000.f) chaut *= 0.9f;
    /* levaut 0 */
    if (chaut > 300.f) chaut = 0.714286f * chaut + 85.71428f;
    float delta = maxmax - chaut;
    delta *= reducdelta;
    /* lissage 0 */
    if (chaut < 200.f) {
        if (delta < 200.f) delta *= 0.95f;
        else if (delta < 400.f) delta *= 0.7f;
        else delta = 280.f;
    } else if (chaut < 400.f) {
        if (delta < 400.f) delta *= 0.6f;
        else delta = 200.f;
    } else if (chaut < 550.f) delta *= 0.3f;
    else if (chaut < 650.f) delta *= 0.2f;
    else delta *= 0.15f;
    if (chromina < 6000.f) delta *= 1.2f;
    if (lumema < 5000.f) delta *= 1.2f;
    *chaut_io = chaut;
    return delta;
}

/* One crop.  crop[3]: the getImage output (crW x crH, contiguous); mat: camera -> working (convertColorSpace); wp: the
 * working-space matrix as floats.  info[16] = {chaut, maxredaut, maxblueaut, minredaut, minblueaut, chromina, lumema,
 * redyel, skinc, nsknc, Nb, redaut, blueaut}. */
void oracle_denoise_info_crop(const float *const crop[3], int crW, int crH, const double mat[9], const float wp[9], double gamma,
                              int aggressive, float *info)
{
    static float *cachef = NULL;
    if (!cachef) { cachef = (float *)malloc(sizeof(float) * 65536); oracle_cachef(cachef); }
    const int wid = (crW + 1) / 2, hei = (crH + 1) / 2;
    const size_t n = (size_t)crW * crH, n2 = (size_t)wid * hei;
    float *hue = (float *)malloc(sizeof(float) * 3 * n2), *chrom = hue + n2, *lum = chrom + n2;
    float *pa = (float *)malloc(sizeof(float) * 2 * n), *pb = pa + n;

    /* provicalc -> Lab (ipdenoise.cc:902-911, 268-283), then the hue / chroma / luminance maps (L384-458) */
    const int nvec = 4 * (crW / 8);     /* half-res columns the 4-lane loop of L395-402 covers */
#pragma omp parallel for
    for (int ii = 0; ii < hei; ++ii)
        for (int jj = 0; jj < wid; ++jj) {
            const size_t o = (size_t)(2 * ii) * crW + 2 * jj;
            const double dr = crop[0][o], dg = crop[1][o], db = crop[2][o];
            const float RL = (float)(mat[0] * dr + mat[1] * dg + mat[2] * db);
            const float GL = (float)(mat[3] * dr + mat[4] * dg + mat[5] * db);
            const float BL = (float)(mat[6] * dr + mat[7] * dg + mat[8] * db);
            float L, a, b;
            oracle_rgb2lab(RL, GL, BL, &L, &a, &b, wp);
            const size_t k = (size_t)ii * wid + jj;
            hue[k] = oracle_xatan2f(b, a);              /* the 4-lane and the scalar xatan2f agree bit for bit (sleef3.npz) */
            float cN = sqrtf(a * a + b * b);
            if (jj < nvec) cN = sse_maxf(cN, 100.f); else if (cN < 100.f) cN = 100.f;
            chrom[k] = cN;
            float Ll = L < 2.f ? 2.f : L;
            Ll = Ll > 32768.f ? 32768.f : Ll;
            lum[k] = Ll;
        }

    /* gamma + YUV on the full crop (L460-482) */
    const float gam = (float)gamma, gamthresh = 0.001f;
    const float gamslope = exp(log((double)gamthresh) / gam) / gamthresh;
    float *gamcurve = (float *)malloc(sizeof(float) * 65536);
    oracle_gamma_lut(gamcurve, gam, gamthresh, gamslope, 65535.f, 32768.f);
    const double expcomp = logf(5.f) / logf(2.f);       /* L936 */
    const float gain = powf(2.0f, (float)expcomp);
#pragma omp parallel for
    for (int i = 0; i < crH; ++i)
        for (int j = 0; j < crW; ++j) {
            const size_t o = (size_t)i * crW + j;
            float X = gain * crop[0][o], Y = gain * crop[1][o], Z = gain * crop[2][o];
#define GAM(v) v = v < 65535.f ? oracle_lutf_noclip(gamcurve, 65536, v) : ((v / 65535.f <= gamthresh ? (v / 65535.f) * gamslope : oracle_xexpf_s(oracle_xlogf_s(v / 65535.f) / gam)) * 32768.f)
            GAM(X); GAM(Y); GAM(Z);
#undef GAM
            const float l = X * wp[3] + Y * wp[4] + Z * wp[5];
            pa[o] = X - l;      /* v -> labdn->a */
            pb[o] = l - Z;      /* u -> labdn->b */
        }
    free(gamcurve);

    /* levwav = max(2, 5 - ceil(log(1))) = 5 */
    oracle_wavelet *ad = oracle_wavelet_decompose(pa, crW, crH, 5), *bd = oracle_wavelet_decompose(pb, crW, crH, 5);
    float chau = 0.f, chred = 0.f, chblue = 0.f, maxchred = 0.f, maxchblue = 0.f, minchred = 100000000.f, minchblue = 100000000.f;
    float chaut = 0.f, redaut = 0.f, blueaut = 0.f, maxredaut = 0.f, maxblueaut = 0.f, minredaut = 0.f, minblueaut = 0.f;
    float chromina = 0.f, lumema = 0.f, redyel = 0.f, skinc = 0.f, nsknc = 0.f;
    int nb = 0;
    const float reduc = aggressive ? (float)0.9 : 1.f;
    for (int lvl = 0; lvl < 5; ++lvl) {
        if (lvl == 1) {
            float chro = 0.f, lume = 0.f, red_yel = 0.f, skin_c = 0.f;
            int nc = 0, nL = 0, nry = 0, nsk = 0;
            for (size_t k = 0; k < n2; ++k) {
                chro += chrom[k];
                ++nc;
                if (hue[k] > -0.8f && hue[k] < 2.0f && chrom[k] > 10000.f) { red_yel += chrom[k]; ++nry; }
                if (hue[k] > 0.f && hue[k] < 1.6f && chrom[k] < 10000.f) { skin_c += chrom[k]; ++nsk; }
                lume += lum[k];
                ++nL;
            }
            if (nc > 0) { chromina = chro / nc; nsknc = (float)nsk / (float)nc; } else nsknc = (float)nsk;
            if (nL > 0) lumema = lume / nL;
            if (nry > 0) redyel = red_yel / nry;
            if (nsk > 0) skinc = skin_c / nsk;
        }
        for (int dir = 1; dir < 4; ++dir) {
            float m = oracle_madrgb(ad->band[lvl][dir], (int)n2);
            const float mada = m * m;
            chred += mada;
            if (mada > maxchred) maxchred = mada;
            if (mada < minchred) minchred = mada;
            maxredaut = sqrtf(reduc * maxchred);
            minredaut = sqrtf(reduc * minchred);
            m = oracle_madrgb(bd->band[lvl][dir], (int)n2);
            const float madb = m * m;
            chblue += madb;
            if (madb > maxchblue) maxchblue = madb;
            if (madb < minchblue) minchblue = madb;
            maxblueaut = sqrtf(reduc * maxchblue);
            minblueaut = sqrtf(reduc * minchblue);
            chau += (mada + madb);
            ++nb;
            chaut = sqrtf(reduc * chau / (nb + nb));
            redaut = sqrtf(reduc * chred / nb);
            blueaut = sqrtf(reduc * chblue / nb);
        }
    }
    oracle_wavelet_free(ad); oracle_wavelet_free(bd);
    free(pa); free(hue);
    info[0] = chaut; info[1] = maxredaut; info[2] = maxblueaut; info[3] = minredaut; info[4] = minblueaut;
    info[5] = chromina; info[6] = lumema; info[7] = redyel; info[8] = skinc; info[9] = nsknc; info[10] = (float)nb;
    info[11] = redaut; info[12] = blueaut;
}

/* store_out[30] = {chrominance, chrominanceRedGreen, chrominanceBlueYellow, ch_M[9], max_r[9], max_b[9]};
 * info_out (nullable) = 9 x 16 per-crop values, crop k = hcr*3 + wcr.  Returns 0, or -1 when the image is too small for the
 * crop layout. */
int oracle_denoise_compute_params(const float *const planes[3], size_t ss, int W, int H, int border, const float mul[3], int do_clip,
                                  const double mat[9], const float wp[9], double gamma, int aggressive, float *store_out, float *info_out)
{
    const int widIm = W - 2 * border, heiIm = H - 2 * border;
    const int crW = widIm / 2, crH = heiIm / 2;
    if (crW < 16 || crH < 16 || widIm - crW - 50 < 0 || heiIm - crH - 50 < 0) return -1;
    const int coordW[3] = {50, widIm / 2 - crW / 2, widIm - crW - 50}, coordH[3] = {50, heiIm / 2 - crH / 2, heiIm - crH - 50};
    float info[9][16];
    memset(info, 0, sizeof info);
    float *crop = (float *)malloc(sizeof(float) * 3 * (size_t)crW * crH);
    float *const cp[3] = {crop, crop + (size_t)crW * crH, crop + 2 * (size_t)crW * crH};
    for (int wcr = 0; wcr <= 2; ++wcr)
        for (int hcr = 0; hcr <= 2; ++hcr) {
            oracle_get_image(planes, ss, coordW[wcr] + border, coordH[hcr] + border, cp, (size_t)crW, crW, crH, mul, do_clip);
            oracle_denoise_info_crop((const float *const *)cp, crW, crH, mat, wp, gamma, aggressive, info[hcr * 3 + wcr]);
        }
    free(crop);
    if (info_out) memcpy(info_out, info, sizeof info);

    /* the reduction, ipdenoise.cc:960-1072, with autoNR 10, autoNRmax 40, multip = adjustr = lowdenoise = 1 (raw) */
    float ch_M[9], max_r[9], max_b[9], Max_R[9], Max_B[9], Min_R[9], Min_B[9];
    const float nrmax = 40.f * 1.f * 1.f * 1.f;
    for (int k = 0; k < 9; ++k) {
        ch_M[k] = 1.0f * info[k][0]; max_r[k] = 1.0f * info[k][1]; max_b[k] = 1.0f * info[k][2];
        const float min_r = 1.0f * info[k][3], min_b = 1.0f * info[k][4];
        const float maxmax = rt_maxf(max_r[k], max_b[k]);
        const float delta = oracle_autodn_adjust(&ch_M[k], (int)info[k][10], maxmax, info[k][6], info[k][5], info[k][7], info[k][8], info[k][9], aggressive);
        if (max_r[k] > max_b[k]) {
            Max_R[k] = delta / (nrmax / 2.f);
            Min_B[k] = -(ch_M[k] - min_b) / nrmax;
            Max_B[k] = 0.f; Min_R[k] = 0.f;
        } else {
            Max_B[k] = delta / (nrmax / 2.f);
            Min_R[k] = -(ch_M[k] - min_r) / nrmax;
            Min_B[k] = 0.f; Max_R[k] = 0.f;
        }
    }
    float chM = 0.f, MaxR = 0.f, MaxB = 0.f, MinR = 100000000000.f, MinB = 100000000000.f;
    float MaxRMoy = 0.f, MaxBMoy = 0.f, MinRMoy = 0.f, MinBMoy = 0.f;
    for (int k = 0; k < 9; ++k) {
        chM += ch_M[k]; MaxBMoy += Max_B[k]; MaxRMoy += Max_R[k]; MinRMoy += Min_R[k]; MinBMoy += Min_B[k];
        if (Max_R[k] > MaxR) MaxR = Max_R[k];
        if (Max_B[k] > MaxB) MaxB = Max_B[k];
        if (Min_R[k] < MinR) MinR = Min_R[k];
        if (Min_B[k] < MinB) MinB = Min_B[k];
    }
    chM /= 9; MaxBMoy /= 9; MaxRMoy /= 9; MinBMoy /= 9; MinRMoy /= 9;
    float maxr, maxb;
    if (MaxR > MaxB) {
        maxr = MaxRMoy + (MaxR - MaxRMoy) * 0.66f;
        maxb = MinBMoy + (MinB - MinBMoy) * 0.66f;
    } else {
        maxb = MaxBMoy + (MaxB - MaxBMoy) * 0.66f;
        maxr = MinRMoy + (MinR - MinRMoy) * 0.66f;
    }
    store_out[0] = chM / (10.f * 1.f * 1.f);
    store_out[1] = maxr;
    store_out[2] = maxb;
    for (int k = 0; k < 9; ++k) { store_out[3 + k] = ch_M[k]; store_out[12 + k] = max_r[k]; store_out[21 + k] = max_b[k]; }
    return 0;
}
