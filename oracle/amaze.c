/*
 * oracle/amaze.c -- CPU restatement of RawImageSource::amaze_demosaic_RT, following the
 * reference's x86-64 (#ifdef __SSE2__) branches, which are what every x86-64 build runs
 * (reference: rtengine/amaze_demosaic_RT.cc:41-1595).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle_common.h).  PARITY UNPINNED: the reference has no
 * tests or golden vectors for this function and its translation unit cannot be compiled in
 * this image without stand-in headers (rtengine.h -> glibmm/lcms2), so this restatement is
 * checked by reading only.
 *
 * What is kept from the reference, because it is part of the numerical result:
 *  - tile grid 160x160, origin (-16,-16), stride 128, 16 px mirrored image border
 *    (amaze_demosaic_RT.cc:63,182-334);
 *  - the per-tile work arena: same plane order, same 128-byte gaps and the same aliasing
 *    (Dgrb on vcdalt, delp/nyquist2 on cddiffsq, delm/rbint, Dgrb2 on dgintv, pmwt on
 *    delhvsqsum, rbm/rbp on vcd; L124-174), so reads of aliased/stale positions give what
 *    the reference reads;
 *  - the vector loop shapes: 4-lane groups that run past the scalar loop bounds, in-place
 *    updates seen by later lanes/rows (hcd/vcd L540-583, hvwt L958-974, pmwt L1213-1223),
 *    byte-offset neighbour addressing of the nyquist map (L888-901), rbint[indx1 +- v1]
 *    (L1253-1260);
 *  - operation order of every fp32 expression (compile with -ffp-contract=off).
 *  The arena is zeroed at the start of every tile (the reference callocs it once per
 *  thread; positions a tile does not write keep values of the thread's previous tile,
 *  which no in-range output depends on -- tests/test_oracle_demosaic.py runs tiles in different
 *  orders on an uncleared arena to check that).
 */
#include "oracle.h"
#include "oracle_common.h"
#include <stdlib.h>

#define TS 160
#define TSH 80
#define GAP 32 /* 128-byte gap between planes, in floats */

typedef struct {
    float *rgbgreen, *delhvsqsum, *dirwts0, *dirwts1, *vcd, *hcd, *vcdalt, *hcdalt, *cddiffsq, *hvwt;
    float *Dgrb0, *Dgrb1, *delp, *delm, *rbint, *dgintv, *dginth, *Dgrbsq1m, *Dgrbsq1p, *cfa;
    float *pmwt, *rbm, *rbp, *nyqutest, *Dgrb2; /* Dgrb2: interleaved {h,v} pairs on dgintv */
    unsigned char *nyquist, *nyquist2;
} amaze_planes;

size_t oracle_amaze_arena_floats(void)
{
    /* 14*ts*ts floats + ts*tsh bytes + 18 gaps (amaze_demosaic_RT.cc:124), rounded up */
    return 14 * (size_t)TS * TS + (TS * TSH) / 4 + 18 * GAP + 16;
}

static void carve(float *base, amaze_planes *p)
{
    const size_t F = (size_t)TS * TS, Hh = (size_t)TS * TSH;
    p->rgbgreen = base;
    p->delhvsqsum = p->rgbgreen + F + GAP;
    p->dirwts0 = p->delhvsqsum + F + GAP;
    p->dirwts1 = p->dirwts0 + F + GAP;
    p->vcd = p->dirwts1 + F + GAP;
    p->hcd = p->vcd + F + GAP;
    p->vcdalt = p->hcd + F + GAP;
    p->hcdalt = p->vcdalt + F + GAP;
    p->cddiffsq = p->hcdalt + F + GAP;
    p->hvwt = p->cddiffsq + F + 2 * GAP;
    p->Dgrb0 = p->vcdalt;
    p->Dgrb1 = p->vcdalt + Hh;
    p->delp = p->cddiffsq;
    p->delm = p->delp + Hh + GAP;
    p->rbint = p->delm;
    p->dgintv = p->hvwt + Hh + GAP;
    p->Dgrb2 = p->dgintv;
    p->dginth = p->dgintv + F + GAP;
    p->Dgrbsq1m = p->dginth + F + GAP;
    p->Dgrbsq1p = p->Dgrbsq1m + Hh + GAP;
    p->cfa = p->Dgrbsq1p + Hh + GAP;
    p->pmwt = p->delhvsqsum;
    p->rbm = p->vcd;
    p->rbp = p->rbm + Hh + GAP;
    p->nyquist = (unsigned char *)(p->cfa + F + GAP);
    p->nyquist2 = (unsigned char *)p->cddiffsq;
    p->nyqutest = (float *)(p->nyquist + Hh + 4 * GAP);
}

int oracle_amaze_debug_stop = 0; /* test hook: return after phase N (0 = run all) */
/* test hook: tiles seen, tiles with doNyquist, nyquist flags, nyquist2 sites, P14 sites taken, P14 sites tested */
long long oracle_amaze_stats[6] = {0, 0, 0, 0, 0, 0};
#define STOP_AFTER(n) do { if (oracle_amaze_debug_stop == (n)) return; } while (0)

static inline int sat_add_i8(int a, int b) { int s = a + b; return s > 127 ? 127 : s; }

void oracle_amaze_tile(const float *raw, size_t rs, int width, int height, unsigned filters,
                       float clip_pt, float clip_pt8, int top, int left,
                       float *red, float *green, float *blue, size_t os,
                       float *arena, int poison)
{
    enum { ts = TS, tsh = TSH, v1 = TS, v2 = 2 * TS, v3 = 3 * TS,
           p1 = -TS + 1, p2 = -2 * TS + 2, p3 = -3 * TS + 3, m1 = TS + 1, m2 = 2 * TS + 2, m3 = 3 * TS + 3 };
    const float eps = 1e-5f, epssq = 1e-10f, arthresh = 0.75f;
    const float gaussodd[4] = {0.14659727707323927f, 0.103592713382435f, 0.0732036125103057f, 0.0365543548389495f};
    const float nyqthresh = 0.5f;
    const float gaussgrad[6] = {nyqthresh * 0.07384411893421103f, nyqthresh * 0.06207511968171489f, nyqthresh * 0.0521818194747806f,
                                nyqthresh * 0.03687419286733595f, nyqthresh * 0.03099732204057846f, nyqthresh * 0.018413194161458882f};
    const float gausseven[2] = {0.13719494435797422f, 0.05640252782101291f};
    const float gquinc[4] = {0.169917f, 0.108947f, 0.069855f, 0.0287182f};
    const int winx = 0, winy = 0;
#define FCT(r, c) fc(filters, (unsigned)(r), (unsigned)(c))
#define RAW(r, c) raw[(size_t)(r) * rs + (c)]

    int ex, ey;
    if (FCT(0, 0) == 1) {
        if (FCT(0, 1) == 0) { ey = 0; ex = 1; } else { ey = 1; ex = 0; }
    } else {
        if (FCT(0, 0) == 0) { ey = 0; ex = 0; } else { ey = 1; ex = 1; }
    }

    amaze_planes P;
    const size_t nfl = oracle_amaze_arena_floats();
    if (!poison) memset(arena, 0, nfl * sizeof(float));
    /* poison != 0 (test hook): keep whatever the previous tile left in the arena, which is
       what a reference thread sees on every tile after its first */
    carve(arena, &P);
    float *cfa = P.cfa, *rgbgreen = P.rgbgreen;
    /* the reference clears these nyquist rows at the start of every tile (L184) */
    memset(&P.nyquist[3 * tsh], 0, (size_t)(ts - 6) * tsh);

    const int bottom = (top + ts < winy + height + 16) ? top + ts : winy + height + 16;
    const int right = (left + ts < winx + width + 16) ? left + ts : winx + width + 16;
    const int rr1 = bottom - top, cc1 = right - left;
    const int rrmin = top < winy ? 16 : 0;
    const int ccmin = left < winx ? 16 : 0;
    const int rrmax = bottom > (winy + height) ? winy + height - top : rr1;
    const int ccmax = right > (winx + width) ? winx + width - left : cc1;

    /* ---- tile initialisation (L205-334) ---- */
#define SETCFA(i, v) do { float t_ = (v) / 65535.f; cfa[i] = t_; rgbgreen[i] = t_; } while (0)
    if (rrmin > 0)
        for (int rr = 0; rr < 16; rr++)
            for (int cc = ccmin, row = 32 - rr + top; cc < ccmax; cc++) SETCFA(rr * ts + cc, RAW(row, cc + left));
    for (int rr = rrmin; rr < rrmax; rr++)
        for (int cc = ccmin, row = rr + top; cc < ccmax; cc++) SETCFA(rr * ts + cc, RAW(row, cc + left));
    if (rrmax < rr1)
        for (int rr = 0; rr < 16; rr++)
            for (int cc = ccmin; cc < ccmax; cc++) SETCFA((rrmax + rr) * ts + cc, RAW(winy + height - rr - 2, left + cc));
    if (ccmin > 0)
        for (int rr = rrmin; rr < rrmax; rr++)
            for (int cc = 0, row = rr + top; cc < 16; cc++) SETCFA(rr * ts + cc, RAW(row, 32 - cc + left));
    if (ccmax < cc1)
        for (int rr = rrmin; rr < rrmax; rr++)
            for (int cc = 0; cc < 16; cc++) SETCFA(rr * ts + ccmax + cc, RAW(top + rr, winx + width - cc - 2));
    if (rrmin > 0 && ccmin > 0)
        for (int rr = 0; rr < 16; rr++)
            for (int cc = 0; cc < 16; cc++) SETCFA(rr * ts + cc, RAW(winy + 32 - rr, winx + 32 - cc));
    if (rrmax < rr1 && ccmax < cc1)
        for (int rr = 0; rr < 16; rr++)
            for (int cc = 0; cc < 16; cc++) SETCFA((rrmax + rr) * ts + ccmax + cc, RAW(winy + height - rr - 2, winx + width - cc - 2));
    if (rrmin > 0 && ccmax < cc1)
        for (int rr = 0; rr < 16; rr++)
            for (int cc = 0; cc < 16; cc++) SETCFA(rr * ts + ccmax + cc, RAW(winy + 32 - rr, winx + width - cc - 2));
    if (rrmax < rr1 && ccmin > 0)
        for (int rr = 0; rr < 16; rr++)
            for (int cc = 0; cc < 16; cc++) SETCFA((rrmax + rr) * ts + cc, RAW(winy + height - rr - 2, winx + 32 - cc));
#undef SETCFA

    /* ---- P1: horizontal/vertical gradients (L342-351); 4-lane groups over [0, cc1) ---- */
    for (int rr = 2; rr < rr1 - 2; rr++)
        for (int i0 = rr * ts; i0 < rr * ts + cc1; i0 += 4)
            for (int k = 0; k < 4; ++k) {
                int i = i0 + k;
                float delh = fabsf(cfa[i + 1] - cfa[i - 1]);
                float delv = fabsf(cfa[i + v1] - cfa[i - v1]);
                P.dirwts1[i] = eps + fabsf(cfa[i + 2] - cfa[i]) + fabsf(cfa[i] - cfa[i - 2]) + delh;
                P.dirwts0[i] = eps + fabsf(cfa[i + v2] - cfa[i]) + fabsf(cfa[i] - cfa[i - v2]) + delv;
                P.delhvsqsum[i] = sqrf(delh) + sqrf(delv);
            }

    STOP_AFTER(1);
    /* ---- P2: colour differences, vertical/horizontal (L380-434) ---- */
    for (int rr = 4; rr < rr1 - 4; rr++)
        for (int i0 = rr * ts + 4; i0 < rr * ts + cc1 - 7; i0 += 4)
            for (int k = 0; k < 4; ++k) {
                int i = i0 + k;
                const float sgn = (FCT(rr, i - rr * ts) & 1) ? -1.f : 1.f;
                const float *d0 = P.dirwts0, *d1 = P.dirwts1;
                float cfav = cfa[i];
                float cru = cfa[i - v1] * (d0[i - v2] + d0[i]) / (d0[i - v2] * (eps + cfav) + d0[i] * (eps + cfa[i - v2]));
                float crd = cfa[i + v1] * (d0[i + v2] + d0[i]) / (d0[i + v2] * (eps + cfav) + d0[i] * (eps + cfa[i + v2]));
                float crl = cfa[i - 1] * (d1[i - 2] + d1[i]) / (d1[i - 2] * (eps + cfav) + d1[i] * (eps + cfa[i - 2]));
                float crr = cfa[i + 1] * (d1[i + 2] + d1[i]) / (d1[i + 2] * (eps + cfav) + d1[i] * (eps + cfa[i + 2]));
                float guha = cfa[i - v1] + 0.5f * (cfav - cfa[i - v2]);
                float gdha = cfa[i + v1] + 0.5f * (cfav - cfa[i + v2]);
                float glha = cfa[i - 1] + 0.5f * (cfav - cfa[i - 2]);
                float grha = cfa[i + 1] + 0.5f * (cfav - cfa[i + 2]);
                float guar = fabsf(1.f - cru) < arthresh ? cfav * cru : guha;
                float gdar = fabsf(1.f - crd) < arthresh ? cfav * crd : gdha;
                float glar = fabsf(1.f - crl) < arthresh ? cfav * crl : glha;
                float grar = fabsf(1.f - crr) < arthresh ? cfav * crr : grha;
                float hwt = d1[i - 1] / (d1[i - 1] + d1[i + 1]);
                float vwt = d0[i - v1] / (d0[i + v1] + d0[i - v1]);
                float Ginthha = intpf(hwt, grha, glha);
                float Gintvha = intpf(vwt, gdha, guha);
                float hcdaltv = sgn * (Ginthha - cfav);
                float vcdaltv = sgn * (Gintvha - cfav);
                P.hcdalt[i] = hcdaltv;
                P.vcdalt[i] = vcdaltv;
                int clip = (cfav > clip_pt8) || (Gintvha > clip_pt8) || (Ginthha > clip_pt8);
                if (clip) { guar = guha; gdar = gdha; glar = glha; grar = grha; }
                P.vcd[i] = clip ? vcdaltv : sgn * (intpf(vwt, gdar, guar) - cfav);
                P.hcd[i] = clip ? hcdaltv : sgn * (intpf(hwt, grar, glar) - cfav);
                P.dgintv[i] = sse_minf(sqrf(guha - gdha), sqrf(guar - gdar));
                P.dginth[i] = sse_minf(sqrf(glha - grha), sqrf(glar - grar));
            }

    STOP_AFTER(2);
    /* ---- P3: variance-based choice + highlight bounding, IN PLACE (L540-583).
       A 4-lane group loads hcd[i-2..i+5] and vcd[i-v2], vcd[i], vcd[i+v2] before it stores,
       so lanes 0,1 see the previous group's updated hcd, every lane sees row rr-2's
       updated vcd. ---- */
    for (int rr = 4; rr < rr1 - 4; rr++)
        for (int i0 = rr * ts + 4; i0 < rr * ts + cc1 - 4; i0 += 4) {
            float nh[4], nv[4], nd[4];
            for (int k = 0; k < 4; ++k) {
                int i = i0 + k;
                const float sgn = (FCT(rr, i - rr * ts) & 1) ? -1.f : 1.f;
                const float nsgn = -sgn, sgn3 = sgn + sgn + sgn;
                float hcdv = P.hcd[i];
                float hcdvar = sqrf(P.hcd[i - 2] - hcdv) + sqrf(P.hcd[i - 2] - P.hcd[i + 2]) + sqrf(hcdv - P.hcd[i + 2]);
                float hcdaltv = P.hcdalt[i];
                float hcdaltvar = sqrf(P.hcdalt[i - 2] - hcdaltv) + sqrf(P.hcdalt[i - 2] - P.hcdalt[i + 2]) + sqrf(hcdaltv - P.hcdalt[i + 2]);
                float vcdv = P.vcd[i];
                float vcdvar = sqrf(P.vcd[i - v2] - vcdv) + sqrf(P.vcd[i - v2] - P.vcd[i + v2]) + sqrf(vcdv - P.vcd[i + v2]);
                float vcdaltv = P.vcdalt[i];
                float vcdaltvar = sqrf(P.vcdalt[i - v2] - vcdaltv) + sqrf(P.vcdalt[i - v2] - P.vcdalt[i + v2]) + sqrf(vcdaltv - P.vcdalt[i + v2]);
                hcdv = hcdaltvar < hcdvar ? hcdaltv : hcdv;
                vcdv = vcdaltvar < vcdvar ? vcdaltv : vcdv;

                float c = cfa[i];
                float Ginth = sgn * hcdv + c;
                float temp2 = sgn3 * hcdv;
                float hwt = 1.f + temp2 / (eps + Ginth + c);
                int hmask = (nsgn * hcdv) > 0.f;
                float hold = hcdv;
                float temp = nsgn * (c - median3_sse(Ginth, cfa[i - 1], cfa[i + 1]));
                hcdv = (temp2 < -(c + Ginth)) ? temp : intpf(hwt, hcdv, temp);
                hcdv = hmask ? hcdv : hold;
                hcdv = (Ginth > clip_pt) ? temp : hcdv;

                float Gintv = sgn * vcdv + c;
                temp2 = sgn3 * vcdv;
                float vwt = 1.f + temp2 / (eps + Gintv + c);
                int vmask = (nsgn * vcdv) > 0.f;
                float vold = vcdv;
                temp = nsgn * (c - median3_sse(Gintv, cfa[i - v1], cfa[i + v1]));
                vcdv = (temp2 < -(c + Gintv)) ? temp : intpf(vwt, vcdv, temp);
                vcdv = vmask ? vcdv : vold;
                vcdv = (Gintv > clip_pt) ? temp : vcdv;
                nh[k] = hcdv; nv[k] = vcdv; nd[k] = sqrf(vcdv - hcdv);
            }
            for (int k = 0; k < 4; ++k) { P.hcd[i0 + k] = nh[k]; P.vcd[i0 + k] = nv[k]; P.cddiffsq[i0 + k] = nd[k]; }
        }

    STOP_AFTER(3);
    /* ---- P4: h/v interpolation weight at R/B sites (L680-728); 4 sites per group ---- */
    for (int rr = 6; rr < rr1 - 6; rr++)
        for (int i0 = rr * ts + 6 + (FCT(rr, 2) & 1); i0 < rr * ts + cc1 - 6; i0 += 8)
            for (int k = 0; k < 4; ++k) {
                int i = i0 + 2 * k;
                const float *vcd = P.vcd, *hcd = P.hcd, *d0 = P.dirwts0, *d1 = P.dirwts1;
                float t = vcd[i];
                float uave = t + vcd[i - v1] + vcd[i - v2] + vcd[i - v3];
                float dave = t + vcd[i + v1] + vcd[i + v2] + vcd[i + v3];
                float Dvu = sqrf(t - uave) + sqrf(vcd[i - v1] - uave) + sqrf(vcd[i - v2] - uave) + sqrf(vcd[i - v3] - uave);
                float Dvd = sqrf(t - dave) + sqrf(vcd[i + v1] - dave) + sqrf(vcd[i + v2] - dave) + sqrf(vcd[i + v3] - dave);
                float hwt = d1[i - 1] / (d1[i - 1] + d1[i + 1]);
                float vwt = d0[i - v1] / (d0[i - v1] + d0[i + v1]);
                t = hcd[i];
                float lave = t + (hcd[i - 3] + hcd[i - 2]) + hcd[i - 1];
                float rave = t + (hcd[i + 1] + hcd[i + 2]) + hcd[i + 3];
                float Dhl = sqrf(t - lave) + sqrf(hcd[i - 1] - lave) + sqrf(hcd[i - 2] - lave) + sqrf(hcd[i - 3] - lave);
                float Dhr = sqrf(t - rave) + sqrf(hcd[i + 1] - rave) + sqrf(hcd[i + 2] - rave) + sqrf(hcd[i + 3] - rave);
                float vcdvar = epssq + intpf(vwt, Dvd, Dvu);
                float hcdvar = epssq + intpf(hwt, Dhr, Dhl);
                Dvu = P.dgintv[i - v1] + P.dgintv[i - v2];
                Dvd = P.dgintv[i + v1] + P.dgintv[i + v2];
                Dhl = P.dginth[i - 2] + P.dginth[i - 1];
                Dhr = P.dginth[i + 1] + P.dginth[i + 2];
                float vcdvar1 = epssq + P.dgintv[i] + intpf(vwt, Dvd, Dvu);
                float hcdvar1 = epssq + P.dginth[i] + intpf(hwt, Dhr, Dhl);
                float varwt = hcdvar / (vcdvar + hcdvar);
                float diffwt = hcdvar1 / (vcdvar1 + hcdvar1);
                int dec = ((0.5f - varwt) * (0.5f - diffwt) > 0.f) && (fabsf(0.5f - diffwt) < fabsf(0.5f - varwt));
                P.hvwt[i >> 1] = dec ? varwt : diffwt;
            }

    STOP_AFTER(4);
    /* ---- P5: nyquist test value (L746-803): vector groups, then scalar tail ---- */
    for (int rr = 6; rr < rr1 - 6; rr++) {
        int cc = 6 + (FCT(rr, 2) & 1);
        int i0 = rr * ts + cc;
        const float *c = P.cddiffsq, *d = P.delhvsqsum;
        for (; cc < cc1 - 7; cc += 8, i0 += 8)
            for (int k = 0; k < 4; ++k) {
                int i = i0 + 2 * k;
                P.nyqutest[i >> 1] =
                    (gaussodd[0] * c[i] +
                     gaussodd[1] * (c[i - m1] + c[i + p1] + c[i - p1] + c[i + m1]) +
                     gaussodd[2] * (c[i - v2] + c[i - 2] + c[i + 2] + c[i + v2]) +
                     gaussodd[3] * (c[i - m2] + c[i + p2] + c[i - p2] + c[i + m2])) -
                    (gaussgrad[0] * d[i] +
                     gaussgrad[1] * (d[i - v1] + d[i - 1] + d[i + 1] + d[i + v1]) +
                     gaussgrad[2] * (d[i - m1] + d[i + p1] + d[i - p1] + d[i + m1]) +
                     gaussgrad[3] * (d[i - v2] + d[i - 2] + d[i + 2] + d[i + v2]) +
                     gaussgrad[4] * (d[i - v2 - 1] + d[i - v2 + 1] + d[i - ts - 2] + d[i - ts + 2] +
                                     d[i + ts - 2] + d[i + ts + 2] + d[i + v2 - 1] + d[i + v2 + 1]) +
                     gaussgrad[5] * (d[i - m2] + d[i + p2] + d[i - p2] + d[i + m2]));
            }
        for (; cc < cc1 - 6; cc += 2, i0 += 2) {
            int i = i0;
            P.nyqutest[i >> 1] =
                (gaussodd[0] * c[i] +
                 gaussodd[1] * (c[i - m1] + c[i + p1] + c[i - p1] + c[i + m1]) +
                 gaussodd[2] * (c[i - v2] + c[i - 2] + c[i + 2] + c[i + v2]) +
                 gaussodd[3] * (c[i - m2] + c[i + p2] + c[i - p2] + c[i + m2])) -
                (gaussgrad[0] * d[i] +
                 gaussgrad[1] * (d[i - v1] + d[i + 1] + d[i - 1] + d[i + v1]) +
                 gaussgrad[2] * (d[i - m1] + d[i + p1] + d[i - p1] + d[i + m1]) +
                 gaussgrad[3] * (d[i - v2] + d[i - 2] + d[i + 2] + d[i + v2]) +
                 gaussgrad[4] * (d[i - v2 - 1] + d[i - v2 + 1] + d[i - ts - 2] + d[i - ts + 2] +
                                 d[i + ts - 2] + d[i + ts + 2] + d[i + v2 - 1] + d[i + v2 + 1]) +
                 gaussgrad[5] * (d[i - m2] + d[i + p2] + d[i - p2] + d[i + m2]));
        }
    }

    STOP_AFTER(5);
    /* ---- P6: nyquist flags + bounding box (L806-825) ---- */
    int nystartrow = 0, nyendrow = 0, nystartcol = ts + 1, nyendcol = 0;
    for (int rr = 6; rr < rr1 - 6; rr++)
        for (int cc = 6 + (FCT(rr, 2) & 1), i = rr * ts + cc; cc < cc1 - 6; cc += 2, i += 2)
            if (P.nyqutest[i >> 1] > 0.f) {
                P.nyquist[i >> 1] = 1;
                nystartrow = nystartrow ? nystartrow : rr;
                nyendrow = rr;
                nystartcol = nystartcol > cc ? cc : nystartcol;
                nyendcol = nyendcol < cc ? cc : nyendcol;
            }
    const int doNyquist = nystartrow != nyendrow && nystartcol != nyendcol;
    {
        long long nfl_ = 0;
        for (int rr = 6; rr < rr1 - 6; rr++)
            for (int cc = 6 + (FCT(rr, 2) & 1), i = rr * ts + cc; cc < cc1 - 6; cc += 2, i += 2) nfl_ += P.nyquist[i >> 1];
#pragma omp atomic
        oracle_amaze_stats[0] += 1;
#pragma omp atomic
        oracle_amaze_stats[1] += doNyquist;
#pragma omp atomic
        oracle_amaze_stats[2] += nfl_;
    }

    if (doNyquist) {
        nyendrow++;
        nyendcol++;
        nystartcol -= (nystartcol & 1);
        nystartrow = nystartrow > 8 ? nystartrow : 8;
        nyendrow = nyendrow < rr1 - 8 ? nyendrow : rr1 - 8;
        nystartcol = nystartcol > 8 ? nystartcol : 8;
        nyendcol = nyendcol < cc1 - 8 ? nyendcol : cc1 - 8;
        memset(&P.nyquist2[4 * tsh], 0, (size_t)(ts - 8) * tsh);

        /* ---- P7: majority vote on the nyquist map, 16-byte vectors with byte offsets
           that do not depend on the row's site parity (L888-901) ---- */
        for (int rr = nystartrow; rr < nyendrow; rr++)
            for (int i0 = rr * ts; i0 < rr * ts + cc1; i0 += 32)
                for (int k = 0; k < 16; ++k) {
                    int b = (i0 >> 1) + k;
                    const unsigned char *n = P.nyquist;
                    int t1 = sat_add_i8(n[((i0 - v2) >> 1) + k], n[((i0 - m1) >> 1) + k]);
                    int t2 = sat_add_i8(n[((i0 + p1) >> 1) + k], n[((i0 - 2) >> 1) + k]);
                    int t3 = sat_add_i8(n[((i0 + 2) >> 1) + k], n[((i0 - p1) >> 1) + k]);
                    int t4 = sat_add_i8(n[((i0 + m1) >> 1) + k], n[((i0 + v2) >> 1) + k]);
                    t1 = sat_add_i8(t1, t3);
                    t2 = sat_add_i8(t2, t4);
                    t1 = sat_add_i8(t1, t2);
                    unsigned char val = n[b];
                    if (t1 > 4) val = 1;
                    if (t1 < 4) val = 0;
                    P.nyquist2[b] = val;
                }

        /* ---- P8: area interpolation in nyquist regions (L914-951) ---- */
        for (int rr = nystartrow; rr < nyendrow; rr++)
            for (int i = rr * ts + nystartcol + (FCT(rr, 2) & 1); i < rr * ts + nyendcol; i += 2)
                if (P.nyquist2[i >> 1]) {
#pragma omp atomic
                    oracle_amaze_stats[3] += 1;
                    float sumcfa = 0.f, sumh = 0.f, sumv = 0.f, sumsqh = 0.f, sumsqv = 0.f, areawt = 0.f;
                    for (int a = -6; a < 7; a += 2) {
                        int i1 = i + (a * ts) - 6;
                        for (int b = -6; b < 7; b += 2, i1 += 2)
                            if (P.nyquist2[i1 >> 1]) {
                                float ct = cfa[i1];
                                sumcfa += ct;
                                sumh += (cfa[i1 - 1] + cfa[i1 + 1]);
                                sumv += (cfa[i1 - v1] + cfa[i1 + v1]);
                                sumsqh += sqrf(ct - cfa[i1 - 1]) + sqrf(ct - cfa[i1 + 1]);
                                sumsqv += sqrf(ct - cfa[i1 - v1]) + sqrf(ct - cfa[i1 + v1]);
                                areawt += 1;
                            }
                    }
                    sumh = sumcfa - xdiv2f(sumh);
                    sumv = sumcfa - xdiv2f(sumv);
                    areawt = xdiv2f(areawt);
                    float hcdvar = epssq + fabsf(areawt * sumsqh - sumh * sumh);
                    float vcdvar = epssq + fabsf(areawt * sumsqv - sumv * sumv);
                    P.hvwt[i >> 1] = hcdvar / (vcdvar + hcdvar);
                }
    }

    STOP_AFTER(8);
    /* ---- P9: G at R/B sites; hvwt refined IN PLACE, row rr reads updated row rr-1 (L957-974) ---- */
    for (int rr = 8; rr < rr1 - 8; rr++)
        for (int i = rr * ts + 8 + (FCT(rr, 2) & 1); i < rr * ts + cc1 - 8; i += 2) {
            float *hvwt = P.hvwt;
            float hvwtalt = xdivf(hvwt[(i - m1) >> 1] + hvwt[(i + p1) >> 1] + hvwt[(i - p1) >> 1] + hvwt[(i + m1) >> 1], 2);
            hvwt[i >> 1] = fabsf(0.5f - hvwt[i >> 1]) < fabsf(0.5f - hvwtalt) ? hvwtalt : hvwt[i >> 1];
            P.Dgrb0[i >> 1] = intpf(hvwt[i >> 1], P.vcd[i], P.hcd[i]);
            rgbgreen[i] = cfa[i] + P.Dgrb0[i >> 1];
            P.Dgrb2[2 * (i >> 1)] = P.nyquist2[i >> 1] ? sqrf(rgbgreen[i] - xdiv2f(rgbgreen[i - 1] + rgbgreen[i + 1])) : 0.f;
            P.Dgrb2[2 * (i >> 1) + 1] = P.nyquist2[i >> 1] ? sqrf(rgbgreen[i] - xdiv2f(rgbgreen[i - v1] + rgbgreen[i + v1])) : 0.f;
        }

    STOP_AFTER(9);
    /* ---- P10: refine nyquist areas using G curvature (L979-999) ---- */
    if (doNyquist)
        for (int rr = nystartrow; rr < nyendrow; rr++)
            for (int i = rr * ts + nystartcol + (FCT(rr, 2) & 1); i < rr * ts + nyendcol; i += 2)
                if (P.nyquist2[i >> 1]) {
#define DH(j) P.Dgrb2[2 * ((j) >> 1)]
#define DV(j) P.Dgrb2[2 * ((j) >> 1) + 1]
                    float gvarh = epssq + (gquinc[0] * DH(i) +
                                           gquinc[1] * (DH(i - m1) + DH(i + p1) + DH(i - p1) + DH(i + m1)) +
                                           gquinc[2] * (DH(i - v2) + DH(i - 2) + DH(i + 2) + DH(i + v2)) +
                                           gquinc[3] * (DH(i - m2) + DH(i + p2) + DH(i - p2) + DH(i + m2)));
                    float gvarv = epssq + (gquinc[0] * DV(i) +
                                           gquinc[1] * (DV(i - m1) + DV(i + p1) + DV(i - p1) + DV(i + m1)) +
                                           gquinc[2] * (DV(i - v2) + DV(i - 2) + DV(i + 2) + DV(i + v2)) +
                                           gquinc[3] * (DV(i - m2) + DV(i + p2) + DV(i - p2) + DV(i + m2)));
#undef DH
#undef DV
                    P.Dgrb0[i >> 1] = (P.hcd[i] * gvarv + P.vcd[i] * gvarh) / (gvarv + gvarh);
                    rgbgreen[i] = cfa[i] + P.Dgrb0[i >> 1];
                }

    STOP_AFTER(10);
    /* ---- P11: diagonal gradients (L1004-1027): pairs (even col i, i+1), 4 pairs per group ---- */
    for (int rr = 6; rr < rr1 - 6; rr++) {
        const int rbEven = (FCT(rr, 2) & 1) == 0; /* even columns are R/B sites */
        for (int cc = 6, i0 = rr * ts + cc; cc < cc1 - 6; cc += 8, i0 += 8)
            for (int k = 0; k < 4; ++k) {
                int i = i0 + 2 * k;
                if (rbEven) {
                    float t = cfa[i + 1];
                    float sp = sqrf(t - cfa[i + 1 - p1]) + sqrf(t - cfa[i + 1 + p1]);
                    P.delp[i >> 1] = fabsf(cfa[i + p1] - cfa[i - p1]);
                    P.delm[i >> 1] = fabsf(cfa[i + m1] - cfa[i - m1]);
                    float sm = sqrf(t - cfa[i + 1 - m1]) + sqrf(t - cfa[i + 1 + m1]);
                    P.Dgrbsq1m[i >> 1] = sm;
                    P.Dgrbsq1p[i >> 1] = sp;
                } else {
                    float t = cfa[i];
                    float sp = sqrf(t - cfa[i - p1]) + sqrf(t - cfa[i + p1]);
                    P.delp[i >> 1] = fabsf(cfa[i + 1 + p1] - cfa[i + 1 - p1]);
                    P.delm[i >> 1] = fabsf(cfa[i + 1 + m1] - cfa[i + 1 - m1]);
                    float sm = sqrf(t - cfa[i - m1]) + sqrf(t - cfa[i + m1]);
                    P.Dgrbsq1m[i >> 1] = sm;
                    P.Dgrbsq1p[i >> 1] = sp;
                }
            }
    }

    STOP_AFTER(11);
    /* ---- P12: diagonal interpolation of R+B, plus/minus weights (L1061-1121) ---- */
    for (int rr = 8; rr < rr1 - 8; rr++)
        for (int i0 = rr * ts + 8 + (FCT(rr, 2) & 1); i0 < rr * ts + cc1 - 8; i0 += 8)
            for (int k = 0; k < 4; ++k) {
                int i = i0 + 2 * k, i1 = i >> 1;
                const float *delm = P.delm, *delp = P.delp, *Dm = P.Dgrbsq1m, *Dp = P.Dgrbsq1p;
                float cfav = cfa[i];
                float t1 = cfa[i + m1], t2 = cfa[i + m2];
                float rbse = (t1 + t1) / (eps + cfav + t2);
                rbse = fabsf(1.f - rbse) < arthresh ? cfav * rbse : t1 + 0.5f * (cfav - t2);
                t1 = cfa[i - m1]; t2 = cfa[i - m2];
                float rbnw = (t1 + t1) / (eps + cfav + t2);
                rbnw = fabsf(1.f - rbnw) < arthresh ? cfav * rbnw : t1 + 0.5f * (cfav - t2);
                t1 = eps + delm[i1];
                float wtse = t1 + delm[(i + m1) >> 1] + delm[(i + m2) >> 1];
                float wtnw = t1 + delm[(i - m1) >> 1] + delm[(i - m2) >> 1];
                float rbmv = (wtse * rbnw + wtnw * rbse) / (wtse + wtnw);
                t1 = median3_sse(rbmv, cfa[i - m1], cfa[i + m1]);
                float wt = ((cfav - rbmv) + (cfav - rbmv)) / (eps + rbmv + cfav);
                t2 = intpf(wt, rbmv, t1);
                t2 = (rbmv + rbmv < cfav) ? t1 : t2;
                t2 = (rbmv < cfav) ? t2 : rbmv;
                float rbm_out = (t2 > clip_pt) ? median3_sse(t2, cfa[i - m1], cfa[i + m1]) : t2;

                t1 = cfa[i + p1]; t2 = cfa[i + p2];
                float rbne = (t1 + t1) / (eps + cfav + t2);
                rbne = fabsf(1.f - rbne) < arthresh ? cfav * rbne : t1 + 0.5f * (cfav - t2);
                t1 = cfa[i - p1]; t2 = cfa[i - p2];
                float rbsw = (t1 + t1) / (eps + cfav + t2);
                rbsw = fabsf(1.f - rbsw) < arthresh ? cfav * rbsw : t1 + 0.5f * (cfav - t2);
                t1 = eps + delp[i1];
                float wtne = t1 + delp[(i + p1) >> 1] + delp[(i + p2) >> 1];
                float wtsw = t1 + delp[(i - p1) >> 1] + delp[(i - p2) >> 1];
                float rbpv = (wtne * rbsw + wtsw * rbne) / (wtne + wtsw);
                t1 = median3_sse(rbpv, cfa[i - p1], cfa[i + p1]);
                wt = ((cfav - rbpv) + (cfav - rbpv)) / (eps + rbpv + cfav);
                t2 = intpf(wt, rbpv, t1);
                t2 = (rbpv + rbpv < cfav) ? t1 : t2;
                t2 = (rbpv < cfav) ? t2 : rbpv;
                float rbp_out = (t2 > clip_pt) ? median3_sse(t2, cfa[i - p1], cfa[i + p1]) : t2;

                float rbvarm = epssq + (gausseven[0] * (Dm[(i - v1) >> 1] + Dm[(i - 1) >> 1] + Dm[(i + 1) >> 1] + Dm[(i + v1) >> 1]) +
                                        gausseven[1] * (Dm[(i - v2 - 1) >> 1] + Dm[(i - v2 + 1) >> 1] + Dm[(i - 2 - v1) >> 1] + Dm[(i + 2 - v1) >> 1] +
                                                        Dm[(i - 2 + v1) >> 1] + Dm[(i + 2 + v1) >> 1] + Dm[(i + v2 - 1) >> 1] + Dm[(i + v2 + 1) >> 1]));
                float rbvarp = epssq + (gausseven[0] * (Dp[(i - v1) >> 1] + Dp[(i - 1) >> 1] + Dp[(i + 1) >> 1] + Dp[(i + v1) >> 1]) +
                                        gausseven[1] * (Dp[(i - v2 - 1) >> 1] + Dp[(i - v2 + 1) >> 1] + Dp[(i - 2 - v1) >> 1] + Dp[(i + 2 - v1) >> 1] +
                                                        Dp[(i - 2 + v1) >> 1] + Dp[(i + 2 + v1) >> 1] + Dp[(i + v2 - 1) >> 1] + Dp[(i + v2 + 1) >> 1]));
                /* stores come last: rbm/rbp alias vcd, pmwt aliases delhvsqsum (not read here) */
                P.rbm[i1] = rbm_out;
                P.rbp[i1] = rbp_out;
                P.pmwt[i1] = rbvarm / (rbvarp + rbvarm);
            }

    STOP_AFTER(12);
    /* ---- P13: pmwt refined IN PLACE (row rr reads updated row rr-1), rbint (L1213-1223) ---- */
    for (int rr = 10; rr < rr1 - 10; rr++)
        for (int i0 = rr * ts + 10 + (FCT(rr, 2) & 1); i0 < rr * ts + cc1 - 10; i0 += 8) {
            float nt[4];
            for (int k = 0; k < 4; ++k) {
                int i = i0 + 2 * k, i1 = i >> 1;
                float alt = 0.25f * (P.pmwt[(i - m1) >> 1] + P.pmwt[(i + p1) >> 1] + P.pmwt[(i - p1) >> 1] + P.pmwt[(i + m1) >> 1]);
                float t = P.pmwt[i1];
                nt[k] = fabsf(0.5f - t) < fabsf(0.5f - alt) ? alt : t;
            }
            for (int k = 0; k < 4; ++k) {
                int i = i0 + 2 * k, i1 = i >> 1;
                P.pmwt[i1] = nt[k];
                P.rbint[i1] = 0.5f * (cfa[i] + intpf(nt[k], P.rbp[i1], P.rbm[i1]));
            }
        }

    STOP_AFTER(13);
    /* ---- P14: G re-interpolated from R+B where the diagonal weight is more decisive (L1241-1297) ---- */
    for (int rr = 12; rr < rr1 - 12; rr++)
        for (int i0 = rr * ts + 12 + (FCT(rr, 2) & 1); i0 < rr * ts + cc1 - 12; i0 += 8)
            for (int k = 0; k < 4; ++k) {
                int i = i0 + 2 * k, i1 = i >> 1;
                const float *rbint = P.rbint, *d0 = P.dirwts0, *d1 = P.dirwts1;
#pragma omp atomic
                oracle_amaze_stats[5] += 1;
                if (!(fabsf(0.5f - P.pmwt[i1]) >= fabsf(0.5f - P.hvwt[i1]))) continue;
#pragma omp atomic
                oracle_amaze_stats[4] += 1;
                float rb = rbint[i1];
                float cru = (cfa[i - v1] + cfa[i - v1]) / (eps + rb + rbint[i1 - v1]);
                float gu = rb * cru;
                float gu2 = cfa[i - v1] + 0.5f * (rb - rbint[i1 - v1]);
                gu = fabsf(1.f - cru) < arthresh ? gu : gu2;
                float crd = (cfa[i + v1] + cfa[i + v1]) / (eps + rb + rbint[i1 + v1]);
                float gd = rb * crd;
                float gd2 = cfa[i + v1] + 0.5f * (rb - rbint[i1 + v1]);
                gd = fabsf(1.f - crd) < arthresh ? gd : gd2;
                float Gintv = (d0[i - v1] * gd + d0[i + v1] * gu) / (d0[i + v1] + d0[i - v1]);
                float G1 = median3_sse(Gintv, cfa[i - v1], cfa[i + v1]);
                float vwt = ((rb - Gintv) + (rb - Gintv)) / (eps + Gintv + rb);
                float G2 = intpf(vwt, Gintv, G1);
                G1 = ((Gintv + Gintv) < rb) ? G1 : G2;
                Gintv = (Gintv < rb) ? G1 : Gintv;
                Gintv = (Gintv > clip_pt) ? median3_sse(Gintv, cfa[i - v1], cfa[i + v1]) : Gintv;

                float crl = (cfa[i - 1] + cfa[i - 1]) / (eps + rb + rbint[i1 - 1]);
                float gl = rb * crl;
                float gl2 = cfa[i - 1] + 0.5f * (rb - rbint[i1 - 1]);
                gl = fabsf(1.f - crl) < arthresh ? gl : gl2;
                float crr = (cfa[i + 1] + cfa[i + 1]) / (eps + rb + rbint[i1 + 1]);
                float gr = rb * crr;
                float gr2 = cfa[i + 1] + 0.5f * (rb - rbint[i1 + 1]);
                gr = fabsf(1.f - crr) < arthresh ? gr : gr2;
                float Ginth = (d1[i - 1] * gr + d1[i + 1] * gl) / (d1[i - 1] + d1[i + 1]);
                float H1 = median3_sse(Ginth, cfa[i - 1], cfa[i + 1]);
                float hwt = ((rb - Ginth) + (rb - Ginth)) / (eps + Ginth + rb);
                float H2 = intpf(hwt, Ginth, H1);
                H1 = ((Ginth + Ginth) < rb) ? H1 : H2;
                Ginth = (Ginth < rb) ? H1 : Ginth;
                Ginth = (Ginth > clip_pt) ? median3_sse(Ginth, cfa[i - 1], cfa[i + 1]) : Ginth;

                float g = intpf(P.hvwt[i1], Gintv, Ginth);
                rgbgreen[i] = g;
                P.Dgrb0[i1] = g - cfa[i];
            }

    STOP_AFTER(14);
    /* ---- P15: split G-B out of G-R on the B rows (L1381-1386) ---- */
    for (int rr = 13 - ey; rr < rr1 - 12; rr += 2)
        for (int i1 = (rr * ts + 13 - ex) >> 1; i1 < (rr * ts + cc1 - 12) >> 1; i1++) {
            P.Dgrb1[i1] = P.Dgrb0[i1];
            P.Dgrb0[i1] = 0;
        }

    STOP_AFTER(15);
    /* ---- P16: chrominance at the opposite-colour sites (L1394-1408) ---- */
    for (int rr = 14; rr < rr1 - 14; rr++) {
        int cc = 14 + (FCT(rr, 2) & 1);
        const int c = 1 - FCT(rr, cc) / 2;
        float *D = c ? P.Dgrb1 : P.Dgrb0;
        for (int i0 = rr * ts + cc; cc < cc1 - 14; cc += 8, i0 += 8) {
            float out[4];
            for (int k = 0; k < 4; ++k) {
                int i = i0 + 2 * k;
#define DG(j) D[(j) >> 1]
                float temp = eps + fabsf(DG(i - m1) - DG(i + m1));
                float temp2 = eps + fabsf(DG(i + p1) - DG(i - p1));
                float wtnw = 1.f / (temp + fabsf(DG(i - m1) - DG(i - m3)) + fabsf(DG(i + m1) - DG(i - m3)));
                float wtne = 1.f / (temp2 + fabsf(DG(i + p1) - DG(i + p3)) + fabsf(DG(i - p1) - DG(i + p3)));
                float wtsw = 1.f / (temp2 + fabsf(DG(i - p1) - DG(i + m3)) + fabsf(DG(i + p1) - DG(i - p3)));
                float wtse = 1.f / (temp + fabsf(DG(i + m1) - DG(i - p3)) + fabsf(DG(i - m1) - DG(i + m3)));
                out[k] = (wtnw * (1.325f * DG(i - m1) - 0.175f * DG(i - m3) - 0.075f * (DG(i - m1 - 2) + DG(i - m1 - v2))) +
                          wtne * (1.325f * DG(i + p1) - 0.175f * DG(i + p3) - 0.075f * (DG(i + p1 + 2) + DG(i + p1 + v2))) +
                          wtsw * (1.325f * DG(i - p1) - 0.175f * DG(i - p3) - 0.075f * (DG(i - p1 - 2) + DG(i - p1 - v2))) +
                          wtse * (1.325f * DG(i + m1) - 0.175f * DG(i + m3) - 0.075f * (DG(i + m1 + 2) + DG(i + m1 + v2)))) /
                         (wtnw + wtne + wtsw + wtse);
#undef DG
            }
            for (int k = 0; k < 4; ++k) D[(i0 + 2 * k) >> 1] = out[k];
        }
    }

    STOP_AFTER(16);
    /* ---- P17: R and B (L1441-1548), P18: G (L1551-1565) written for [16,rr1-16)x[16,cc1-16) ---- */
    for (int rr = 16; rr < rr1 - 16; rr++) {
        const int row = rr + top;
        for (int cc = 16; cc < cc1 - 16; cc++) {
            const int i = rr * ts + cc, col = left + cc;
            const float *hvwt = P.hvwt;
            float r, b;
            if (FCT(rr, cc) & 1) { /* G site */
                float h_up = hvwt[(i - v1) >> 1], h_dn = hvwt[(i + v1) >> 1], h_r = hvwt[(i + 1) >> 1], h_l = hvwt[(i - 1) >> 1];
                float temp = 1.f / (h_up + 2.f - h_r - h_l + h_dn);
                r = rgbgreen[i] - (h_up * P.Dgrb0[(i - v1) >> 1] + (1.f - h_r) * P.Dgrb0[(i + 1) >> 1] + (1.f - h_l) * P.Dgrb0[(i - 1) >> 1] + h_dn * P.Dgrb0[(i + v1) >> 1]) * temp;
                b = rgbgreen[i] - (h_up * P.Dgrb1[(i - v1) >> 1] + (1.f - h_r) * P.Dgrb1[(i + 1) >> 1] + (1.f - h_l) * P.Dgrb1[(i - 1) >> 1] + h_dn * P.Dgrb1[(i + v1) >> 1]) * temp;
            } else {
                r = rgbgreen[i] - P.Dgrb0[i >> 1];
                b = rgbgreen[i] - P.Dgrb1[i >> 1];
            }
            red[(size_t)row * os + col] = sse_maxf(65535.f * r, 0.f);
            blue[(size_t)row * os + col] = sse_maxf(65535.f * b, 0.f);
            green[(size_t)row * os + col] = sse_maxf(rgbgreen[i] * 65535.f, 0.f);
        }
    }
#undef FCT
#undef RAW
}

int oracle_amaze_demosaic(const float *raw, size_t raw_stride, int W, int H, unsigned filters,
                          double initialGain, int border,
                          float *red, float *green, float *blue, size_t out_stride)
{
    const float clip_pt = 1.0 / initialGain;
    const float clip_pt8 = 0.8 / initialGain;
    const int nty = (H + 16 + (TS - 32) - 1) / (TS - 32); /* tops: -16, 112, ... < H */
    const int ntx = (W + 16 + (TS - 32) - 1) / (TS - 32);
    int fail = 0;
#pragma omp parallel
    {
        float *arena = (float *)malloc(oracle_amaze_arena_floats() * sizeof(float));
        if (!arena) {
#pragma omp atomic write
            fail = 1;
        }
#pragma omp for schedule(dynamic) collapse(2)
        for (int ty = 0; ty < nty; ++ty)
            for (int tx = 0; tx < ntx; ++tx)
                if (arena)
                    oracle_amaze_tile(raw, raw_stride, W, H, filters, clip_pt, clip_pt8,
                                      -16 + ty * (TS - 32), -16 + tx * (TS - 32),
                                      red, green, blue, out_stride, arena, 0);
        free(arena);
    }
    if (fail) return -2;
    if (border < 4) oracle_border_interpolate2(W, H, 3, raw, raw_stride, filters, red, green, blue, out_stride);
    return 0;
}
