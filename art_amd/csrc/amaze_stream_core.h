// art_amd/csrc/amaze_stream_core.h -- AMaZE v2: the tile state lives in LDS.
//
// Replaces RawImageSource::amaze_demosaic_RT (reference: rtengine/amaze_demosaic_RT.cc:41-1595) for FULL tiles
// (160x160, the reference's tile grid).  One 1024-thread workgroup walks a tile top to bottom ONCE, two rows per step.
// Every phase of the reference runs a fixed number of rows ("off") behind the load front on ring buffers in LDS, so the
// CFA is read once from memory, R/G/B are written once, and no intermediate plane ever leaves the CU.
//
//   step t, sub-step a:  load rows (2t,2t+1) | P2 @6 | P4 @14 | P5+P6 @12 | P8 @22 (window rows -6..0) | P10/P14/P15 site classification @30 | P12 @24
//   ---- LDS barrier ----
//   step t, sub-step b:  P1 @2 | P3 @8 | P7 @14 | P8 @22 (window rows 2..6) | P9 @26 | P11 @20 | P13 @26 | P10, P14 at the listed sites @30 | P16 @36 | output @40 | P8 site list @20
//   ---- LDS barrier ----
//
// A stage at offset "off" handles tile rows (2t-off, 2t-off+1).  A stage of sub-step b may read what sub-step a of the SAME
// step wrote; everything else reads rows completed in earlier steps, so the two barriers per step are the only
// synchronisation.  The three row recurrences of the reference (vcd L540-583: stride 2, so the two rows of a step are
// independent; hvwt L958-974 and pmwt L1213-1223: one wave walks the two rows in order) keep their exact order.
//
// What the arena formulation (amaze.hip, v1) got for free from the reference's plane layout is reproduced explicitly:
//   * positions the reference never writes but reads as 0 (cleared per tile): columns 0-3/156-159 and rows 2,3,156,157 of
//     vcd/hcd/vcdalt/hcdalt, the nyquist flag bytes without a site, nyquist2 rows 2-7 and 152-157;
//   * pmwt aliases delhvsqsum (L169): P13's over-run sites read pmwt half-columns 76..79 that P12 never writes, i.e.
//     delhvsqsum(rr/2, (rr&1)*80 + 76..79) -- kept in a side table when P1 produces those rows;
//   * nyquist2 aliases cddiffsq and its memset (L879) stops at row 155: rows 156,157 are the bytes of cddiffsq(19, 80..119);
//   * Dgrb2 aliases dgintv, Dgrb1 aliases vcdalt: only positions outside what any output depends on.
// The Nyquist bounding box (L806-876) is tile-global and only known after the whole tile went through P6.  The stream
// assumes "every nyquist2 site lies inside the box"; it records the box and the extent of the sites it processed, and a
// tile where that does not hold is re-done by the arena kernel (exact fallback, amaze.hip).
//
// The same source compiles for the device (amaze_stream.hip) and, with AMZ_EMUL, as a sequential CPU emulation used by
// tests/test_amaze_stream_emul.py to check the schedule against the oracle without a GPU (test harness only).
#pragma once

#ifdef AMZ_EMUL
#include <math.h>
#include <stdint.h>
#include <string.h>
#define AMZ_DEV static inline
typedef float *amz_lf;
typedef unsigned char *amz_lb;
typedef int *amz_li;
typedef unsigned short *amz_ls;
typedef char *amz_lc;
typedef const float *amz_gcf;
typedef float *amz_gf;
namespace amz {
static inline unsigned fc(unsigned filters, unsigned row, unsigned col) { return (filters >> (((((row) << 1) & 14u) + ((col) & 1u)) << 1)) & 3u; }
static inline float sse_min(float x, float y) { return x < y ? x : y; }
static inline float sse_max(float x, float y) { return x > y ? x : y; }
static inline float sqr(float x) { return x * x; }
static inline float intp(float a, float b, float c) { return a * b + (1.f - a) * c; }
static inline float median3(float a, float b, float c) { return sse_max(sse_min(a, b), sse_min(c, sse_max(a, b))); }
static inline float xdiv2f(float d) { int32_t i; memcpy(&i, &d, 4); if (i & 0x7FFFFFFF) i -= 1 << 23; memcpy(&d, &i, 4); return d; }
static inline float xdivf(float d, int n) { int32_t i; memcpy(&i, &d, 4); if (i & 0x7FFFFFFF) i -= n << 23; memcpy(&d, &i, 4); return d; }
static inline int ngroups(int start, int bound, int step) { return bound > start ? (bound - start + step - 1) / step : 0; }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
}
#else
#include <hip/hip_runtime.h>
#include "devmath.h"
#define AMZ_DEV __device__ __forceinline__
typedef __attribute__((address_space(3))) float *amz_lf;
typedef __attribute__((address_space(3))) unsigned char *amz_lb;
typedef __attribute__((address_space(3))) int *amz_li;
typedef __attribute__((address_space(3))) unsigned short *amz_ls;
typedef __attribute__((address_space(3))) char *amz_lc;
typedef const __attribute__((address_space(1))) float *amz_gcf;
typedef __attribute__((address_space(1))) float *amz_gf;
namespace amz {
using artgpu::fc; using artgpu::sse_min; using artgpu::sse_max; using artgpu::sqr; using artgpu::intp; using artgpu::median3;
using artgpu::xdiv2f; using artgpu::xdivf; using artgpu::ngroups;
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
}
#endif

namespace amz {

constexpr int TS = 160, TSH = 80;
constexpr float eps = 1e-5f, epssq = 1e-10f, arthresh = 0.75f;

// ---------------------------------------------------------------------------------------------------------------------
// Ring buffers: name, depth in rows, row stride in floats.  depth >= (offset of the last reader + its upward reach)
// - (offset of the writer) + 2; the emulation checks every read against a row tag (AMZ_EMUL).
// ---------------------------------------------------------------------------------------------------------------------
#define AMZ_RINGS(X)                                                                                                     \
    X(CFA, 42, 160)  /* load @0 .. output @40 */                                                                       \
    X(D0, 8, 160)    /* dirwts0: P1 @2 -> P2 @6 (+-2 rows) */                                                          \
    X(D1, 6, 160)    /* dirwts1: P1 @2 -> P2 @6 */                                                                     \
    X(DHV, 14, 160)  /* delhvsqsum: P1 @2 -> P5 @12 (+-2) */                                                           \
    X(HCA, 4, 160)   /* hcdalt: P2 @6 -> P3 @8 */                                                                      \
    X(VCA, 6, 160)   /* vcdalt: P2 @6 -> P3 @8 (+-2) */                                                                \
    X(VCD, 14, 160)  /* vcd: P2 @6, updated in place by P3 @8 -> P4 @14 (+-3) */                                       \
    X(HCO, 4, 160)   /* hcd as P2 leaves it -> P3 @8 */                                                                \
    X(HCN, 8, 160)   /* hcd after P3 @8 -> P4 @14 */                                                                   \
    X(CDD, 8, 160)   /* cddiffsq: P3 @8 -> P5 @12 (+-2) */                                                             \
    X(DGV, 12, 160)  /* dgintv: P2 @6 -> P4 @14 (+-2) */                                                               \
    X(DGH, 10, 160)  /* dginth: P2 @6 -> P4 @14 */                                                                     \
    X(HVW, 10, 160)  /* hwt [idx], vwt [80+idx] at R/B sites: P2 @6 -> P4 @14 */                                       \
    X(VH, 18, 160)   /* vcd [idx], hcd [80+idx] at R/B sites: P4 @14 -> P9 @26, P10 @30 */                             \
    X(HVWT, 30, 80)  /* hvwt: P4 @14, P8 @22, P9 @26 in place -> P14 @30, output @40 (+-1) */                          \
    X(NYQ, 6, 20)    /* nyquist flag bytes: P5 @12 -> P7 @14 (+-2) */                                                  \
    X(NYQ2, 16, 20)  /* nyquist2 bytes: P7 @14 -> P8 @22 (+-6), P9 @24, P10 @28 */                                     \
    X(DG2, 8, 160)   /* Dgrb2 {h,v} pairs: P9 @24 -> P10 @28 (+-2) */                                                  \
    X(DG0, 18, 80)   /* Dgrb[0]: P9 @26, P10/P14/P15 @30, P16 @36 -> output @40 (+-1) */                               \
    X(DG1, 14, 80)   /* Dgrb[1]: P15 @30, P16 @36 -> output @40 (+-1) */                                               \
    X(RGBG, 16, 80)  /* rgbgreen at R/B sites: P9 @26, P10/P14 @30 -> output @40 */                                    \
    X(DELP, 8, 80)   /* P11 @18 -> P12 @22 (+-2) */                                                                    \
    X(DELM, 8, 80)                                                                                                      \
    X(DM, 8, 80)     /* Dgrbsq1m */                                                                                     \
    X(DP, 8, 80)     /* Dgrbsq1p */                                                                                     \
    X(RBM, 4, 80)    /* P12 @22 -> P13 @24 */                                                                          \
    X(RBP, 4, 80)                                                                                                       \
    X(PMWT, 8, 80)   /* P12 @22, P13 @24 in place -> P14 @28 */                                                        \
    X(RBINT, 8, 80)  /* P13 @24 -> P14 @28 (+-2 rows) */

enum RingId {
#define X(n, d, s) R_##n,
    AMZ_RINGS(X)
#undef X
    R_COUNT
};
#define X(n, d, s) constexpr int n##_D = d, n##_S = s;
AMZ_RINGS(X)
#undef X
// float offsets of the rings inside the workgroup's LDS block
struct RingLayout {
    int off[R_COUNT + 1];
    constexpr RingLayout() : off()
    {
        int o = 0, k = 0;
#define X(n, d, s) off[k++] = o; o += d * s;
        AMZ_RINGS(X)
#undef X
        off[k] = o;
    }
};
constexpr RingLayout LAYOUT{};
#define X(n, d, s) constexpr int n##_OFF = LAYOUT.off[R_##n];
AMZ_RINGS(X)
#undef X
constexpr int SIDE_OFF = LAYOUT.off[R_COUNT];   // delhvsqsum(row 4..75, cols 76..79 and 156..159): what pmwt[.., 76..79] aliases
constexpr int SIDE_FLOATS = 72 * 8;
constexpr int NQA_OFF = SIDE_OFF + SIDE_FLOATS;  // cddiffsq(row 19, cols 80..119): the bytes nyquist2 rows 156,157 alias (never memset, L879)
constexpr int NQA_FLOATS = 40;
constexpr int LIST_OFF = NQA_OFF + NQA_FLOATS;   // P8 site list of one step (row << 8 | col, 16 bits each), at most 144 entries
constexpr int LIST_INTS = 80;                    // two lists: the one step t consumes is rebuilt (for step t+2) only after step t
constexpr int RED_OFF = LIST_OFF + 2 * LIST_INTS;    // int words: [0..3] flag box (min row, max row, min col, max col), [4..7] extent of the
constexpr int RED_INTS = 24;                     // nyquist2 sites P8 processed; [8..15]: the same for the other tile in flight; [16], [17] list counts
constexpr int STG_OFF = RED_OFF + RED_INTS;      // CFA staging: the raw values of the next step's two rows, written by LDS-DMA (two buffers)
constexpr int STG_ROW = 192, STG_FLOATS = 2 * 2 * STG_ROW;
constexpr int HOT_OFF = STG_OFF + STG_FLOATS;      // P10 / P14 site lists of one step: [0] n10, [1] n14, [2..161] list10, [162..321] list14
constexpr int HOT_INTS = 2 + 2 * 160;
constexpr int DYN_OFF = HOT_OFF + HOT_INTS;      // the tiles of the workgroup's sequence: four slots of eight words, position k in slot k & 3 (back, front,
constexpr int DYN_SLOT = 8;                      // next and the one being pulled): [0] k, [1] tile (-1: none), [2] top, [3] left, [4] rows, [5] Nyquist box,
constexpr int DYN_INTS = 4 * DYN_SLOT + 4;       // [6] second attempt; word 32: the tile counter still has tiles
// Ring positions (round 6).  Every ring depth is even and a step starts at an even global row G0 = 2 T, so the rows of a PAIR
// (G0 - 2 j, G0 - 2 j + 1) are neighbours in every ring and never straddle its wrap-around.  Where pair j of ring n lives in this step --
// the byte address of its first row -- is one entry of a table in LDS; the table of step T + 1 is written during step T by the lanes of
// the lightest column role (each lane owns one entry and advances it by one pair slot with a compare and a select), two tables alternate.
// A row pointer is then ONE table word plus a literal (second row of the pair: + one row stride) instead of eight scalar instructions
// for (gbase + row) % depth * stride + offset -- those were ~90 % of the kernel's scalar instruction stream (1.05e9 per 45 MP frame).
struct PosTable {
    int first[R_COUNT + 1];        // first entry of ring n; entry first[n] + j = pair j of ring n, 0 <= j < depth / 2
    short h[192], j[192];          // per entry: pairs in its ring, its pair distance
    int stride2[192], off4[192];   // bytes per pair, byte offset of the ring
    constexpr PosTable() : first(), h(), j(), stride2(), off4()
    {
        int e = 0, k = 0;
#define X(n, d, s) first[k++] = e; for (int i = 0; i < d / 2; ++i) { h[e] = d / 2; j[e] = (short)i; stride2[e] = 2 * s * 4; off4[e] = n##_OFF * 4; ++e; }
        AMZ_RINGS(X)
#undef X
        first[k] = e;
        for (; e < 192; ++e) { h[e] = 1; j[e] = 0; stride2[e] = 0; off4[e] = 0; }
    }
};
constexpr PosTable POSTAB{};
constexpr int POS_PAIRS = POSTAB.first[R_COUNT];
static_assert(POS_PAIRS <= 192, "one entry per lane of the producing column role");
constexpr int POS_OFF = DYN_OFF + DYN_INTS;      // two tables of POS_PAIRS words (byte addresses)
constexpr int LDS_FLOATS = POS_OFF + 2 * POS_PAIRS;
static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");

constexpr int NTHREADS = 1024;
constexpr int LAST_OFF = 40;


#ifdef AMZ_EMUL
// row tags: which tile row a ring slot holds (emulation only)
struct Tags { int row[R_COUNT][64]; long long errors; int first_ring, first_want, first_have; };
static Tags g_tags;
static inline void tag_write(int ring, int depth, int row) { g_tags.row[ring][(unsigned)row % depth] = row; }
static inline void tag_read(int ring, int depth, int row)
{
    const int have = g_tags.row[ring][(unsigned)row % depth];
    if (have != row) {
        if (!g_tags.errors) { g_tags.first_ring = ring; g_tags.first_want = row; g_tags.first_have = have; }
        g_tags.errors++;
    }
}
#define AMZ_TAGW(n, row) amz::tag_write(amz::R_##n, amz::n##_D, (row))
#define AMZ_TAGR(n, row) amz::tag_read(amz::R_##n, amz::n##_D, (row))
#else
#define AMZ_TAGW(n, row) ((void)0)
#define AMZ_TAGR(n, row) ((void)0)
#endif
// the workgroup streams a SEQUENCE of tiles: tile k of the sequence occupies the global rows 160 k .. 160 k + 159 of the rings
#define AMZ_GB (a.gbase)
// row pointers: W = this stage writes the row, R = it reads a row that must hold exactly that tile row,
// X = it reads a row whose content cannot reach an output (no check)
#define AMZ_ROWP(n, row) amz::ring_ptr<amz::R_##n, amz::n##_D, amz::n##_S, amz::n##_OFF>(lds, a, (row))
#define ROWW(n, row) (AMZ_TAGW(n, AMZ_GB + (row)), AMZ_ROWP(n, row))
#define ROWR(n, row) (AMZ_TAGR(n, AMZ_GB + (row)), AMZ_ROWP(n, row))
#define ROWX(n, row) (AMZ_ROWP(n, row))
// the same for a row r + k + s where r is uniform over the wave and s (0 or 1) differs per lane: with r + k even the two candidates are the
// rows of one pair (one table word, the lane adds s row strides), with r + k odd they are the last row of one pair and the first of the next
#define AMZ_SROWP(n, r, k, s) amz::ring_ptr_s<amz::R_##n, amz::n##_D, amz::n##_S, amz::n##_OFF>(lds, a, (r) + (k), (s))
#define SROWW(n, r, k, s) (AMZ_TAGW(n, AMZ_GB + (r) + (k) + (s)), AMZ_SROWP(n, r, k, s))
#define SROWR(n, r, k, s) (AMZ_TAGR(n, AMZ_GB + (r) + (k) + (s)), AMZ_SROWP(n, r, k, s))
#define SROWX(n, r, k, s) (AMZ_SROWP(n, r, k, s))

struct TileArgs {
    amz_gcf raw;        // CFA plane
    long rs;            // row stride in floats
    amz_gf red, green, blue;
    long os;
    int top, left;      // tile origin in the frame (may be negative: mirrored border, L205-334)
    int rr1;            // rows of the tile (160, less at the bottom edge of the frame; 0: no tile)
    int gbase;          // 160 * (position of the tile in the workgroup's sequence): ring rows are gbase + tile row
    int rbase;          // the tile row the step starts at: gbase + rbase = g0 (negative rows and rows beyond the tile are fine); a stage at
                        // offset off gets r = rbase - off, so that (its rows) - rbase are literal constants
    int g0;             // 2 T, the global row the step starts at (even)
    int pos;            // byte offset of this step's ring position table
    int ny_box;         // the Nyquist box P7 / P8 / P10 work in (L827-876), [8, rr1-8) x [8, 152) on the first attempt, one byte per bound
                        // (r0 | r1 << 8 | c0 << 16 | c1 << 24: every value the kernel keeps per tile in flight is a scalar register it does not have)
    int W, H;
    unsigned filters;
    float clip_pt, clip_pt8;
    int g00;            // 1 if (0,0) is a green site
    int ey;             // row parity of the red sites (L1381: ey, ex)
};

AMZ_DEV int ny_pack(int r0, int r1, int c0, int c1) { return r0 | (r1 << 8) | (c0 << 16) | (c1 << 24); }
AMZ_DEV int ny_r0(const TileArgs &a) { return a.ny_box & 255; }
AMZ_DEV int ny_r1(const TileArgs &a) { return (a.ny_box >> 8) & 255; }
AMZ_DEV int ny_c0(const TileArgs &a) { return (a.ny_box >> 16) & 255; }
AMZ_DEV int ny_c1(const TileArgs &a) { return (int)((unsigned)a.ny_box >> 24); }

// pointer to tile row `row` of a ring (depth D rows of S floats at OFF) -- see "Ring positions" at the LDS layout
template <int RING, int D, int S, int OFF>
AMZ_DEV amz_lf ring_ptr(amz_lf lds, const TileArgs &a, int row)
{
    static_assert(D % 2 == 0, "ring depths are even: the two rows of a step are neighbours in every ring");
    const int kc = row - a.rbase;                  // a literal once the stage is inlined (rows are r + const, r = rbase - off): <= 1
    const int j = (-(kc >> 1)) % (D / 2);          // pair distance from the step's first row, modulo the ring's pairs
    const int byte = *(amz_li)((amz_lc)lds + a.pos + 4 * (POSTAB.first[RING] + j)) + (kc & 1) * (S * 4);
#ifdef AMZ_EMUL
    if (kc > 1 || byte != OFF * 4 + (int)((unsigned)(a.gbase + row) % (unsigned)D) * S * 4) {
        if (!g_tags.errors) { g_tags.first_ring = 100 + RING; g_tags.first_want = a.gbase + row; g_tags.first_have = byte; }
        g_tags.errors++;
        return lds + OFF + (int)((unsigned)(a.gbase + row) % (unsigned)D) * S;
    }
#endif
    return (amz_lf)((amz_lc)lds + byte);
}
template <int RING, int D, int S, int OFF>
AMZ_DEV amz_lf ring_ptr_s(amz_lf lds, const TileArgs &a, int rk, int s)
{
    const int kc = rk - a.rbase;
    if ((kc & 1) == 0) return ring_ptr<RING, D, S, OFF>(lds, a, rk) + s * S;
    amz_lf p0 = ring_ptr<RING, D, S, OFF>(lds, a, rk), p1 = ring_ptr<RING, D, S, OFF>(lds, a, rk + 1);
    return s ? p1 : p0;
}

// parity helpers: site (r,c) is green iff ((r + c) & 1) ^ g00; the R/B sites of row r are the columns cc = par(r) mod 2
AMZ_DEV int row_par(const TileArgs &a, int r) { return (r & 1) ^ a.g00; }
AMZ_DEV bool is_green(const TileArgs &a, int r, int c) { return (((r + c) & 1) ^ a.g00) != 0; }

// highlight bounding of a colour difference (amaze_demosaic_RT.cc:555-581)
AMZ_DEV float bound_cd(float cdv, float sgn, float c, float nA, float nB, float clip_pt)
{
    const float nsgn = -sgn, sgn3 = sgn + sgn + sgn;
    const float Gint = sgn * cdv + c;
    const float temp2 = sgn3 * cdv;
    const float wt = 1.f + temp2 / (eps + Gint + c);
    const bool mask = (nsgn * cdv) > 0.f;
    const float old = cdv;
    const float temp = nsgn * (c - median3(Gint, nA, nB));
    cdv = (temp2 < -(c + Gint)) ? temp : intp(wt, cdv, temp);
    cdv = mask ? cdv : old;
    cdv = (Gint > clip_pt) ? temp : cdv;
    return cdv;
}
AMZ_DEV float var3(float a, float b, float c) { return sqr(a - b) + sqr(a - c) + sqr(b - c); }
AMZ_DEV float rb_ratio(float cfav, float t1, float t2)
{
    float r = (t1 + t1) / (eps + cfav + t2);
    return fabsf(1.f - r) < arthresh ? cfav * r : t1 + 0.5f * (cfav - t2);
}
AMZ_DEV float rb_bound(float rbv, float cfav, float nA, float nB, float clip_pt)
{
    float t1 = median3(rbv, nA, nB);
    float wt = ((cfav - rbv) + (cfav - rbv)) / (eps + rbv + cfav);
    float t2 = intp(wt, rbv, t1);
    t2 = (rbv + rbv < cfav) ? t1 : t2;
    t2 = (rbv < cfav) ? t2 : rbv;
    return (t2 > clip_pt) ? median3(t2, nA, nB) : t2;
}
AMZ_DEV float g_dir(float rb, float cn, float rbn)
{
    float cr = (cn + cn) / (eps + rb + rbn);
    float g = rb * cr;
    float g2 = cn + 0.5f * (rb - rbn);
    return fabsf(1.f - cr) < arthresh ? g : g2;
}
AMZ_DEV float g_bound(float Gint, float rb, float nA, float nB, float clip_pt)
{
    float G1 = median3(Gint, nA, nB);
    float wt = ((rb - Gint) + (rb - Gint)) / (eps + Gint + rb);
    float G2 = intp(wt, Gint, G1);
    G1 = ((Gint + Gint) < rb) ? G1 : G2;
    Gint = (Gint < rb) ? G1 : Gint;
    return (Gint > clip_pt) ? median3(Gint, nA, nB) : Gint;
}

// ---------------------------------------------------------------------------------------------------------------------
// source pixel of tile position (rr, cc): the reference's mirrored 16-pixel image border (L205-334)
// ---------------------------------------------------------------------------------------------------------------------
// (the four corner blocks use 32 - rr / 32 - cc WITHOUT the tile origin, L282-334, the edge strips 32 - rr + top)
AMZ_DEV int src_row(const TileArgs &a, int rr, int cc)
{
    const int y = a.top + rr, x = a.left + cc;
    if (y < 0) return (x < 0 || x >= a.W) ? 32 - rr : 32 - rr + a.top;
    if (y >= a.H) return 2 * a.H - 2 - y;      // below the frame: H - (rr - rrmax) - 2
    return y;
}
AMZ_DEV int src_col(const TileArgs &a, int rr, int cc)
{
    const int y = a.top + rr, x = a.left + cc;
    if (x < 0) return (y < 0 || y >= a.H) ? 32 - cc : 32 - cc + a.left;
    if (x >= a.W) return 2 * a.W - 2 - x;
    return x;
}

// ---------------------------------------------------------------------------------------------------------------------
// Stages.  "r" is the first (even) row of the pair a stage handles in this step; c is the thread's column 0..159
// (or 160..191 for the helper lanes of a 192-thread group).
// ---------------------------------------------------------------------------------------------------------------------

// P1 (L342-351): gradients of row r, all 160 columns (4-lane groups over [0, cc1)); neighbours past a row end continue in
// the adjacent row of the flat tile
AMZ_DEV void st_p1(amz_lf lds, const TileArgs &a, int r, int c)
{
    if (r < 2 || r >= a.rr1 - 2 || c >= TS) return;
    amz_lf cm2 = ROWR(CFA, r - 2), cm1 = ROWR(CFA, r - 1), c0r = ROWR(CFA, r), cp1 = ROWR(CFA, r + 1), cp2 = ROWR(CFA, r + 2);
    const float cl1 = c >= 1 ? c0r[c - 1] : cm1[TS + c - 1];
    const float cl2 = c >= 2 ? c0r[c - 2] : cm1[TS + c - 2];
    const float cr1 = c + 1 < TS ? c0r[c + 1] : cp1[c + 1 - TS];
    const float cr2 = c + 2 < TS ? c0r[c + 2] : cp1[c + 2 - TS];
    const float c0 = c0r[c];
    const float cu1 = cm1[c], cu2 = cm2[c], cd1 = cp1[c], cd2 = cp2[c];
    const float delh = fabsf(cr1 - cl1);
    const float delv = fabsf(cd1 - cu1);
    const float w1 = eps + fabsf(cr2 - c0) + fabsf(c0 - cl2) + delh;
    const float w0 = eps + fabsf(cd2 - c0) + fabsf(c0 - cu2) + delv;
    const float dq = sqr(delh) + sqr(delv);
    ROWW(D0, r)[c] = w0;
    ROWW(D1, r)[c] = w1;
    ROWW(DHV, r)[c] = dq;
    if (r >= 4 && r < 76) {
        if (c >= 76 && c < 80) lds[SIDE_OFF + (r - 4) * 8 + (c - 76)] = dq;
        if (c >= 156) lds[SIDE_OFF + (r - 4) * 8 + 4 + (c - 156)] = dq;
    }
}

// P2 (L380-434): colour differences of row r, columns 4..155; zero where the reference's cleared planes are read unwritten
AMZ_DEV void st_p2(amz_lf lds, const TileArgs &a, int r, int c)
{
    if (r < 2 || r >= a.rr1 - 2 || c >= TS) return;
    float o_hcdalt = 0.f, o_vcdalt = 0.f, o_vcd = 0.f, o_hcd = 0.f;
    if (r >= 4 && r < a.rr1 - 4 && c >= 4 && c < TS - 4) {
        amz_lf cr = ROWR(CFA, r), d0r = ROWR(D0, r), d1r = ROWR(D1, r);
        const float sgn = is_green(a, r, c) ? -1.f : 1.f;
        const float cfav = cr[c];
        const float cu1 = ROWR(CFA, r - 1)[c], cu2 = ROWR(CFA, r - 2)[c];
        const float cd1 = ROWR(CFA, r + 1)[c], cd2 = ROWR(CFA, r + 2)[c];
        const float cl1 = cr[c - 1], cl2 = cr[c - 2], cr1 = cr[c + 1], cr2 = cr[c + 2];
        const float d0c = d0r[c], d1c = d1r[c];
        const float d0u2 = ROWR(D0, r - 2)[c], d0d2 = ROWR(D0, r + 2)[c], d1l2 = d1r[c - 2], d1r2 = d1r[c + 2];
        const float cru = cu1 * (d0u2 + d0c) / (d0u2 * (eps + cfav) + d0c * (eps + cu2));
        const float crd = cd1 * (d0d2 + d0c) / (d0d2 * (eps + cfav) + d0c * (eps + cd2));
        const float crl = cl1 * (d1l2 + d1c) / (d1l2 * (eps + cfav) + d1c * (eps + cl2));
        const float crr = cr1 * (d1r2 + d1c) / (d1r2 * (eps + cfav) + d1c * (eps + cr2));
        const float guha = cu1 + 0.5f * (cfav - cu2);
        const float gdha = cd1 + 0.5f * (cfav - cd2);
        const float glha = cl1 + 0.5f * (cfav - cl2);
        const float grha = cr1 + 0.5f * (cfav - cr2);
        float guar = fabsf(1.f - cru) < arthresh ? cfav * cru : guha;
        float gdar = fabsf(1.f - crd) < arthresh ? cfav * crd : gdha;
        float glar = fabsf(1.f - crl) < arthresh ? cfav * crl : glha;
        float grar = fabsf(1.f - crr) < arthresh ? cfav * crr : grha;
        const float d1l = d1r[c - 1], d1rr = d1r[c + 1], d0u = ROWR(D0, r - 1)[c], d0d = ROWR(D0, r + 1)[c];
        const float hwt = d1l / (d1l + d1rr);
        const float vwt = d0u / (d0d + d0u);
        const float Ginthha = intp(hwt, grha, glha);
        const float Gintvha = intp(vwt, gdha, guha);
        o_hcdalt = sgn * (Ginthha - cfav);
        o_vcdalt = sgn * (Gintvha - cfav);
        const bool clip = (cfav > a.clip_pt8) || (Gintvha > a.clip_pt8) || (Ginthha > a.clip_pt8);
        if (clip) { guar = guha; gdar = gdha; glar = glha; grar = grha; }
        o_vcd = clip ? o_vcdalt : sgn * (intp(vwt, gdar, guar) - cfav);
        o_hcd = clip ? o_hcdalt : sgn * (intp(hwt, grar, glar) - cfav);
        ROWW(DGV, r)[c] = sse_min(sqr(guha - gdha), sqr(guar - gdar));
        ROWW(DGH, r)[c] = sse_min(sqr(glha - grha), sqr(glar - grar));
        if (!is_green(a, r, c)) {
            // P4 (L698-699) forms the same two quotients from the same operands (a + b == b + a): hand them over instead of the weights
            amz_lf hv = ROWW(HVW, r);
            hv[c >> 1] = hwt;
            hv[TSH + (c >> 1)] = vwt;
        }
    }
    ROWW(HCA, r)[c] = o_hcdalt;
    ROWW(VCA, r)[c] = o_vcdalt;
    ROWW(VCD, r)[c] = o_vcd;
    ROWW(HCO, r)[c] = o_hcd;
}

// new hcd of a column whose 4-lane-group lane is 2 or 3: reads original values only (L540-583)
AMZ_DEV float p3_hcd_pure(amz_lf ho, amz_lf ha, amz_lf cf, const TileArgs &a, int r, int c)
{
    const float sgn = is_green(a, r, c) ? -1.f : 1.f;
    float hcdv = ho[c];
    const float hv = var3(ho[c - 2], hcdv, ho[c + 2]);
    const float a0 = ha[c];
    const float hav = var3(ha[c - 2], a0, ha[c + 2]);
    hcdv = hav < hv ? a0 : hcdv;
    return bound_cd(hcdv, sgn, cf[c], cf[c - 1], cf[c + 1], a.clip_pt);
}
// P3 (L540-583): variance choice + highlight bounding of row r.  hcd: lanes 0,1 of a 4-lane group read the UPDATED lanes 2,3 of
// the group to their left (recomputed here: they only depend on original values); vcd: row r reads the updated row r-2.
#ifdef AMZ_EMUL
#define AMZ_P3_COL(c) (c)
#else
// On the device the hcd value a lane 0 / 1 of a 4-lane group needs from the group to its left is the one lane - 2 of the same wave has
// just computed: it comes over by a wave shuffle instead of being evaluated a second time (with the two halves of a group in two divergent
// branches the wave paid for p3_hcd_pure twice).  For that a wave's 64 columns start two columns early -- 64 part - 2 --, so that the
// source lane is always in the same wave; the three parts still cover columns 0 .. 159.  (The emulation, which runs one thread at a time,
// keeps the recomputation: same values.)
#define AMZ_P3_COL(c) ((c) - 2)
#endif
AMZ_DEV void st_p3(amz_lf lds, const TileArgs &a, int r, int c_in)
{
    const int c = AMZ_P3_COL(c_in);
#ifdef AMZ_EMUL
    if (r < 4 || r >= a.rr1 - 4 || c >= TS) return;
#else
    if (r < 4 || r >= a.rr1 - 4) return;          // (uniform over the wave: every lane reaches the shuffle)
#endif
    float nh = 0.f;
    const bool dom = c >= 4 && c < TS - 4;
    amz_lf ho = ROWR(HCO, r), ha = ROWR(HCA, r), cf = ROWR(CFA, r);
    const int idx = c - 4, k = idx & 3, g = idx >> 2;
#ifndef AMZ_EMUL
    float pure = 0.f;
    if (dom) pure = p3_hcd_pure(ho, ha, cf, a, r, c);
    const float left2 = __shfl_up(pure, 2);          // lane - 2 = column c - 2 (lanes 0, 1 of a wave are lanes 2, 3 of their group: not used)
    if (c < 0 || c >= TS) return;
#endif
    if (dom) {
        if (k >= 2) {
#ifdef AMZ_EMUL
            nh = p3_hcd_pure(ho, ha, cf, a, r, c);
#else
            nh = pure;
#endif
        } else {
#ifdef AMZ_EMUL
            const float hm2 = g > 0 ? p3_hcd_pure(ho, ha, cf, a, r, c - 2) : ho[c - 2];
#else
            const float hm2 = g > 0 ? left2 : ho[c - 2];
#endif
            const float sgn = is_green(a, r, c) ? -1.f : 1.f;
            float hcdv = ho[c];
            const float hv = var3(hm2, hcdv, ho[c + 2]);
            const float a0 = ha[c];
            const float hav = var3(ha[c - 2], a0, ha[c + 2]);
            hcdv = hav < hv ? a0 : hcdv;
            nh = bound_cd(hcdv, sgn, cf[c], cf[c - 1], cf[c + 1], a.clip_pt);
        }
        const float sgn = is_green(a, r, c) ? -1.f : 1.f;
        amz_lf vr = ROWR(VCD, r);
        const float n2 = ROWR(VCD, r - 2)[c];
        const float o0 = vr[c], o2 = ROWR(VCD, r + 2)[c];
        const float am2 = ROWR(VCA, r - 2)[c], a0 = ROWR(VCA, r)[c], a2 = ROWR(VCA, r + 2)[c];
        const float cm1 = ROWR(CFA, r - 1)[c], c0 = cf[c], cp1 = ROWR(CFA, r + 1)[c];
        float vcdv = o0;
        const float vv = var3(n2, vcdv, o2);
        const float vav = var3(am2, a0, a2);
        vcdv = vav < vv ? a0 : vcdv;
        const float nv = bound_cd(vcdv, sgn, c0, cm1, cp1, a.clip_pt);
        vr[c] = nv;
        const float cdq = sqr(nv - nh);
        ROWW(CDD, r)[c] = cdq;
        if (r == 19 && c >= 80 && c < 120) lds[NQA_OFF + (c - 80)] = cdq;
    }
    ROWW(HCN, r)[c] = nh;      // columns 0-3 and 156-159: the cleared plane
}

// the R/B site a column thread owns in the row pair (r, r+1): row r + site_sel
AMZ_DEV int site_sel(const TileArgs &a, int r, int c) { return (c ^ row_par(a, r)) & 1; }

// P4 (L680-728): h/v weight at the R/B site of column c
AMZ_DEV void st_p4(amz_lf lds, const TileArgs &a, int r, int c)
{
    if (c >= TS) return;
    const int sl = site_sel(a, r, c), rr = r + sl;
    if (rr < 6 || rr >= a.rr1 - 6) return;
    const int par = row_par(a, rr);
    if (c < 6 + par || c > 156 + par) return;          // 4 * ngroups(6 + par, 154, 8) = 76 sites: the last group overruns
    amz_lf hn = SROWR(HCN, r, 0, sl), dh = SROWR(DGH, r, 0, sl);
    float tv = SROWR(VCD, r, 0, sl)[c];
    const float vu1 = SROWR(VCD, r, -1, sl)[c], vu2 = SROWR(VCD, r, -2, sl)[c], vu3 = SROWR(VCD, r, -3, sl)[c];
    const float vd1 = SROWR(VCD, r, +1, sl)[c], vd2 = SROWR(VCD, r, +2, sl)[c], vd3 = SROWR(VCD, r, +3, sl)[c];
    const float uave = tv + vu1 + vu2 + vu3;
    const float dave = tv + vd1 + vd2 + vd3;
    float Dvu = sqr(tv - uave) + sqr(vu1 - uave) + sqr(vu2 - uave) + sqr(vu3 - uave);
    float Dvd = sqr(tv - dave) + sqr(vd1 - dave) + sqr(vd2 - dave) + sqr(vd3 - dave);
    amz_lf hv = SROWR(HVW, r, 0, sl);                         // (over-run sites read slots P2 does not write: never used)
    const float hwt = hv[c >> 1], vwt = hv[TSH + (c >> 1)];
    const float vcd_c = tv;
    tv = hn[c];
    // columns past the row end (over-run sites only): what the flat tile holds there never reaches a site anyone reads
#define COLZ(row, cc) ((cc) < TS ? (row)[cc] : 0.f)
    const float hl1 = hn[c - 1], hl2 = hn[c - 2], hl3 = hn[c - 3];
    const float hr1 = COLZ(hn, c + 1), hr2 = COLZ(hn, c + 2), hr3 = COLZ(hn, c + 3);
    const float lave = tv + (hl3 + hl2) + hl1;
    const float rave = tv + (hr1 + hr2) + hr3;
    float Dhl = sqr(tv - lave) + sqr(hl1 - lave) + sqr(hl2 - lave) + sqr(hl3 - lave);
    float Dhr = sqr(tv - rave) + sqr(hr1 - rave) + sqr(hr2 - rave) + sqr(hr3 - rave);
    const float vcdvar = epssq + intp(vwt, Dvd, Dvu);
    const float hcdvar = epssq + intp(hwt, Dhr, Dhl);
    Dvu = SROWR(DGV, r, -1, sl)[c] + SROWR(DGV, r, -2, sl)[c];
    Dvd = SROWR(DGV, r, +1, sl)[c] + SROWR(DGV, r, +2, sl)[c];
    Dhl = dh[c - 2] + dh[c - 1];
    Dhr = COLZ(dh, c + 1) + COLZ(dh, c + 2);
#undef COLZ
    const float vcdvar1 = epssq + SROWR(DGV, r, 0, sl)[c] + intp(vwt, Dvd, Dvu);
    const float hcdvar1 = epssq + dh[c] + intp(hwt, Dhr, Dhl);
    const float varwt = hcdvar / (vcdvar + hcdvar);
    const float diffwt = hcdvar1 / (vcdvar1 + hcdvar1);
    const bool dec = ((0.5f - varwt) * (0.5f - diffwt) > 0.f) && (fabsf(0.5f - diffwt) < fabsf(0.5f - varwt));
    SROWW(HVWT, r, 0, sl)[c >> 1] = dec ? varwt : diffwt;
    amz_lf vh = SROWW(VH, r, 0, sl);
    vh[c >> 1] = vcd_c;
    vh[TSH + (c >> 1)] = tv;
}

// per-thread registers that live across the steps of a tile: the CFA rows in flight (loader threads) and the thread's own
// bounding box of the Nyquist flags it set / the nyquist2 sites it processed (min row, max row, min col, max col), merged into the
// workgroup's boxes once per tile.  (An LDS atomic per flag would be turned into a scalar loop over the active lanes by the
// compiler: four such loops per step cost more than the stage itself.)
struct ThreadRegs { int bb[4]; int pos_p, pos_h, pos_stride2, pos_off4; };    // pos_*: the ring position table entry a lane of the producing role owns
AMZ_DEV void bb_reset(int *bb) { bb[0] = 1 << 30; bb[1] = 0; bb[2] = 1 << 30; bb[3] = 0; }
AMZ_DEV void bb_add(int *bb, int rr, int cc) { bb[0] = imin(bb[0], rr); bb[1] = imax(bb[1], rr); bb[2] = imin(bb[2], cc); bb[3] = imax(bb[3], cc); }
AMZ_DEV void bb_flush(amz_lf lds, int base, const int *bb)
{
    if (bb[1] == 0 && bb[3] == 0) return;
    amz_li red = (amz_li)(lds + RED_OFF) + base;
#ifdef AMZ_EMUL
    red[0] = imin(red[0], bb[0]); red[1] = imax(red[1], bb[1]); red[2] = imin(red[2], bb[2]); red[3] = imax(red[3], bb[3]);
#else
    __hip_atomic_fetch_min(&red[0], bb[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_max(&red[1], bb[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_min(&red[2], bb[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_max(&red[3], bb[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
}

// the ring position table ("Ring positions" at the LDS layout): lane c of the producing column role owns entry c.  At step T the entry of
// pair j of a ring is the ring's pair slot (T - j) mod h; a lane carries its slot and advances it by one per step.
AMZ_DEV void pos_init(amz_lf lds, int c, ThreadRegs &rg)
{
    if (c >= POS_PAIRS) return;
    rg.pos_h = POSTAB.h[c]; rg.pos_stride2 = POSTAB.stride2[c]; rg.pos_off4 = POSTAB.off4[c];
    rg.pos_p = (rg.pos_h - POSTAB.j[c]) % rg.pos_h;
    ((amz_li)(lds + POS_OFF))[c] = rg.pos_off4 + rg.pos_p * rg.pos_stride2;      // the table of step 0
}
AMZ_DEV void pos_produce(amz_lf lds, int T, int c, ThreadRegs &rg)
{
    if (c >= POS_PAIRS) return;
    rg.pos_p = rg.pos_p + 1 == rg.pos_h ? 0 : rg.pos_p + 1;
    ((amz_li)(lds + POS_OFF + ((T + 1) & 1) * POS_PAIRS))[c] = rg.pos_off4 + rg.pos_p * rg.pos_stride2;       // the table of step T + 1
}

// P5 + P6 (L746-825): nyquist test value and flag of the site of column c; helper lanes 160..171 clear the flag bytes without a site
AMZ_DEV void st_p5(amz_lf lds, const TileArgs &a, int r, int c, ThreadRegs &rg)
{
    if (r + 1 < 6 || r >= a.rr1 - 6) return;
    if (c >= TS) {
        const int h = c - TS;
        if (h < 12) {
            const int sl = h >= 6, k = h % 6;
            if (r + sl < 6 || r + sl >= a.rr1 - 6) return;
            amz_lb nb = (amz_lb)SROWW(NYQ, r, 0, sl);
            nb[k < 3 ? k : 74 + k] = 0;                 // bytes 0,1,2 and 77,78,79
        }
        return;
    }
    const int sl = site_sel(a, r, c), rr = r + sl;
    if (rr < 6 || rr >= a.rr1 - 6) return;
    const int par = row_par(a, rr);
    if (c < 6 + par) return;
    amz_lb nb = (amz_lb)SROWW(NYQ, r, 0, sl);
    if (c >= TS - 6) { nb[c >> 1] = 0; return; }        // sites 154..159: no test (L806-808), the byte stays cleared
    const float go0 = 0.14659727707323927f, go1 = 0.103592713382435f, go2 = 0.0732036125103057f, go3 = 0.0365543548389495f;
    const float nyqthresh = 0.5f;
    const float gg0 = nyqthresh * 0.07384411893421103f, gg1 = nyqthresh * 0.06207511968171489f, gg2 = nyqthresh * 0.0521818194747806f;
    const float gg3 = nyqthresh * 0.03687419286733595f, gg4 = nyqthresh * 0.03099732204057846f, gg5 = nyqthresh * 0.018413194161458882f;
    const int si = (c - 6 - par) >> 1;
    const bool vec = si < 4 * ngroups(6 + par, TS - 7, 8);
    amz_lf c0 = SROWR(CDD, r, 0, sl), cu1 = SROWR(CDD, r, -1, sl), cu2 = SROWR(CDD, r, -2, sl), cd1 = SROWR(CDD, r, +1, sl), cd2 = SROWR(CDD, r, +2, sl);
    amz_lf d0 = SROWR(DHV, r, 0, sl), du1 = SROWR(DHV, r, -1, sl), du2 = SROWR(DHV, r, -2, sl), dd1 = SROWR(DHV, r, +1, sl), dd2 = SROWR(DHV, r, +2, sl);
    const float gA = go0 * c0[c] +
                     go1 * (cu1[c - 1] + cu1[c + 1] + cd1[c - 1] + cd1[c + 1]) +
                     go2 * (cu2[c] + c0[c - 2] + c0[c + 2] + cd2[c]) +
                     go3 * (cu2[c - 2] + cu2[c + 2] + cd2[c - 2] + cd2[c + 2]);
    const float s1 = vec ? (du1[c] + d0[c - 1] + d0[c + 1] + dd1[c])
                         : (du1[c] + d0[c + 1] + d0[c - 1] + dd1[c]);
    const float gB = gg0 * d0[c] + gg1 * s1 +
                     gg2 * (du1[c - 1] + du1[c + 1] + dd1[c - 1] + dd1[c + 1]) +
                     gg3 * (du2[c] + d0[c - 2] + d0[c + 2] + dd2[c]) +
                     gg4 * (du2[c - 1] + du2[c + 1] + du1[c - 2] + du1[c + 2] +
                            dd1[c - 2] + dd1[c + 2] + dd2[c - 1] + dd2[c + 1]) +
                     gg5 * (du2[c - 2] + du2[c + 2] + dd2[c - 2] + dd2[c + 2]);
    const bool flag = gA - gB > 0.f;
    nb[c >> 1] = flag ? 1 : 0;
    if (flag) bb_add(rg.bb, rr, c);
}

// P7 (L888-901): majority vote, item = one byte of rows (r, r+1); byte offsets independent of the row parity.  Rows 2..7 and
// 152..157 are the memset (L879) / never-written part of nyquist2 that P8's window reads.
AMZ_DEV void st_p7(amz_lf lds, const TileArgs &a, int r, int item)
{
    if (item >= 2 * TSH) return;
    const int sl = item >= TSH, rr = r + sl, b = item - (sl ? TSH : 0);
    if (rr < 2 || rr >= a.rr1 - 2) return;
    amz_lb out = (amz_lb)SROWW(NYQ2, r, 0, sl);
    if (rr < ny_r0(a) || rr >= ny_r1(a)) {
        // outside [nystartrow, nyendrow): the memset value (L879); rows 156..159 lie behind the memset (L879): they still hold the bytes of cddiffsq(19, 80..159), and P8's window at rows
        // 150, 151 tests them
        out[b] = rr >= TS - 4 ? ((amz_lb)(lds + NQA_OFF))[(rr - (TS - 4)) * TSH + b] : 0;   // (only tiles of 159 / 160 rows get there)
        return;
    }
    amz_lb n0 = (amz_lb)SROWR(NYQ, r, 0, sl), nu1 = (amz_lb)SROWR(NYQ, r, -1, sl), nu2 = (amz_lb)SROWR(NYQ, r, -2, sl);
    amz_lb nd1 = (amz_lb)SROWR(NYQ, r, +1, sl), nd2 = (amz_lb)SROWR(NYQ, r, +2, sl);
    // bytes before/after a row belong to the neighbouring rows of the flat map; those bytes (site index 79 / 0) are never set
    const int bl = b - 1, br = b + 1;
    const int t = (int)nu2[b] + (bl >= 0 ? (int)nu1[bl] : 0) + (int)nu1[b] + (bl >= 0 ? (int)n0[bl] : 0) +
                  (br < TSH ? (int)n0[br] : 0) + (bl >= 0 ? (int)nd1[bl] : 0) + (int)nd1[b] + (int)nd2[b];
    unsigned char val = n0[b];
    if (t > 4) val = 1;
    if (t < 4) val = 0;
    out[b] = val;
}

// is column c's site in rows (r, r+1) one that P8 / P10 process (inside the tile's Nyquist box, nyquist2 set)?
AMZ_DEV bool nyq_site(amz_lf lds, const TileArgs &a, int r, int c, int *prr)
{
    if (c >= TS) return false;
    const int sl = site_sel(a, r, c), rr = r + sl;
    *prr = rr;
    if (rr < ny_r0(a) || rr >= ny_r1(a) || c < ny_c0(a) + row_par(a, rr) || c >= ny_c1(a)) return false;     // L914-915, L980-981
    return ((amz_lb)SROWR(NYQ2, r, 0, sl))[c >> 1] != 0;
}

// P8 (L914-951): area interpolation at one flagged site.  The 49 window sites are visited in the reference's order; a site that is
// not flagged contributes +0 to every sum (exact: the sums are non-negative), so the window is a straight-line, branch-free block
// whose loads the compiler can batch.  One wave owns the step's sites; the window rows -6..0 are summed in sub-step a, rows 2..6 in
// sub-step b (the partial sums stay in registers across the barrier), which halves the longest serial chain of a sub-step.
struct P8Acc { float sumcfa, sumh, sumv, sumsqh, sumsqv, areawt; };
template <int A0, int A1>
AMZ_DEV void p8_accumulate(amz_lf lds, const TileArgs &a, int r, int sl, int cc, P8Acc &s)
{
    const int idx0 = (cc - 6) >> 1;
#pragma unroll
    for (int ai = A0; ai <= A1; ai += 2) {
        amz_lb nq = (amz_lb)SROWR(NYQ2, r, ai, sl);
        amz_lf cr = SROWR(CFA, r, ai, sl), cu = SROWR(CFA, r, ai - 1, sl), cd = SROWR(CFA, r, ai + 1, sl);
#pragma unroll
        for (int bj = 0; bj < 7; ++bj) {
            const bool f = nq[idx0 + bj] != 0;
            const int col = cc - 6 + 2 * bj;
            const float ct = cr[col], cl = cr[col - 1], cq = cr[col + 1], cn = cu[col], cs = cd[col];
            const float t1 = f ? ct : 0.f;
            const float t2 = f ? (cl + cq) : 0.f;
            const float t3 = f ? (cn + cs) : 0.f;
            const float t4 = f ? (sqr(ct - cl) + sqr(ct - cq)) : 0.f;
            const float t5 = f ? (sqr(ct - cn) + sqr(ct - cs)) : 0.f;
            s.sumcfa += t1;
            s.sumh += t2;
            s.sumv += t3;
            s.sumsqh += t4;
            s.sumsqv += t5;
            s.areawt += f ? 1.f : 0.f;
        }
    }
}
AMZ_DEV void p8_finish(amz_lf lds, const TileArgs &a, int r, int sl, int cc, const P8Acc &s, int *bb)
{
    const int rr = r + sl;
    const float sumh = s.sumcfa - xdiv2f(s.sumh);
    const float sumv = s.sumcfa - xdiv2f(s.sumv);
    const float areawt = xdiv2f(s.areawt);
    const float hcdvar = epssq + fabsf(areawt * s.sumsqh - sumh * sumh);
    const float vcdvar = epssq + fabsf(areawt * s.sumsqv - sumv * sumv);
    SROWR(HVWT, r, 0, sl)[cc >> 1] = hcdvar / (vcdvar + hcdvar);
    bb_add(bb, rr, cc);
}
// the P8 wave, lane 0..63: first half of the window for list entry `lane` (sub-step a) ...
struct P8Regs { P8Acc acc; int sl, cc; };
AMZ_DEV void p8_wave_a(amz_lf lds, const TileArgs &a, int t, int r, int lane, P8Regs &pr)
{
    amz_li red = (amz_li)(lds + RED_OFF);
    amz_ls list = (amz_ls)(lds + LIST_OFF + (t & 1) * LIST_INTS);
    const int n = red[16 + (t & 1)];
    pr.cc = -1;
    if (lane < n) {
        const int e = list[lane];
        pr.sl = (e >> 8) - r; pr.cc = e & 255;
        pr.acc = P8Acc{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        p8_accumulate<-6, 0>(lds, a, r, pr.sl, pr.cc, pr.acc);
    }
}
// ... second half and the result (sub-step b); list entries beyond the wave width (dense Nyquist regions) are done here in full
AMZ_DEV void p8_wave_b(amz_lf lds, const TileArgs &a, int t, int r, int lane, P8Regs &pr, int *bb)
{
    amz_li red = (amz_li)(lds + RED_OFF);
    amz_ls list = (amz_ls)(lds + LIST_OFF + (t & 1) * LIST_INTS);
    const int n = red[16 + (t & 1)];
    if (pr.cc >= 0) {
        p8_accumulate<2, 6>(lds, a, r, pr.sl, pr.cc, pr.acc);
        p8_finish(lds, a, r, pr.sl, pr.cc, pr.acc, bb);
    }
    for (int q = 64 + lane; q < n; q += 64) {
        const int e = list[q];
        const int sl = (e >> 8) - r, cc = e & 255;
        P8Acc acc{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        p8_accumulate<-6, 0>(lds, a, r, sl, cc, acc);
        p8_accumulate<2, 6>(lds, a, r, sl, cc, acc);
        p8_finish(lds, a, r, sl, cc, acc, bb);
    }
}

// P9 (L957-974), part 1: the refined weight of site j (0..71) of row rr; reads the UPDATED row rr-1 and the old row rr+1
AMZ_DEV float p9_new_weight(amz_lf lds, const TileArgs &a, int rr, int j)
{
    const int cc = 8 + row_par(a, rr) + 2 * j;
    amz_lf hu = ROWR(HVWT, rr - 1), hd = ROWR(HVWT, rr + 1);
    const float hvwtalt = xdivf(hu[(cc - 1) >> 1] + hu[(cc + 1) >> 1] + hd[(cc - 1) >> 1] + hd[(cc + 1) >> 1], 2);
    const float h0 = ROWR(HVWT, rr)[cc >> 1];
    return fabsf(0.5f - h0) < fabsf(0.5f - hvwtalt) ? hvwtalt : h0;
}
// part 2: everything that merely consumes the refined weight of the site
AMZ_DEV void p9_site(amz_lf lds, const TileArgs &a, int rr, int j, float h)
{
    const int cc = 8 + row_par(a, rr) + 2 * j, idx = cc >> 1;
    ROWR(HVWT, rr)[idx] = h;
    amz_lf vh = ROWR(VH, rr), cr = ROWR(CFA, rr);
    const float dg = intp(h, vh[idx], vh[TSH + idx]);
    ROWW(DG0, rr)[idx] = dg;
    const float gval = cr[cc] + dg;
    ROWW(RGBG, rr)[idx] = gval;
    const bool ny = ((amz_lb)ROWR(NYQ2, rr))[idx] != 0;
    amz_lf d2 = ROWW(DG2, rr);
    d2[2 * idx] = ny ? sqr(gval - xdiv2f(cr[cc - 1] + cr[cc + 1])) : 0.f;
    d2[2 * idx + 1] = ny ? sqr(gval - xdiv2f(ROWR(CFA, rr - 1)[cc] + ROWR(CFA, rr + 1)[cc])) : 0.f;
}

// P11 (L1004-1027): diagonal gradients; item = one (even column, odd column) pair of rows (r, r+1)
AMZ_DEV void st_p11(amz_lf lds, const TileArgs &a, int r, int item)
{
    if (item >= 152) return;
    const int sl = item >= 76, rr = r + sl, j = item - (sl ? 76 : 0);
    if (rr < 6 || rr >= a.rr1 - 6) return;
    const int e = 6 + 2 * j, idx = e >> 1;
    const bool rbEven = row_par(a, rr) == 0;
    const int g = rbEven ? e + 1 : e; // green site of the pair
    const int q = rbEven ? e : e + 1; // red/blue site of the pair
    amz_lf cr = SROWR(CFA, r, 0, sl), cu = SROWR(CFA, r, -1, sl), cd = SROWR(CFA, r, +1, sl);
    const float t = cr[g];
    const float sp = sqr(t - cd[g - 1]) + sqr(t - cu[g + 1]);
    const float sm = sqr(t - cu[g - 1]) + sqr(t - cd[g + 1]);
    SROWW(DELP, r, 0, sl)[idx] = fabsf(cu[q + 1] - cd[q - 1]);
    SROWW(DELM, r, 0, sl)[idx] = fabsf(cd[q + 1] - cu[q - 1]);
    SROWW(DM, r, 0, sl)[idx] = sm;
    SROWW(DP, r, 0, sl)[idx] = sp;
}

// P12 (L1061-1121): diagonal interpolation of R+B and plus/minus weight at the site of column c; helper lanes 160..167 put what
// pmwt[.., 76..79] aliases (delhvsqsum, L169) into the ring row
AMZ_DEV void st_p12(amz_lf lds, const TileArgs &a, int r, int c)
{
    if (r + 1 < 8 || r >= a.rr1 - 8) return;
    if (c >= TS) {
        const int h = c - TS;
        if (h < 8) {
            const int sl = h >= 4, rr = r + sl, k = h & 3;
            if (rr < 8 || rr >= a.rr1 - 8) return;
            SROWW(PMWT, r, 0, sl)[76 + k] = lds[SIDE_OFF + ((rr >> 1) - 4) * 8 + (rr & 1) * 4 + k];
        }
        return;
    }
    const int sl = site_sel(a, r, c), rr = r + sl;
    if (rr < 8 || rr >= a.rr1 - 8) return;
    const int par = row_par(a, rr);
    if (c < 8 + par || c > 150 + par) return;
    const int idx = c >> 1;
    const float ge0 = 0.13719494435797422f, ge1 = 0.05640252782101291f;
    amz_lf c0 = SROWR(CFA, r, 0, sl), cu1 = SROWR(CFA, r, -1, sl), cu2 = SROWR(CFA, r, -2, sl), cd1 = SROWR(CFA, r, +1, sl), cd2 = SROWR(CFA, r, +2, sl);
    const float cfav = c0[c];
    const float cse = cd1[c + 1], cnw = cu1[c - 1], cne = cu1[c + 1], csw = cd1[c - 1];
    const float rbse = rb_ratio(cfav, cse, cd2[c + 2]);
    const float rbnw = rb_ratio(cfav, cnw, cu2[c - 2]);
    amz_lf dm0 = SROWR(DELM, r, 0, sl), dmu1 = SROWR(DELM, r, -1, sl), dmu2 = SROWR(DELM, r, -2, sl), dmd1 = SROWR(DELM, r, +1, sl), dmd2 = SROWR(DELM, r, +2, sl);
    float t1 = eps + dm0[idx];
    const float wtse = t1 + dmd1[(c + 1) >> 1] + dmd2[(c + 2) >> 1];
    const float wtnw = t1 + dmu1[(c - 1) >> 1] + dmu2[(c - 2) >> 1];
    const float rbmv = (wtse * rbnw + wtnw * rbse) / (wtse + wtnw);
    const float rbm_out = rb_bound(rbmv, cfav, cnw, cse, a.clip_pt);
    const float rbne = rb_ratio(cfav, cne, cu2[c + 2]);
    const float rbsw = rb_ratio(cfav, csw, cd2[c - 2]);
    amz_lf dp0 = SROWR(DELP, r, 0, sl), dpu1 = SROWR(DELP, r, -1, sl), dpu2 = SROWR(DELP, r, -2, sl), dpd1 = SROWR(DELP, r, +1, sl), dpd2 = SROWR(DELP, r, +2, sl);
    t1 = eps + dp0[idx];
    const float wtne = t1 + dpu1[(c + 1) >> 1] + dpu2[(c + 2) >> 1];
    const float wtsw = t1 + dpd1[(c - 1) >> 1] + dpd2[(c - 2) >> 1];
    const float rbpv = (wtne * rbsw + wtsw * rbne) / (wtne + wtsw);
    const float rbp_out = rb_bound(rbpv, cfav, csw, cne, a.clip_pt);
    amz_lf m0 = SROWR(DM, r, 0, sl), mu1 = SROWR(DM, r, -1, sl), mu2 = SROWR(DM, r, -2, sl), md1 = SROWR(DM, r, +1, sl), md2 = SROWR(DM, r, +2, sl);
    amz_lf q0 = SROWR(DP, r, 0, sl), qu1 = SROWR(DP, r, -1, sl), qu2 = SROWR(DP, r, -2, sl), qd1 = SROWR(DP, r, +1, sl), qd2 = SROWR(DP, r, +2, sl);
    const float rbvarm = epssq + (ge0 * (mu1[c >> 1] + m0[(c - 1) >> 1] + m0[(c + 1) >> 1] + md1[c >> 1]) +
                                  ge1 * (mu2[(c - 1) >> 1] + mu2[(c + 1) >> 1] + mu1[(c - 2) >> 1] + mu1[(c + 2) >> 1] +
                                         md1[(c - 2) >> 1] + md1[(c + 2) >> 1] + md2[(c - 1) >> 1] + md2[(c + 1) >> 1]));
    const float rbvarp = epssq + (ge0 * (qu1[c >> 1] + q0[(c - 1) >> 1] + q0[(c + 1) >> 1] + qd1[c >> 1]) +
                                  ge1 * (qu2[(c - 1) >> 1] + qu2[(c + 1) >> 1] + qu1[(c - 2) >> 1] + qu1[(c + 2) >> 1] +
                                         qd1[(c - 2) >> 1] + qd1[(c + 2) >> 1] + qd2[(c - 1) >> 1] + qd2[(c + 1) >> 1]));
    SROWW(RBM, r, 0, sl)[idx] = rbm_out;
    SROWW(RBP, r, 0, sl)[idx] = rbp_out;
    SROWW(PMWT, r, 0, sl)[idx] = rbvarm / (rbvarp + rbvarm);
}

// P13 (L1213-1223), part 1: refined pmwt of site j (0..71; the last group of four over-runs the scalar bound) of row rr
AMZ_DEV float p13_new_weight(amz_lf lds, const TileArgs &a, int rr, int j)
{
    const int cc = 10 + row_par(a, rr) + 2 * j;
    amz_lf pu = ROWR(PMWT, rr - 1), pd = ROWR(PMWT, rr + 1);
    const float alt = 0.25f * (pu[(cc - 1) >> 1] + pu[(cc + 1) >> 1] + pd[(cc - 1) >> 1] + pd[(cc + 1) >> 1]);
    const float t = ROWR(PMWT, rr)[cc >> 1];
    return fabsf(0.5f - t) < fabsf(0.5f - alt) ? alt : t;
}
AMZ_DEV void p13_site(amz_lf lds, const TileArgs &a, int rr, int j, float nt)
{
    const int cc = 10 + row_par(a, rr) + 2 * j, idx = cc >> 1;
    ROWR(PMWT, rr)[idx] = nt;
    ROWW(RBINT, rr)[idx] = 0.5f * (ROWR(CFA, rr)[cc] + intp(nt, ROWX(RBP, rr)[idx], ROWX(RBM, rr)[idx]));
}

// dirwts0 / dirwts1 of P1 (L342-351) re-evaluated where P14 needs them (the planes are long gone)
AMZ_DEV float dirwt_v(amz_lf cm2, amz_lf cm1, amz_lf c0r, amz_lf cp1, amz_lf cp2, int c)
{
    const float c0 = c0r[c];
    const float delv = fabsf(cp1[c] - cm1[c]);
    return eps + fabsf(cp2[c] - c0) + fabsf(c0 - cm2[c]) + delv;
}
AMZ_DEV float dirwt_h(amz_lf cr, int c)
{
    const float c0 = cr[c];
    const float delh = fabsf(cr[c + 1] - cr[c - 1]);
    return eps + fabsf(cr[c + 2] - c0) + fabsf(c0 - cr[c - 2]) + delh;
}

// P10 (L979-999) + P14 (L1241-1297) + P15 (L1381-1386): all three only touch the site's own Dgrb / rgbgreen.  P10 (sites with
// nyquist2 set) and P14 (sites where the diagonal weight is the more decisive one) are expensive and apply to ~1 site in 8 each:
// the column threads only classify their site (sub-step a) and put the ones that need work into two lists; ONE wave per list then
// does the arithmetic with full lanes (sub-step b).  P14 overrides whatever P10 produced for the site, so a site that passes P14's
// test goes to that list only.
AMZ_DEV void p15_store(amz_lf lds, const TileArgs &a, int r, int sl, int c, bool in14, float dg, float gval)
{
    const int rr = r + sl, idx = c >> 1;
    SROWR(RGBG, r, 0, sl)[idx] = gval;
    if (in14 && (rr & 1) != a.ey) {       // a blue row: G-B moves to Dgrb[1] (L1381-1386)
        SROWW(DG1, r, 0, sl)[idx] = dg;
        SROWR(DG0, r, 0, sl)[idx] = 0.f;
    } else {
        SROWR(DG0, r, 0, sl)[idx] = dg;
    }
}
AMZ_DEV bool p14_in_range(const TileArgs &a, int rr, int c) { const int par = row_par(a, rr); return rr >= 12 && rr < a.rr1 - 12 && c >= 12 + par && c <= 146 + par; }
// classification of the site of column c in rows (r, r+1): 0 nothing to compute (P15's bookkeeping is done here), 1 P10, 2 P14
AMZ_DEV int st_p1014_classify(amz_lf lds, const TileArgs &a, int r, int c, int *entry)
{
    *entry = 0;
    if (c >= TS) return 0;
    const int sl = site_sel(a, r, c), rr = r + sl;
    *entry = (sl << 8) | c;
    if (rr < 8 || rr >= a.rr1 - 8) return 0;
    const int par = row_par(a, rr), idx = c >> 1;
    if (c < 8 + par || c >= TS - 8) return 0;
    const bool in14 = p14_in_range(a, rr, c);
    if (in14 && fabsf(0.5f - SROWR(PMWT, r, 0, sl)[idx]) >= fabsf(0.5f - SROWR(HVWT, r, 0, sl)[idx])) return 2;
    if (rr >= ny_r0(a) && rr < ny_r1(a) && c >= ny_c0(a) + par && c < ny_c1(a) && ((amz_lb)SROWR(NYQ2, r, 0, sl))[idx]) return 1;
    p15_store(lds, a, r, sl, c, in14, SROWR(DG0, r, 0, sl)[idx], SROWR(RGBG, r, 0, sl)[idx]);
    return 0;
}
// P10 at one listed site (L979-999)
AMZ_DEV void p10_site(amz_lf lds, const TileArgs &a, int r, int sl, int c)
{
    const int idx = c >> 1;
    const float gq0 = 0.169917f, gq1 = 0.108947f, gq2 = 0.069855f, gq3 = 0.0287182f;
    // Dgrb2 rows 6,7,152,153 and columns < 8 / >= 152 alias dgintv in the reference: only sites no output depends on read them
    amz_lf e0 = SROWX(DG2, r, 0, sl), eu1 = SROWX(DG2, r, -1, sl), eu2 = SROWX(DG2, r, -2, sl), ed1 = SROWX(DG2, r, +1, sl), ed2 = SROWX(DG2, r, +2, sl);
#define DH(row, col) (row)[(col) & ~1]
#define DV(row, col) (row)[(col) | 1]
    const float gvarh = epssq + (gq0 * DH(e0, c) +
                                 gq1 * (DH(eu1, c - 1) + DH(eu1, c + 1) + DH(ed1, c - 1) + DH(ed1, c + 1)) +
                                 gq2 * (DH(eu2, c) + DH(e0, c - 2) + DH(e0, c + 2) + DH(ed2, c)) +
                                 gq3 * (DH(eu2, c - 2) + DH(eu2, c + 2) + DH(ed2, c - 2) + DH(ed2, c + 2)));
    const float gvarv = epssq + (gq0 * DV(e0, c) +
                                 gq1 * (DV(eu1, c - 1) + DV(eu1, c + 1) + DV(ed1, c - 1) + DV(ed1, c + 1)) +
                                 gq2 * (DV(eu2, c) + DV(e0, c - 2) + DV(e0, c + 2) + DV(ed2, c)) +
                                 gq3 * (DV(eu2, c - 2) + DV(eu2, c + 2) + DV(ed2, c - 2) + DV(ed2, c + 2)));
#undef DH
#undef DV
    amz_lf vh = SROWR(VH, r, 0, sl);
    const float dg = (vh[TSH + idx] * gvarv + vh[idx] * gvarh) / (gvarv + gvarh);
    p15_store(lds, a, r, sl, c, p14_in_range(a, r + sl, c), dg, SROWR(CFA, r, 0, sl)[c] + dg);
}
// P14 at one listed site (L1241-1297)
AMZ_DEV void p14_site(amz_lf lds, const TileArgs &a, int r, int sl, int c)
{
    const int idx = c >> 1;
    const float hw = SROWR(HVWT, r, 0, sl)[idx];
    amz_lf rb0 = SROWR(RBINT, r, 0, sl);
    const float rb = rb0[idx];
    amz_lf f3u = SROWR(CFA, r, -3, sl), f2u = SROWR(CFA, r, -2, sl), f1u = SROWR(CFA, r, -1, sl), f0 = SROWR(CFA, r, 0, sl);
    amz_lf f1d = SROWR(CFA, r, +1, sl), f2d = SROWR(CFA, r, +2, sl), f3d = SROWR(CFA, r, +3, sl);
    const float cfav = f0[c];
    const float cu = f1u[c], cd = f1d[c], cl = f0[c - 1], cr = f0[c + 1];
    const float gu = g_dir(rb, cu, SROWR(RBINT, r, -2, sl)[idx]);      // rbint[indx1 -+ v1] is two rows away (L1253-1260)
    const float gd = g_dir(rb, cd, SROWR(RBINT, r, +2, sl)[idx]);
    const float d0u = dirwt_v(f3u, f2u, f1u, f0, f1d, c), d0d = dirwt_v(f1u, f0, f1d, f2d, f3d, c);
    float Gintv = (d0u * gd + d0d * gu) / (d0d + d0u);
    Gintv = g_bound(Gintv, rb, cu, cd, a.clip_pt);
    const float gl = g_dir(rb, cl, rb0[idx - 1]);
    const float gr = g_dir(rb, cr, rb0[idx + 1]);
    const float d1l = dirwt_h(f0, c - 1), d1r = dirwt_h(f0, c + 1);
    float Ginth = (d1l * gr + d1r * gl) / (d1l + d1r);
    Ginth = g_bound(Ginth, rb, cl, cr, a.clip_pt);
    const float gval = intp(hw, Gintv, Ginth);
    p15_store(lds, a, r, sl, c, true, gval - cfav, gval);
}
// appending to a list: a thread at a time in the emulation, one LDS atomic per wave on the device (every lane of the wave calls it)
AMZ_DEV void hot_append(amz_lf lds, int which, bool flag, int entry)
{
    amz_li hot = (amz_li)(lds + HOT_OFF);
#ifdef AMZ_EMUL
    if (flag) hot[2 + which * 160 + hot[which]++] = entry;
#else
    const unsigned long long m = __ballot(flag);
    if (m) {
        const int lane = (int)(threadIdx.x & 63u);
        int base = 0;
        if (lane == 0) base = __hip_atomic_fetch_add(&hot[which], __popcll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        base = __builtin_amdgcn_readfirstlane(base);
        if (flag) hot[2 + which * 160 + base + __popcll(m & ((1ull << lane) - 1ull))] = entry;
    }
#endif
}
AMZ_DEV void st_p1014_light(amz_lf lds, const TileArgs &a, int r, int c)
{
    int entry;
    const int kind = st_p1014_classify(lds, a, r, c, &entry);
    hot_append(lds, 0, kind == 1, entry);
    hot_append(lds, 1, kind == 2, entry);
}
// the worker waves of sub-step b: lane-strided over a list; lane 0 empties the list afterwards
AMZ_DEV void p10_worker(amz_lf lds, const TileArgs &a, int r, int lane)
{
    amz_li hot = (amz_li)(lds + HOT_OFF);
    const int n = hot[0];
    for (int q = lane; q < n; q += 64) { const int e = hot[2 + q]; p10_site(lds, a, r, e >> 8, e & 255); }
}
AMZ_DEV void p14_worker(amz_lf lds, const TileArgs &a, int r, int lane)
{
    amz_li hot = (amz_li)(lds + HOT_OFF);
    const int n = hot[1];
    for (int q = lane; q < n; q += 64) { const int e = hot[2 + 160 + q]; p14_site(lds, a, r, e >> 8, e & 255); }
}
AMZ_DEV void hot_reset(amz_lf lds, int which) { ((amz_li)(lds + HOT_OFF))[which] = 0; }

// P16 (L1394-1408): chrominance of the opposite colour at the site of column c
AMZ_DEV void st_p16(amz_lf lds, const TileArgs &a, int r, int c)
{
    if (c >= TS) return;
    const int sl = site_sel(a, r, c), rr = r + sl;
    if (rr < 14 || rr >= a.rr1 - 14) return;
    const int par = row_par(a, rr);
    if (c < 14 + par || c > 148 + par) return;
    const bool red_row = (rr & 1) == a.ey;   // c = 1 - FC/2: red sites get G-B from the blue rows above and below
    // rows rr+-1, rr+-3 hold the native differences; Dgrb[1] positions P15 never wrote alias vcdalt in the reference and
    // only reach sites no output depends on
#define DROW(k) (red_row ? SROWX(DG1, r, (k), sl) : SROWX(DG0, r, (k), sl))
    amz_lf u1 = DROW(-1), u3 = DROW(-3), d1 = DROW(1), d3 = DROW(3);
#undef DROW
#define AT(row, col) (row)[(col) >> 1]
    const float dnw = AT(u1, c - 1), dse = AT(d1, c + 1), dne = AT(u1, c + 1), dsw = AT(d1, c - 1);
    const float dnw3 = AT(u3, c - 3), dse3 = AT(d3, c + 3), dne3 = AT(u3, c + 3), dsw3 = AT(d3, c - 3);
    const float temp = eps + fabsf(dnw - dse);
    const float temp2 = eps + fabsf(dne - dsw);
    const float wtnw = 1.f / (temp + fabsf(dnw - dnw3) + fabsf(dse - dnw3));
    const float wtne = 1.f / (temp2 + fabsf(dne - dne3) + fabsf(dsw - dne3));
    const float wtsw = 1.f / (temp2 + fabsf(dsw - dse3) + fabsf(dne - dsw3));
    const float wtse = 1.f / (temp + fabsf(dse - dsw3) + fabsf(dnw - dse3));
    // DG(i-m1-2), DG(i-m1-v2); DG(i+p1+2), DG(i+p1+v2); DG(i-p1-2), DG(i-p1-v2); DG(i+m1+2), DG(i+m1+v2)
    const float val = (wtnw * (1.325f * dnw - 0.175f * dnw3 - 0.075f * (AT(u1, c - 3) + AT(u3, c - 1))) +
                       wtne * (1.325f * dne - 0.175f * dne3 - 0.075f * (AT(u1, c + 3) + AT(d1, c + 1))) +
                       wtsw * (1.325f * dsw - 0.175f * dsw3 - 0.075f * (AT(d1, c - 3) + AT(u1, c - 1))) +
                       wtse * (1.325f * dse - 0.175f * dse3 - 0.075f * (AT(d1, c + 3) + AT(d3, c + 1)))) /
                      (wtnw + wtne + wtsw + wtse);
#undef AT
    if (red_row) SROWW(DG1, r, 0, sl)[c >> 1] = val; else SROWR(DG0, r, 0, sl)[c >> 1] = val;
}

// P17/P18 (L1441-1565): R, G, B of tile row rr, column c
AMZ_DEV void st_out(amz_lf lds, const TileArgs &a, int rr, int c)
{
    if (rr < 16 || rr >= a.rr1 - 16 || c < 16 || c >= TS - 16) return;
    float gval, rv, bv;
    if (is_green(a, rr, c)) {
        gval = ROWR(CFA, rr)[c];
        amz_lf hu = ROWR(HVWT, rr - 1), h0 = ROWR(HVWT, rr), hd = ROWR(HVWT, rr + 1);
        const float h_up = hu[c >> 1], h_dn = hd[c >> 1], h_r = h0[(c + 1) >> 1], h_l = h0[(c - 1) >> 1];
        const float temp = 1.f / (h_up + 2.f - h_r - h_l + h_dn);
        amz_lf a0u = ROWR(DG0, rr - 1), a00 = ROWR(DG0, rr), a0d = ROWR(DG0, rr + 1);
        amz_lf a1u = ROWR(DG1, rr - 1), a10 = ROWR(DG1, rr), a1d = ROWR(DG1, rr + 1);
        rv = gval - (h_up * a0u[c >> 1] + (1.f - h_r) * a00[(c + 1) >> 1] + (1.f - h_l) * a00[(c - 1) >> 1] + h_dn * a0d[c >> 1]) * temp;
        bv = gval - (h_up * a1u[c >> 1] + (1.f - h_r) * a10[(c + 1) >> 1] + (1.f - h_l) * a10[(c - 1) >> 1] + h_dn * a1d[c >> 1]) * temp;
    } else {
        gval = ROWR(RGBG, rr)[c >> 1];
        rv = gval - ROWR(DG0, rr)[c >> 1];
        bv = gval - ROWR(DG1, rr)[c >> 1];
    }
    const long o = (long)(rr + a.top) * a.os + (a.left + c);
    a.red[o] = sse_max(65535.f * rv, 0.f);
    a.blue[o] = sse_max(65535.f * bv, 0.f);
    a.green[o] = sse_max(gval * 65535.f, 0.f);
}

// ---------------------------------------------------------------------------------------------------------------------
// The schedule.  Groups of 192 threads (three waves) G0..G4 = tid / 192 for tid < 960; wave 15 (tid >= 960) = U.
// ---------------------------------------------------------------------------------------------------------------------


// tile validity: every nyquist2 site the stream processed lies inside the reference's bounding box (L827-876)
AMZ_DEV bool tile_valid(amz_lf lds, int par, int rr1, int *box)
{
    amz_li red = (amz_li)(lds + RED_OFF) + 8 * par;
    int nystartrow = red[0] == (1 << 30) ? 0 : red[0], nyendrow = red[1], nystartcol = red[2], nyendcol = red[3];
    const bool doNyquist = nystartrow != nyendrow && nystartcol != nyendcol;
    if (doNyquist) {
        nyendrow++;
        nyendcol++;
        nystartcol -= (nystartcol & 1);
        nystartrow = imax(8, nystartrow);
        nyendrow = imin(rr1 - 8, nyendrow);
        nystartcol = imax(8, nystartcol);
        nyendcol = imin(TS - 8, nyendcol);
    } else {
        nystartrow = nyendrow = nystartcol = nyendcol = 0;       // P7, P8 and P10 do not run at all (L844)
    }
    box[0] = nystartrow; box[1] = nyendrow; box[2] = nystartcol; box[3] = nyendcol;
    if (red[5] == 0 && red[7] == 0) return true;                 // no nyquist2 site at all
    return red[4] >= nystartrow && red[5] < nyendrow && red[6] >= nystartcol && red[7] < nyendcol;
}

// reduction words of the tile with sequence parity `par` (flag box, extent of the processed nyquist2 sites)
AMZ_DEV void red_reset(amz_lf lds, int par)
{
    amz_li red = (amz_li)(lds + RED_OFF) + 8 * par;
    red[0] = 1 << 30; red[1] = 0; red[2] = TS + 1; red[3] = 0;
    red[4] = 1 << 30; red[5] = 0; red[6] = 1 << 30; red[7] = 0;
}
AMZ_DEV void seq_begin(amz_lf lds, int tid)
{
    if (tid == 0) {
        red_reset(lds, 0);
        red_reset(lds, 1);
        amz_li red = (amz_li)(lds + RED_OFF);
        red[16] = 0; red[17] = 0;
        hot_reset(lds, 0); hot_reset(lds, 1);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The tile sequence.  A workgroup streams its tiles back to back: tile k occupies the global rows 160 k .. 160 k + 159, and a
// stage at offset "off" handles global rows (2T - off, 2T - off + 1) at global step T -- so while the late stages finish tile k
// the early stages already work on tile k + 1 (no pipeline fill / drain per tile: 80 steps per tile instead of 99).  At most
// two tiles are in flight (the deepest offset is 38 rows); the loader prefetches one step ahead and needs the next one too.
// ---------------------------------------------------------------------------------------------------------------------
struct TileRef { int top, left, rr1, gbase, tile, box; };   // tile: index in the frame, bit 30 = second attempt (the box is then the true one); box: ny_pack
constexpr int TILE_REDO = 1 << 30;
AMZ_DEV int tile_index(const TileRef &t) { return t.tile < 0 ? t.tile : (t.tile & (TILE_REDO - 1)); }
AMZ_DEV bool tile_redo(const TileRef &t) { return t.tile >= 0 && (t.tile & TILE_REDO) != 0; }
struct TileSeq { TileRef back, front, next; int t2; };      // t2 = 2 T, the global row the current step starts at (set by the driver per step)
AMZ_DEV TileArgs with_tile(const TileArgs &frame, const TileRef &t, int t2)
{
    TileArgs a = frame;
    a.top = t.top; a.left = t.left; a.rr1 = t.rr1; a.gbase = t.gbase;
    a.rbase = t2 - t.gbase; a.g0 = t2; a.pos = POS_OFF * 4 + ((t2 & 2) ? POS_PAIRS * 4 : 0);
    a.ny_box = t.box;
    return a;
}
// the tile a stage at global row G works on
AMZ_DEV TileArgs stage_tile(const TileArgs &frame, const TileSeq &q, int G) { return with_tile(frame, G >= q.front.gbase ? q.front : q.back, q.t2); }
#define AMZ_STAGE(fn, off, ...)                                              \
    {                                                                        \
        const int G_ = 2 * T - (off);                                        \
        const TileArgs A_ = stage_tile(frame, q, G_);                        \
        fn(lds, A_, A_.rbase - (off), __VA_ARGS__);                          \
    }

// load: the CFA rows travel global memory -> LDS staging buffer (asynchronous LDS-DMA, issued one step ahead, no register in
// between: values kept in registers across the step loop made the compiler wait for the loads in the step that issued them) ->
// ring (scaled by 1/65535, L205-334).  Thread c of the loader role puts the two rows fetched during the previous step into the ring
// and starts the fetch of the next two (of this tile or, at its end, the first two of the next tile).
AMZ_DEV void stage_fetch(amz_lf lds, const TileArgs &b, int buf, int n0, int n1, int c)
{
    amz_gcf p0 = b.raw + ((long)src_row(b, n0, c) * b.rs + src_col(b, n0, c));
    amz_gcf p1 = b.raw + ((long)src_row(b, n1, c) * b.rs + src_col(b, n1, c));
#ifdef AMZ_EMUL
    lds[STG_OFF + (buf * 2 + 0) * STG_ROW + c] = *p0;
    lds[STG_OFF + (buf * 2 + 1) * STG_ROW + c] = *p1;
#else
    // one dword per lane to (wave-uniform LDS base) + 4 * lane
    const int base = STG_OFF + buf * 2 * STG_ROW + (c & ~63);
    __builtin_amdgcn_global_load_lds(p0, lds + base, 4, 0, 0);
    __builtin_amdgcn_global_load_lds(p1, lds + base + STG_ROW, 4, 0, 0);
#endif
}
AMZ_DEV void st_load(amz_lf lds, const TileArgs &frame, const TileSeq &q, int T, int c, ThreadRegs &rg)
{
    if (c >= TS) return;
    const TileArgs a = with_tile(frame, q.front, q.t2);
    const int r = a.rbase;
#ifndef AMZ_EMUL
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's own DMA of the previous step has landed
#endif
    if (r < a.rr1) {
        const int sb = STG_OFF + (T & 1) * 2 * STG_ROW;
        ROWW(CFA, r)[c] = lds[sb + c] / 65535.f;
        if (r + 1 < a.rr1) ROWW(CFA, r + 1)[c] = lds[sb + STG_ROW + c] / 65535.f;
    }
    // unconditional (row index clamped)
    const bool nxt = r + 2 >= TS;
    const TileArgs b = with_tile(frame, nxt ? q.next : q.front, q.t2);
    const int lim = b.rr1 > 0 ? b.rr1 - 1 : 0;
    stage_fetch(lds, b, (T + 1) & 1, imin(nxt ? 0 : r + 2, lim), imin(nxt ? 1 : r + 3, lim), c);
}
AMZ_DEV void st_load_first(amz_lf lds, const TileArgs &frame, const TileSeq &q, int c)
{
    if (c >= TS) return;
    stage_fetch(lds, with_tile(frame, q.front, 0), 0, 0, 1, c);
}

// a stage that keeps a per-thread box enters a new tile (its global row reaches the front tile's first row): merge the box into
// the finished tile's reduction words
AMZ_DEV void bb_tile_change(amz_lf lds, const TileSeq &q, int T, int off, int base, int *bb)
{
    if (2 * T - off == q.front.gbase && q.front.gbase > 0) {
        bb_flush(lds, base + 8 * ((q.back.gbase / TS) & 1), bb);
        bb_reset(bb);
    }
}

// Roles.  A column role needs three waves (columns 0..63, 64..127, 128..191: part 0..2); a few roles are one wave.  Waves w, w+4,
// w+8, w+12 of a workgroup share a SIMD, and a sub-step lasts as long as its busiest SIMD issues VALU instructions, so the table
// spreads each sub-step's roles such that the four SIMDs carry about the same number of instructions (counts per wave and step
// in DESIGN.md section 10: P2 358, P5+load 291, P12 390, P4 267, site classification 150, P8 766 / 650; P3 304 per row,
// P16+output 351, P1+P11 163, P9 272, P13 + P14 sites 486, P7 + lists + P10 sites 365).
enum RoleA { A_P2 = 0, A_P5L, A_P12, A_P4, A_LIGHT, A_P8 };
enum RoleB { B_P3R0 = 0, B_P3R1, B_P16OUT, B_P1P11, B_P9, B_P13_P14, B_P7_P10, B_P8 };
struct WaveRole { int a, apart, b, bpart; };
AMZ_DEV WaveRole wave_role(int wave)
{
    // Round 6: a wave keeps ONE pair of roles (a, b) for the whole kernel and every pair has a step loop of its own (amaze_stream.hip), so
    // the pairs are fixed -- loop 0: P2 + P16OUT, 1: P5L + P3R0, 2: P12 + P3R1, 3: LIGHT (+ the ring position table) + P1P11 + P7, each on
    // three waves (parts 0..2; part 0 of loop 3 leads: tile counter, redo queue), 4 / 5 / 6: P4 (part 0 / 1 / 2) + one of the single-wave jobs
    // P13+P14 / site list + P10 / P9, 7: the P8 wave -- and only their placement on the SIMDs is free (waves w, w+4, w+8, w+12 share one).
    // The table is the best of 300 random placements followed by a pair-swap descent, timed on the 45 MP benchmark frame
    // (scripts/amz_roles_search.py; -5 % against "LIGHT x 3 + P8 | one wave of each other loop per SIMD"):
    //   SIMD class 0: P2.1 P5L.0 P5L.2 P4.2+P9 | class 1: P2.0 P4.1+LIST+P10 P12.0 P5L.1 | class 2: P2.2 P12.1 L.2 P4.0+P13P14 | class 3: L.1 P12.2 L.0 P8
    const unsigned char tab[16] = {1, 0, 2, 13, 4, 20, 9, 10, 6, 8, 14, 12, 24, 5, 16, 28};       // by wave: loop << 2 | part
    const int ka[8] = {A_P2, A_P5L, A_P12, A_LIGHT, A_P4, A_P4, A_P4, A_P8};
    const int kb[8] = {B_P16OUT, B_P3R0, B_P3R1, B_P1P11, B_P13_P14, B_P7_P10, B_P9, B_P8};
    const int lp = tab[wave] >> 2, pt = tab[wave] & 3;
    WaveRole r;
    r.a = ka[lp]; r.b = kb[lp];
    r.apart = lp >= 4 && lp < 7 ? lp - 4 : pt; r.bpart = lp >= 4 ? 0 : pt;
    return r;
}
constexpr int LOADER_ROLE = A_P4;

// sub-step a of a column role; c = 64 * part + lane
AMZ_DEV void substep_a(amz_lf lds, const TileArgs &frame, const TileSeq &q, int T, int role, int c, ThreadRegs &rg)
{
    switch (role) {
    case A_P2:
        AMZ_STAGE(st_p2, 6, c)
        AMZ_STAGE(st_p2, 5, c)
        break;
    case A_P5L:
        bb_tile_change(lds, q, T, 12, 0, rg.bb);
        AMZ_STAGE(st_p5, 12, c, rg)
        break;
    case A_P12:
        AMZ_STAGE(st_p12, 24, c)
        break;
    case A_LIGHT:
        AMZ_STAGE(st_p1014_light, 30, c)
        pos_produce(lds, T, c, rg);
        break;
    case A_P4:
        AMZ_STAGE(st_p4, 14, c)
        st_load(lds, frame, q, T, c, rg);       // (round 6: the loader rides on the lightest column role of sub-step a, it used to be P5's)
        break;
    default: break;     // A_P8: p8_step_a (it carries registers into sub-step b, so the driver calls it)
    }
}

// the column roles of sub-step b
AMZ_DEV void substep_b_threads(amz_lf lds, const TileArgs &frame, const TileSeq &q, int T, int role, int c)
{
    switch (role) {
    case B_P3R0: AMZ_STAGE(st_p3, 8, c) break;
    case B_P3R1: AMZ_STAGE(st_p3, 7, c) break;
    case B_P16OUT:
        AMZ_STAGE(st_p16, 36, c)
        AMZ_STAGE(st_out, 40, c)
        AMZ_STAGE(st_out, 39, c)
        break;
    case B_P1P11:
        AMZ_STAGE(st_p1, 2, c)
        AMZ_STAGE(st_p1, 1, c)
        AMZ_STAGE(st_p11, 20, c)
        AMZ_STAGE(st_p7, 14, c)         // (192 items: one per lane of the role's three waves)
        break;
    default: break;
    }
}
// wave 15
AMZ_DEV void p8_step_a(amz_lf lds, const TileArgs &frame, const TileSeq &q, int T, int lane, P8Regs &pr, int *bb)
{
    bb_tile_change(lds, q, T, 22, 4, bb);
    const int G = 2 * T - 22;
    const TileArgs a = stage_tile(frame, q, G);
    p8_wave_a(lds, a, T, a.rbase - 22, lane, pr);
}
AMZ_DEV void p8_step_b(amz_lf lds, const TileArgs &frame, const TileSeq &q, int T, int lane, P8Regs &pr, int *bb)
{
    const int G = 2 * T - 22;
    const TileArgs a = stage_tile(frame, q, G);
    p8_wave_b(lds, a, T, a.rbase - 22, lane, pr, bb);
}
// the last stage (output, offset 38) has just left tile q.back: is the tile valid?  (called by one thread, between barriers)
// (one step after the output stage wrote its last rows of q.back: at that earlier step every wave drains its global stores, see the
// driver -- the tile's pixels must have left this XCD's L2 before another workgroup may write them again)
AMZ_DEV bool tile_done(const TileSeq &q, int T) { return 2 * T - LAST_OFF - 2 == q.front.gbase && q.front.gbase > 0 && q.back.rr1 > 0; }
AMZ_DEV bool tile_drain(const TileSeq &q, int T) { return 2 * T - LAST_OFF == q.front.gbase && q.front.gbase > 0 && q.back.rr1 > 0; }
AMZ_DEV void tile_ref_set(TileRef &t, int k, int tile, int top, int left, int rr1)
{
    t.gbase = TS * k; t.tile = tile; t.top = top; t.left = left; t.rr1 = rr1;
    t.box = ny_pack(8, rr1 > 8 ? rr1 - 8 : 0, 8, TS - 8);
}
AMZ_DEV void tile_ref_none(TileRef &t, int k) { tile_ref_set(t, k, -1, 0, 0, 0); }
constexpr int STEPS_PER_TILE = TS / 2;
constexpr int TAIL_STEPS = LAST_OFF / 2 + 2;   // after the last tile's rows were loaded: the deepest stage, the store drain, the last validity check

} // namespace amz
