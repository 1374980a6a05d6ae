// art_amd/csrc/nlm_sweep.hip -- NL-means tile kernel (reference: rtengine/nlmeans.cc:138-276).
//
// THIS TRANSLATION UNIT IS COMPILED WITH -fgpu-flush-denormals-to-zero: the reference runs the whole tile loop with
// MXCSR flush-to-zero on (nlmeans.cc:157-160), so every fp32 result below is flushed by the hardware instead of by
// explicit ftz() calls.  (Flushing denormal INPUTS as well changes nothing here: every operand is either the result
// of a flushed operation, a normal-range value, or a denormal exp-LUT entry whose products and sums are themselves
// flushed by SSE's FTZ -- see DESIGN.md.)
//
// One WORKGROUP per reference tile (150x150, stride 150 - 2*border); the 2*sr+1 offsets of one search row (same ty) are in flight
// at once, one wave each.  For every chunk of 8 anti-diagonal steps:
//   stage      : the source rows the chunk needs go to LDS ONCE (strip_a: the pixels themselves, strip_b: the rows shifted by ty
//                with a 21-column window that covers every tx of the row and the patch-centre shift); all eleven waves take their
//                squared differences and, later, the weighted sample from there;
//   sweep      : lane l of a wave owns tile rows 3l..3l+2 and walks x = d - l (skewed coordinates).  The integral image uses the
//                reference's association (left + up) - (upleft - s) (nlmeans.cc:192-204).  S never leaves the registers: a lane keeps
//                the last eight steps of its three rows (and of the row above, which it receives anyway) in a ring whose index is
//                the step number -- static once the eight steps of a chunk are unrolled -- and everything a step needs from the
//                rows above (the row above itself and the two upper corners of the three box sums) is some entry of the PREVIOUS
//                lane's ring, one DPP wave shift away: lane l-1 is one column ahead, so "column x of its rows" is its previous
//                step and "column x - 2pr" the one 2pr before that.  A step is straight-line code without an LDS round trip on
//                its dependent chain; the four-corner box sum (L219/236) of the pixel whose lower-right corner was just completed
//                goes to the chunk ring in LDS;
//   accumulate : ALL threads walk the chunk's pixels once: mask, SW and the weighted sum are loaded once, the eleven offsets are
//                applied in the reference's order (tx ascending inside ty; weight from the exp LUT, vector / scalar lane forms of
//                L213-243) and both accumulators are stored once.
// Then the final estimate (L252-273).  Per offset that is ~3 B per pixel of global traffic.  Same arithmetic per pixel, same
// order as the reference: bit-identical.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "devmath.h"
#include "kernels.h"

namespace artgpu {

namespace {
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int TS = 150;          // reference tile size
constexpr int RPL = 3;           // tile rows per lane
constexpr int RP = 156;          // chunk ring pitch (floats per step): >= 152 (a lane's three rows are stored whether they exist or not), and
                                 // = 4 mod 8 so that the accumulate pass (32 lanes = 4 rows x 8 steps per LDS cycle) touches 32 banks

// c ? a : b on the bit patterns: always a select, never a branch
__device__ __forceinline__ float bsel(bool c, float a, float b)
{
    const int m = -(int)c;
    return __int_as_float((__float_as_int(a) & m) | (__float_as_int(b) & ~m));
}
// the previous lane's value (0 for lane 0): one DPP operand, no LDS
__device__ __forceinline__ float lane_above(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, true));
}
// x - (the previous lane's v) for three pairs, the wave shift folded into the subtraction as its DPP operand (the compiler folds the shift
// into additions only and leaves a v_mov_b32_dpp in front of a subtraction).  Inline assembly is outside the compiler's hazard
// bookkeeping: a VALU result needs two wait states before a DPP instruction may read it, hence the s_nop.
__device__ __forceinline__ void sub_lane_above3(float &r0, float &r1, float &r2, float x0, float x1, float x2, float v0, float v1, float v2)
{
    asm("s_nop 1\n\t"
        "v_subrev_f32_dpp %0, %6, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_subrev_f32_dpp %1, %7, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_subrev_f32_dpp %2, %8, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2) : "v"(x0), "v"(x1), "v"(x2), "v"(v0), "v"(v1), "v"(v2));
}
// _mm_max_ps(x, 0) for an x that is the result of an arithmetic instruction (never a signalling NaN): NaN -> 0, like the compare-and-select
// form, in one instruction (the compiler's fmaxf would canonicalise the LDS operand first)
__device__ __forceinline__ float max0(float x)
{
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}
} // namespace


namespace {
constexpr int G_CH = 8;                 // anti-diagonal steps per chunk
constexpr int G_NW = 11;                // waves per workgroup = offsets in flight (2 * 5 + 1)
constexpr int G_NT = G_NW * 64;
constexpr int G_SB = 21;                // strip_b columns: x from (d0 - row/RPL - 8)
constexpr int G_SA = G_CH + 1;          // strip_a row pitch (odd: the sweep reads three rows per lane)
constexpr int G_NLUT = 8192;            // exp table entries; in LDS as 8192 pairs (e[i], e[i+1]): one 8-byte read per weight
constexpr int G_LDS_FLOATS = 2 * G_NLUT + G_NW * G_CH * RP + TS * G_SB + (TS + 42) * G_SA;
static_assert(G_CH == 8, "the register ring of the sweep is indexed by the step inside a chunk");
} // namespace

template <int PR>
__global__ void __launch_bounds__(G_NT) nlm_group_kernel(NlmArgs a)
{
    extern __shared__ float g_lds[];
    f32x2 *const exppair = reinterpret_cast<f32x2 *>(g_lds);
    float *const cring_all = g_lds + 2 * G_NLUT;
    float *const strip_b = cring_all + G_NW * G_CH * RP;
    float *const strip_a = strip_b + TS * G_SB;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < G_NLUT; i += G_NT) exppair[i] = f32x2{a.explut[i], a.explut[min(i + 1, G_NLUT - 1)]};
    float *const cring = cring_all + wave * (G_CH * RP);
    const float *__restrict__ src = a.src;
    const float *__restrict__ mask = a.mask;
    float *__restrict__ SW = a.SW;
    float *__restrict__ img = a.img;

    const int tile = blockIdx.x;
    const int tile_y = tile / a.ntiles_x, tile_x = tile - tile_y * a.ntiles_x;
    const int border = a.border, WW = a.WW, HH = a.HH, W = a.W;
    const int step = TS - 2 * border;
    const int start_y = tile_y * step, end_y = min(start_y + TS, HH), TH = end_y - start_y;
    const int start_x = tile_x * step, end_x = min(start_x + TS, WW), TW = end_x - start_x;
    constexpr int pr = PR, pr2 = 2 * PR;
    const int sr = a.search_radius, nt = 2 * sr + 1;
    const int xx0 = start_x + border, xvec_end = end_x - border - 3;
    const int nvec = xvec_end > xx0 ? (xvec_end - xx0 + 3) / 4 * 4 : 0;
    const int nsteps = TW + (TH + RPL - 1) / RPL - 1;
    const int row0 = lane * RPL;
    const bool lane_has_rows = row0 < TH;
    const bool sweeper = wave < nt;
    const int tx = wave - sr;

    // One flat loop over (ty, chunk) so that the global loads of iteration it+1 (its source strips) and the accumulator loads of
    // iteration it are in flight while the waves sweep iteration it.
    constexpr int NPB = (TS * G_SB + G_NT - 1) / G_NT, NPA = (TS * G_CH + G_NT - 1) / G_NT;
    const int nchunks = (nsteps + G_CH - 1) / G_CH, niter = nt * nchunks;
    float pb[NPB], pa[NPA];
    // A tile whose strips (the 21-column window of every search row, the skew of the anti-diagonals) lie inside the padded image -- nine
    // tiles in ten -- fetches them without clamps: a per-thread element offset that never changes, added by the load itself to a
    // uniform base that moves with (ty, d0).  strip_a's zero fill outside the tile is not needed: lanes left of the tile are switched
    // off in the sweep and what lanes right of it compute is never read.
    constexpr int SKEW = (TS - 1) / RPL;
    const bool interior = TH == TS && TW == TS && start_y >= sr && start_y + TS - 1 + sr <= HH - 1 && start_x >= 8 + SKEW &&
                          start_x + (nchunks - 1) * G_CH - 8 + G_SB - 1 <= WW - 1;
    unsigned pbo[NPB], pao[NPA];
#pragma unroll
    for (int q = 0; q < NPB; ++q) {
        const int e = min(tid + q * G_NT, TS * G_SB - 1);
        const int row = e / G_SB, k = e - row * G_SB;
        pbo[q] = (unsigned)(row * WW + k + SKEW - row / RPL) * 4u;
    }
#pragma unroll
    for (int q = 0; q < NPA; ++q) {
        const int e = min(tid + q * G_NT, TS * G_CH - 1);
        const int row = e / G_CH, st = e - row * G_CH;
        pao[q] = (unsigned)(row * WW + st + SKEW - row / RPL) * 4u;
    }
    // a uniform base plus a 32-bit byte offset per lane: the addressing form the load / store instructions have (no 64-bit vector arithmetic)
    auto at = [](float *base, unsigned byte_off) { return reinterpret_cast<float *>(reinterpret_cast<char *>(base) + byte_off); };
    auto atc = [](const float *base, unsigned byte_off) { return reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + byte_off); };
    auto fetch_strips = [&](int it) {
        const int ty = it / nchunks - sr, d0 = (it % nchunks) * G_CH;
        if (interior) {
            const float *const ub = src + ((size_t)(ty + start_y) * WW + (d0 - 8 + start_x - SKEW));
            const float *const ua = src + ((size_t)start_y * WW + (d0 + start_x - SKEW));
#pragma unroll
            for (int q = 0; q < NPB; ++q) pb[q] = *atc(ub, pbo[q]);
#pragma unroll
            for (int q = 0; q < NPA; ++q) pa[q] = *atc(ua, pao[q]);
            return;
        }
#pragma unroll
        for (int q = 0; q < NPB; ++q) {
            const int e = tid + q * G_NT;
            const int row = e / G_SB, k = e - row * G_SB;
            float v = 0.f;
            if (e < TS * G_SB && row < TH) {
                const int gy = min(max(row + ty + start_y, 0), HH - 1);
                const int gx = min(max(d0 - row / RPL - 8 + k + start_x, 0), WW - 1);
                v = src[(size_t)gy * WW + gx];
            }
            pb[q] = v;
        }
#pragma unroll
        for (int q = 0; q < NPA; ++q) {
            const int e = tid + q * G_NT;
            const int row = e / G_CH, st = e - row * G_CH;
            const int xx = d0 + st - row / RPL;
            float v = 0.f;
            if (e < TS * G_CH && row < TH && xx >= 0 && xx < TW) {
                const int gy = min(max(row + start_y, 0), HH - 1), gx = min(max(xx + start_x, 0), WW - 1);
                v = src[(size_t)gy * WW + gx];
            }
            pa[q] = v;
        }
    };
#ifdef NLM_PROFILE
    long long tacc[6] = {0, 0, 0, 0, 0, 0}, tp = wall_clock64();
#define TICK(k) do { long long now_ = wall_clock64(); tacc[k] += now_ - tp; tp = now_; } while (0)
#else
#define TICK(k) do { } while (0)
#endif
    // per-thread constants of the accumulate pass: thread t of slot q holds pixel (row, step) = (t / 8, t % 8) of the chunk, i.e. tile
    // column d0 + step - row / 3; everything but d0 is fixed for the whole tile
    bool arow[NPA];
    int axc[NPA];
    unsigned aoc[NPA], aic[NPA];
#pragma unroll
    for (int q = 0; q < NPA; ++q) {
        const int t = tid + q * G_NT;
        const int row = t / G_CH, st = t - row * G_CH;
        const int sty = row - pr;
        arow[q] = t < TS * G_CH && sty >= border && sty < TH - border;
        axc[q] = st - row / RPL - pr - border;                   // stx - border = d0 + this
        // byte offsets of the pixel at d0 = 0, modulo 2^32: together with 4 * d0 they are in range whenever the pixel is valid
        const int y = max(sty + start_y - border, 0), x0 = axc[q] + start_x;
        aoc[q] = ((unsigned)y * (unsigned)W + (unsigned)x0) * 4u;
        aic[q] = ((unsigned)y * (unsigned)a.img_stride + (unsigned)x0) * 4u;
    }
    const unsigned ncol = (unsigned)max(TW - 2 * border, 0);       // a sliver tile at the right edge (narrower than two borders) writes nothing
    fetch_strips(0);
    // the sweep's ring: S of the lane's three rows and of the row above at the last eight steps, indexed by step & 7
    float h0[G_CH], h1[G_CH], h2[G_CH], hu[G_CH];
#pragma unroll
    for (int j = 0; j < G_CH; ++j) h0[j] = h1[j] = h2[j] = hu[j] = 0.f;
    for (int it = 0; it < niter; ++it) {
        const int d0 = (it % nchunks) * G_CH;
        if (d0 == 0) {                                                           // a new search row: new integral images
#pragma unroll
            for (int j = 0; j < G_CH; ++j) h0[j] = h1[j] = h2[j] = hu[j] = 0.f;
        }
        __syncthreads();            // the previous iteration's accumulate pass is done with the strips and the rings
        TICK(0);
#pragma unroll
        for (int q = 0; q < NPB; ++q) { const int e = tid + q * G_NT; if (e < TS * G_SB) strip_b[e] = pb[q]; }
#pragma unroll
        for (int q = 0; q < NPA; ++q) { const int e = tid + q * G_NT; if (e < TS * G_CH) strip_a[(e / G_CH) * G_SA + (e % G_CH)] = pa[q]; }
        __syncthreads();
        TICK(1);
        TICK(2);
        TICK(3);
        // ---- loads that do not depend on the sweep: the next iteration's strips, this iteration's accumulators
        if (it + 1 < niter) fetch_strips(it + 1);
        bool aok[NPA];
        float am[NPA], asw[NPA], aim[NPA];
        unsigned aoo[NPA], aio[NPA];
#pragma unroll
        for (int q = 0; q < NPA; ++q) {
            aok[q] = arow[q] && (unsigned)(d0 + axc[q]) < ncol;
            aoo[q] = aoc[q] + 4u * (unsigned)d0; aio[q] = aic[q] + 4u * (unsigned)d0;
            am[q] = asw[q] = aim[q] = 0.f;
            if (aok[q]) { am[q] = *atc(mask, aoo[q]); asw[q] = *at(SW, aoo[q]); aim[q] = *at(img, aio[q]); }
        }
        if (sweeper && lane_has_rows) {
            // ---- sweep.  Step t = d0 + s puts lane l on column t - l.  A lane that has not started (t < l) is switched off, so its ring
            //      still holds the zeros of the reset: that is the "everything to the left is +0" its first column needs -- the
            //      reference's first-column forms (up + s, above + s) are what the general form gives then: (0 + u) - (0 - s) = u + s
            //      with the same single rounding, and likewise the first row with lane 0's wave shift delivering 0.  Only S(0,0) = 0
            //      is a case of its own.  A lane past its last column keeps computing: what it stores is never read (the accumulate
            //      pass reads box sums of columns < TW only), and nothing here can trap.  From step 49 on every lane with rows has
            //      started: those chunks run without the per-step test.
            const float *const pa = strip_a + row0 * G_SA, *const pb = strip_b + row0 * G_SB + tx + 8;
            float *const cr = cring + row0;
            float na0 = pa[0], na1 = pa[G_SA], na2 = pa[2 * G_SA], nb0 = pb[0], nb1 = pb[G_SB], nb2 = pb[2 * G_SB];
            auto step = [&](auto sc_, auto starting_) {
                constexpr int s = decltype(sc_)::value;
                constexpr bool starting = decltype(starting_)::value;
                constexpr int p = (s + 7) & 7, q = (s + 8 - pr2) & 7, qq = (s + 7 - pr2) & 7;
                // the strips' samples one step ahead, so that a step's LDS latency is not at the head of its dependent chain
                const float a0 = na0, a1 = na1, a2 = na2, b0 = nb0, b1 = nb1, b2 = nb2;
                if (s + 1 < G_CH) {
                    na0 = pa[s + 1]; na1 = pa[G_SA + s + 1]; na2 = pa[2 * G_SA + s + 1];
                    nb0 = pb[s + 1]; nb1 = pb[G_SB + s + 1]; nb2 = pb[2 * G_SB + s + 1];
                }
                if (starting && lane > d0 + s) return;
                const float up0 = lane_above(h2[p]);
                const float df0 = a0 - b0, df1 = a1 - b1, df2 = a2 - b2;
                float sc0 = df0 * df0;
                const float sc1 = df1 * df1, sc2 = df2 * df2;
                if (starting && s == 0 && d0 == 0) sc0 = 0.f;        // step 0: lane 0 alone, at S(0,0)
                const float st0 = (h0[p] + up0) - (hu[p] - sc0);
                const float st1 = (h1[p] + st0) - (h0[p] - sc1);
                const float st2 = (h2[p] + st1) - (h1[p] - sc2);
                // corners of the three box sums: rows k - 2pr at this column (cc) and at column - 2pr (ca), the lane's own rows at column - 2pr (cb)
                float r0, r1, r2;
                if (pr2 == 4) {         // rows -4 (two lanes up: the previous lane's "row above"), -3, -2 (the previous lane's rows 0, 1)
                    const float ca0 = lane_above(hu[qq]), ca1 = lane_above(h0[qq]), ca2 = lane_above(h1[qq]);
                    sub_lane_above3(r0, r1, r2, (st0 + ca0) - h0[q], (st1 + ca1) - h1[q], (st2 + ca2) - h2[q], hu[p], h0[p], h1[p]);
                } else {                // rows -2, -1 (the previous lane's rows 1, 2) and the lane's own row 0
                    const float ca0 = lane_above(h1[qq]), ca1 = lane_above(h2[qq]), ca2 = h0[q];
                    r0 = ((st0 + ca0) - h0[q]) - lane_above(h1[p]);
                    r1 = ((st1 + ca1) - h1[q]) - up0;
                    r2 = ((st2 + ca2) - h2[q]) - st0;
                }
                cr[s * RP] = r0; cr[s * RP + 1] = r1; cr[s * RP + 2] = r2;
                h0[s] = st0; h1[s] = st1; h2[s] = st2; hu[s] = up0;
            };
            auto chunk = [&](auto starting_) {
                step(std::integral_constant<int, 0>{}, starting_); step(std::integral_constant<int, 1>{}, starting_);
                step(std::integral_constant<int, 2>{}, starting_); step(std::integral_constant<int, 3>{}, starting_);
                step(std::integral_constant<int, 4>{}, starting_); step(std::integral_constant<int, 5>{}, starting_);
                step(std::integral_constant<int, 6>{}, starting_); step(std::integral_constant<int, 7>{}, starting_);
            };
            if (d0 < 56) chunk(std::true_type{});
            else chunk(std::false_type{});
        }
        TICK(4);
        __syncthreads();
        // ---- accumulate: every thread, the chunk's pixels once, the row's offsets in order
#pragma unroll
        for (int q = 0; q < NPA; ++q) {
            if (!aok[q]) continue;
            const int t = tid + q * G_NT;
            const int row = t / G_CH, s = t - row * G_CH;
            const int sty = row - pr;
            const bool vec = d0 + axc[q] < nvec;        // column - (start_x + border), nlmeans.cc:207-212
            const float m = am[q];
            float swv = asw[q], imv = aim[q];
            const float *sbp = strip_b + sty * G_SB + (s - row / RPL + sty / RPL - pr + 8 - sr);    // + w: the sample at offset w
            const float *crp = cring_all + s * RP + row;
            // weight of offset w in the reference's two lane forms (vector lanes L213-228, scalar tail L229-243), added in offset order
            auto add_vec = [&](int w) {
                // The reference clamps twice (L213-228): index = (int)max(min(8190, dd), 0), fraction = max(min(8191, dd), 0) - index.  One clamp
                // to [0, 8191] gives the same weight: below 8191 the integer part and the fraction (x - floor x, exact) are the reference's;
                // at 8191 it reads pair 8191 = (e[8191], e[8191]) with fraction 0 where the reference reads pair 8190 with fraction 1 --
                // 0 * e + 1 * e[8191] either way.  _mm_max_ps(_mm_min_ps(c, x), 0) is the median of (x, 0, c) for every x: a NaN gives 0
                // both ways (v_med3_f32 returns the minimum of the three when one is a NaN); -0 and +0 give the same weight.
                const float dd = max0(crp[w * (G_CH * RP)]) * m;
                const float t = __builtin_amdgcn_fmed3f(dd, 0.f, 8191.f);
                const int idx = (int)t;
                const float diff = __builtin_amdgcn_fractf(t);
                const f32x2 pr = exppair[idx] * f32x2{1.f - diff, diff};       // v_pk_mul_f32
                const float weight = pr.y + pr.x;
                swv = swv + weight;
                imv = imv + (weight * sbp[w]);
            };
            auto add_any = [&](int w) {
                if (vec) { add_vec(w); return; }
                const float dd = std_max(crp[w * (G_CH * RP)], 0.f) * m;
                float weight;
                if (dd < 0.f || !(dd == dd)) weight = exppair[0].x;
                else if (dd > 8190.f) weight = exppair[8190].y;
                else {
                    const int idx = (int)dd;
                    const float diff = dd - (float)idx;
                    const f32x2 e = exppair[idx];
                    const float p1 = e.x, p2 = e.y - p1;
                    weight = p1 + (p2 * diff);
                }
                swv = swv + weight;
                imv = imv + (weight * sbp[w]);
            };
            // Nearly every wave holds vector-form pixels only (the scalar form is the last < 4 columns of a tile): that case runs
            // without a per-offset branch, and with the full search row as straight-line code -- the exec-mask bookkeeping and the
            // loop control of the general form were a third of this pass's instructions.
            if (__builtin_amdgcn_ballot_w64(!vec) == 0) {
                if (nt == G_NW) {
#pragma unroll
                    for (int w = 0; w < G_NW; ++w) add_vec(w);
                } else {
                    for (int w = 0; w < nt; ++w) add_vec(w);
                }
            } else {
                for (int w = 0; w < nt; ++w) add_any(w);
            }
            *at(SW, aoo[q]) = swv;
            *at(img, aio[q]) = imv;
        }
        TICK(5);
    }
#ifdef NLM_PROFILE
    if (blockIdx.x == 700 && tid == 0)
        printf("nlm tile 700, %d iterations, ticks (100 MHz): top-barrier %lld  strip-write+barrier %lld  prefetch-issue %lld  sc %lld  sweep %lld  barrier+accumulate %lld\n",
               niter, tacc[0], tacc[1], tacc[2], tacc[3], tacc[4], tacc[5]);
#endif
#undef TICK
    __syncthreads();
    // final estimate (nlmeans.cc:252-273)
    const int ow = TW - 2 * border, oh = TH - 2 * border;
    if (ow > 0 && oh > 0)
        for (int t = tid; t < oh * ow; t += G_NT) {
            const int ry = t / ow, rx = t - ry * ow;
            const int y = start_y + ry, x = start_x + rx;
            const size_t io = (size_t)y * a.img_stride + x;
            const float f = 1e-5f + SW[(size_t)y * W + x];
            img[io] = (img[io] / f) * a.factor;
        }
}

bool nlm_group_supported(const NlmArgs &a)
{
    return a.border * 2 < TS && a.search_radius <= 5 && a.patch_radius <= 2 && a.patch_radius >= 1;
}

hipError_t launch_nlm_group(const NlmArgs &a, hipStream_t s)
{
    constexpr size_t dyn = (size_t)G_LDS_FLOATS * sizeof(float);
    static_assert(dyn <= 160 * 1024, "LDS budget");
    const void *const k = a.patch_radius == 2 ? reinterpret_cast<const void *>(&nlm_group_kernel<2>) : reinterpret_cast<const void *>(&nlm_group_kernel<1>);
    if (hipError_t e = dyn_lds_once(k, (int)dyn); e != hipSuccess) return e;
    if (a.patch_radius == 2) hipLaunchKernelGGL(nlm_group_kernel<2>, dim3(a.ntiles_x * a.ntiles_y), dim3(G_NT), dyn, s, a);
    else hipLaunchKernelGGL(nlm_group_kernel<1>, dim3(a.ntiles_x * a.ntiles_y), dim3(G_NT), dyn, s, a);
    return hipGetLastError();
}

} // namespace artgpu
