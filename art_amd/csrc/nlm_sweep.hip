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
//                reference's association (left + up) - (upleft - s) (nlmeans.cc:192-204); `up` of the lane's first row comes from
//                the previous lane by a wave shift, everything else from registers.  S goes to an 8-column LDS ring, the
//                four-corner box sum (L219/236) of the pixel whose lower-right corner was just completed goes to the chunk ring;
//   accumulate : ALL threads walk the chunk's pixels once: mask, SW and the weighted sum are loaded once, the eleven offsets are
//                applied in the reference's order (tx ascending inside ty; weight from the exp LUT, vector / scalar lane forms of
//                L213-243) and both accumulators are stored once.
// Then the final estimate (L252-273).  Per offset that is ~3 B per pixel of global traffic.  Same arithmetic per pixel, same
// order as the reference: bit-identical.
#include <hip/hip_runtime.h>
#include "devmath.h"
#include "kernels.h"

namespace artgpu {

namespace {
constexpr int TS = 150;          // reference tile size
constexpr int RPL = 3;           // tile rows per lane
constexpr int RP = 151;          // ring pitch (floats per step)
constexpr int SCOLS = 8;         // S ring depth in columns (needs >= 2*pr + 4)
constexpr int SP = 153;          // S ring pitch

// c ? a : b on the bit patterns: always a select, never a branch
__device__ __forceinline__ float bsel(bool c, float a, float b)
{
    const int m = -(int)c;
    return __int_as_float((__float_as_int(a) & m) | (__float_as_int(b) & ~m));
}
__device__ __forceinline__ void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }
} // namespace


namespace {
constexpr int G_CH = 8;                 // anti-diagonal steps per chunk
constexpr int G_NW = 11;                // waves per workgroup = offsets in flight (2 * 5 + 1)
constexpr int G_NT = G_NW * 64;
constexpr int G_SB = 21;                // strip_b columns: x from (d0 - row/RPL - 8)
constexpr int G_SA = G_CH + 1;          // strip_a row pitch (odd: the sweep reads three rows per lane)
constexpr int G_LDS_FLOATS = 8192 + G_NW * G_CH * RP + G_NW * SCOLS * SP + TS * G_SB + (TS + 42) * G_SA;
} // namespace

__global__ void __launch_bounds__(G_NT) nlm_group_kernel(NlmArgs a)
{
    extern __shared__ float g_lds[];
    float *const explut = g_lds;
    float *const cring_all = explut + 8192;
    float *const sring_all = cring_all + G_NW * G_CH * RP;
    float *const strip_b = sring_all + G_NW * SCOLS * SP;
    float *const strip_a = strip_b + TS * G_SB;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < 8192; i += G_NT) explut[i] = a.explut[i];
    float *const cring = cring_all + wave * (G_CH * RP), *const sring = sring_all + wave * (SCOLS * SP);
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
    const int pr = a.patch_radius, sr = a.search_radius, pr2 = 2 * pr, nt = 2 * sr + 1;
    const int xx0 = start_x + border, xvec_end = end_x - border - 3;
    const int nvec = xvec_end > xx0 ? (xvec_end - xx0 + 3) / 4 * 4 : 0;
    const int nsteps = TW + (TH + RPL - 1) / RPL - 1;
    const int row0 = lane * RPL;
    const bool lane_has_rows = row0 < TH;
    const bool has1 = row0 + 1 < TH, has2 = row0 + 2 < TH;
    const bool sweeper = wave < nt;
    const int tx = wave - sr;

    // One flat loop over (ty, chunk) so that the global loads of iteration it+1 (its source strips) and the accumulator loads of
    // iteration it are in flight while the waves sweep iteration it.
    constexpr int NPB = (TS * G_SB + G_NT - 1) / G_NT, NPA = (TS * G_CH + G_NT - 1) / G_NT;
    const int nchunks = (nsteps + G_CH - 1) / G_CH, niter = nt * nchunks;
    float pb[NPB], pa[NPA];
    auto fetch_strips = [&](int it) {
        const int ty = it / nchunks - sr, d0 = (it % nchunks) * G_CH;
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
            const int row = e / G_CH, s = e - row * G_CH;
            const int xx = d0 + s - row / RPL;
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
    fetch_strips(0);
    float left0 = 0.f, left1 = 0.f, left2 = 0.f, upleft0 = 0.f, s2_latest = 0.f;
    for (int it = 0; it < niter; ++it) {
        const int d0 = (it % nchunks) * G_CH;
        if (d0 == 0) { left0 = left1 = left2 = upleft0 = s2_latest = 0.f; }     // a new search row: new integral images
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
        size_t aoo[NPA], aio[NPA];
#pragma unroll
        for (int q = 0; q < NPA; ++q) {
            const int t = tid + q * G_NT;
            const int row = t / G_CH, s = t - row * G_CH;
            const int xx = d0 + s - row / RPL;
            const int sty = row - pr, stx = xx - pr;
            aok[q] = t < TS * G_CH && row < TH && xx < TW && sty >= border && sty < TH - border && stx >= border && stx < TW - border;
            am[q] = asw[q] = aim[q] = 0.f; aoo[q] = aio[q] = 0;
            if (aok[q]) {
                const int y = sty + start_y - border, x = stx + start_x - border;
                aoo[q] = (size_t)y * W + x;
                aio[q] = (size_t)y * a.img_stride + x;
                am[q] = mask[aoo[q]]; asw[q] = SW[aoo[q]]; aim[q] = img[aio[q]];
            }
        }
        if (sweeper) {
            // ---- sweep (the recurrence, written without branches: every LDS read of a step is an unconditional load issued up
            //      front -- any address stays inside this kernel's LDS block -- and conditions only select values or mask stores, so a
            //      step is one LDS round trip; the row above comes from the previous lane through a DPP wave shift)
            for (int s = 0; s < G_CH; ++s) {
                const int xx = d0 + s - lane;
                const float up0 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s2_latest), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
                const bool act = lane_has_rows && xx >= 0 && xx < TW;
                float *cr = cring + s * RP + row0;
                float *sw = sring + (xx & (SCOLS - 1)) * SP + row0;
                const float *sb = sring + ((xx - pr2) & (SCOLS - 1)) * SP + row0;
                // squared differences of this wave's offset straight from the staged source rows (strip_a: the pixel, strip_b: the
                // pixel at the offset); positions outside the tile are only ever computed by lanes whose results are discarded
                const float *pa0 = strip_a + row0 * G_SA + s, *pb0 = strip_b + row0 * G_SB + s + tx + 8;
                const float df0 = pa0[0] - pb0[0], df1 = pa0[G_SA] - pb0[G_SB], df2 = pa0[2 * G_SA] - pb0[2 * G_SB];
                const float sc0 = df0 * df0, sc1 = df1 * df1, sc2 = df2 * df2;
                // corners of the three box sums: column xx - 2pr at rows -2pr.. and 0.., and this column at the rows above
                float ca[3], cb[3], cc[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) { ca[k] = sb[k - pr2]; cb[k] = sb[k]; cc[k] = sw[k - pr2]; }
                // The reference's first-column forms (up + s, above + s) are what the general form gives when everything to the left is
                // +0: (0 + u) - (0 - s) = u + s with the same single rounding.  The state of a lane that has not started (xx < 0) is
                // kept at 0 below, so only S(0,0) = 0 needs its own case.  Arms are chosen with bit masks (no branches).
                const float st0 = bsel(row0 == 0, bsel(xx == 0, 0.f, left0 + sc0), (left0 + up0) - (upleft0 - sc0));
                const float st1 = bsel(has1, (left1 + st0) - (left0 - sc1), 0.f);
                const float st2 = bsel(has2, (left2 + st1) - (left1 - sc2), 0.f);
                if (pr2 == 2) cc[2] = st0;            // index 0 of this column is this step's own result (2pr >= 4: all rows above)
                const float r0 = ((st0 + ca[0]) - cb[0]) - cc[0];
                const float r1 = ((st1 + ca[1]) - cb[1]) - cc[1];
                const float r2 = ((st2 + ca[2]) - cb[2]) - cc[2];
                if (act) {
                    sw[0] = st0; sw[1] = st1; sw[2] = st2;
                    if (xx >= pr2) {
                        if (row0 >= pr2) cr[0] = r0;
                        if (has1 && row0 + 1 >= pr2) cr[1] = r1;
                        if (has2 && row0 + 2 >= pr2) cr[2] = r2;
                    }
                }
                // state: zero until the lane starts; what a lane holds after its last column is never read
                const int started = -(int)(xx >= 0);
                upleft0 = __int_as_float(__float_as_int(up0) & started);
                left0 = __int_as_float(__float_as_int(st0) & started);
                left1 = __int_as_float(__float_as_int(st1) & started);
                left2 = __int_as_float(__float_as_int(st2) & started);
                s2_latest = left2;
                wave_fence();
            }
        }
        TICK(4);
        __syncthreads();
        // ---- accumulate: every thread, the chunk's pixels once, the row's offsets in order
#pragma unroll
        for (int q = 0; q < NPA; ++q) {
            if (!aok[q]) continue;
            const int t = tid + q * G_NT;
            const int row = t / G_CH, s = t - row * G_CH;
            const int xx = d0 + s - row / RPL;
            const int sty = row - pr, px = xx - pr + start_x;
            const bool vec = (px - xx0) < nvec;
            const float m = am[q];
            float swv = asw[q], imv = aim[q];
            const float *sbp = strip_b + sty * G_SB + (s - row / RPL + sty / RPL - pr + 8 - sr);    // + w: the sample at offset w
            const float *crp = cring_all + s * RP + row;
            // weight of offset w in the reference's two lane forms (vector lanes L213-228, scalar tail L229-243), added in offset order
            auto add_vec = [&](int w) {
                const float dd = sse_max(crp[w * (G_CH * RP)], 0.f) * m;
                const float clamped = sse_max(sse_min(8190.f, dd), 0.f);
                const int idx = (int)clamped;
                const float diff = sse_max(sse_min(8191.f, dd), 0.f) - (float)idx;
                const float weight = (diff * explut[idx + 1]) + ((1.f - diff) * explut[idx]);
                swv = swv + weight;
                imv = imv + (weight * sbp[w]);
            };
            auto add_any = [&](int w) {
                if (vec) { add_vec(w); return; }
                const float dd = std_max(crp[w * (G_CH * RP)], 0.f) * m;
                float weight;
                if (dd < 0.f || !(dd == dd)) weight = explut[0];
                else if (dd > 8190.f) weight = explut[8191];
                else {
                    const int idx = (int)dd;
                    const float diff = dd - (float)idx;
                    const float p1 = explut[idx], p2 = explut[idx + 1] - p1;
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
            SW[aoo[q]] = swv;
            img[aio[q]] = imv;
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
    return 2 * a.patch_radius + 4 <= SCOLS && a.border * 2 < TS && a.search_radius <= 5 && a.patch_radius <= 2 && a.patch_radius >= 1;
}

hipError_t launch_nlm_group(const NlmArgs &a, hipStream_t s)
{
    constexpr size_t dyn = (size_t)G_LDS_FLOATS * sizeof(float);
    static_assert(dyn <= 160 * 1024, "LDS budget");
    if (hipError_t e = dyn_lds_once(reinterpret_cast<const void *>(&nlm_group_kernel), (int)dyn); e != hipSuccess) return e;
    hipLaunchKernelGGL(nlm_group_kernel, dim3(a.ntiles_x * a.ntiles_y), dim3(G_NT), dyn, s, a);
    return hipGetLastError();
}

} // namespace artgpu
