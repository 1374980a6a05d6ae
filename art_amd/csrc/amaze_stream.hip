// art_amd/csrc/amaze_stream.hip -- AMaZE v2 for gfx950: one 1024-thread workgroup streams a reference tile (160x160) through
// LDS ring buffers, two rows per step, two LDS-only barriers per step.  The stages, their offsets and the ring layout are in
// amaze_stream_core.h (shared with the CPU emulation of the schedule under tests/emul); this file is the device driver:
// wave roles, the wave-level row recurrences, the barrier, the tile loop and the hand-over of tiles the stream cannot take
// (Nyquist sites outside the tile's bounding box) to the arena kernel of amaze.hip.
//
// Replaces RawImageSource::amaze_demosaic_RT (reference: rtengine/amaze_demosaic_RT.cc:41-1595, x86-64 / __SSE2__ branches).
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "amaze_stream_core.h"

namespace artgpu {

namespace {

using namespace amz;

// workgroup barrier that orders LDS traffic only: __syncthreads() would also drain the output stores and the CFA prefetch
// (s_waitcnt vmcnt(0)), a memory round trip per sub-step
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
__device__ __forceinline__ void wave_order()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// P9 (L957-974): ONE wave walks the two rows of the step in order (row rr reads the updated row rr-1); 72 sites = 64 + 8 lanes
__device__ __forceinline__ void wave_p9(amz_lf lds, const TileArgs &a, int r, int lane)
{
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int rr = r + k;
        if (rr < 8 || rr >= TS - 8) continue;
        const float h0 = p9_new_weight(lds, a, rr, lane);
        float h1 = 0.f;
        if (lane < 8) h1 = p9_new_weight(lds, a, rr, 64 + lane);
        p9_site(lds, a, rr, lane, h0);
        if (lane < 8) p9_site(lds, a, rr, 64 + lane, h1);
        wave_order();
    }
}
// P13 (L1213-1223): the same scheme for pmwt
__device__ __forceinline__ void wave_p13(amz_lf lds, const TileArgs &a, int r, int lane)
{
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int rr = r + k;
        if (rr < 10 || rr >= TS - 10) continue;
        const float h0 = p13_new_weight(lds, a, rr, lane);
        float h1 = 0.f;
        if (lane < 8) h1 = p13_new_weight(lds, a, rr, 64 + lane);
        p13_site(lds, a, rr, lane, h0);
        if (lane < 8) p13_site(lds, a, rr, 64 + lane, h1);
        wave_order();
    }
}
// the nyquist2 sites of rows (r, r+1), compacted for P8 of the next step
__device__ __forceinline__ void wave_list(amz_lf lds, const TileArgs &a, int t, int lane)
{
    const int r = 2 * t - 20, buf = (t + 1) & 1;     // the list step t+1 consumes
    amz_li red = (amz_li)(lds + RED_OFF), list = (amz_li)(lds + LIST_OFF + buf * LIST_INTS);
    int n = 0;
    if (r >= 8 && r < TS - 8) {
#pragma unroll
        for (int pass = 0; pass < 3; ++pass) {
            const int c = pass * 64 + lane;
            int rr = 0;
            const bool f = nyq_site(lds, a, r, c, &rr);
            const unsigned long long m = __ballot(f);
            const int pos = n + __popcll(m & ((1ull << lane) - 1ull));
            if (f) list[pos] = (rr << 8) | c;
            n += __popcll(m);
        }
    }
    if (lane == 0) red[8 + buf] = n;
}

} // namespace

__global__ void __launch_bounds__(amz::NTHREADS)
amaze_stream_kernel(AmazeStreamArgs s)
{
    extern __shared__ float dyn_lds[];
    amz_lf lds = (amz_lf)dyn_lds;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane_ = tid & 63;
    const int grp = wave / 3;                    // 0..4: column groups of three waves; 5: wave 15
    const int c_ = tid - grp * 192;              // column 0..159 (160..191: helper lanes); wave 15: the lane

    // the single-wave roles (row recurrences, the Nyquist area sums) are the longest serial chains of a sub-step: they win the
    // issue arbitration on their SIMD, the column roles fill the gaps
    if (wave >= 12) __builtin_amdgcn_s_setprio(2);

    TileArgs a;
    a.raw = (amz_gcf)s.raw; a.rs = (long)s.raw_stride;
    a.red = (amz_gf)s.red; a.green = (amz_gf)s.green; a.blue = (amz_gf)s.blue; a.os = (long)s.out_stride;
    a.W = s.W; a.H = s.H; a.filters = s.filters; a.clip_pt = s.clip_pt; a.clip_pt8 = s.clip_pt8; a.g00 = s.g00; a.ey = s.ey;

    for (int k = blockIdx.x; k < s.ntiles; k += gridDim.x) {
        const int tile = s.tiles[k];
        const int ty = tile / s.ntx, tx = tile - ty * s.ntx;
        a.top = -16 + ty * (TS - 32);
        a.left = -16 + tx * (TS - 32);
        ThreadRegs rg;
        rg.pf0 = rg.pf1 = 0.f;
        bb_reset(rg.bb);
        P8Regs p8;
        p8.cc = -1;
        bb_reset(p8.bb);
        tile_begin(lds, tid);
        if (grp == 1) st_load_first(a, c_, rg);
        lds_barrier();
#ifdef AMZ_PROFILE
        long long ta = 0, tb = 0, tw = 0;
#define AMZ_T0 const long long t0_ = __builtin_amdgcn_s_memtime();
#define AMZ_T1(acc) { const long long t1_ = __builtin_amdgcn_s_memtime(); acc += t1_ - t0_; }
#else
#define AMZ_T0
#define AMZ_T1(acc)
#endif
        for (int t = 0; t < NSTEPS; ++t) {
            // the column / lane are made opaque per step: otherwise every role's column-derived addresses are hoisted out of the
            // step loop and kept live across all the other roles (the kernel then spills into scratch inside the loop)
            int c = c_, lane = lane_;
            asm volatile("" : "+v"(c), "+v"(lane));
            { AMZ_T0 if (grp < 5) substep_a(lds, a, t, grp, c, rg); else p8_wave_a(lds, t, lane, p8); AMZ_T1(ta) }
            { AMZ_T0 lds_barrier(); AMZ_T1(tw) }
            AMZ_T0
            if (grp < 4) {
                substep_b_threads(lds, a, t, grp, c);
            } else if (wave == 12) {
                wave_p9(lds, a, 2 * t - 26, lane);
            } else if (wave == 13) {
                wave_p13(lds, a, 2 * t - 26, lane);
            } else if (wave == 14) {
                st_p7(lds, a, 2 * t - 14, lane);
                st_p7(lds, a, 2 * t - 14, 64 + lane);
                st_p7(lds, a, 2 * t - 14, 128 + lane);
                wave_list(lds, a, t, lane);
            } else {
                p8_wave_b(lds, t, lane, p8);
            }
            AMZ_T1(tb)
            { AMZ_T0 lds_barrier(); AMZ_T1(tw) }
        }
#ifdef AMZ_PROFILE
        if (blockIdx.x == 1000 && lane_ == 0) printf("wave %2d: a %8lld  b %8lld  barrier-wait %8lld cycles (%d steps)\n", wave, ta, tb, tw, NSTEPS);
#endif
        if (grp == 1) bb_flush(lds, 0, rg.bb);
        if (grp == 5) bb_flush(lds, 4, p8.bb);
        lds_barrier();
        if (tid == 0 && !tile_valid(lds)) {
            const int slot = atomicAdd(&s.fallback[0], 1);
            s.fallback[1 + slot] = tile;
        }
        lds_barrier();      // the next tile's tile_begin rewrites the reduction words
    }
}

hipError_t launch_amaze_stream(const AmazeStreamArgs &s, int grid, hipStream_t stream)
{
    constexpr size_t dyn = (size_t)amz::LDS_FLOATS * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&amaze_stream_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(amaze_stream_kernel, dim3(grid), dim3(amz::NTHREADS), dyn, stream, s);
    return hipGetLastError();
}

} // namespace artgpu
