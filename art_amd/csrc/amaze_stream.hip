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
        if (rr < 8 || rr >= a.rr1 - 8) continue;
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
        if (rr < 10 || rr >= a.rr1 - 10) continue;
        const float h0 = p13_new_weight(lds, a, rr, lane);
        float h1 = 0.f;
        if (lane < 8) h1 = p13_new_weight(lds, a, rr, 64 + lane);
        p13_site(lds, a, rr, lane, h0);
        if (lane < 8) p13_site(lds, a, rr, 64 + lane, h1);
        wave_order();
    }
}
// the nyquist2 sites of tile rows (r, r+1), compacted for P8 of the next step
__device__ __forceinline__ void wave_list(amz_lf lds, const TileArgs &a, int T, int r, int lane)
{
    const int buf = (T + 1) & 1;     // the list step T+1 consumes
    amz_li red = (amz_li)(lds + RED_OFF), list = (amz_li)(lds + LIST_OFF + buf * LIST_INTS);
    int n = 0;
    if (r + 1 >= 8 && r < a.rr1 - 8) {
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
    if (lane == 0) red[16 + buf] = n;
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
    const WaveRole role = wave_role(wave);       // what this wave does in sub-step a / b, and which 64 columns of a column role
    const int ca_ = role.apart * 64 + lane_, cb_ = role.bpart * 64 + lane_;

    // the single-wave roles (row recurrences, the Nyquist area sums) are the longest serial chains of a sub-step: they win the
    // issue arbitration on their SIMD, the column roles fill the gaps
    if (role.b >= B_P9 || role.a == A_P8) __builtin_amdgcn_s_setprio(2);

    TileArgs frame;
    frame.raw = (amz_gcf)s.raw; frame.rs = (long)s.raw_stride;
    frame.red = (amz_gf)s.red; frame.green = (amz_gf)s.green; frame.blue = (amz_gf)s.blue; frame.os = (long)s.out_stride;
    frame.W = s.W; frame.H = s.H; frame.filters = s.filters; frame.clip_pt = s.clip_pt; frame.clip_pt8 = s.clip_pt8; frame.g00 = s.g00; frame.ey = s.ey;
    frame.top = 0; frame.left = 0; frame.rr1 = 0; frame.gbase = 0;

    // this workgroup's tile sequence: tile blockIdx.x, then whatever the shared tile counter hands out (the workgroups do not start
    // together when the arena kernel's tiles occupy some CUs at first, and a fixed share per workgroup made the last starter the
    // kernel's length), then whatever the redo queue holds
    int nk = ((int)blockIdx.x < s.ntiles) ? 1 : 0;
    const int nk_static = nk;
    amz_li dyn = (amz_li)(lds + DYN_OFF);          // the redo entry tid 0 pulled for sequence position dyn[0]
    auto tile_ref = [&](int k) -> TileRef {
        TileRef t;
        if (k < nk_static) {
            const int tile = s.tiles[blockIdx.x];
            const int ty = tile / s.ntx, tx = tile - ty * s.ntx;
            const int top = -16 + ty * (TS - 32);
            tile_ref_set(t, k, tile, top, -16 + tx * (TS - 32), min(top + TS, s.H + 16) - top);
        } else if (dyn[0] == k && dyn[1] >= 0) {
            const int tile = dyn[1];
            const int ty = tile / s.ntx, tx = tile - ty * s.ntx;
            const int top = -16 + ty * (TS - 32);
            tile_ref_set(t, k, tile, top, -16 + tx * (TS - 32), min(top + TS, s.H + 16) - top);
            if (dyn[6]) { t.tile |= TILE_REDO; t.box = ny_pack(dyn[2], dyn[3], dyn[4], dyn[5]); }
        } else {
            tile_ref_none(t, k);
        }
        return t;
    };
    // tid 0: take one entry of the redo queue for sequence position k (none: dyn[1] = -1)
    int *const cnt = reinterpret_cast<int *>(s.queue_words + s.ntiles);    // eight bookkeeping counters behind the queue (artgpu_get_option)
    bool tiles_left = true;
    auto pull = [&](int k) {
        int tile = -1;
        unsigned long long w = 0;
        if (tiles_left) {
            // tiles gridDim.x .. ntiles - 1 of the list are handed out in order (queue_hdr[2], cleared per launch)
            const int i = (int)gridDim.x + atomicAdd(&s.queue_hdr[2], 1);
            if (i < s.ntiles) {
                dyn[0] = k; dyn[1] = s.tiles[i]; dyn[6] = 0;
                return;
            }
            tiles_left = false;
        }
        // every read of the queue is a read-modify-write (+0): the per-XCD L2s are not coherent with each other, and a plain or sc1
        // load can return what an earlier launch left in this XCD's L2 -- a consumer that trusted a stale "reserved" count took a
        // slot that was never published in this launch, and the real entry published there later was lost
        const int taken = __hip_atomic_fetch_add(&s.queue_hdr[1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int reserved = __hip_atomic_fetch_add(&s.queue_hdr[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (taken < reserved) {
            int expect = taken;
            if (__hip_atomic_compare_exchange_strong(&s.queue_hdr[1], &expect, taken + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                // the producer publishes the word right after reserving the slot; if it does not show up, abandon the slot (the
                // producer's publishing compare-and-swap then fails and it hands the tile to the arena kernel instead)
                for (int spin = 0; spin < 4096; ++spin) {
                    w = __hip_atomic_fetch_or(&s.queue_words[taken], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (w) break;
                    __builtin_amdgcn_s_sleep(8);
                }
                if (!w) {
                    unsigned long long zero = 0;
                    if (!__hip_atomic_compare_exchange_strong(&s.queue_words[taken], &zero, ~0ull, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                        w = zero;      // published in the meantime
                }
                if (w && w != ~0ull) tile = (int)(w & 0xffffffu);
            }
        }
        atomicAdd(&cnt[tile >= 0 ? 0 : 1], 1);
        dyn[0] = k; dyn[1] = tile; dyn[6] = 1;
        dyn[2] = (int)((w >> 24) & 0xff); dyn[3] = (int)((w >> 32) & 0xff); dyn[4] = (int)((w >> 40) & 0xff); dyn[5] = (int)((w >> 48) & 0xff);
    };
    if (nk == 0) return;       // (more workgroups than tiles)
    if (tid == 0) { dyn[0] = -1; dyn[1] = -1; if (nk_static == 1) pull(1); }
    seq_begin(lds, tid);
    lds_barrier();
    TileSeq q;
    tile_ref_none(q.back, -1);
    q.front = tile_ref(0);
    q.next = tile_ref(1);
    if (q.next.rr1 > 0 && nk < 2) nk = 2;

    ThreadRegs rg;
    bb_reset(rg.bb);
    P8Regs p8;
    p8.cc = -1;
    if (role.a == LOADER_ROLE) st_load_first(lds, frame, q, ca_);
#ifdef AMZ_PROFILE
    long long ta = 0, tb = 0, tw = 0;
#define AMZ_T0 const long long t0_ = __builtin_amdgcn_s_memtime();
#define AMZ_T1(acc) { const long long t1_ = __builtin_amdgcn_s_memtime(); acc += t1_ - t0_; }
#else
#define AMZ_T0
#define AMZ_T1(acc)
#endif
#ifdef AMZ_DEBUG
    int dbg[3][8]; int ndbg = 0;
#endif
    for (int T = 0; T < STEPS_PER_TILE * nk + TAIL_STEPS; ++T) {
        if (T > 0 && T % STEPS_PER_TILE == 0) {      // the load front enters the next tile
            q.back = q.front;
            q.front = q.next;
            const int kn = T / STEPS_PER_TILE + 1;
            q.next = tile_ref(kn);
            if (q.next.rr1 > 0 && nk < kn + 1) nk = kn + 1;
        }
        // the column / lane are made opaque per step: otherwise every role's column-derived addresses are hoisted out of the
        // step loop and kept live across all the other roles (the kernel then spills into scratch inside the loop)
        int ca = ca_, cb = cb_, lane = lane_;
        asm volatile("" : "+v"(ca), "+v"(cb), "+v"(lane));
        // The per-XCD L2s are not coherent with each other: if tile q.back has to be streamed again, the second attempt runs on another
        // CU, possibly another XCD, and BOTH L2s would hold dirty copies of the tile's output lines -- whichever is written back last
        // wins.  So before the tile is offered again its pixels leave this XCD's L2: every wave drains its stores one step ahead
        // (the output stage finished the tile eight steps ago), then thread 0 writes the L2 back (release, agent scope) and publishes.
        if (tile_drain(q, T)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) {
            if (tile_done(q, T)) {                   // every stage has left tile q.back
                const int par = (q.back.gbase / TS) & 1;
                int box[4];
#ifdef AMZ_DEBUG
                { amz_li red = (amz_li)(lds + RED_OFF) + 8 * par; const bool v_ = tile_valid(lds, par, q.back.rr1, box);
                  if ((!v_ || tile_redo(q.back)) && ndbg < 3) { dbg[ndbg][0] = tile_index(q.back); dbg[ndbg][1] = tile_redo(q.back); dbg[ndbg][2] = (int)v_; dbg[ndbg][3] = box[2]; dbg[ndbg][4] = box[3]; dbg[ndbg][5] = red[6]; dbg[ndbg][6] = red[7]; dbg[ndbg][7] = T; ++ndbg; } }
#endif
                if (!tile_valid(lds, par, q.back.rr1, box) && !tile_redo(q.back)) {
                    // stream it again with the true box: publish a queue entry; if the slot was abandoned, the arena kernel takes the tile
                    const unsigned long long w = (unsigned long long)tile_index(q.back) | ((unsigned long long)box[0] << 24) | ((unsigned long long)box[1] << 32) |
                                                 ((unsigned long long)box[2] << 40) | ((unsigned long long)box[3] << 48) | (1ull << 63);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    const int slot = atomicAdd(&s.queue_hdr[0], 1);
                    unsigned long long zero = 0;
                    atomicAdd(&cnt[2], 1);
                    if (!__hip_atomic_compare_exchange_strong(&s.queue_words[slot], &zero, w, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                        const int fs = atomicAdd(&s.fallback[0], 1);
                        s.fallback[1 + fs] = tile_index(q.back);
                        atomicAdd(&cnt[3], 1);
                    }
                }
                red_reset(lds, par);
            }
            // one step before the load front needs a sequence position beyond the static tiles: look into the redo queue
            const int kn = (T + 1) / STEPS_PER_TILE + 1;
            if ((T + 1) % STEPS_PER_TILE == 0 && kn >= nk_static) pull(kn);
        }
        { AMZ_T0 if (role.a != A_P8) substep_a(lds, frame, q, T, role.a, ca, rg); else p8_step_a(lds, frame, q, T, lane, p8, rg.bb); AMZ_T1(ta) }
        { AMZ_T0 lds_barrier(); AMZ_T1(tw) }
        AMZ_T0
        if (role.b < B_P9) {
            substep_b_threads(lds, frame, q, T, role.b, cb);
        } else if (role.b == B_P9) {
            const TileArgs a = stage_tile(frame, q, 2 * T - 26);
            wave_p9(lds, a, 2 * T - 26 - a.gbase, lane);
        } else if (role.b == B_P13_P14) {
            {
                const TileArgs a = stage_tile(frame, q, 2 * T - 26);
                wave_p13(lds, a, 2 * T - 26 - a.gbase, lane);
            }
            const TileArgs a = stage_tile(frame, q, 2 * T - 30);
            p14_worker(lds, a, 2 * T - 30 - a.gbase, lane);
            wave_order();
            if (lane == 0) hot_reset(lds, 1);
        } else if (role.b == B_P7_P10) {
            {
                const TileArgs a = stage_tile(frame, q, 2 * T - 14);
                st_p7(lds, a, 2 * T - 14 - a.gbase, lane);
                st_p7(lds, a, 2 * T - 14 - a.gbase, 64 + lane);
                st_p7(lds, a, 2 * T - 14 - a.gbase, 128 + lane);
            }
            {
                const TileArgs a = stage_tile(frame, q, 2 * T - 20);
                wave_list(lds, a, T, 2 * T - 20 - a.gbase, lane);
            }
            const TileArgs a = stage_tile(frame, q, 2 * T - 30);
            p10_worker(lds, a, 2 * T - 30 - a.gbase, lane);
            wave_order();
            if (lane == 0) hot_reset(lds, 0);
        } else {
            p8_step_b(lds, frame, q, T, lane, p8, rg.bb);
        }
        AMZ_T1(tb)
        { AMZ_T0 lds_barrier(); AMZ_T1(tw) }
    }
#ifdef AMZ_DEBUG
    if (tid == 0) for (int k = 0; k < ndbg; ++k) printf("wg %d: tile %d redo %d valid %d box cols %d %d sites cols %d %d at T %d (nk %d)\n", (int)blockIdx.x, dbg[k][0], dbg[k][1], dbg[k][2], dbg[k][3], dbg[k][4], dbg[k][5], dbg[k][6], dbg[k][7], nk);
#endif
#ifdef AMZ_PROFILE
    if (blockIdx.x == 100 && lane_ == 0) printf("wave %2d: a %8lld  b %8lld  barrier-wait %8lld cycles (%d tiles)\n", wave, ta, tb, tw, nk);
#endif
}

hipError_t launch_amaze_stream(const AmazeStreamArgs &s, int grid, hipStream_t stream)
{
    constexpr size_t dyn = (size_t)amz::LDS_FLOATS * sizeof(float);
    if (hipError_t e = dyn_lds_once(reinterpret_cast<const void *>(&amaze_stream_kernel), (int)dyn); e != hipSuccess) return e;
    hipLaunchKernelGGL(amaze_stream_kernel, dim3(grid), dim3(amz::NTHREADS), dyn, stream, s);
    return hipGetLastError();
}

} // namespace artgpu
