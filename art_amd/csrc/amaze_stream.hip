// art_amd/csrc/amaze_stream.hip -- AMaZE v2 for gfx950: one 1024-thread workgroup streams a reference tile (160x160) through
// LDS ring buffers, two rows per step, two LDS-only barriers per step.  The stages, their offsets and the ring layout are in
// amaze_stream_core.h (shared with the CPU emulation of the schedule under tests/emul); this file is the device driver:
// wave roles, the wave-level row recurrences, the barrier, the tile loop and the hand-over of tiles the stream cannot take
// (Nyquist sites outside the tile's bounding box) to the arena kernel of amaze.hip.
//
// Replaces RawImageSource::amaze_demosaic_RT (reference: rtengine/amaze_demosaic_RT.cc:41-1595, x86-64 / __SSE2__ branches).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
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
    amz_li red = (amz_li)(lds + RED_OFF);
    amz_ls list = (amz_ls)(lds + LIST_OFF + buf * LIST_INTS);
    int n = 0;
    if (r + 1 >= 8 && r < a.rr1 - 8) {
#pragma unroll
        for (int pass = 0; pass < 3; ++pass) {
            const int c = pass * 64 + lane;
            int rr = 0;
            const bool f = nyq_site(lds, a, r, c, &rr);
            const unsigned long long m = __ballot(f);
            const int pos = n + __popcll(m & ((1ull << lane) - 1ull));
            if (f) list[pos] = (unsigned short)((rr << 8) | c);
            n += __popcll(m);
        }
    }
    if (lane == 0) red[16 + buf] = n;
}

} // namespace

// ---- the tile sequence of a workgroup: the leader (lane 0 of part 0 of the LIGHT role) fills the slots, every wave reads them ----
// tile blockIdx.x first, then whatever the shared tile counter hands out (the workgroups do not start together when the arena kernel's
// tiles occupy some CUs at first, and a fixed share per workgroup made the last starter the kernel's length), then the redo queue
__device__ __forceinline__ TileRef seq_tile_ref(amz_lf lds, int k)
{
    amz_li slot = (amz_li)(lds + DYN_OFF) + (k & 3) * DYN_SLOT;
    TileRef t;
    const int sk = __builtin_amdgcn_readfirstlane(slot[0]), tile = __builtin_amdgcn_readfirstlane(slot[1]);
    if (sk == k && tile >= 0) {
        t.gbase = TS * k;
        t.tile = tile | (__builtin_amdgcn_readfirstlane(slot[6]) ? TILE_REDO : 0);
        t.top = __builtin_amdgcn_readfirstlane(slot[2]); t.left = __builtin_amdgcn_readfirstlane(slot[3]);
        t.rr1 = __builtin_amdgcn_readfirstlane(slot[4]); t.box = __builtin_amdgcn_readfirstlane(slot[5]);
    } else {
        tile_ref_none(t, k);
    }
    return t;
}
__device__ __forceinline__ void seq_slot_set(const AmazeStreamArgs &s, amz_lf lds, int k, int tile, int redo, int box)
{
    amz_li slot = (amz_li)(lds + DYN_OFF) + (k & 3) * DYN_SLOT;
    TileRef t;
    tile_ref_none(t, k);
    if (tile >= 0) {
        const int ty = tile / s.ntx, tx = tile - ty * s.ntx;
        const int top = -16 + ty * (TS - 32);
        tile_ref_set(t, k, tile, top, -16 + tx * (TS - 32), min(top + TS, s.H + 16) - top);
        if (redo) t.box = box;
    }
    slot[0] = k; slot[1] = tile; slot[2] = t.top; slot[3] = t.left; slot[4] = t.rr1; slot[5] = t.box; slot[6] = redo;
}
// leader: take the next fresh tile, or one entry of the redo queue, for sequence position k (none: tile -1)
__device__ __forceinline__ void seq_pull(const AmazeStreamArgs &s, amz_lf lds, int k)
{
    amz_li dyn = (amz_li)(lds + DYN_OFF);
    int *const cnt = reinterpret_cast<int *>(s.queue_words + s.ntiles);    // eight bookkeeping counters behind the queue (artgpu_get_option)
    int tile = -1;
    unsigned long long w = 0;
    if (dyn[4 * DYN_SLOT]) {
        // tiles gridDim.x .. ntiles - 1 of the list are handed out in order (queue_hdr[2], cleared per launch)
        const int i = (int)gridDim.x + atomicAdd(&s.queue_hdr[2], 1);
        if (i < s.ntiles) {
            seq_slot_set(s, lds, k, s.tiles[i], 0, 0);
            return;
        }
        dyn[4 * DYN_SLOT] = 0;
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
    seq_slot_set(s, lds, k, tile, 1, ny_pack((int)((w >> 24) & 0xff), (int)((w >> 32) & 0xff), (int)((w >> 40) & 0xff), (int)((w >> 48) & 0xff)));
}
// leader, every stage has left tile q.back: is it valid?  If not, stream it again with the true box: publish a queue entry; if the slot
// was abandoned, the arena kernel takes the tile.
// The per-XCD L2s are not coherent with each other: the second attempt runs on another CU, possibly another XCD, and BOTH L2s would hold
// dirty copies of the tile's output lines -- whichever is written back last wins.  So before the tile is offered again its pixels leave
// this XCD's L2: the storing waves drain their stores one step ahead (the output stage finished the tile eight steps ago), then the
// leader writes the L2 back (release, agent scope) and publishes.
__device__ __forceinline__ void seq_tile_done(const AmazeStreamArgs &s, amz_lf lds, int par, int rr1, int tile, int redo)
{
    int *const cnt = reinterpret_cast<int *>(s.queue_words + s.ntiles);
    int box[4];
    if (!tile_valid(lds, par, rr1, box) && !redo) {
        const unsigned long long w = (unsigned long long)tile | ((unsigned long long)box[0] << 24) | ((unsigned long long)box[1] << 32) |
                                     ((unsigned long long)box[2] << 40) | ((unsigned long long)box[3] << 48) | (1ull << 63);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int slot = atomicAdd(&s.queue_hdr[0], 1);
        unsigned long long zero = 0;
        atomicAdd(&cnt[2], 1);
        if (!__hip_atomic_compare_exchange_strong(&s.queue_words[slot], &zero, w, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            const int fs = atomicAdd(&s.fallback[0], 1);
            s.fallback[1 + fs] = tile;
            atomicAdd(&cnt[3], 1);
        }
    }
    red_reset(lds, par);
}

#ifdef AMZ_PROFILE
#define AMZ_T0 const long long t0_ = __builtin_amdgcn_s_memtime();
#define AMZ_T1(acc) { const long long t1_ = __builtin_amdgcn_s_memtime(); acc += t1_ - t0_; }
#else
#define AMZ_T0
#define AMZ_T1(acc)
#endif

// One step loop per pair of roles (round 6).  A wave plays role KA in sub-step a and KB in sub-step b for the whole kernel, and the
// workgroup barrier only counts arrivals, so every pair of roles gets a loop of its own: it keeps what ITS two stages need in registers
// (the one loop of rounds 2 - 5 carried every role's scalars through every role's code: 106 scalar registers, 43 of them spilled to
// lanes), hoists its lane offsets out of the loop, and only the leader's loop carries the tile bookkeeping.
template <int KA, int KB, bool LEADER>
__device__ __forceinline__ void role_loop(const AmazeStreamArgs &s, amz_lf lds, const int wave, const int ca_, const int cb_, const int lane_)
{
    TileArgs frame;
    frame.raw = (amz_gcf)s.raw; frame.rs = (long)s.raw_stride;
    frame.red = (amz_gf)s.red; frame.green = (amz_gf)s.green; frame.blue = (amz_gf)s.blue; frame.os = (long)s.out_stride;
    frame.W = s.W; frame.H = s.H; frame.filters = s.filters; frame.clip_pt = s.clip_pt; frame.clip_pt8 = s.clip_pt8; frame.g00 = s.g00; frame.ey = s.ey;
    frame.top = 0; frame.left = 0; frame.rr1 = 0; frame.gbase = 0; frame.rbase = 0; frame.g0 = 0; frame.pos = 0; frame.ny_box = 0;

    TileSeq q;
    q.t2 = 0;
    tile_ref_none(q.back, -1);
    q.front = seq_tile_ref(lds, 0);
    q.next = seq_tile_ref(lds, 1);
    int nk = q.next.rr1 > 0 ? 2 : 1;

    ThreadRegs rg;
    bb_reset(rg.bb);
    rg.pos_p = rg.pos_h = rg.pos_stride2 = rg.pos_off4 = 0;
    P8Regs p8;
    p8.cc = -1;
    if (KA == LOADER_ROLE) st_load_first(lds, frame, q, ca_);
    if (KA == A_LIGHT) pos_init(lds, ca_, rg);
    lds_barrier();                                   // (the ring position table of step 0)
#ifdef AMZ_PROFILE
    long long ta = 0, tb = 0, tw = 0;
#endif
    int tt = 0, kf = 0;                              // T % STEPS_PER_TILE, the front tile's position in the sequence
    for (int T = 0; T < STEPS_PER_TILE * nk + TAIL_STEPS; ++T, ++tt) {
        q.t2 = 2 * T;
        if (tt == STEPS_PER_TILE) {                  // the load front enters the next tile
            tt = 0;
            ++kf;
            q.back = q.front;
            q.front = seq_tile_ref(lds, kf);
            if (seq_tile_ref(lds, kf + 1).rr1 > 0) nk = kf + 2;
        }
        // (only the loader looks at the tile after the front one, and only in the front tile's last step: not carried through the loop)
        q.next = q.front;
        if (KA == LOADER_ROLE && tt == STEPS_PER_TILE - 1) q.next = seq_tile_ref(lds, kf + 1);
        int ca = ca_, cb = cb_, lane = lane_;
#ifdef AMZ_OPAQUE_LANES
        // (rounds 2 - 5, one loop for all roles: the column / lane were made opaque per step, otherwise every role's column-derived addresses
        // were hoisted out of the step loop and kept live across all the other roles.  With a loop per pair of roles the hoisting is what is
        // wanted: 3.84 -> 3.63 ms.)
        asm volatile("" : "+v"(ca), "+v"(cb), "+v"(lane));
#endif
        if (KB == B_P16OUT) { if (tile_drain(q, T)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        if (LEADER && ca_ == 0) {          // (lane 0 of the role's part 0)
            if (tile_done(q, T)) seq_tile_done(s, lds, (kf - 1) & 1, q.back.rr1, tile_index(q.back), tile_redo(q.back) ? 1 : 0);
            // one step before the load front needs a sequence position beyond the first tile: ask the tile counter / the redo queue
            if (tt + 1 == STEPS_PER_TILE) seq_pull(s, lds, kf + 2);
        }
        { AMZ_T0 if (KA != A_P8) substep_a(lds, frame, q, T, KA, ca, rg); else p8_step_a(lds, frame, q, T, lane, p8, rg.bb); AMZ_T1(ta) }
        { AMZ_T0 lds_barrier(); AMZ_T1(tw) }
        AMZ_T0
        if (KB < B_P9) {
            substep_b_threads(lds, frame, q, T, KB, cb);
        } else if (KB == B_P9) {
            const TileArgs a = stage_tile(frame, q, 2 * T - 26);
            wave_p9(lds, a, a.rbase - 26, lane);
        } else if (KB == B_P13_P14) {
            {
                const TileArgs a = stage_tile(frame, q, 2 * T - 26);
                wave_p13(lds, a, a.rbase - 26, lane);
            }
            const TileArgs a = stage_tile(frame, q, 2 * T - 30);
            p14_worker(lds, a, a.rbase - 30, lane);
            wave_order();
            if (lane == 0) hot_reset(lds, 1);
        } else if (KB == B_P7_P10) {
            {
                const TileArgs a = stage_tile(frame, q, 2 * T - 20);
                wave_list(lds, a, T, a.rbase - 20, lane);
            }
            const TileArgs a = stage_tile(frame, q, 2 * T - 30);
            p10_worker(lds, a, a.rbase - 30, lane);
            wave_order();
            if (lane == 0) hot_reset(lds, 0);
        } else {
            p8_step_b(lds, frame, q, T, lane, p8, rg.bb);
        }
        AMZ_T1(tb)
        { AMZ_T0 lds_barrier(); AMZ_T1(tw) }
    }
#ifdef AMZ_PROFILE
    if (blockIdx.x == 100 && lane_ == 0) printf("wave %2d (a %d b %d): a %8lld  b %8lld  barrier-wait %8lld cycles (%d tiles)\n", wave, KA, KB, ta, tb, tw, nk);
#endif
}

#ifdef AMZ_ROLE_ENV
// experiment only (scripts/amz_roles_search.py): the wave -> (loop, part) table from the environment, one byte per wave = loop << 2 | part;
// loops: 0 P2+P16OUT, 1 P5L+P3R0, 2 P12+P3R1, 3 LIGHT+P1P11 (its part 0 leads), 4 P4+P13P14, 5 P4+LIST+P10, 6 P4+P9, 7 P8
__device__ unsigned char amz_roles_dev[16];
#endif

__global__ void __launch_bounds__(amz::NTHREADS)
amaze_stream_kernel(AmazeStreamArgs s)
{
    extern __shared__ float dyn_lds[];
    amz_lf lds = (amz_lf)dyn_lds;
    if ((int)blockIdx.x >= s.ntiles) return;       // (more workgroups than tiles)
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane_ = tid & 63;
    WaveRole role = wave_role(wave);             // what this wave does in sub-step a / b, and which 64 columns of a column role
#ifdef AMZ_ROLE_ENV
    {
        const int e = __builtin_amdgcn_readfirstlane((int)amz_roles_dev[wave]);
        const int lp = e >> 2, pt = e & 3;
        const int ka[8] = {A_P2, A_P5L, A_P12, A_LIGHT, A_P4, A_P4, A_P4, A_P8};
        const int kb[8] = {B_P16OUT, B_P3R0, B_P3R1, B_P1P11, B_P13_P14, B_P7_P10, B_P9, B_P8};
        role.a = ka[lp]; role.b = kb[lp]; role.apart = lp >= 4 && lp < 7 ? lp - 4 : pt; role.bpart = lp >= 4 ? 0 : pt;
    }
#endif
    const int ca_ = role.apart * 64 + lane_, cb_ = role.bpart * 64 + lane_;

    // the single-wave roles (row recurrences, the Nyquist area sums) are the longest serial chains of a sub-step: they win the
    // issue arbitration on their SIMD, the column roles fill the gaps
    if (role.b >= B_P9 || role.a == A_P8) __builtin_amdgcn_s_setprio(2);

    if (tid == 0) {
        amz_li dyn = (amz_li)(lds + DYN_OFF);
        for (int k = 0; k < 4; ++k) { dyn[k * DYN_SLOT] = -1; dyn[k * DYN_SLOT + 1] = -1; }
        dyn[4 * DYN_SLOT] = 1;                     // the tile counter still has tiles
        seq_slot_set(s, lds, 0, s.tiles[blockIdx.x], 0, 0);
        seq_pull(s, lds, 1);
    }
    seq_begin(lds, tid);
    lds_barrier();
    switch (role.a) {
    case A_P2: role_loop<A_P2, B_P16OUT, false>(s, lds, wave, ca_, cb_, lane_); break;
    case A_P5L: role_loop<A_P5L, B_P3R0, false>(s, lds, wave, ca_, cb_, lane_); break;
    case A_P12: role_loop<A_P12, B_P3R1, false>(s, lds, wave, ca_, cb_, lane_); break;
    case A_LIGHT: role_loop<A_LIGHT, B_P1P11, true>(s, lds, wave, ca_, cb_, lane_); break;
    case A_P8: role_loop<A_P8, B_P8, false>(s, lds, wave, ca_, cb_, lane_); break;
    default:
        if (role.b == B_P13_P14) role_loop<A_P4, B_P13_P14, false>(s, lds, wave, ca_, cb_, lane_);
        else if (role.b == B_P7_P10) role_loop<A_P4, B_P7_P10, false>(s, lds, wave, ca_, cb_, lane_);
        else role_loop<A_P4, B_P9, false>(s, lds, wave, ca_, cb_, lane_);
        break;
    }
}

hipError_t launch_amaze_stream(const AmazeStreamArgs &s, int grid, hipStream_t stream)
{
    constexpr size_t dyn = (size_t)amz::LDS_FLOATS * sizeof(float);
    if (hipError_t e = dyn_lds_once(reinterpret_cast<const void *>(&amaze_stream_kernel), (int)dyn); e != hipSuccess) return e;
#ifdef AMZ_ROLE_ENV
    if (const char *e = getenv("AMZ_ROLES")) {
        unsigned char t[16];
        for (int i = 0; i < 16; ++i) { unsigned v = 0; sscanf(e + 2 * i, "%2x", &v); t[i] = (unsigned char)v; }
        if (hipError_t er = hipMemcpyToSymbol(HIP_SYMBOL(amz_roles_dev), t, 16); er != hipSuccess) return er;
    }
#endif
    hipLaunchKernelGGL(amaze_stream_kernel, dim3(grid), dim3(amz::NTHREADS), dyn, stream, s);
    return hipGetLastError();
}

} // namespace artgpu
