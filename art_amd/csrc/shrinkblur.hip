// art_amd/csrc/shrinkblur.hip -- ShrinkAllL / ShrinkAllAB in ONE pass over the coefficients (FTblockDN.cc:638-839).
//
// The reference does, per band: sfave = shrink factor of every coefficient (L664-690 / L760-790), boxblur(sfave, sfaved, radius) --
// boxblur.h:558-742: a running sum along every row into a temporary, then a running sum down every column --, and
// coef *= (sfaved^2 + sfave^2) / (sfaved + sfave + eps) (L698-714 / L803-836).  The running sums are fp32 accumulations whose value
// depends on their order, so a row has to be walked from column 0 and a column from row 0: the three-kernel form (shrink_sf_*,
// hblur_kernel, vblur_combine_kernel in denoise.hip) stores sfave and the row-blurred plane and reads them back -- 32 to 36 bytes per
// coefficient in three dependent passes, 4.3 of the 8.5 ms the FTblockDN chain took on a 45 MP frame.
//
// Here a workgroup owns a STRIP of 64 rows of one band and walks it left to right in blocks of 64 columns.  Per block: all threads
// evaluate the shrink factors of the block's coefficients into LDS; one wave (lane = row) advances the 64 row sums across the block;
// one wave (lane = column) advances the block's 64 column sums down the strip; all threads update the coefficients.  What a column sum
// needs from above the strip -- its accumulator and the 2 * rad + 1 row-blurred values it is about to drop -- is handed down by the strip
// above through a small buffer in global memory: strips of a band form a wavefront, strip s working on block j while strip s - 1 is at
// block j + 1 or further.  Workgroups take (strip, band) from a ticket counter, strip-major, so a workgroup only ever waits for one
// that started before it (no assumption about the order in which the hardware dispatches blockIdx).  The lags: the row sum of column c
// needs the factor of column c + rad, the column sum of row r the row-blurred value of row r + rad, so block j produces columns
// [64 j - rad, 64 j + 64 - rad) and a strip that holds rows [R0, R0 + 64) produces rows [R0 - rad, R0 + 64 - rad): it evaluates the
// factors of the rad rows above itself a second time (they only enter the coefficient update), the last strip runs to the bottom.
// Every coefficient is read once (plus rad / 64 of them twice) and written once: 8 - 12 bytes instead of 32 - 36, same operations in
// the same order on every row and every column: same bits (tests/test_gpu_denoise.py compares the two forms and both with the oracle).
#include "kernels.h"
#include "devmath.h"
#include "devsleef.h"
#include <type_traits>
#include <mutex>

// Hardware assumption of the strip-to-strip hand-over below (the hand-over wave's `s_waitcnt vmcnt(0)` between its data stores and its flag
// store, relaxed agent-scope accesses on both sides): on the gfx9 family -- gfx942, gfx950 -- stores count in vmcnt, so the wait covers them,
// and agent-scope (sc1) stores write through the XCD's L2 while agent-scope loads miss it; a target with a separate store counter (vscnt)
// could publish the flag ahead of the data.  This translation unit is therefore only valid for those two targets.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx942__) && !defined(__gfx950__)
#error "shrinkblur.hip: the strip hand-over relies on the gfx942 / gfx950 memory model (stores counted in vmcnt, write-through sc1 stores)"
#endif

namespace artgpu {

namespace {

constexpr int FS_R = 64, FS_C = 64, FS_T = 1024, FS_NE = FS_T / 64 - 3;   // 16 waves: row sums, column sums, hand-over, 13 elementwise
constexpr int FS_SWIN = 256, FS_SWS = FS_SWIN + 1;                          // factor window: 256 columns (circular), odd row stride
// Hand-over slots per band.  Strip s reads slot s % FS_RING (written by strip s - 1) and writes slot (s + 1) % FS_RING.  Two are enough: strip
// s + 1 stores block j of ITS hand-over -- into the slot strip s reads -- only behind its own wait for `progress[s] >= j + 1` (the hand-over wave
// polls before it stores, and a strip's steps are separated by workgroup barriers), and strip s publishes j + 1 at its step j + 4, three steps
// after its column sums consumed block j of that slot.  So a slot is rewritten only behind its reader, and nobody waits for anything the
// per-strip slots of rounds 4 did not make them wait for: 45 bands x 2 x 266 KB = 24 MB instead of 0.5 GB for a 45 MP frame.
// RELIANCE ON THE MEMORY MODEL (recorded here on purpose): with two slots the same global addresses are written and read several times within
// one launch -- strip s + 2 reads where strip s read.  The write-after-read order is the protocol above; the FRESHNESS of a read rests on the
// hand-over's `sc1` (agent-scope, relaxed) loads never being served from a stale line: on gfx942 / gfx950 an `sc1` load bypasses the CU's L1
// and an `sc1` store is written through the XCD's L2 to the fabric, so a line strip s left in some XCD's L2 cannot satisfy strip s + 2's read
// on another -- or the same -- XCD (MI355X_MICROARCH.md, "inter-workgroup visibility"; the translation unit refuses any other target at compile time).  The
// per-strip slots of round 4 wrote and read each address once per launch and did not depend on it.  Guard:
// tests/test_gpu_denoise.py::test_fused_hand_over_ring_under_many_short_strips (47 strips of two blocks, fresh data per repetition, every
// repetition against the three-kernel form) and scripts/soak_fused.py.
constexpr int FS_RING = 2;
constexpr unsigned FS_DIAG_MAGIC = 0xF5D1A600u;
constexpr long long FS_WAIT_TICKS = 500000000LL;                           // 5 s of the 100 MHz s_memrealtime clock

// Barrier for data that travels through LDS only: __syncthreads() would also wait for this wave's global loads -- the coefficients of the
// next block, fetched a step ahead precisely so that nobody waits for them
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
__device__ __forceinline__ float ld_agent(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// band constants of the shrink factor (shrink_sf_L_kernel / shrink_sf_AB_kernel)
struct SfConst { float mad_L, levelFactor, madab, rmadLm9, nv_const, nv_scale; int has_nv; };

// TAIL: the block holds coefficients of the band's last n % 4 (the reference's scalar loop tail, other operation order); everywhere else the
// vector form is the only one, and the factors of a thread's rows are independent straight-line chains the scheduler can interleave
// FASTEXP: the exponential's scaling by 2^q as one v_ldexp_f32 instead of sleef's five multiplications by powers of two (xexpf_v_ldexp,
// devsleef.h: a third of its instructions).  scripts/exp_ldexp_check.c walks every float: the two forms return the same bits except (i) for
// 35 497 arguments in [-99.36, -89.07], where the result is a subnormal below 2.1e-39 whose last bits depend on how often it was rounded on the way
// down, and (ii) from +398.9 on, where ldexpk's factors leave the exponent field and the chain returns noise instead of +inf.  The band's
// constants are tested once per ticket (`fastexp` in the kernel) so that the argument cannot be positive -- it is -(x / y) - z with x, z >= 0
// by construction and y, z's scale >= 0 by that test --, and a result below 2.1e-39 never reaches the factor: for L it is multiplied by madv <=
// mag / 801 (the argument is -mag / (9 madv) <= -89) and added to mag, 2^80 times larger; for chroma it is subtracted from 1.  So the factors are
// the same bits whichever form runs (tests: the fused pass against the three-kernel form and the oracle, both of which scale with ldexpk; strong
// edges over faint noise in test_fused_shrink_pass_strong_edges_faint_noise put thousands of arguments below -89).
template <bool AB, bool TAIL, bool FASTEXP = false>
__device__ __forceinline__ float shrink_factor(const SfConst &k, float c, float cl, float nvv, bool vecform)
{
    const float eps = 0.01f;
    auto expv = [](float d) { return FASTEXP ? xexpf_v_ldexp(d) : xexpf_v(d); };
    if constexpr (!AB) {
        const float nv = k.has_nv ? nvv : k.nv_const;
        const float mag = sqr(c);
        if (!TAIL || vecform) {
            const float madv = nv * k.levelFactor;
            return mag / (mag + madv * expv(-mag / (9.0f * madv)) + eps);
        }
        return mag / (mag + k.levelFactor * nv * xexpf_s(-mag / (9 * k.levelFactor * nv)) + eps);
    } else {
        const float nvc = k.has_nv ? k.nv_scale * nvv : 1.f;
        const float mag_ab = sqr(c);
        if (!TAIL || vecform) {
            const float mad_abv = nvc * k.madab;
            const float mag_L = sqr(cl) * k.rmadLm9;
            return 1.f - expv(-(mag_ab / mad_abv) - mag_L);
        }
        const float mag_L = sqr(cl);
        return 1.f - xexpf_s(-(mag_ab / (nvc * k.madab)) - (mag_L / (9.f * k.mad_L)));
    }
}

// One workgroup per CU, the roles of a step run side by side (a step = one LDS-only barrier):
//   step T:  elementwise waves   coefficient update of block T - 3, then the shrink factors of block T (coefficients fetched at step T - 1)
//            row-sum wave        block T - 1
//            column-sum wave     block T - 2 (+ the hand-over: what the strip above left for block T - 1 is sent for, block T - 2's rows
//                                go to the strip below, block T - 3 -- whose stores have left by now -- is published)
// A block's factors live in a circular window of 256 columns; its row-blurred values in one of three buffers (written by the row sums,
// turned into column sums in place, read by the update).
template <int MAXR>
__global__ void __launch_bounds__(FS_T) shrink_blur_kernel(FusedShrinkArgs a)
{
    constexpr int SROWS = FS_R + MAXR;             // window rows: up to rad rows above the strip + the strip
    constexpr int HROWS = FS_R + 2 * MAXR + 1;     // 2 rad + 1 rows handed down + the strip
    constexpr int HBUF = (HROWS + 2) * (FS_C + 1);  // + the column sums entering the strip (row HROWS) and leaving it (row HROWS + 1)
    constexpr int HS = FS_C + 1;
    constexpr int NQ = (SROWS + FS_NE - 1) / FS_NE;   // window rows per elementwise wave
    constexpr int NQB = FS_R / FS_NE;                 // passes whose window rows (< 64) every strip but the first and the last holds and writes
    constexpr int NOVMAX = 2 * MAXR + 1;
    extern __shared__ float fs_lds[];
    float *const S = fs_lds;                       // [SROWS][FS_SWS]: factor of window row wr (image row R0 - rad + wr), image column c at c & 255
    float *const HB0 = S + SROWS * FS_SWS;         // 3 x [HROWS][HS]
    __shared__ int s_ticket;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);      // (wave-uniform, and known to be)
    const int ew = wv - 3;                         // elementwise wave index
    const int rbase = ew, rstride = FS_NE;         // window rows rbase + rstride q
    if (wv < 3) __builtin_amdgcn_s_setprio(3);     // the serial roles go first on their SIMDs
#ifdef FS_PROFILE
    long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define FS_T0 const long long t0_ = __builtin_readcyclecounter();
#define FS_T1(k) pt[k] += __builtin_readcyclecounter() - t0_;
#else
#define FS_T0
#define FS_T1(k)
#endif
    const int W = a.w, H = a.h;
    const int n = (int)a.n;                        // (shrink_blur_supported: a band fits 31 bits)
    const int nv4 = (n / 4) * 4;
    const float eps = 0.01f;
    // persistent workgroups: a workgroup that has finished its strip takes the next ticket (strip-major: strip s of every band before
    // strip s + 1 of any), so the strips it may have to wait for belong to tickets taken before its own, i.e. to running workgroups
  for (;;) {
    __syncthreads();
    if (tid == 0) s_ticket = atomicAdd(a.ticket, 1);
    __syncthreads();
    const int ticket = s_ticket;
    if (ticket >= a.nsub * a.nstrips) break;
    // a launch holds nL bands of L (ShrinkAllL) followed by the bands of the chroma channels (ShrinkAllAB), channel after channel
    const int strip = ticket / a.nsub, sub = ticket - strip * a.nsub;
    const bool AB = sub >= a.nL;
    const int sc = sub - a.nL;                                          // chroma: band of the launch's chroma part,
    const int ch = AB ? sc / a.nsub_ch : 0, subc = AB ? sc - ch * a.nsub_ch : sub;   // its channel, and the band of the channel
    const int rad = a.rad[a.level0 + subc / 3];
    const float *coef = AB ? a.coefC + (size_t)sc * a.n : a.coef + (size_t)sub * a.n;
    float *out = AB ? a.coefC + (size_t)sc * a.n : a.coef_out + (size_t)sub * a.n;
    const float *coefL = a.coefL + (size_t)subc * a.n;
    SfConst kc;
    kc.mad_L = a.madL[subc];
    kc.levelFactor = kc.mad_L * 5.f / (float)(subc / 3 + 1);
    kc.madab = 0.f; kc.rmadLm9 = 0.f;
    kc.nv_const = a.noisevar_const; kc.nv_scale = a.noisevar_scale; kc.has_nv = AB && a.noisevar != nullptr;
    if (AB) {
        const float m = a.madab[ch * a.mad_ch_stride + subc];
        kc.madab = a.useNoiseCCurve ? m : m * a.noisevar_ab[ch];
        kc.rmadLm9 = 1.f / (kc.mad_L * 9.f);
    }
    // (see shrink_factor: with these signs the exponential's argument is <= 0 or NaN; a NaN constant fails the test)
    const bool fastexp = AB ? ((!kc.has_nv || (a.noisevar_nonneg && kc.nv_scale >= 0.f)) && kc.madab >= 0.f && kc.rmadLm9 >= 0.f)
                            : kc.nv_const * kc.levelFactor >= 0.f;
    const int R0 = strip * FS_R, Rb = min(R0 + FS_R, H);
    const bool first = strip == 0, last = Rb == H;
    const bool inner = !first && !last;            // window rows [0, 64 + rad) are image rows, rows [0, 64) of them are written
    const int rs0 = max(0, R0 - rad);              // first image row with a factor in the window
    const int ro0 = rs0, ro1 = last ? H : Rb - rad;    // rows this strip writes
    const int nov = 2 * rad + 1;
    const int NB = (W + rad + FS_C - 1) / FS_C;
    const size_t slot = (size_t)(2 * MAXR + 2) * a.wpad;
    float *const hand_hb = a.hand + (size_t)(sub * FS_RING + strip % FS_RING) * slot;        // what the strip above left for this one, block-major:
    float *const hand_nx = a.hand + (size_t)(sub * FS_RING + (strip + 1) % FS_RING) * slot;  // [block][2 MAXR + 2 rows][64 columns]
    int *const prog = a.progress + sub * a.nstrips;

    // ---- elementwise waves: registers that travel a step ahead (clamped addresses: every load is unconditional)
    float pc[NQ], pl[NQ], pn[NQ], cu[NQ];
    auto fetch = [&](int J) {                      // coefficients of block J for its factors
        const int col = min(J * FS_C + lane, W - 1);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int r = min(max(R0 - rad + rbase + rstride * q, 0), Rb - 1);
            const int i = r * W + col;
            pc[q] = coef[i];
            pl[q] = AB ? coefL[i] : 0.f;
            pn[q] = kc.has_nv ? a.noisevar[i] : 0.f;
        }
    };
    auto fetch_upd = [&](int J) {                  // coefficients of block J's (lagged) columns for their update
        const int col = min(max(J * FS_C - rad + lane, 0), W - 1);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int r = min(max(R0 - rad + rbase + rstride * q, 0), H - 1);
            cu[q] = coef[r * W + col];
        }
    };
    if (ew >= 0) fetch(0);

    // ---- column-sum wave: what the strip above hands down for a block, fetched a step ahead
    float hpre[NOVMAX], tvpre = 0.f;
    int seen = 0, flagpre = 0;                     // flagpre: the strip above's counter, read a step ahead as well (a poll is a memory round trip
                                                   // even when the counter has long moved on; only a counter that is still short is polled for)
    auto prefetch_hand = [&](int J) {
        seen = max(seen, flagpre);
        if (seen < J + 1) {
            FS_T0
            int v, spins = 0;
            long long t_wait = 0;
            while ((v = __hip_atomic_load(prog + strip - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < J + 1) {
                __builtin_amdgcn_s_sleep(2);
                // the strip above belongs to a ticket taken before this one, i.e. to a running workgroup: the wait is bounded by that strip's
                // progress.  Should that ever not hold -- FIVE SECONDS of the constant 100 MHz clock without the counter getting there, not a
                // number of polls: under a debugger, thread tracing, power capping or CU masking a healthy run is merely slow --, say where
                // (pinned host words) and GO ON with whatever the slot holds: the launch then ends in finite time with wrong coefficients in
                // this band, and the library reports the fault at its next synchronisation point (artgpu_synchronize / any call that waits
                // for the stream: check_async_faults in artgpu_api.hip; artgpu_last_error names band, strip and block).  Round 5 trapped here;
                // a wave trap surfaces as a queue exception that aborts the host PROCESS inside the runtime before any HIP call returns, so
                // the attribution was never read and a host application lost more than one frame (round-5 advisor).
                if ((++spins & 1023) == 0) {
                    const long long now = (long long)__builtin_amdgcn_s_memrealtime();
                    if (t_wait == 0) t_wait = now;
                    else if (now - t_wait > (a.wait_ticks > 0 ? a.wait_ticks : FS_WAIT_TICKS)) {
                        // (the strips below a waiting strip wait as well and give up moments later: the FIRST one to give up -- elected through
                        // a device word behind the ticket, cleared with it per launch -- is the one next to the cause and the one that is reported)
                        if (a.diag && lane == 0 && atomicCAS(a.ticket + 1, 0, 1) == 0) {
                            a.diag[1] = sub; a.diag[2] = strip; a.diag[3] = J; a.diag[4] = v;
                            __hip_atomic_store(a.diag, (int)FS_DIAG_MAGIC, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                        }
                        v = 1 << 30;          // (no further waits in this strip)
                        break;
                    }
                }
            }
            seen = v;
            FS_T1(6)
#ifdef FS_PROFILE
            pt[7] += 1;
#endif
        }
        const float *hp = hand_hb + (size_t)J * (2 * MAXR + 2) * FS_C + lane;
#pragma unroll
        for (int k = 0; k < NOVMAX; ++k) hpre[k] = k < nov ? ld_agent(hp + k * FS_C) : 0.f;
        tvpre = ld_agent(hp + (2 * MAXR + 1) * FS_C);
        flagpre = __hip_atomic_load(prog + strip - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    if (wv == 2 && !first) prefetch_hand(0);

    // ---- row-sum wave (lane = row of the strip)
    float tempval = 0.f, reclen = 0.f;
    int hlen = rad + 1;

    // One loop per ROLE (round 5; the roles used to be the arms of an if-chain inside one step loop).  The wait-count pass works on the instruction
    // stream, not on waves: behind the chain it had to assume that the loads and stores of EVERY arm may be in flight, so the first use of a
    // prefetched coefficient in the next step waited for vmcnt(0) -- for the stores and the prefetches the wave had issued a moment ago as well.
    // Every role still meets the others at one LDS-only barrier per step: the barrier counts arrivals, wherever in the code they happen.
    if (ew >= 0) {
    for (int T = 0; T < NB + 4; ++T) {
#ifdef FS_PROFILE
        const long long ts_ = __builtin_readcyclecounter();
#endif
        {
            FS_T0
            // ---- coefficient update of block J = T - 3 (FTblockDN.cc:698-714, 803-836), rows [ro0, ro1)
            if (T >= 3) {
                const int J = T - 3;
                const float *HBj = HB0 + (J % 3) * HBUF;
                const int col = J * FS_C - rad + lane;
                const bool tailblk = last && J * FS_C - rad + FS_C > W - 4;
                if (col >= 0 && col < W) {
                    if (!tailblk) {
                        // rows of an inner strip's first NQF passes are all there: straight-line code the scheduler can interleave (behind a
                        // row test every row is a dependent chain of its own, one after the other)
                        auto upd = [&](int q) {
                            const int wr = rbase + rstride * q;
                            const float sfd = HBj[wr * HS + lane];
                            const float sf = S[wr * FS_SWS + (col & (FS_SWIN - 1))];
                            const float num = sqr(sfd) + sqr(sf), den = sfd + sf + eps;
                            out[(R0 - rad + wr) * W + col] = cu[q] * num / den;
                        };
                        if (inner) {
#pragma unroll
                            for (int q = 0; q < NQB; ++q) upd(q);
                        }
                        {
#pragma unroll
                            for (int q = 0; q < NQ; ++q) {
                                const int r = R0 - rad + rbase + rstride * q;
                                if ((!inner || q >= NQB) && r >= ro0 && r < ro1) upd(q);
                            }
                        }
                    } else {
#pragma unroll 1
                        for (int q = 0; q < NQ; ++q) {
                            const int wr = rbase + rstride * q, r = R0 - rad + wr;
                            if (r >= ro0 && r < ro1) {
                                const float sfd = HBj[wr * HS + lane];
                                const float sf = S[wr * FS_SWS + (col & (FS_SWIN - 1))], c = cu[q];
                                const int i = r * W + col;
                                const float num = sqr(sfd) + sqr(sf), den = sfd + sf + eps;
                                if (i < nv4) out[i] = c * num / den; else out[i] = c * (num / den);
                            }
                        }
                    }
                }
            }
            if (T >= 2 && T - 2 < NB) fetch_upd(T - 2);
            // ---- shrink factors of block T (after the update in program order: the window slots they overwrite hold columns the update
            //      of block T - 3 has just read -- in the same rows, i.e. in this wave)
            if (T < NB) {
                const int col = T * FS_C + lane;
                const bool tailblk = last && T * FS_C + FS_C > W - 4;
                auto factors = [&](auto abtag, auto fasttag) {       // (one instantiation per kind of band: no per-row branch on the kind)
                    constexpr bool ABc = decltype(abtag)::value, FASTc = decltype(fasttag)::value;
                    if (!tailblk) {
                        auto fac = [&](int q) {
                            const float sf = shrink_factor<ABc, false, FASTc>(kc, pc[q], pl[q], pn[q], true);
                            if (col < W) S[(rbase + rstride * q) * FS_SWS + (col & (FS_SWIN - 1))] = sf;
                        };
                        if (inner) {
#pragma unroll
                            for (int q = 0; q < NQB; ++q) fac(q);
                        }
                        {
#pragma unroll
                            for (int q = 0; q < NQ; ++q) {
                                const int r = R0 - rad + rbase + rstride * q;
                                if ((!inner || q >= NQB) && r >= 0 && r < Rb) fac(q);     // (uniform: whole rows)
                            }
                        }
                    } else {
#pragma unroll 1
                        for (int q = 0; q < NQ; ++q) {
                            const int wr = rbase + rstride * q, r = R0 - rad + wr;
                            if (r >= 0 && r < Rb && col < W)
                                S[wr * FS_SWS + (col & (FS_SWIN - 1))] = shrink_factor<ABc, true>(kc, pc[q], pl[q], pn[q], r * W + col < nv4);
                        }
                    }
                };
                if (fastexp) { if (AB) factors(std::true_type{}, std::true_type{}); else factors(std::false_type{}, std::true_type{}); }
                else { if (AB) factors(std::true_type{}, std::false_type{}); else factors(std::false_type{}, std::false_type{}); }
                if (T + 1 < NB) fetch(T + 1);
            }
            FS_T1(3)
        }
        lds_barrier();
#ifdef FS_PROFILE
        pt[0] += __builtin_readcyclecounter() - ts_;
#endif
    }
    } else if (wv == 0) {
    for (int T = 0; T < NB + 4; ++T) {
#ifdef FS_PROFILE
        const long long ts_ = __builtin_readcyclecounter();
#endif
        {
            // ---- row sums of block J = T - 1 over columns [X0 - rad, X0 + 64 - rad) (boxblur.h:565-600, hblur_kernel)
            if (T >= 1 && T <= NB && R0 + lane < Rb) {
                FS_T0
                const int J = T - 1, X0 = J * FS_C;
                const float *srow = S + (rad + lane) * FS_SWS;
                auto s = [&](int k) -> float { return srow[(X0 + k) & (FS_SWIN - 1)]; };     // factor at column X0 + k
                float *hb = HB0 + (J % 3) * HBUF + (nov + lane) * HS;
                int jj = 0;
                if (J == 0) jj = rad;                                  // columns < 0 do not exist
                while (jj < FS_C) {
                    const int col = X0 - rad + jj;
                    if (col >= W) break;
                    if (col > rad && col + 8 <= W - rad && jj + 8 <= FS_C) {
                        float hi[8], lo[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) { hi[k] = s(jj + k); lo[k] = s(jj + k - nov); }
#pragma unroll
                        for (int k = 0; k < 8; ++k) { tempval = tempval + (hi[k] - lo[k]) * reclen; hb[jj + k] = tempval; }
                        jj += 8;
                        continue;
                    }
                    if (col == 0) {
                        tempval = s(jj - rad);
                        for (int q = 1; q <= rad; q++) tempval += s(jj - rad + q);
                        tempval = tempval / hlen;
                    } else if (col <= rad) {
                        tempval = (tempval * hlen + s(jj)) / (hlen + 1);
                        hlen++;
                        if (col == rad) reclen = 1.f / hlen;
                    } else if (col < W - rad) {
                        tempval = tempval + (s(jj) - s(jj - nov)) * reclen;
                    } else {
                        tempval = (tempval * hlen - s(jj - nov)) / (hlen - 1);
                        hlen--;
                    }
                    hb[jj] = tempval;
                    ++jj;
                }
                FS_T1(1)
            }
        }
        lds_barrier();
#ifdef FS_PROFILE
        pt[0] += __builtin_readcyclecounter() - ts_;
#endif
    }
    } else if (wv == 1) {
    for (int T = 0; T < NB + 4; ++T) {
#ifdef FS_PROFILE
        const long long ts_ = __builtin_readcyclecounter();
#endif
        {
            // ---- column sums of block J = T - 2 over rows [ro0, ro1) (boxblur.h:602-742, vblur_combine_kernel); the value of row r replaces
            //      the row-blurred value of row r - rad - 1, which that step was the last to need
            const int J = T - 2;
            if (J >= 0 && J < NB) {
                FS_T0
                float *HBj = HB0 + (J % 3) * HBUF;
                const int X0 = J * FS_C;
                const int col = X0 - rad + lane;
                float tv = first ? 0.f : HBj[HROWS * HS + lane];
                if (col >= 0 && col < W) {
                    const bool vec = col < (W / 4) * 4;
                    const bool allvec = X0 - rad + FS_C <= (W / 4) * 4;
                    float lenf = first ? (float)(rad + 1) : (float)nov;
                    int leni = first ? rad + 1 : nov;
                    const float rlen = 1.f / (float)nov;
                    const float *hbc = HBj + (nov - R0) * HS + lane;       // hbc[r * HS] = row-blurred value of row r
                    float *vout = HBj + (rad - R0) * HS + lane;            // vout[r * HS] <- column sum of row r
                    int r = ro0;
                    while (r < ro1) {
                        if (allvec && r > rad && r + 8 <= H - rad && r + 8 <= ro1) {       // (no lane of the block takes the scalar form)
                            float hi[8], lo[8];
#pragma unroll
                            for (int k = 0; k < 8; ++k) { hi[k] = hbc[(r + k + rad) * HS]; lo[k] = hbc[(r + k - rad - 1) * HS]; }
#pragma unroll
                            for (int k = 0; k < 8; ++k) { tv = tv + (hi[k] - lo[k]) * rlen; vout[(r + k) * HS] = tv; }
                            r += 8;
                            continue;
                        }
                        if (r == 0) {
                            if (vec) {
                                tv = hbc[0];
                                for (int i = 1; i <= rad; i++) tv = tv + hbc[i * HS];
                                tv = tv / lenf;
                            } else {
                                tv = hbc[0] / leni;
                                for (int i = 1; i <= rad; i++) tv += hbc[i * HS] / leni;
                            }
                        } else if (r <= rad) {
                            if (vec) {
                                const float lenp1 = lenf + 1.f;
                                tv = (tv * lenf + hbc[(r + rad) * HS]) / lenp1;
                                lenf = lenp1;
                            } else {
                                tv = (tv * leni + hbc[(r + rad) * HS]) / (leni + 1);
                                leni++;
                            }
                        } else if (r < H - rad) {
                            const float d = hbc[(r + rad) * HS] - hbc[(r - rad - 1) * HS];
                            tv = vec ? tv + d * rlen : tv + d / leni;
                        } else {
                            if (vec) {
                                const float lenm1 = lenf - 1.f;
                                tv = (tv * lenf - hbc[(r - rad - 1) * HS]) / lenm1;
                                lenf = lenm1;
                            } else {
                                tv = (tv * leni - hbc[(r - rad - 1) * HS]) / (leni - 1);
                                leni--;
                            }
                        }
                        vout[r * HS] = tv;
                        ++r;
                    }
                }
                HBj[(HROWS + 1) * HS + lane] = tv;                        // leaves the strip through the hand-over wave
                FS_T1(2)
            }
        }
        lds_barrier();
#ifdef FS_PROFILE
        pt[0] += __builtin_readcyclecounter() - ts_;
#endif
    }
    } else {
    for (int T = 0; T < NB + 4; ++T) {
#ifdef FS_PROFILE
        const long long ts_ = __builtin_readcyclecounter();
#endif
        {
            // ---- the hand-over.  Everything that crosses to another workgroup goes through THIS wave: its write-through (sc1) stores are
            //      the only stores it has in flight, so waiting for them a step later costs nothing and neither holds up the coefficient
            //      traffic of the other waves nor needs a release fence (which would write back the XCD's whole dirty L2, full of this
            //      kernel's coefficient stores); the strip below reads with sc1 loads and needs no acquire.
            //      step T: what the strip above left for block T - 1 (sent for a step ago) goes to that block's buffer, where the row sums
            //      of the block are being written; block T - 4 -- its stores have left -- is published; block T is sent for; block T - 2's
            //      last row-blurred rows and block T - 3's column sums go to the strip below.
            FS_T0
            if (!first && T >= 1 && T - 1 < NB) {
                float *HBj = HB0 + ((T - 1) % 3) * HBUF;
#pragma unroll
                for (int k = 0; k < NOVMAX; ++k)
                    if (k < nov) HBj[k * HS + lane] = hpre[k];
                HBj[HROWS * HS + lane] = tvpre;
            }
            if (!last && T >= 4) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (gfx942 / gfx950: this wave's data stores have left -- see the #error at the top)
                if (lane == 0 && !(sub == a.stall_band && strip == a.stall_strip)) __hip_atomic_store(prog + strip, T - 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (!first && T < NB) prefetch_hand(T);
            if (!last) {
                if (T >= 2 && T - 2 < NB) {
                    const float *HBj = HB0 + ((T - 2) % 3) * HBUF;
                    float *hp = hand_nx + (size_t)(T - 2) * (2 * MAXR + 2) * FS_C + lane;
#pragma unroll
                    for (int k = 0; k < NOVMAX; ++k)
                        if (k < nov) st_agent(hp + k * FS_C, HBj[(FS_R + k) * HS + lane]);
                }
                if (T >= 3 && T - 3 < NB)      // (every lane stores: the strip below reads all 64 slots of the block)
                    st_agent(hand_nx + (size_t)(T - 3) * (2 * MAXR + 2) * FS_C + (2 * MAXR + 1) * FS_C + lane, HB0[((T - 3) % 3) * HBUF + (HROWS + 1) * HS + lane]);
            }
            FS_T1(4)
        }
        lds_barrier();
#ifdef FS_PROFILE
        pt[0] += __builtin_readcyclecounter() - ts_;
#endif
    }
    }
  }
#ifdef FS_PROFILE
    if (a.prof) {
        unsigned long long *pp = reinterpret_cast<unsigned long long *>(a.prof);
        if (tid == 0) { atomicAdd(pp + 0, (unsigned long long)pt[0]); atomicAdd(pp + 1, (unsigned long long)pt[1]); }
        if (tid == 64) atomicAdd(pp + 2, (unsigned long long)pt[2]);
        if (tid == 128) { atomicAdd(pp + 4, (unsigned long long)pt[4]); atomicAdd(pp + 6, (unsigned long long)pt[6]); atomicAdd(pp + 7, (unsigned long long)pt[7]); }
        if (lane == 0) atomicAdd(pp + 8 + wv, (unsigned long long)(pt[1] + pt[2] + pt[3] + pt[4]));      // busy time per wave
    }
#endif
}

template <int MAXR>
constexpr int fs_lds_bytes()
{
    constexpr int SROWS = FS_R + MAXR, HROWS = FS_R + 2 * MAXR + 1, HS = FS_C + 1;
    return (SROWS * FS_SWS + 3 * (HROWS + 2) * HS) * (int)sizeof(float);
}

template <int MAXR>
hipError_t launch_one(const FusedShrinkArgs &a, hipStream_t s)
{
    constexpr int lds = fs_lds_bytes<MAXR>();
    static_assert(lds <= 160 * 1024 - 64, "LDS");
    hipError_t e = dyn_lds_once(reinterpret_cast<const void *>(shrink_blur_kernel<MAXR>), lds);
    if (e != hipSuccess) return e;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int total = a.nsub * a.nstrips;
    hipLaunchKernelGGL((shrink_blur_kernel<MAXR>), dim3(total < cus ? total : cus), dim3(FS_T), lds, s, a);
    return hipGetLastError();
}

} // namespace

// What a workgroup of the current device may have: the kernel needs 137 KB (radii up to 7) or 157 KB (up to 15) of dynamic LDS and 1024
// threads; a device (or a build for another target) with less gets the three-kernel form of the passes instead of a failing launch.
static bool fs_device_fits(int lds_bytes) { return device_block_fits(lds_bytes, FS_T); }

bool shrink_blur_supported(int w, int h, const int *rad, int level0, int nsub)
{
    if (w < 64 || h < 64 || nsub < 1 || (long long)w * h >= (1LL << 31)) return false;
    int maxr = 0;
    for (int sub = 0; sub < nsub; ++sub) {
        const int r = rad[level0 + sub / 3];
        if (r < 1 || r > 15) return false;
        maxr = r > maxr ? r : maxr;
    }
    return fs_device_fits(maxr > 7 ? fs_lds_bytes<15>() : fs_lds_bytes<7>());
}
static int fs_strips(int h) { return (h + FS_R - 1) / FS_R; }
static int fs_wpad(int w) { return ((w + 15 + FS_C - 1) / FS_C) * FS_C; }      // NB blocks of 64 columns (rad <= 15)
size_t shrink_blur_scratch_floats(int w, int h, int nsub, int maxr)
{
    const int rows = 2 * (maxr > 7 ? 15 : 7) + 2;
    // FS_RING hand-over slots of `rows` x wpad floats per band (live only while the two strips either side of them are in flight), then the
    // progress counters and the ticket (ints)
    return (size_t)nsub * FS_RING * rows * fs_wpad(w) + (size_t)nsub * fs_strips(h) + 64;
}

hipError_t launch_shrink_blur(FusedShrinkArgs a, float *scratch, hipStream_t s)
{
    int maxr = 0;
    if (a.nsub_ch <= 0) a.nsub_ch = a.nsub - a.nL > 0 ? a.nsub - a.nL : 1;
    const int nper = a.nL > a.nsub_ch ? a.nL : a.nsub_ch;
    for (int sub = 0; sub < nper; ++sub) { const int r = a.rad[a.level0 + sub / 3]; maxr = r > maxr ? r : maxr; }
    const int rows = 2 * (maxr > 7 ? 15 : 7) + 2;
    a.nstrips = fs_strips(a.h);
    a.wpad = fs_wpad(a.w);
    a.hand = scratch;
    a.progress = reinterpret_cast<int *>(scratch + (size_t)a.nsub * FS_RING * rows * a.wpad);
    a.ticket = a.progress + a.nsub * a.nstrips;
    hipError_t e = hipMemsetAsync(a.progress, 0, ((size_t)a.nsub * a.nstrips + 2) * sizeof(int), s);     // + the ticket and the word that elects the first strip to report a fault
    if (e != hipSuccess) return e;
    return maxr > 7 ? launch_one<15>(a, s) : launch_one<7>(a, s);
}

} // namespace artgpu
