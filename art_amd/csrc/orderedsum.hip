// orderedsum.hip -- fp32 sums in index order, evaluated in parallel and bit-exactly.
//
// The reference accumulates several statistics as `float acc = 0; for (i...) acc += x[i];` over millions of values
// (ShrinkAll_info, rtengine/FTblockDN.cc:1237-1290).  fp32 addition is not associative, so the result is defined by the
// order; a tree reduction gives a different number.  A serial chain costs one dependent v_add_f32 per value (~8 cycles on
// one otherwise idle SIMD: 18 ms for the 2.8 M values of a 45 MP frame).  This file evaluates the same chain with a scan:
//
//   While the accumulator s stays inside one binade [2^e, 2^(e+1)), it is an integer m = s/u in [2^23, 2^24) with
//   u = 2^(e-23), and adding x >= 0 with round-to-nearest-even is   m' = m + q + c,   q = floor(x/u), f = x/u - q,
//   c = [f > 1/2], or, on a tie f == 1/2, whatever makes m + q + c even.  So every value is a map  m -> m + d[m & 1]
//   with two integers (d[0], d[1]); such maps are closed under composition (the parity of the result is known from the
//   parity of the input), composition is associative, and an inclusive scan over the maps gives every prefix of the chain.
//   Ties are part of the map, not an exception (they are common: the maps are full of clamped constants like 100.f).
//   What the scan cannot absorb is handled one value at a time with a real v_add_f32 and a restart: s == 0 or tiny,
//   a value >= s/16 (q would not fit the 32-bit budget of a chunk), the value that carries m to >= 2^24
//   (the binade changes: at most ~30 times per chain), and negative/NaN/inf inputs.
//
// All arithmetic on the data is exact by construction (power-of-two scaling, floor, an exact difference), so the result is the
// reference's bit for bit; tests/test_gpu_dninfo.py checks it against numpy's sequential float32 cumsum on adversarial inputs.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "devmath.h"
#include "kernels.h"

namespace artgpu {

namespace {

constexpr int OS_E = 24;                 // consecutive values per lane (2 x 4 staged chunks of 64 x (OS_E + 1) floats fit 64 KB of LDS)
constexpr int OS_CHUNK = 64 * OS_E;      // values one wave consumes per pass
constexpr int OS_PAD = OS_E + 1;         // LDS row pitch: odd, so the 64 lanes of a column read 64 different banks
constexpr int OS_LDS = 64 * OS_PAD;      // floats per staged chunk

__device__ __forceinline__ int os_padded(int i) { return i + i / OS_E; }

struct OsMap { int d0, d1; };            // m -> m + (m & 1 ? d1 : d0)
__device__ __forceinline__ OsMap os_then(OsMap f, OsMap g)      // apply f, then g
{
    OsMap h;
    h.d0 = f.d0 + ((f.d0 & 1) ? g.d1 : g.d0);
    h.d1 = f.d1 + (((1 + f.d1) & 1) ? g.d1 : g.d0);
    return h;
}
// the map of one value in units of u (y = x/u, 0 <= y < 2^19)
__device__ __forceinline__ OsMap os_map_of(float y)
{
    const float f = __builtin_amdgcn_fractf(y);    // y - floor(y), exact
    const int qi = (int)y;                         // truncation = floor for y >= 0
    const bool tie = f == 0.5f;
    const int c = f > 0.5f ? 1 : 0;
    OsMap g;
    g.d0 = qi + (tie ? (qi & 1) : c);
    g.d1 = qi + (tie ? ((qi & 1) ^ 1) : c);
    return g;
}
// 0 <= y < 2^19 in one unsigned compare: negatives (sign bit), NaN and inf all land above.  (-0.f counts as "does not fit":
// it is then added for real, which is still exact.)
__device__ __forceinline__ bool os_fits(float y) { return __float_as_uint(y) < 0x49000000u; }

// s <- (((s + x[0]) + x[1]) + ... + x[cnt-1]) with fp32 rounding at every step.  `lds` holds the values at os_padded()
// positions; values at index >= cnt must be absent (they are read as 0).  Wave-uniform in and out; call with all 64 lanes.
__device__ __forceinline__ float ordered_sum_chunk(float s, const float *lds, int cnt, int lane)
{
    float x[OS_E];
#pragma unroll
    for (int j = 0; j < OS_E; ++j) {
        const int idx = lane * OS_E + j;
        x[j] = idx < cnt ? lds[lane * OS_PAD + j] : 0.f;
    }
    int pos = 0;                                      // values below pos are already in s
    for (;;) {
        if (s != s) return s;                         // NaN absorbs everything
        if (s == __builtin_inff()) {                  // +inf stays unless a NaN follows
            bool nan = false;
#pragma unroll
            for (int j = 0; j < OS_E; ++j) nan |= (lane * OS_E + j >= pos) && (x[j] != x[j]);
            return __ballot(nan) ? __builtin_nanf("") : s;
        }
        int first = OS_E;                             // first value of this lane that must be added for real
        int mb = 0;                                   // m just before it
        float u = 0.f;
        if (!(s >= 1e-30f)) {
            // zero (or denormal-range) accumulator: adding zeros changes nothing; the first non-zero value is added for real
#pragma unroll
            for (int j = OS_E - 1; j >= 0; --j)
                if (lane * OS_E + j >= pos && !(x[j] == 0.f)) first = j;
        } else {
            const uint32_t es = __float_as_uint(s) & 0x7f800000u;          // biased exponent of s, in place
            u = __uint_as_float(es - (23u << 23));
            const float inv_u = __uint_as_float((277u << 23) - es);       // 2^(23-e)
            const int m_in = (int)(s * inv_u);
            OsMap d = {0, 0};
            bool ev = false;
#pragma unroll
            for (int j = 0; j < OS_E; ++j) {
                float y = (lane * OS_E + j >= pos ? x[j] : 0.f) * inv_u;
                ev |= !os_fits(y);
                y = ev ? 0.f : y;                     // past an exception the lane's map is never used
                // a = q + [f > 1/2] = ceil(y - 1/2) (the subtraction is exact for y < 2^23); a tie is y - 1/2 being an integer
                const float t = y - 0.5f, ct = ceilf(t);
                const int ai = (int)ct;
                const bool tie = ct == t;
                if (__ballot(tie) == 0) {             // wave-uniform: no lane has a tie at this position -> plain shifts
                    d.d0 += ai; d.d1 += ai;
                } else {
                    OsMap g;
                    g.d0 = ai + (tie ? (ai & 1) : 0);
                    g.d1 = ai + (tie ? ((ai & 1) ^ 1) : 0);
                    d = os_then(d, g);
                }
            }
            // inclusive scan over the lanes in lane order: row_shr 1,2,4,8 inside each row of 16, then row_bcast 15 / 31
            OsMap incl = d;
#define OS_STEP(ctrl, rmask)                                                                        \
            {                                                                                        \
                OsMap p;                                                                             \
                p.d0 = __builtin_amdgcn_update_dpp(0, incl.d0, ctrl, rmask, 0xf, false);             \
                p.d1 = __builtin_amdgcn_update_dpp(0, incl.d1, ctrl, rmask, 0xf, false);             \
                incl = os_then(p, incl);        /* lanes without a source see the identity map */   \
            }
            OS_STEP(0x111, 0xf) OS_STEP(0x112, 0xf) OS_STEP(0x114, 0xf) OS_STEP(0x118, 0xf)
            OS_STEP(0x142, 0xa) OS_STEP(0x143, 0xc)
#undef OS_STEP
            const int pin = m_in & 1;
            const int after = pin ? incl.d1 : incl.d0;
            const int total = __shfl(after, 63);
            if (__ballot(ev) == 0 && m_in + total < (1 << 24)) return (float)(m_in + total) * u;    // the common case
            // something in this chunk needs a real addition: find the first such value
            int m = m_in + __shfl_up(after, 1);
            if (lane == 0) m = m_in;
            bool open = true;
#pragma unroll
            for (int j = 0; j < OS_E; ++j) {
                const float y = (lane * OS_E + j >= pos ? x[j] : 0.f) * inv_u;
                if (open) {
                    bool stop = !os_fits(y);
                    int mn = m;
                    if (!stop) {
                        const OsMap g = os_map_of(y);
                        mn = m + ((m & 1) ? g.d1 : g.d0);
                        stop = mn >= (1 << 24);
                    }
                    if (stop) { first = j; mb = m; open = false; }
                    else m = mn;
                }
            }
        }
        const unsigned long long who = __ballot(first < OS_E);
        if (who == 0) return u != 0.f ? __builtin_nanf("") : s;   // zero accumulator and nothing but zeros left (NaN: cannot happen, see above)
        const int l0 = __ffsll((long long)who) - 1;
        const int j0 = __shfl(first, l0);
        if (u != 0.f) s = (float)__shfl(mb, l0) * u;  // the chain up to that value
        s = s + lds[l0 * OS_PAD + j0];                // the reference's own operation
        pos = l0 * OS_E + j0 + 1;
        if (pos >= cnt) return s;
    }
}

} // namespace

// sum of x[0..n) in index order (one workgroup; wave 0 runs the chain, all four waves stage)
__global__ void __launch_bounds__(256) ordered_sum_kernel(const float *__restrict__ x, long long n, float *out)
{
    __shared__ float buf[OS_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float s = 0.f;
    float pre[OS_CHUNK / 256];
#pragma unroll
    for (int i = 0; i < OS_CHUNK / 256; ++i) { const long long e = tid + 256 * i; pre[i] = e < n ? x[e] : 0.f; }
    for (long long base = 0; base < n; base += OS_CHUNK) {
        const int m = (int)((n - base) < OS_CHUNK ? (n - base) : OS_CHUNK);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < OS_CHUNK / 256; ++i) buf[os_padded(tid + 256 * i)] = pre[i];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < OS_CHUNK / 256; ++i) { const long long e = base + OS_CHUNK + tid + 256 * i; pre[i] = e < n ? x[e] : 0.f; }
        if (wave == 0) s = ordered_sum_chunk(s, buf, m, lane);
    }
    if (tid == 0) *out = s;
}

// ShrinkAll_info's lvl == 1 statistics (FTblockDN.cc:1237-1290) for the nine crops: chro, lume, red_yel, skin_c in scan order
// (one wave each), the two counts as integers.  `sigma`/`sigma_L` are dead in the reference (ipdenoise.cc:935-937).
// One workgroup of eight waves per crop: waves 4-7 stage chunk c (their global loads for chunk c+1 already in flight) into one
// half of the LDS while waves 0-3 run their chains over chunk c-1 in the other half.
__global__ void __launch_bounds__(512) dninfo_stats_kernel(DnInfoArgs a)
{
    __shared__ float buf[2][4][OS_LDS];
    __shared__ int s_cnt[2];
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool producer = wave >= 4;
    const int pt = tid - 256;
    const long long n2 = (long long)a.wid * a.hei;
    const long long nchunks = (n2 + OS_CHUNK - 1) / OS_CHUNK;
    const float *hue = a.maps + (size_t)k * 3 * n2, *chrom = hue + n2, *lum = chrom + n2;
    if (tid < 2) s_cnt[tid] = 0;
    float s = 0.f;          // consumer wave w: 0 chro, 1 lume, 2 red_yel, 3 skin_c
    int nry = 0, nsk = 0;
    constexpr int PER = OS_CHUNK / 256;
    float ph[PER], pc[PER], pl[PER];
    if (producer) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const long long e = pt + 256 * i;
            const bool in = e < n2;
            ph[i] = in ? hue[e] : 0.f; pc[i] = in ? chrom[e] : 0.f; pl[i] = in ? lum[e] : 0.f;
        }
    }
    __syncthreads();
    for (long long c = 0; c <= nchunks; ++c) {
        if (producer) {
            if (c < nchunks) {
                const long long base = c * OS_CHUNK;
                const int m = (int)((n2 - base) < OS_CHUNK ? (n2 - base) : OS_CHUNK);
                float (*dst)[OS_LDS] = buf[c & 1];
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const int e = pt + 256 * i;
                    float ry = 0.f, sk = 0.f;
                    if (e < m) {
                        const float h = ph[i], cv = pc[i];
                        if (h > -0.8f && h < 2.0f && cv > 10000.f) { ry = cv; ++nry; }
                        if (h > 0.f && h < 1.6f && cv < 10000.f) { sk = cv; ++nsk; }
                    }
                    const int o = os_padded(e);
                    dst[0][o] = pc[i]; dst[1][o] = pl[i]; dst[2][o] = ry; dst[3][o] = sk;
                }
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const long long e = base + OS_CHUNK + pt + 256 * i;
                    const bool in = e < n2;
                    ph[i] = in ? hue[e] : 0.f; pc[i] = in ? chrom[e] : 0.f; pl[i] = in ? lum[e] : 0.f;
                }
            }
        } else if (c >= 1) {
            const long long base = (c - 1) * OS_CHUNK;
            const int m = (int)((n2 - base) < OS_CHUNK ? (n2 - base) : OS_CHUNK);
            s = ordered_sum_chunk(s, buf[(c - 1) & 1][wave], m, lane);
        }
        __syncthreads();
    }
    if (producer) { atomicAdd(&s_cnt[0], nry); atomicAdd(&s_cnt[1], nsk); }
    __syncthreads();
    if (!producer && lane == 0) a.stats[k * 8 + wave] = s;
    if (tid == 0) { a.stats[k * 8 + 4] = __int_as_float(s_cnt[0]); a.stats[k * 8 + 5] = __int_as_float(s_cnt[1]); }
}

hipError_t launch_dninfo_stats(const DnInfoArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(dninfo_stats_kernel, dim3(9), dim3(512), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_ordered_sum(const float *x, long long n, float *out, hipStream_t s)
{
    hipLaunchKernelGGL(ordered_sum_kernel, dim3(1), dim3(256), 0, s, x, n, out);
    return hipGetLastError();
}

} // namespace artgpu
