// art_amd/csrc/dct64.h -- register-resident fast DCT-II / DCT-III of length 64 (Lee's recursion),
// used by detail.hip.  Each lane transforms its own 64-sample line held in registers; all indices
// are compile-time constants after template expansion, so the code is straight-line VALU work
// (~2 N log2 N operations instead of N^2) with no table traffic.
//   lee_fwd<N>(x): x <- F,  F[k] = sum_n x[n] cos(pi (2n+1) k / (2N))        (unnormalised DCT-II)
//   lee_inv<N>(z): z <- y,  y[n] = sum_k z[k] cos(pi (2n+1) k / (2N))        (its transpose, DCT-III)
// FFTW's REDFT10 = 2 * lee_fwd ; REDFT01(X) = lee_inv(X[0], 2 X[1], 2 X[2], ...)  (FTblockDN.cc:1604,1614).
// The twiddles 1/(2 cos(pi (2n+1) / (2N))) are generated in double precision (tools in the file header of
// detail.hip); the stage is tolerance-checked against a double-accumulated direct DCT (oracle/detail.c).
#pragma once
#include <hip/hip_runtime.h>

namespace artgpu {

template <int N> struct LeeTw;
template <> struct LeeTw<2> { static constexpr float c[1] = {0.707106781f}; };

template <> struct LeeTw<4> { static constexpr float c[2] = {0.5411961f, 1.30656296f}; };

template <> struct LeeTw<8> { static constexpr float c[4] = {0.509795579f, 0.601344887f, 0.899976223f, 2.56291545f}; };

template <> struct LeeTw<16> { static constexpr float c[8] = {0.502419286f, 0.522498615f, 0.566944035f, 0.646821783f, 0.788154623f, 1.06067769f, 1.7224471f, 5.10114862f}; };

template <> struct LeeTw<32> { static constexpr float c[16] = {0.500602998f, 0.50547096f, 0.51544731f, 0.531042591f, 0.553103896f, 0.582934968f, 0.622504123f, 0.674808341f, 0.744536271f, 0.839349645f, 0.972568238f, 1.16943993f, 1.48416462f, 2.05778101f, 3.40760842f, 10.1900081f}; };

template <> struct LeeTw<64> { static constexpr float c[32] = {0.500150636f, 0.501358452f, 0.503788726f, 0.507471172f, 0.512451479f, 0.518792713f, 0.526577315f, 0.535909817f, 0.546920438f, 0.559769813f, 0.574655184f, 0.591818536f, 0.611557348f, 0.634238937f, 0.660319808f, 0.690372128f, 0.725120522f, 0.765494165f, 0.812702091f, 0.868344715f, 0.934583597f, 1.01440826f, 1.11207162f, 1.23383274f, 1.38929396f, 1.59397228f, 1.87467598f, 2.28205007f, 2.92462843f, 4.08461108f, 6.79675071f, 20.3738782f}; };

// The recursion over an element type T: float, or a pair of floats (two independent transforms in lockstep, every operation a packed
// v_pk_add_f32 / v_pk_mul_f32 -- the same IEEE additions and multiplications on each half, so the same bits).
typedef float lee_f2 __attribute__((ext_vector_type(2)));

template <int N, typename T>
__device__ __forceinline__ void lee_fwd_t(T (&x)[N])
{
    if constexpr (N == 1) {
        return;
    } else {
        T a[N / 2], b[N / 2];
#pragma unroll
        for (int n = 0; n < N / 2; ++n) {
            a[n] = x[n] + x[N - 1 - n];
            b[n] = (x[n] - x[N - 1 - n]) * LeeTw<N>::c[n];
        }
        lee_fwd_t<N / 2, T>(a);
        lee_fwd_t<N / 2, T>(b);
#pragma unroll
        for (int k = 0; k < N / 2; ++k) {
            x[2 * k] = a[k];
            x[2 * k + 1] = (k + 1 < N / 2) ? b[k] + b[k + 1] : b[k];
        }
    }
}

template <int N, typename T>
__device__ __forceinline__ void lee_inv_t(T (&z)[N])
{
    if constexpr (N == 1) {
        return;
    } else {
        T a[N / 2], b[N / 2];
#pragma unroll
        for (int k = 0; k < N / 2; ++k) {
            a[k] = z[2 * k];
            b[k] = (k > 0) ? z[2 * k + 1] + z[2 * k - 1] : z[1];
        }
        lee_inv_t<N / 2, T>(a);
        lee_inv_t<N / 2, T>(b);
#pragma unroll
        for (int n = 0; n < N / 2; ++n) {
            const T t = b[n] * LeeTw<N>::c[n];
            z[n] = a[n] + t;
            z[N - 1 - n] = a[n] - t;
        }
    }
}

// One line of N samples: the two half-size transforms of the first recursion step are independent and have the same shape, so they run
// as ONE transform over pairs (even half, odd half) -- about half the vector instructions of the scalar recursion, the same operations.
#ifndef LEE_SCALAR
template <int N>
__device__ __forceinline__ void lee_fwd(float (&x)[N])
{
    lee_f2 p[N / 2];
#pragma unroll
    for (int n = 0; n < N / 2; ++n) {
        p[n].x = x[n] + x[N - 1 - n];
        p[n].y = (x[n] - x[N - 1 - n]) * LeeTw<N>::c[n];
    }
    lee_fwd_t<N / 2, lee_f2>(p);
#pragma unroll
    for (int k = 0; k < N / 2; ++k) {
        x[2 * k] = p[k].x;
        x[2 * k + 1] = (k + 1 < N / 2) ? p[k].y + p[k + 1].y : p[k].y;
    }
}

template <int N>
__device__ __forceinline__ void lee_inv(float (&z)[N])
{
    lee_f2 p[N / 2];
#pragma unroll
    for (int k = 0; k < N / 2; ++k) {
        p[k].x = z[2 * k];
        p[k].y = (k > 0) ? z[2 * k + 1] + z[2 * k - 1] : z[1];
    }
    lee_inv_t<N / 2, lee_f2>(p);
#pragma unroll
    for (int n = 0; n < N / 2; ++n) {
        const float t = p[n].y * LeeTw<N>::c[n];
        z[n] = p[n].x + t;
        z[N - 1 - n] = p[n].x - t;
    }
}
#else
template <int N> __device__ __forceinline__ void lee_fwd(float (&x)[N]) { lee_fwd_t<N, float>(x); }
template <int N> __device__ __forceinline__ void lee_inv(float (&z)[N]) { lee_inv_t<N, float>(z); }
#endif

} // namespace artgpu
