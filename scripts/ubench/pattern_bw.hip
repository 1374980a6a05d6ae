// scripts/ubench/pattern_bw.hip -- what does a COPY reach when the plane is walked the way the box blurs have to walk it?
//   linear          every thread 16 bytes, grid-stride
//   row-march R x C a workgroup owns R rows of a band and marches along them in chunks of C columns (hblur: the recurrence runs along the row)
//   col-march C     a workgroup owns C columns and marches down the rows, RB rows per step (vblur)
// Every variant keeps D steps of loads in flight per thread (registers), so latency is not what is measured.
// build: hipcc --offload-arch=gfx950 -O3 pattern_bw.hip -o pattern_bw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int W = 4096, H = 2732, NB = 15;

__global__ void __launch_bounds__(256) lin(const v4f *s, v4f *d, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i0 = blockIdx.x * (size_t)256 + threadIdx.x; i0 < n; i0 += stride * 4) {
        v4f x[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = s[min(i0 + k * stride, n - 1)];
#pragma unroll
        for (int k = 0; k < 4; ++k) if (i0 + k * stride < n) d[i0 + k * stride] = x[k];
    }
}
// R rows x C columns per step; 256 threads; each thread moves (R*C/4)/256 float4 per step; D steps in flight
template <int R, int C, int D>
__global__ void __launch_bounds__(256) rowmarch(const float *s, float *d)
{
    constexpr int PER = R * C / 4 / 256;        // float4 per thread and step
    static_assert(PER >= 1, "");
    const int band = blockIdx.y, r0 = blockIdx.x * R;
    const float *sp = s + (size_t)band * W * H;
    float *dp = d + (size_t)band * W * H;
    const int t = threadIdx.x;
    v4f buf[D][PER];
    auto addr = [&](int step, int k) -> size_t {
        const int e = t + 256 * k;                 // float4 index inside the R x C tile
        const int row = e / (C / 4), c4 = e % (C / 4);
        return (size_t)min(r0 + row, H - 1) * W + min(step * C + 4 * c4, W - 4);
    };
    constexpr int NS = W / C;
#pragma unroll
    for (int q = 0; q < D - 1; ++q)
#pragma unroll
        for (int k = 0; k < PER; ++k) buf[q][k] = *(const v4f *)(sp + addr(q, k));
    for (int s0 = 0; s0 < NS; s0 += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const int st = s0 + u;
#pragma unroll
            for (int k = 0; k < PER; ++k) buf[(u + D - 1) % D][k] = *(const v4f *)(sp + addr(st + D - 1, k));
#pragma unroll
            for (int k = 0; k < PER; ++k) *(v4f *)(dp + addr(st, k)) = buf[u][k];
        }
    }
}
// C columns, RB rows per step
template <int C, int RB, int D>
__global__ void __launch_bounds__(256) colmarch(const float *s, float *d)
{
    constexpr int PER = RB * C / 4 / 256;
    static_assert(PER >= 1, "");
    const int band = blockIdx.y, c0 = blockIdx.x * C;
    const float *sp = s + (size_t)band * W * H;
    float *dp = d + (size_t)band * W * H;
    const int t = threadIdx.x;
    v4f buf[D][PER];
    auto addr = [&](int step, int k) -> size_t {
        const int e = t + 256 * k;
        const int row = e / (C / 4), c4 = e % (C / 4);
        return (size_t)min(step * RB + row, H - 1) * W + c0 + 4 * c4;
    };
    const int NS = (H + RB - 1) / RB;
#pragma unroll
    for (int q = 0; q < D - 1; ++q)
#pragma unroll
        for (int k = 0; k < PER; ++k) buf[q][k] = *(const v4f *)(sp + addr(q, k));
    for (int s0 = 0; s0 < NS; s0 += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const int st = s0 + u;
#pragma unroll
            for (int k = 0; k < PER; ++k) buf[(u + D - 1) % D][k] = *(const v4f *)(sp + addr(st + D - 1, k));
#pragma unroll
            for (int k = 0; k < PER; ++k) *(v4f *)(dp + addr(st, k)) = buf[u][k];
        }
    }
}
template <typename F>
void timeit(const char *name, F f)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) f();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 2.0 * W * H * NB * 4;
    printf("%-28s %.3f ms -> %.2f TB/s (read + write)\n", name, ms / 5, bytes / 1e9 / (ms / 5));
}
int main()
{
    const size_t n = (size_t)W * H * NB;
    float *s, *d;
    (void)hipMalloc(&s, n * 4 + 4096); (void)hipMalloc(&d, n * 4 + 4096); (void)hipMemset(s, 0, n * 4);
    timeit("linear", [&] { hipLaunchKernelGGL(lin, dim3(8192), dim3(256), 0, 0, (const v4f *)s, (v4f *)d, n / 4); });
    timeit("hipMemcpyDtoD", [&] { (void)hipMemcpyAsync(d, s, n * 4, hipMemcpyDeviceToDevice, 0); });
#define RM(R, C, D) timeit("row-march " #R "x" #C " D" #D, [&] { hipLaunchKernelGGL((rowmarch<R, C, D>), dim3((H + R - 1) / R, NB), dim3(256), 0, 0, s, d); })
    RM(64, 32, 3); RM(64, 64, 2); RM(32, 64, 3); RM(16, 64, 4); RM(16, 128, 3); RM(16, 256, 2); RM(8, 256, 3); RM(8, 512, 2); RM(4, 1024, 2);
#define CM(C, RB, D) timeit("col-march " #C " x" #RB " D" #D, [&] { hipLaunchKernelGGL((colmarch<C, RB, D>), dim3(W / C, NB), dim3(256), 0, 0, s, d); })
    CM(64, 16, 4); CM(64, 32, 3); CM(128, 16, 3); CM(256, 8, 3); CM(256, 16, 2); CM(512, 4, 3); CM(1024, 4, 2);
    return 0;
}
