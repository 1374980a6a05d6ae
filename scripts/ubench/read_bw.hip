// scripts/ubench/read_bw.hip -- what does a read-only streaming kernel reach on this chip?  (The ceiling for MadRgb and the other
// reduction-style passes; hipMemcpy device-to-device reaches 5.3 TB/s counting read + write.)
// build: hipcc --offload-arch=gfx950 -O3 read_bw.hip -o read_bw
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T, int U>
__global__ void __launch_bounds__(256) rd(const T *p, size_t n, float *out)
{
    const size_t stride = (size_t)gridDim.x * 256;
    float acc = 0.f;
    for (size_t i0 = blockIdx.x * (size_t)256 + threadIdx.x; i0 < n; i0 += stride * U) {
        T x[U];
#pragma unroll
        for (int k = 0; k < U; ++k) { const size_t i = i0 + k * stride; x[k] = i < n ? p[i] : T{}; }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            if constexpr (sizeof(T) == 4) acc += *(float *)&x[k];
            else { const float4 v = *(float4 *)&x[k]; acc += v.x + v.y + v.z + v.w; }
        }
    }
    if (acc == 123.456f) out[0] = acc;
}
template <typename T, int U>
void run(const char *name, const void *d, size_t bytes, int grid, float *out)
{
    const size_t n = bytes / sizeof(T);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((rd<T, U>), dim3(grid), dim3(256), 0, 0, (const T *)d, n, out);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((rd<T, U>), dim3(grid), dim3(256), 0, 0, (const T *)d, n, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-10s U=%2d grid %5d: %.3f ms per %.0f MB -> %.2f TB/s\n", name, U, grid, ms / 5, bytes / 1e6, bytes / 1e9 / (ms / 5));
}
int main()
{
    const size_t bytes = 672ull << 20;
    void *d; float *out;
    (void)hipMalloc(&d, bytes); (void)hipMalloc(&out, 4); (void)hipMemset(d, 0, bytes);
    for (int grid : {1024, 2880, 8192, 32768}) {
        run<float, 8>("dword", d, bytes, grid, out);
        run<float, 16>("dword", d, bytes, grid, out);
        run<float4, 4>("dwordx4", d, bytes, grid, out);
        run<float4, 8>("dwordx4", d, bytes, grid, out);
    }
    return 0;
}
