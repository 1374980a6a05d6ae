// scripts/ubench/pk_rate.hip -- does a packed fp32 instruction (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: two IEEE operations per lane) cost a
// SIMD of gfx950 as much as one plain fp32 instruction, or as much as two?  Inline asm, so that the compiler neither packs nor unpacks anything.
// 1024-thread workgroups, one per CU (4 waves per SIMD), 8 independent chains per thread.
// build: hipcc --offload-arch=gfx950 -O3 pk_rate.hip -o pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ void __launch_bounds__(1024) k(float *out, int iters, float s)
{
    float a[8]; f2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i + s; p[i] = f2{a[i], a[i] + 0.5f}; }
    const float b = s + 1.0001f; const f2 b2 = {b, b + 0.25f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < 16; ++rep)             // 128 instructions per iteration: the loop's own scalar instructions do not count
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            else if (KIND == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(b2));
            else if (KIND == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(b2));
            else if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(b2));
            else if (KIND == 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
        }
    }
    float r = 0;
    for (int i = 0; i < 8; ++i) r += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 1024 + threadIdx.x] = r;
}
template <int KIND>
void run(const char *name, float *d)
{
    const int iters = 1000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(1024), 0, 0, d, 10, 0.f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(1024), 0, 0, d, iters, 0.f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-14s %8.3f ms -> %.2f ns per wave-instruction per SIMD\n", name, ms, ms * 1e6 / (4.0 * iters * 8 * 16));
}
int main()
{
    float *d; (void)hipMalloc(&d, 256 * 1024 * 4);
    run<0>("v_add_f32", d); run<1>("v_pk_add_f32", d); run<2>("v_pk_mul_f32", d); run<3>("v_pk_fma_f32", d); run<4>("v_fma_f32", d);
    return 0;
}
