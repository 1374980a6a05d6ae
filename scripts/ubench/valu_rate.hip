// scripts/ubench/valu_rate.hip -- how many cycles does a SIMD of gfx950 spend per wave64 VALU instruction of the kinds the AMaZE stream
// kernel is made of?  (A sizing aid for DESIGN.md: "is the kernel VALU-bound?")  1024-thread workgroups, one per CU, 4 waves per SIMD,
// 8 independent chains per thread so that latency is hidden.  build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int KIND>
__global__ void __launch_bounds__(1024) k(float *out, int iters, float s)
{
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i + s;
    const float b = s + 1.0001f, c = s + 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) a[i] = a[i] + b;
            else if (KIND == 1) a[i] = a[i] * b;
            else if (KIND == 2) a[i] = __builtin_fmaf(a[i], b, c);
            else if (KIND == 3) a[i] = a[i] < c ? b : a[i] + c;          // cmp + add + cndmask (3 instr)
            else if (KIND == 4) a[i] = a[i] / b;                          // IEEE division sequence
            else if (KIND == 5) a[i] = fabsf(a[i] - b);
            else if (KIND == 6) a[i] = fmaxf(a[i], b) + c;                // 2 instr
            else if (KIND == 7) a[i] = __builtin_amdgcn_rcpf(a[i]) + b;   // rcp + add
        }
    }
    float r = 0;
    for (int i = 0; i < 8; ++i) r += a[i];
    out[blockIdx.x * 1024 + threadIdx.x] = r;
}
template <int KIND>
void run(const char *name, int per_iter_instr, float *d)
{
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(1024), 0, 0, d, 10, 0.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(1024), 0, 0, d, iters, 0.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr_per_simd = 4.0 * iters * 8 * per_iter_instr;   // 4 waves per SIMD
    printf("%-28s %8.3f ms  -> %.2f ns per wave-instruction per SIMD (%d instr/element assumed)\n", name, ms, ms * 1e6 / wave_instr_per_simd, per_iter_instr);
}
int main()
{
    float *d; hipMalloc(&d, 256 * 1024 * 4);
    run<0>("v_add_f32", 1, d); run<1>("v_mul_f32", 1, d); run<2>("v_fma_f32", 1, d); run<3>("cmp+add+cndmask", 3, d);
    run<4>("IEEE fdiv (sequence)", 1, d); run<5>("sub+abs", 1, d); run<6>("max+add", 2, d); run<7>("rcp+add", 2, d);
    return 0;
}
