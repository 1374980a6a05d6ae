// scripts/ubench/issue_mix.hip -- does a SIMD of gfx950 issue scalar (SALU) and LDS instructions "for free" next to its vector
// instructions, or does every instruction of a wave cost issue time?  1024-thread workgroups, one per CU, 4 waves per SIMD.  Each
// iteration runs NV independent v_add_f32, NS s_add_u32 on wave-private scalars and NL ds_read_b32 (conflict-free).  Inline asm so that
// the compiler neither removes nor merges anything.
// build: hipcc --offload-arch=gfx950 -O3 issue_mix.hip -o issue_mix
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NV, int NS, int NL>
__global__ void __launch_bounds__(1024) k(float *out, int iters)
{
    __shared__ float lds[1024];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, l0 = 0;
    int s0 = 1, s1 = 2, s2 = 3, s3 = 4;
    const unsigned addr = threadIdx.x * 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (NV > 0) asm volatile("v_add_f32 %0, %0, %0" : "+v"(a0));
            if (NS > 0) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s0) : : "scc");
            if (NV > 1) asm volatile("v_add_f32 %0, %0, %0" : "+v"(a1));
            if (NS > 1) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s1) : : "scc");
            if (NL > 0) asm volatile("ds_read_b32 %0, %1" : "=v"(l0) : "v"(addr));
            if (NV > 2) asm volatile("v_add_f32 %0, %0, %0" : "+v"(a2));
            if (NS > 2) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s2) : : "scc");
            if (NV > 3) asm volatile("v_add_f32 %0, %0, %0" : "+v"(a3));
            if (NS > 3) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s3) : : "scc");
        }
        if (NL > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    out[blockIdx.x * 1024 + threadIdx.x] = a0 + a1 + a2 + a3 + l0 + (float)(s0 + s1 + s2 + s3);
}
template <int NV, int NS, int NL>
void run(float *d)
{
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, NS, NL>), dim3(256), dim3(1024), 0, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, NS, NL>), dim3(256), dim3(1024), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_simd_iters = 4.0 * iters * 8;          // 4 waves per SIMD, 8 groups per iteration
    printf("per group: %d VALU + %d SALU + %d LDS : %7.3f ms -> %.2f ns per group per SIMD (= %.2f ns per instruction of any kind)\n", NV, NS, NL, ms,
           ms * 1e6 / per_simd_iters, ms * 1e6 / per_simd_iters / (NV + NS + NL));
}
int main()
{
    float *d; hipMalloc(&d, 256 * 1024 * 4);
    run<4, 0, 0>(d); run<4, 2, 0>(d); run<4, 4, 0>(d); run<2, 4, 0>(d); run<0, 4, 0>(d); run<4, 0, 1>(d); run<4, 4, 1>(d); run<1, 4, 0>(d); run<1, 0, 1>(d);
    return 0;
}
