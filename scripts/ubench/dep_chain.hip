// scripts/ubench/dep_chain.hip -- how long does ONE dependent fp32 operation take on gfx950 when a SIMD has only few waves?
// (The YvV gaussian's recurrence wave, DESIGN.md section 12: one wave per SIMD running mul -> add -> add -> add per step.)
// One workgroup per CU with W waves per SIMD (W = 1, 2, 4), each thread one chain of dependent v_add_f32 (or the recurrence's
// mul, add, add, add with two independent multiplies beside it, as the gaussian does it).
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off dep_chain.hip -o dep_chain
#include <hip/hip_runtime.h>
#include <cstdio>
// KIND 2 / 3: the recurrence with its input read from LDS a batch of 16 ahead and its results written back to LDS, interleaved with the
// steps (2) or in one burst after the batch (3) -- the loop of gauss_stream_kernel
template <int KIND>
__global__ void klds(float *out, int iters, float s)
{
    __shared__ float buf[64 * 65];
    for (int i = threadIdx.x; i < 64 * 65; i += blockDim.x) buf[i] = i * 1e-4f;
    __syncthreads();
    if (threadIdx.x >= 64) return;
    float *const in = buf + threadIdx.x;
    float a = threadIdx.x * 0.001f + s, m2 = a + 1.f, m3 = a + 2.f;
    const float b1 = s + 0.3f, b2 = s + 0.2f, b3 = s + 0.1f, B = s + 0.4f;
    float x[16], xn[16], v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = in[k * 65];
    for (int it = 0; it < iters; ++it) {
        const int sn = ((it + 1) & 3) * 16;
#pragma unroll
        for (int k = 0; k < 16; ++k) xn[k] = in[(sn + k) * 65];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            v[k] = x[k] * B + a * b1 + m2 * b2 + m3 * b3;
            if (KIND == 2) in[((it & 3) * 16 + k) * 65] = v[k];
            m3 = m2; m2 = a; a = v[k];
        }
        if (KIND == 3) {
#pragma unroll
            for (int k = 0; k < 16; ++k) in[((it & 3) * 16 + k) * 65] = v[k];
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = xn[k];
    }
    out[blockIdx.x * 64 + threadIdx.x] = a + m2 + m3;
}
template <int KIND>
void runlds(const char *name, float *d)
{
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(klds<KIND>, dim3(86), dim3(512), 0, 0, d, 10, 0.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(klds<KIND>, dim3(86), dim3(512), 0, 0, d, iters, 0.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-60s %7.2f ns per step\n", name, ms * 1e6 / ((double)iters * 16));
}
template <int KIND>
__global__ void k(float *out, int iters, float s)
{
    float a = threadIdx.x * 0.001f + s, m2 = a + 1.f, m3 = a + 2.f;
    const float b = s + 1.0001f, b1 = s + 0.3f, b2 = s + 0.2f, b3 = s + 0.1f, B = s + 0.4f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (KIND == 0) a = a + b;
            else {
                const float v = b * B + a * b1 + m2 * b2 + m3 * b3;      // ((xB + m1 b1) + m2 b2) + m3 b3
                m3 = m2; m2 = a; a = v;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + m2 + m3;
}
template <int KIND>
void run(const char *name, int waves_per_simd, int dep_ops, float *d, int grid = 256, int threads_override = 0)
{
    const int iters = 20000, threads = threads_override ? threads_override : 64 * 4 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(threads), 0, 0, d, 10, 0.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(threads), 0, 0, d, iters, 0.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double steps = (double)iters * 16;
    if (threads_override) printf("[%d workgroups of %d threads] ", grid, threads);
    printf("%-34s %d wave(s) per SIMD: %7.2f ns per step = %.2f ns per dependent operation\n", name, waves_per_simd, ms * 1e6 / steps, ms * 1e6 / steps / dep_ops);
}
int main()
{
    float *d; hipMalloc(&d, 256 * 1024 * 4);
    for (int w : {1, 2, 4}) run<0>("dependent v_add_f32", w, 1, d);
    for (int w : {1, 2, 4}) run<1>("recurrence step (mul,add,add,add)", w, 4, d);
    // a nearly idle chip, as the gaussian leaves it: 86 workgroups with ONE busy wave each
    run<1>("recurrence step (mul,add,add,add)", 1, 4, d, 86, 64);
    runlds<2>("recurrence, input from LDS, results to LDS step by step", d);
    runlds<3>("recurrence, input from LDS, results to LDS after the batch", d);
    return 0;
}
