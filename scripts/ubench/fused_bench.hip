// scripts/ubench/fused_bench.hip -- the fused ShrinkAll pass (art_amd/csrc/shrinkblur.hip) on its own: 15 bands of 4096 x 2732
// (the band set of one channel of a 45 MP frame), L or AB form, time per launch and -- built with -DFS_PROFILE -- the cycles a
// workgroup spends in each phase of a block and waiting for the strip above.
// build (from the repo root): hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iart_amd/csrc [-DFS_PROFILE] \
//        scripts/ubench/fused_bench.hip art_amd/csrc/shrinkblur.hip -o /tmp/fused_bench
#include "kernels.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace artgpu;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
int main(int argc, char **argv)
{
    // mode 0: 15 L bands; 1: 15 chroma bands; 2: 15 L + 30 chroma bands in one launch
    const int W = argc > 1 ? atoi(argv[1]) : 4096, H = argc > 2 ? atoi(argv[2]) : 2732, mode = argc > 3 ? atoi(argv[3]) : 0;
    const int NL = mode == 1 ? 0 : 15, NC = mode == 0 ? 0 : (mode == 1 ? 15 : 30), NSUB = NL + NC, ab = mode;
    const size_t n = (size_t)W * H;
    std::vector<float> h(n * 2);
    srand(1);
    for (auto &v : h) v = (float)(rand() % 20001 - 10000) * 0.37f;
    float *coef, *coefC, *coef2, *nv, *mad, *scratch;
    long long *prof;
    CK(hipMalloc(&coef, n * 15 * 4)); CK(hipMalloc(&coef2, n * 15 * 4)); CK(hipMalloc(&coefC, n * 30 * 4)); CK(hipMalloc(&nv, n * 4)); CK(hipMalloc(&mad, 128 * 4));
    CK(hipMalloc(&prof, 256));
    for (int b = 0; b < 15; ++b) CK(hipMemcpy(coef + b * n, h.data() + (b * 7919) % n, n * 4, hipMemcpyHostToDevice));
    for (int b = 0; b < 30; ++b) CK(hipMemcpy(coefC + b * n, h.data() + (b * 104729 + 13) % n, n * 4, hipMemcpyHostToDevice));
    std::vector<float> one(n, 1.5f), m(128, 2.5e5f);
    CK(hipMemcpy(nv, one.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(mad, m.data(), 128 * 4, hipMemcpyHostToDevice));
    FusedShrinkArgs a = {};
    a.coef = coef; a.coef_out = coef2; a.coefC = coefC; a.coefL = coef; a.n = n; a.w = W; a.h = H; a.madL = mad; a.madab = mad + 32; a.mad_ch_stride = 32;
    a.noisevar = nv; a.noisevar_nonneg = 1; a.noisevar_const = 1.f; a.noisevar_scale = 1.f; a.noisevar_ab[0] = a.noisevar_ab[1] = 1.f; a.useNoiseCCurve = 1;
    for (int l = 0; l < 10; ++l) a.rad[l] = l + 2;
    a.level0 = 0; a.nsub = NSUB; a.nL = NL; a.nsub_ch = 15; a.prof = prof;
    CK(hipMalloc(&scratch, shrink_blur_scratch_floats(W, H, NSUB, 6) * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 6; ++it) {
        CK(hipMemset(prof, 0, 256));
        CK(hipEventRecord(e0));
        CK(launch_shrink_blur(a, scratch, nullptr));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        long long p[32]; CK(hipMemcpy(p, prof, 256, hipMemcpyDeviceToHost));
        printf("%s %dx%d x%d: %.3f ms  (%.2f TB/s)", ab == 0 ? "L" : ab == 1 ? "AB" : "L+a+b", W, H, NSUB, ms, n * (NL * 8.0 + NC * 16.0) / ms / 1e9);
        if (p[0] > 0) {
            printf("  busy %% of the step: rows %.0f cols %.0f hand-over %.0f (+ waiting for the strip above %.0f, %lld polls) | elementwise waves:", 100.0 * p[1] / p[0],
                   100.0 * p[2] / p[0], 100.0 * (p[4] - p[6]) / p[0], 100.0 * p[6] / p[0], p[7]);
            for (int w = 3; w < 16; ++w) printf(" %.0f", 100.0 * p[8 + w] / p[0]);
        }
        printf("\n");
    }
    return 0;
}
