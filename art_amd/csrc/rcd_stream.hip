// art_amd/csrc/rcd_stream.hip -- RCD demosaic, device driver of the row-streaming schedule in rcd_stream_core.h.
//
// Replaces RawImageSource::rcd_demosaic (reference: rtengine/rcd_demosaic.cc:51-347).  Persistent workgroups take reference tiles
// (194x194, stride 176) from a counter and walk each one top to bottom, R rows per iteration, with the whole tile state in LDS
// rings: the CFA is read once (1.1 x, the tiles' 18-pixel overlap), R/G/B are written once, nothing else touches memory.
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "rcd_stream_core.h"

namespace artgpu {

namespace {
// workgroup barrier that orders LDS traffic only: a __syncthreads() would also drain the output stores and the prefetch
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <int R>
__global__ void __launch_bounds__(rcs::Cfg<R>::NT) __attribute__((amdgpu_waves_per_eu(6, 6))) rcd_stream_kernel(RcdStreamArgs a)
{
    using namespace rcs;
    typedef Sched<R> S;
    extern __shared__ float lds_generic[];
    rcs_lf lds = (rcs_lf)lds_generic;
    __shared__ int s_next;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const S s(wave, lane);
    int tile = blockIdx.x;
    while (tile < a.ntiles) {
        const int tr = tile / a.numTw, tc = tile - tr * a.numTw;
        const int rowStart = tr * TSN, rowEnd = min(rowStart + TS, a.H);
        const int colStart = tc * TSN, colEnd = min(colStart + TS, a.W);
        if (rowEnd - rowStart > 2 * BORDER && colEnd - colStart > 2 * BORDER) {      // tiles that write a pixel (L112-125, L304-316)
            Tile tl;
            tl.raw = (rcs_gcf)(a.raw + (size_t)rowStart * a.raw_stride + colStart);
            tl.rs = (long)a.raw_stride;
            const size_t oo = (size_t)rowStart * a.out_stride + colStart;
            tl.red = (rcs_gf)(a.red + oo); tl.green = (rcs_gf)(a.green + oo); tl.blue = (rcs_gf)(a.blue + oo);
            tl.os = (long)a.out_stride;
            tl.rows = rowEnd - rowStart; tl.cols = colEnd - colStart;
            tl.filters = a.filters;
            tl.vec2 = a.vec2;
            LoadRegs v;
            s.fetch(tl, 0, v);
            s.commit(lds, tl, 0, v);
            lds_barrier();
            for (int A = R; S::more(tl, A); A += R) {
                s.fetch(tl, A, v);
                s.i1(lds, tl, A);
                lds_barrier();
                s.i2(lds, tl, A);
                lds_barrier();
                s.i3(lds, tl, A);
                lds_barrier();
                s.i4(lds, tl, A);
                s.commit(lds, tl, A, v);
                lds_barrier();
            }
        }
        if (threadIdx.x == 0) s_next = (int)gridDim.x + atomicAdd(a.counter, 1);
        lds_barrier();
        tile = s_next;
        lds_barrier();
    }
}

template <int R>
hipError_t launch_r(const RcdStreamArgs &a, int grid, hipStream_t stream)
{
    constexpr size_t dyn = (size_t)rcs::Cfg<R>::LDS_FLOATS * sizeof(float);
    if (hipError_t e = dyn_lds_once(reinterpret_cast<const void *>(&rcd_stream_kernel<R>), (int)dyn); e != hipSuccess) return e;
    hipLaunchKernelGGL(rcd_stream_kernel<R>, dim3(grid), dim3(rcs::Cfg<R>::NT), dyn, stream, a);
    return hipGetLastError();
}
} // namespace

int rcd_stream_workgroups_per_cu(int) { return 2; }

hipError_t launch_rcd_stream(const RcdStreamArgs &a, int rows_per_iter, int grid, hipStream_t stream)
{
    switch (rows_per_iter) {
    case 4: return launch_r<4>(a, grid, stream);
    case 8: return launch_r<8>(a, grid, stream);
    }
    return hipErrorInvalidValue;
}

} // namespace artgpu
