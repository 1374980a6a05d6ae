// art_amd/csrc/selftest.hip -- artgpu_eval_primitive: the device-side math primitives of the path, evaluated one value per lane on caller
// arrays.  Every kernel of the library is built from these inline functions (devmath.h, devsleef.h, paramcurve.h); this entry point exists
// so that they can be checked ON THE GPU, bit for bit, against fixtures generated from the reference's own headers compiled in place
// (tests/golden/*.npz <- oracle/_ref: rtengine/sleef.h:1198-1313, sleefsseavx.h:1232-1345,1435-1442, LUT.h:349-459,
// helpersse2.h:168-179, median.h, halffloat.h:9-46) -- tests/test_gpu_primitives.py.  It computes nothing the product path needs.
#include <hip/hip_runtime.h>
#include "devmath.h"
#include "devsleef.h"
#include "paramcurve.h"
#include "kernels.h"

namespace artgpu {

namespace {
__global__ void __launch_bounds__(256) prim_eval_kernel(PrimArgs p)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= p.n) return;
    const float *a = (const float *)p.a, *b = (const float *)p.b, *c = (const float *)p.c;
    float *o0 = (float *)p.out0, *o1 = (float *)p.out1;
    switch (p.prim) {
    case PRIM_XEXPF_S: o0[i] = xexpf_s(a[i]); break;
    case PRIM_XEXPF_V: o0[i] = xexpf_v(a[i]); break;
    case PRIM_XEXPF_VN: o0[i] = xexpf_v_nocheck(a[i]); break;
    case PRIM_XEXPF_V_LDEXP: o0[i] = xexpf_v_ldexp(a[i]); break;
    case PRIM_XLOGF_S: o0[i] = xlogf_s(a[i]); break;
    case PRIM_XLOGF_V: o0[i] = xlogf_v(a[i]); break;
    case PRIM_XLOGF_VN: o0[i] = xlogf_v_nocheck(a[i]); break;
    case PRIM_POW_F: o0[i] = pow_F(a[i], b[i]); break;
    case PRIM_XLIN2LOG: o0[i] = xlin2log(a[i], p.param); break;
    case PRIM_XLOG2LIN: o0[i] = xlog2lin(a[i], p.param); break;
    case PRIM_XCBRTF: o0[i] = xcbrtf_s(a[i]); break;
    case PRIM_XATAN2F: o0[i] = xatan2f_s(a[i], b[i]); break;
    case PRIM_XSINCOSF: { float sn, cs; xsincosf_v(a[i], sn, cs); o0[i] = sn; o1[i] = cs; break; }
    case PRIM_LUTF_SCALAR: o0[i] = lutf_lookup<true>(p.table, p.table_size, a[i]); break;
    case PRIM_LUTF_VECTOR: o0[i] = lutf_vlookup(p.table, p.table_size, a[i]); break;
    case PRIM_MEDIAN3: o0[i] = median3(a[i], b[i], c[i]); break;
    case PRIM_VMINF: o0[i] = sse_min(a[i], b[i]); break;
    case PRIM_VMAXF: o0[i] = sse_max(a[i], b[i]); break;
    case PRIM_VINTPF: o0[i] = intp(a[i], b[i], c[i]); break;
    case PRIM_XDIV2F: o0[i] = xdiv2f(a[i]); break;
    case PRIM_XDIVF2: o0[i] = xdivf(a[i], 2); break;
    case PRIM_XLOG_D: ((double *)p.out0)[i] = pc_xlog(((const double *)p.a)[i]); break;
    case PRIM_XEXP_D: ((double *)p.out0)[i] = pc_xexp(((const double *)p.a)[i]); break;
    case PRIM_FLOAT_TO_HALF: ((unsigned *)p.out0)[i] = float_to_half_dng(a[i]); break;     // (32-bit words: the half in the low 16 bits)
    default: break;
    }
}
} // namespace

hipError_t launch_prim_eval(const PrimArgs &p, hipStream_t s)
{
    if (p.n <= 0) return hipSuccess;
    hipLaunchKernelGGL(prim_eval_kernel, dim3((unsigned)((p.n + 255) / 256)), dim3(256), 0, s, p);
    return hipGetLastError();
}

} // namespace artgpu
