// force_strict.hip -- bit-exact kernels.  COMPILED WITH -ffp-contract=off (see Makefile).
//
// These reproduce the reference arithmetic operation for operation so that GPU results are
// bit-identical to rs-src/nbody.rs on the same inputs (2-D, z ignored):
//   force()                nbody.rs:164-184   dx=px2-px1; d2=dx*dx+dy*dy; f=(m1*m2)/(d2+EPS); (f*dx,f*dy)
//   brute-force force pass nbody.rs:132-144   ascending j, skip j==i by INDEX, sequential f32 sum
// (the matching integrator, v += (dt*F)/m ; p += dt*v_new, is k_integrate_f2 in bh_eval.hip)
// IEEE-754 binary32 throughout: `/` is the correctly rounded divide (hipcc default
// -fhip-fp32-correctly-rounded-divide-sqrt), no FMA contraction, f32 denormals kept (gfx950
// default float_denorm_mode_32 = 3).  One thread per target body walks ALL sources in ascending
// order (no j-split, no multi-accumulator unrolling), so the summation order is the reference's.
// Sources are staged through LDS in 256-body tiles exactly like the fast kernel.
#include "kernels.h"

namespace nbx {

__device__ __forceinline__ void ref_force(float px1, float py1, float m1, float px2, float py2, float m2,
                                          float& fx, float& fy)
{
    const float dx = __fsub_rn(px2, px1);                                // nbody.rs:174
    const float dy = __fsub_rn(py2, py1);                                // :175
    const float dist_sq = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));  // :176
    const float f = __fmul_rn(m1, m2) / __fadd_rn(dist_sq, kEps);        // :180
    fx = __fmul_rn(f, dx);                                               // :183
    fy = __fmul_rn(f, dy);
}

__global__ __launch_bounds__(kTile) void k_force_strict(const float4* __restrict__ posm, const int n, const int lo,
                                                        const int n_targets, float2* __restrict__ force_out)
{
    __shared__ float4 tile[2][kTile];
    const int tid = threadIdx.x;
    const int it = blockIdx.x * kTile + tid;       // target index within the slab
    const int i = lo + (it < n_targets ? it : n_targets - 1);  // global body index (clamped)
    const float4 pi = posm[i];
    float fx = 0.0f, fy = 0.0f;                    // nbody.rs:130
    const int tiles = (n + kTile - 1) / kTile;
    float4 nxt = posm[tid];
    int buf = 0;
    for (int t = 0; t < tiles; t++) {
        tile[buf][tid] = nxt;
        __syncthreads();
        if (t + 1 < tiles) nxt = posm[(size_t)(t + 1) * kTile + tid];
        const int jbase = t * kTile;
        const int cnt = (n - jbase) < kTile ? (n - jbase) : kTile;  // never touch the zero-mass padding
        for (int k = 0; k < cnt; k++) {
            if (jbase + k == i) continue;          // nbody.rs:136
            const float4 sj = tile[buf][k];
            float fx_add, fy_add;
            ref_force(pi.x, pi.y, pi.w, sj.x, sj.y, sj.w, fx_add, fy_add);  // :140
            fx = __fadd_rn(fx, fx_add);            // :141
            fy = __fadd_rn(fy, fy_add);            // :142
        }
        buf ^= 1;
    }
    if (it < n_targets) force_out[it] = make_float2(fx, fy);
}

hipError_t launch_force_strict(const float4* posm, int n, int lo, int n_targets, float2* force_out,
                               hipStream_t stream)
{
    if (n_targets <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_force_strict, dim3((n_targets + kTile - 1) / kTile), dim3(kTile), 0, stream, posm, n, lo,
                       n_targets, force_out);
    return hipGetLastError();
}

}  // namespace nbx
