// force_strict.hip -- bit-exact kernels.  COMPILED WITH -ffp-contract=off (see Makefile).
//
// These reproduce the reference arithmetic operation for operation so that GPU results are
// bit-identical to rs-src/nbody.rs on the same inputs (2-D, z ignored):
//   force()                nbody.rs:164-184   dx=px2-px1; d2=dx*dx+dy*dy; f=(m1*m2)/(d2+EPS); (f*dx,f*dy)
//   brute-force force pass nbody.rs:132-144   ascending j, skip j==i by INDEX, sequential f32 sum
// (the matching integrator, v += (dt*F)/m ; p += dt*v_new, is k_integrate_f2 in bh_eval.hip)
// IEEE-754 binary32 throughout: `/` is the correctly rounded divide (hipcc default
// -fhip-fp32-correctly-rounded-divide-sqrt), no FMA contraction, f32 denormals kept (gfx950
// default float_denorm_mode_32 = 3).  Every target's running sums advance over ALL sources in ascending
// order (no j-split, no multi-accumulator unrolling), so the summation order is the reference's.
// Sources are staged through LDS in 256-body tiles like the fast LDS kernels.
#include "kernels.h"

namespace nbx {

// Correctly rounded a / b.  FASTDIV = false: the compiler's IEEE expansion (v_div_scale x2, v_rcp, five fma/mul,
// v_div_fmas, v_div_fixup).  FASTDIV = true: the arithmetic core of that same expansion -- reciprocal refined once, quotient
// refined twice -- without the three range-handling instructions.  The two agree bit for bit whenever v_div_scale has
// nothing to scale and v_div_fixup nothing to fix: both operands normal, numerator above 2^-103, exponent difference
// inside (-126, 96).  The kernel takes this path only when the launch has PROVEN that for every pair (mass range known
// on the host, max|coordinate| reduced on the device just before the launch; see strict_fastdiv_ok).
template <bool FASTDIV>
__device__ __forceinline__ float ieee_div(const float a, const float b)
{
    if (!FASTDIV) return a / b;
    float y = __builtin_amdgcn_rcpf(b);
    const float e0 = __builtin_fmaf(-b, y, 1.0f);
    y = __builtin_fmaf(e0, y, y);
    float q = a * y;
    const float r0 = __builtin_fmaf(-b, q, a);
    q = __builtin_fmaf(r0, y, q);
    const float r1 = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(r1, y, q);
}

template <bool FASTDIV>
__device__ __forceinline__ void ref_force(float px1, float py1, float m1, float px2, float py2, float m2,
                                          float& fx, float& fy)
{
    const float dx = __fsub_rn(px2, px1);                                // nbody.rs:174
    const float dy = __fsub_rn(py2, py1);                                // :175
    const float dist_sq = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));  // :176
    const float f = ieee_div<FASTDIV>(__fmul_rn(m1, m2), __fadd_rn(dist_sq, kEps));   // :180
    fx = __fmul_rn(f, dx);                                               // :183
    fy = __fmul_rn(f, dy);
}

// One thread per body leaves the chip idle at the reference's own scales (N = 10 000 gives 157 waves for 1024 SIMDs) and
// the summation order forbids splitting the j loop.  What CAN be shared is the work per TERM: C adjacent lanes (a pair
// or a quad) serve one target, lane c evaluating force(i, j) for the sources j = k + c of every group of C, each term
// with exactly the reference's operations.  The running sums are then advanced in ascending j by ALL lanes of the
// group (they stay identical copies), each add taking its term straight from the owning lane through a DPP quad
// permute (v_add_f32_dpp: no extra instruction, no LDS).  Same terms, same order, same bits -- with C times the waves.
// C = 4 up to 65 536 targets per GPU, 2 up to 131 072, 1 (plain one-thread-per-body) beyond.
// fx += tx(lane 0 of the group); fx += tx(lane 1); ...  -- IEEE v_add_f32 with the term fetched through the DPP operand
// path.  Written as one asm block because the compiler would otherwise pair fx/fy into v_pk_add_f32 (which has no DPP
// form) behind eight v_mov_b32_dpp.  s_nop 1: a VGPR written by a VALU instruction needs two wait states before a DPP
// read; the hazard recognizer does not look inside inline asm.
#define NBX_DPP_ADD(QP)                                                              \
    "v_add_f32_dpp %0, %2, %0 quad_perm:" QP " row_mask:0xf bank_mask:0xf\n\t"       \
    "v_add_f32_dpp %1, %3, %1 quad_perm:" QP " row_mask:0xf bank_mask:0xf\n\t"

template <int C>
__device__ __forceinline__ void add_group_terms(float& fx, float& fy, const float tx, const float ty)
{
    if (C == 1) {
        fx = __fadd_rn(fx, tx);                                         // nbody.rs:141
        fy = __fadd_rn(fy, ty);                                         // :142
    } else if (C == 2) {
        asm("s_nop 1\n\t" NBX_DPP_ADD("[0,0,2,2]") NBX_DPP_ADD("[1,1,3,3]") : "+v"(fx), "+v"(fy) : "v"(tx), "v"(ty));
    } else {
        asm("s_nop 1\n\t" NBX_DPP_ADD("[0,0,0,0]") NBX_DPP_ADD("[1,1,1,1]") NBX_DPP_ADD("[2,2,2,2]") NBX_DPP_ADD("[3,3,3,3]")
            : "+v"(fx), "+v"(fy) : "v"(tx), "v"(ty));
    }
}
#undef NBX_DPP_ADD

template <int C, bool kCheck, bool FASTDIV>
__device__ __forceinline__ void strict_group(const float4 sj, const int j, const int i, const int n, const float4 pi,
                                             float& fx, float& fy)
{
    float tx, ty;
    ref_force<FASTDIV>(pi.x, pi.y, pi.w, sj.x, sj.y, sj.w, tx, ty);     // nbody.rs:140
    if (kCheck && (j == i || j >= n)) { tx = 0.0f; ty = 0.0f; }         // :136; +0 leaves a sum that started at +0 unchanged
    add_group_terms<C>(fx, fy, tx, ty);                                 // ascending j
}

// |coordinate| bound under which the short division is exact for every pair (with the mass bounds of strict_fastdiv_ok):
// d^2 + EPS <= 8 * 1e5^2 + EPS < 1e11
constexpr unsigned kFastDivCoordBits = 0x47C35000u;   // 1.0e5f

template <int C, bool FASTDIV>
__device__ __forceinline__ void strict_sweep(const float4* __restrict__ posm, const int n, const int lo, const int n_targets,
                                             float2* __restrict__ force_out)
{
    constexpr int kTargets = kTile / C;            // targets per workgroup
    __shared__ float4 tile[2][kTile];
    const int tid = threadIdx.x;
    const int c = tid % C;                         // which source of every group of C this lane evaluates
    const int it = blockIdx.x * kTargets + tid / C;            // target index within the slab
    const int i = lo + (it < n_targets ? it : n_targets - 1);  // global body index (clamped)
    const float4 pi = posm[i];
    const int wg_first = lo + blockIdx.x * kTargets;           // this workgroup's targets: [wg_first, wg_first + kTargets)
    float fx = 0.0f, fy = 0.0f;                    // nbody.rs:130
    const int tiles = (n + kTile - 1) / kTile;     // posm is padded with zero-mass records up to a multiple of kTile
    float4 nxt = posm[tid];
    int buf = 0;
    for (int t = 0; t < tiles; t++) {
        tile[buf][tid] = nxt;
        __syncthreads();
        if (t + 1 < tiles) nxt = posm[(size_t)(t + 1) * kTile + tid];
        const int jbase = t * kTile;
        // the index test and the end-of-array test are only compiled into the tiles that need them (uniform per workgroup)
        if (jbase + kTile <= n && (jbase + kTile <= wg_first || jbase >= wg_first + kTargets)) {
#pragma unroll 4
            for (int k = 0; k < kTile; k += C)
                strict_group<C, false, FASTDIV>(tile[buf][k + c], jbase + k + c, i, n, pi, fx, fy);
        } else {
#pragma unroll 2
            for (int k = 0; k < kTile; k += C)
                strict_group<C, true, FASTDIV>(tile[buf][k + c], jbase + k + c, i, n, pi, fx, fy);
        }
        buf ^= 1;
    }
    if (c == 0 && it < n_targets) force_out[it] = make_float2(fx, fy);
}

// guard: device word with max|coordinate| of the sources as float bits (k_max_coord, same stream), or null = always the
// compiler's IEEE division.  The choice is uniform for the whole launch.
template <int C>
__global__ __launch_bounds__(kTile) void k_force_strict(const float4* __restrict__ posm, const int n, const int lo,
                                                        const int n_targets, float2* __restrict__ force_out,
                                                        const unsigned* __restrict__ guard)
{
    if (guard && guard[0] <= kFastDivCoordBits)
        strict_sweep<C, true>(posm, n, lo, n_targets, force_out);
    else
        strict_sweep<C, false>(posm, n, lo, n_targets, force_out);
}

// group size by targets per GPU (measured, profiles/r01_strict_coop_sweep.txt): 4 lanes per target up to 40 960
// targets, 2 up to 98 304, one thread per body beyond
int strict_group_size(int n_targets) { return n_targets <= 40960 ? 4 : (n_targets <= 98304 ? 2 : 1); }

// Masses for which m_i * m_j is a normal float in [1e-20, 1e20] for every pair; together with |coordinates| <= 1e5
// (checked on the device) no pair's division needs the scaling / fix-up steps of the IEEE expansion.
bool strict_fastdiv_ok(float mass_min, float mass_max) { return mass_min >= 1.0e-10f && mass_max <= 1.0e10f; }

hipError_t launch_force_strict(const float4* posm, int n, int lo, int n_targets, float2* force_out,
                               hipStream_t stream, ForceLaunch* info, unsigned* guard)
{
    if (n_targets <= 0) return hipSuccess;
    if (guard) {   // refresh max|coordinate| of the current sources (padding records are zeros)
        const hipError_t e = launch_max_coord(posm, ((n + kTile - 1) / kTile) * kTile, guard, stream);
        if (e != hipSuccess) return e;
    }
    const int c = strict_group_size(n_targets);
    const int per_wg = kTile / c;
    const dim3 grid((n_targets + per_wg - 1) / per_wg);
    if (c == 4)
        hipLaunchKernelGGL(k_force_strict<4>, grid, dim3(kTile), 0, stream, posm, n, lo, n_targets, force_out, guard);
    else if (c == 2)
        hipLaunchKernelGGL(k_force_strict<2>, grid, dim3(kTile), 0, stream, posm, n, lo, n_targets, force_out, guard);
    else
        hipLaunchKernelGGL(k_force_strict<1>, grid, dim3(kTile), 0, stream, posm, n, lo, n_targets, force_out, guard);
    // jsplit = 1: the source loop is never split; variant = -(lanes that share one target)
    if (info) *info = ForceLaunch{(int)grid.x, kTile, 1, 1, 2, -c};
    return hipGetLastError();
}

}  // namespace nbx
