// force_tile.hip -- K1 (all-pairs force sweeps) and K2 (reduce + kick-drift) for gfx950 / MI355X.
//
// Replaces the hot loop of the reference: nbody.rs:132-144 (for i, for j != i: force()) with
// force() = nbody.rs:164-184, and the integrator nbody.rs:153-160.
//
// Law (reference, NOT Newton): F_ij = m_i m_j d / (|d|^2 + EPS), d = p_j - p_i  (magnitude ~ 1/r).
// The fast kernels factor m_i out and accumulate the acceleration a_i = sum_j m_j d /(|d|^2+EPS):
//   d    = p_j - p_i                      3 sub              (2 in 2-D)
//   r2   = fma(dz,dz,fma(dy,dy,fma(dx,dx,EPS)))  3 fma       (2)
//   inv  = v_rcp_f32(r2)                  1 transcendental   (1 ulp; r2 >= EPS > 0 always)
//   s    = m_j * inv                      1 mul
//   a   += s * d                          3 fma              (2)
// = 17 algorithmic flops / interaction (12 in 2-D), SURVEY.md section 8(d).  The shipped kernels do this for
// PAIRS of target bodies with v_pk_*_f32: 12 VALU issues per 2 interactions (10 packed + 2 v_rcp_f32).
// The self term (and any coincident body) contributes s*0 = exactly 0, as in the reference where
// f*dx = 0; zero-mass padding sources contribute 0*d = 0.  So no index test in the inner loop.
//
// Mapping (all variants): one workgroup = 256 threads = 4 wave64; each thread owns B target bodies in
// registers (B*256 per workgroup); the source range of the launch is cut in `jsplit` contiguous tile ranges;
// workgroup w handles (target block w / jsplit, source range w % jsplit).  Workgroups are dispatched
// round-robin over the 8 XCDs (block b -> XCD b % 8), so with jsplit a multiple or divisor of 8 every XCD's
// private L2 only ever sees its own 1/jsplit of the source array.
//
// Variants (NBX_OPT_KERNEL_VARIANT):
//   7 / 6  k_force_smem_pkw  packed math, sources through the scalar cache as SGPR operands, the four waves of a workgroup
//                            share 256 targets and split the source range (7: one common mass)   <- default, >= 16 384 sources
//   1      k_force_tile_pk   packed math, sources staged through LDS tiles                        <- default below that
// (Rounds 1-4 also carried 0 = compiler-scheduled LDS tiles, 2 = scalar-cache scalar math, 3 = 4-source LDS batches, 4 = batched
//  reciprocals, 5 = 6 without the wave split: measured losers, removed in round 5; their A/B numbers: docs/rounds/r01.md, r02.md.)
// LDS tiles: sources stream HBM/L2 -> VGPR (one coalesced 16-B float4 load per lane per tile, issued one
// tile ahead) -> LDS (double buffered, ONE barrier per tile) -> broadcast ds_read_b128 (conflict free).
#include "kernels.h"

namespace nbx {

// variant 1: LDS tiles, explicitly PACKED fp32 math.  Each thread owns P pairs of target bodies; a
// pair lives in 64-bit VGPR pairs (xi = {x_a, x_b}, ...) and every per-interaction VALU op is a
// v_pk_*_f32 processing both bodies at once (12 VALU issues per 2 interactions instead of 22):
//   d  = {sx,sx} - xi          v_pk_add_f32 with op_sel broadcast of the source dword, neg on src1
//   r2 = fma(d,d,...)          3 v_pk_fma_f32 (eps folded into the first)
//   inv = rcp(r2.lo), rcp(r2.hi)   2 v_rcp_f32 (no packed transcendental)
//   s  = {m,m} * inv           v_pk_mul_f32 (op_sel broadcast)
//   a += s * d                 3 v_pk_fma_f32
// The source record (x,y,z,m) read by ds_read_b128 lands in two aligned VGPR pairs {x,y},{z,m}, so the
// broadcasts cost nothing.  Sources are read from LDS in batches of UNROLL ahead of their use.
typedef float v2f __attribute__((ext_vector_type(2)));

template <int DIM>
__device__ __forceinline__ void interact_pk(const float4 sj, const v2f xi, const v2f yi, const v2f zi, v2f& ax,
                                            v2f& ay, v2f& az)
{
    // LDS record order is (x, y, m, z): m in the LOW half of the second register pair, so the
    // {m,m} operand of v_pk_mul is a plain op_sel_hi:[0,..] broadcast (no v_mov)
    const v2f sx = {sj.x, sj.x}, sy = {sj.y, sj.y}, sm = {sj.z, sj.z}, sz = {sj.w, sj.w};
    const v2f eps = {kEps, kEps};
    const v2f dx = sx - xi;
    const v2f dy = sy - yi;
    v2f r2 = __builtin_elementwise_fma(dx, dx, eps);
    r2 = __builtin_elementwise_fma(dy, dy, r2);
    v2f dz = {0.f, 0.f};
    if (DIM == 3) {
        // {z,z} - zi with z taken from the HIGH half of the (m,z) register pair.  hipcc materialises
        // this broadcast with a v_mov; the op_sel form below is the same instruction without it.
        const v2f mz = {sj.z, sj.w};
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dz) : "v"(mz), "v"(zi));
        (void)sz;
        r2 = __builtin_elementwise_fma(dz, dz, r2);
    }
    v2f inv;
    inv.x = __builtin_amdgcn_rcpf(r2.x);
    inv.y = __builtin_amdgcn_rcpf(r2.y);
    const v2f s = sm * inv;
    ax = __builtin_elementwise_fma(s, dx, ax);
    ay = __builtin_elementwise_fma(s, dy, ay);
    if (DIM == 3) az = __builtin_elementwise_fma(s, dz, az);
}

template <int P, int DIM, int UNROLL>
__global__ __launch_bounds__(kTile) void k_force_tile_pk(const float4* __restrict__ posm, const int lo,
                                                         const int n_targets, const int tiles_total,
                                                         const int jsplit, float4* __restrict__ acc_partial,
                                                         const int acc_stride)
{
    __shared__ float4 tile[2][kTile];
    constexpr int B = 2 * P;
    const int tid = threadIdx.x;
    const int split = blockIdx.x % jsplit;
    const int iblk = blockIdx.x / jsplit;
    const int t0 = (int)(((unsigned)tiles_total * (unsigned)split) / (unsigned)jsplit);
    const int t1 = (int)(((unsigned)tiles_total * (unsigned)(split + 1)) / (unsigned)jsplit);

    v2f xi[P], yi[P], zi[P], ax[P], ay[P], az[P];
#pragma unroll
    for (int p = 0; p < P; p++) {
        int ia = iblk * (kTile * B) + (2 * p) * kTile + tid;
        int ib = ia + kTile;
        ia = ia < n_targets ? ia : n_targets - 1;
        ib = ib < n_targets ? ib : n_targets - 1;
        const float4 pa = posm[lo + ia];
        const float4 pb = posm[lo + ib];
        xi[p] = v2f{pa.x, pb.x}; yi[p] = v2f{pa.y, pb.y}; zi[p] = v2f{pa.z, pb.z};
        ax[p] = v2f{0.f, 0.f}; ay[p] = v2f{0.f, 0.f}; az[p] = v2f{0.f, 0.f};
    }

    float4 nxt = posm[(size_t)t0 * kTile + tid];
    int buf = 0;
    for (int t = t0; t < t1; t++) {
        tile[buf][tid] = make_float4(nxt.x, nxt.y, nxt.w, nxt.z);  // (x, y, m, z)
        __syncthreads();
        if (t + 1 < t1) nxt = posm[(size_t)(t + 1) * kTile + tid];
#pragma unroll 1
        for (int k0 = 0; k0 < kTile; k0 += UNROLL) {
            float4 sj[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; u++) sj[u] = tile[buf][k0 + u];
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
#pragma unroll
                for (int p = 0; p < P; p++) interact_pk<DIM>(sj[u], xi[p], yi[p], zi[p], ax[p], ay[p], az[p]);
            }
        }
        buf ^= 1;
    }

#pragma unroll
    for (int p = 0; p < P; p++) {
        const int ia = iblk * (kTile * B) + (2 * p) * kTile + tid;
        const int ib = ia + kTile;
        if (ia < n_targets) acc_partial[(size_t)split * acc_stride + ia] = make_float4(ax[p].x, ay[p].x, az[p].x, 0.0f);
        if (ib < n_targets) acc_partial[(size_t)split * acc_stride + ib] = make_float4(ax[p].y, ay[p].y, az[p].y, 0.0f);
    }
}

// variants 6 / 7: packed math with sources through the SCALAR cache (no LDS, no barriers in the loop: posm[j] with a wave-uniform j
// becomes s_load_dwordx4 and the source record feeds v_pk_* as an SGPR-pair operand), the FOUR WAVES of a workgroup sharing one block of 256 targets (64 lanes x 2 packed
// pairs) and each taking a quarter of the workgroup's source range; the four partial sums meet in LDS once, at the end, and
// are added in wave order (fixed => bit-reproducible).  Same instruction stream per wave, same number of workgroups for a
// given total split, but only a quarter of the partial-acceleration slabs ever reach HBM (N = 262 144: 8 slabs = 34 MB
// per launch instead of 32 = 134 MB) and K2 adds 8 terms per body instead of 32.
// UNIT_MASS (variant 7): every body has the SAME mass (known on the host: min == max) -> a_i = m * sum_j d / (|d|^2 + eps):
// the per-interaction v_pk_mul (m_j * inv) leaves the loop -- 9 packed ops + 2 rcp per 2 interactions instead of 10 + 2 --
// and the common mass multiplies the finished sums.  Sources are then weightless, so the zero-mass padding records cannot be
// swept: the source loop ends at the true body count.
// (Round 3, rejected: ONE v_rcp_f32 for a lane's four interactions with a source -- r = 1/(abcd) from two multiplies, the four
//  inverses back with three packed multiplies: 18 packed + 4 packed + 1 mul + 1 rcp = 100 issue cycles instead of 104.  Measured
//  2 % fewer cycles per step, but the extra packed multiplies cost more power than the three v_rcp_f32 they replace: at the same
//  1.3 kW the clock settles at 2 178 MHz instead of 2 312 and the step takes 12.92 ms instead of 12.42
//  (profiles/r03_k1_grouped_rcp_ab.txt).  The sweep is power-limited: fewer joules per interaction, not fewer issue slots.)
// K4 (fp16 sources): `src` is not posm but the widened fp16 copy of it, so a target's own source record is NOT at the target's
// fp32 position and its term is no exact zero: K2 / the force readout recompute it with the sweep's arithmetic and take it out
// (SelfImage) -- the sweep itself is the same kernel, instruction for instruction.
template <int DIM, int UNROLL, bool UNIT_MASS>
__global__ __launch_bounds__(kTile) void k_force_smem_pkw(const float4* __restrict__ posm, const float4* __restrict__ src,
                                                          const int lo,
                                                          const int n_targets, const int tiles_total, const int n_sources,
                                                          const int jsplit, float4* __restrict__ acc_partial,
                                                          const int acc_stride, const float unit_mass,
                                                          const int* __restrict__ exc_idx,
                                                          float4* __restrict__ exc_rec, const int exc_count)
{
    constexpr int P = 2;
    __shared__ float red[4][3][kTile];
    const int tid = threadIdx.x;
    if (UNIT_MASS && blockIdx.x == 0)   // snapshot of the exceptional sources for K2 (MassExceptions); usually 0 or 1 record
        for (int k = tid; k < exc_count; k += kTile) {
            const float4 s = src[exc_idx[k]];
            exc_rec[k] = make_float4(s.x, s.y, s.z, s.w - unit_mass);   // the weight the sweep left out
        }
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int split = blockIdx.x % jsplit;
    const int iblk = blockIdx.x / jsplit;
    const int j0 = (int)(((unsigned)tiles_total * (unsigned)split) / (unsigned)jsplit) * kTile;
    const int j1 = (int)(((unsigned)tiles_total * (unsigned)(split + 1)) / (unsigned)jsplit) * kTile;
    const int quarter = (j1 - j0) >> 2;             // whole tiles per workgroup: a multiple of 64
    const int ja = j0 + quarter * wave;
    int jb = ja + quarter;
    if (UNIT_MASS) jb = jb < n_sources ? jb : n_sources;
    v2f xi[P], yi[P], zi[P], ax[P], ay[P], az[P];
#pragma unroll
    for (int p = 0; p < P; p++) {
        int ia = iblk * kTile + (2 * p) * 64 + lane;
        int ib = ia + 64;
        ia = ia < n_targets ? ia : n_targets - 1;
        ib = ib < n_targets ? ib : n_targets - 1;
        const float4 pa = posm[lo + ia];
        const float4 pb = posm[lo + ib];
        xi[p] = v2f{pa.x, pb.x}; yi[p] = v2f{pa.y, pb.y}; zi[p] = v2f{pa.z, pb.z};
        ax[p] = v2f{0.f, 0.f}; ay[p] = v2f{0.f, 0.f}; az[p] = v2f{0.f, 0.f};
    }
#pragma unroll UNROLL
    for (int j = ja; j < jb; j++) {
        const float4 s = src[j];
        const v2f sx = {s.x, s.x}, sy = {s.y, s.y}, sz = {s.z, s.z}, sm = {s.w, s.w};
        const v2f eps = {kEps, kEps};
#pragma unroll
        for (int p = 0; p < P; p++) {
            const v2f dx = sx - xi[p];
            const v2f dy = sy - yi[p];
            v2f r2 = __builtin_elementwise_fma(dx, dx, eps);
            r2 = __builtin_elementwise_fma(dy, dy, r2);
            v2f dz = {0.f, 0.f};
            if (DIM == 3) {
                dz = sz - zi[p];
                r2 = __builtin_elementwise_fma(dz, dz, r2);
            }
            v2f sc;
            sc.x = __builtin_amdgcn_rcpf(r2.x);
            sc.y = __builtin_amdgcn_rcpf(r2.y);
            if (!UNIT_MASS) sc = sm * sc;
            ax[p] = __builtin_elementwise_fma(sc, dx, ax[p]);
            ay[p] = __builtin_elementwise_fma(sc, dy, ay[p]);
            if (DIM == 3) az[p] = __builtin_elementwise_fma(sc, dz, az[p]);
        }
    }
#pragma unroll
    for (int p = 0; p < P; p++) {   // body b of lane l is target b * 64 + l of the block: conflict-free rows
        red[wave][0][(2 * p) * 64 + lane] = ax[p].x; red[wave][0][(2 * p + 1) * 64 + lane] = ax[p].y;
        red[wave][1][(2 * p) * 64 + lane] = ay[p].x; red[wave][1][(2 * p + 1) * 64 + lane] = ay[p].y;
        red[wave][2][(2 * p) * 64 + lane] = az[p].x; red[wave][2][(2 * p + 1) * 64 + lane] = az[p].y;
    }
    __syncthreads();
    const int it = iblk * kTile + tid;
    if (it < n_targets) {
        float a0 = red[0][0][tid], a1 = red[0][1][tid], a2 = red[0][2][tid];
#pragma unroll
        for (int w = 1; w < 4; w++) { a0 += red[w][0][tid]; a1 += red[w][1][tid]; a2 += red[w][2][tid]; }
        if (UNIT_MASS) { a0 *= unit_mass; a1 *= unit_mass; a2 *= unit_mass; }
        acc_partial[(size_t)split * acc_stride + it] = make_float4(a0, a1, a2, 0.0f);
    }
}

__device__ __forceinline__ void add_exceptions(const MassExceptions exc, const float4 p, const int self, float4& a)
{
    for (int k = 0; k < exc.count; k++) {
        if (exc.idx[k] == self) continue;   // a body owes itself nothing (with fp16 source images the term would not be an exact zero)
        const float4 s = exc.rec[k];
        const float dx = s.x - p.x, dy = s.y - p.y;
        float r2 = __builtin_fmaf(dx, dx, kEps);
        r2 = __builtin_fmaf(dy, dy, r2);
        float dz = 0.0f;
        if (exc.dim == 3) {
            dz = s.z - p.z;
            r2 = __builtin_fmaf(dz, dz, r2);
        }
        const float sc = s.w * __builtin_amdgcn_rcpf(r2);
        a.x = __builtin_fmaf(sc, dx, a.x);
        a.y = __builtin_fmaf(sc, dy, a.y);
        a.z = __builtin_fmaf(sc, dz, a.z);
    }
}

// K4's self-image term (kernels.h SelfImage): target i's interaction with its own widened fp16 source record, with the sweep's
// arithmetic, taken out of the finished sum.
__device__ __forceinline__ void remove_self_image(const SelfImage si, const float4 p, const int self, float4& a)
{
    if (!si.src) return;
    const float4 s = si.src[self];
    const float dx = s.x - p.x, dy = s.y - p.y;
    float r2 = __builtin_fmaf(dx, dx, kEps);
    r2 = __builtin_fmaf(dy, dy, r2);
    float dz = 0.0f;
    if (si.dim == 3) {
        dz = s.z - p.z;
        r2 = __builtin_fmaf(dz, dz, r2);
    }
    const float sc = (si.unit_mass > 0.0f ? si.unit_mass : s.w) * __builtin_amdgcn_rcpf(r2);
    a.x = __builtin_fmaf(-sc, dx, a.x);
    a.y = __builtin_fmaf(-sc, dy, a.y);
    a.z = __builtin_fmaf(-sc, dz, a.z);
}

// K2: a_i = sum over splits in FIXED ascending order (deterministic), then the reference's
// kick-drift (nbody.rs:153-160) with F/m == a:  v += dt*a ; p += dt*v_new.  Products and sums are
// kept unfused (mul then add, as rustc emits them).
__global__ __launch_bounds__(kTile) void k_integrate(float4* __restrict__ posm, const int lo, const int n_targets,
                                                     float4* __restrict__ vel,
                                                     const float4* __restrict__ acc_partial, const int jsplit,
                                                     const int acc_stride, const float dt, const MassExceptions exc,
                                                     const SelfImage si)
{
    const int i = blockIdx.x * kTile + threadIdx.x;
    if (i >= n_targets) return;
    float4 a = acc_partial[i];
    for (int s = 1; s < jsplit; s++) {
        const float4 q = acc_partial[(size_t)s * acc_stride + i];
        a.x += q.x; a.y += q.y; a.z += q.z;
    }
    float4 v = vel[i];
    float4 p = posm[lo + i];
    remove_self_image(si, p, lo + i, a);
    add_exceptions(exc, p, lo + i, a);
    v.x = __fadd_rn(v.x, __fmul_rn(dt, a.x));
    v.y = __fadd_rn(v.y, __fmul_rn(dt, a.y));
    v.z = __fadd_rn(v.z, __fmul_rn(dt, a.z));
    p.x = __fadd_rn(p.x, __fmul_rn(dt, v.x));
    p.y = __fadd_rn(p.y, __fmul_rn(dt, v.y));
    p.z = __fadd_rn(p.z, __fmul_rn(dt, v.z));
    vel[i] = v;
    posm[lo + i] = p;
}

__global__ __launch_bounds__(kTile) void k_reduce_forces(const float4* __restrict__ posm, const int lo,
                                                         const int n_targets,
                                                         const float4* __restrict__ acc_partial, const int jsplit,
                                                         const int acc_stride, float4* __restrict__ out, const MassExceptions exc,
                                                         const SelfImage si)
{
    const int i = blockIdx.x * kTile + threadIdx.x;
    if (i >= n_targets) return;
    float4 a = acc_partial[i];
    for (int s = 1; s < jsplit; s++) {
        const float4 q = acc_partial[(size_t)s * acc_stride + i];
        a.x += q.x; a.y += q.y; a.z += q.z;
    }
    const float4 p = posm[lo + i];
    remove_self_image(si, p, lo + i, a);
    add_exceptions(exc, p, lo + i, a);
    const float m = p.w;
    out[i] = make_float4(m * a.x, m * a.y, m * a.z, 0.0f);
}

template <int B, int DIM>
static hipError_t launch_tiles(dim3 grid, hipStream_t stream, const float4* posm, int lo, int n_targets, int tiles_total, int jsplit,
                               float4* acc_partial, int acc_stride)
{
    hipLaunchKernelGGL((k_force_tile_pk<B / 2, DIM, 8>), grid, dim3(kTile), 0, stream, posm, lo, n_targets, tiles_total, jsplit,
                       acc_partial, acc_stride);
    return hipGetLastError();
}

// variants 6 / 7: one workgroup = 256 targets x 4 source quarters; `jsplit` partial slabs
hipError_t launch_force_wave_split(const float4* posm, int lo, int n_targets, int tiles_total, int n_sources, int jsplit, int dim,
                                   bool unit_mass, float mass, float4* acc_partial, int acc_stride, hipStream_t stream,
                                   ForceLaunch* info, const int* exc_idx, float4* exc_rec, int exc_count, const float4* widened)
{
    if (n_targets <= 0 || tiles_total <= 0) return hipSuccess;
    if (jsplit < 1) jsplit = 1;
    if (jsplit > tiles_total) jsplit = tiles_total;
    const int iblocks = (n_targets + kTile - 1) / kTile;
    const dim3 grid((unsigned)(iblocks * jsplit));
    // variant codes: 6 / 7 = fp32 sources (general / unit-mass sweep); 17 / 18 = the same kernels on the widened fp16 copy (K4)
    if (info) *info = ForceLaunch{(int)grid.x, kTile, jsplit, 4, dim, (widened ? 11 : 0) + (unit_mass ? 7 : 6)};
    const float4* src = widened ? widened : posm;
#define NBX_WS(DD, UM) \
    hipLaunchKernelGGL((k_force_smem_pkw<DD, 8, UM>), grid, dim3(kTile), 0, stream, posm, src, lo, n_targets, tiles_total, n_sources, \
                       jsplit, acc_partial, acc_stride, mass, exc_idx, exc_rec, exc_count)
    if (dim == 3) { if (unit_mass) NBX_WS(3, true); else NBX_WS(3, false); }
    else          { if (unit_mass) NBX_WS(2, true); else NBX_WS(2, false); }
#undef NBX_WS
    return hipGetLastError();
}

hipError_t launch_force_tile(const float4* posm, int lo, int n_targets, int tiles_total, int jsplit, int bpt, int dim,
                             float4* acc_partial, int acc_stride, hipStream_t stream, ForceLaunch* info)
{
    if (n_targets <= 0 || tiles_total <= 0) return hipSuccess;
    if (bpt != 2 && bpt != 4) return hipErrorInvalidValue;      // packed pairs: two or four targets per thread
    if (jsplit < 1) jsplit = 1;
    if (jsplit > tiles_total) jsplit = tiles_total;
    const int iblocks = (n_targets + kTile * bpt - 1) / (kTile * bpt);
    const dim3 grid((unsigned)(iblocks * jsplit));
    if (info) *info = ForceLaunch{(int)grid.x, kTile, jsplit, bpt, dim, 1};
    if (dim == 3) return bpt == 2 ? launch_tiles<2, 3>(grid, stream, posm, lo, n_targets, tiles_total, jsplit, acc_partial, acc_stride)
                                  : launch_tiles<4, 3>(grid, stream, posm, lo, n_targets, tiles_total, jsplit, acc_partial, acc_stride);
    return bpt == 2 ? launch_tiles<2, 2>(grid, stream, posm, lo, n_targets, tiles_total, jsplit, acc_partial, acc_stride)
                    : launch_tiles<4, 2>(grid, stream, posm, lo, n_targets, tiles_total, jsplit, acc_partial, acc_stride);
}

hipError_t launch_integrate(float4* posm, int lo, int n_targets, float4* vel, const float4* acc_partial, int jsplit,
                            int acc_stride, float dt, hipStream_t stream, MassExceptions exc, SelfImage si)
{
    if (n_targets <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_integrate, dim3((n_targets + kTile - 1) / kTile), dim3(kTile), 0, stream, posm, lo, n_targets,
                       vel, acc_partial, jsplit, acc_stride, dt, exc, si);
    return hipGetLastError();
}

hipError_t launch_reduce_forces(const float4* posm, int lo, int n_targets, const float4* acc_partial, int jsplit,
                                int acc_stride, float4* out, hipStream_t stream, MassExceptions exc, SelfImage si)
{
    if (n_targets <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_reduce_forces, dim3((n_targets + kTile - 1) / kTile), dim3(kTile), 0, stream, posm, lo,
                       n_targets, acc_partial, jsplit, acc_stride, out, exc, si);
    return hipGetLastError();
}

}  // namespace nbx
