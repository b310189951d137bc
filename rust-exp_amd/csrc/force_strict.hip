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
// Two kernels, same bits: k_force_strict_pc (workgroups of term-producing waves feeding one summing wave; the default up to
// ~120 000 targets per GPU) and k_force_strict (one thread per body, sources staged through LDS in 256-body tiles; beyond).
#include <type_traits>
#include "kernels.h"

namespace nbx {

// max over all sources of max(|x|,|y|,|z|) as float bits (NaN counts as +inf), into *out (pre-zeroed): the guard word of the
// short correctly-rounded division below
__global__ __launch_bounds__(kTile) void k_max_coord(const float4* __restrict__ posm, const int n, unsigned* out)
{
    float m = 0.0f;
    for (int i = blockIdx.x * kTile + threadIdx.x; i < n; i += gridDim.x * kTile) {
        const float4 p = posm[i];
        float c = fmaxf(fmaxf(fabsf(p.x), fabsf(p.y)), fabsf(p.z));
        if (!(c == c) || !(p.x == p.x) || !(p.y == p.y) || !(p.z == p.z)) c = __builtin_inff();
        m = fmaxf(m, c);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

hipError_t launch_max_coord(const float4* posm, int n_records, unsigned* guard, hipStream_t stream)
{
    const hipError_t e = hipMemsetAsync(guard, 0, sizeof(unsigned), stream);
    if (e != hipSuccess) return e;
    if (n_records <= 0) return hipSuccess;
    const int blocks = (n_records + kTile - 1) / kTile;
    hipLaunchKernelGGL(k_max_coord, dim3(blocks < 256 ? blocks : 256), dim3(kTile), 0, stream, posm, n_records, guard);
    return hipGetLastError();
}


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

// Two consecutive sources per step with PACKED fp32.  The bit-exact kernels are VALU-bound (one thread per body with scalar
// instructions: 99.8 % busy at 15.3 instructions per 64 pairs, profiles/r02_strict_pmc_before_packing_262144.json); v_pk_*_f32 does two
// lanes' worth per instruction: 11.7 per 64 pairs.  Every packed lane is the same correctly rounded IEEE operation as its
// scalar form, in the same order, and the two terms are added to the running sums in ascending j: bit-identical results.
typedef float v2f_s __attribute__((ext_vector_type(2)));

template <bool FASTDIV>
__device__ __forceinline__ v2f_s ieee_div2(const v2f_s a, const v2f_s b)
{
    if (!FASTDIV) return v2f_s{a.x / b.x, a.y / b.y};
    v2f_s y = {__builtin_amdgcn_rcpf(b.x), __builtin_amdgcn_rcpf(b.y)};
    const v2f_s one = {1.0f, 1.0f};
    const v2f_s e0 = __builtin_elementwise_fma(-b, y, one);
    y = __builtin_elementwise_fma(e0, y, y);
    v2f_s q = a * y;
    const v2f_s r0 = __builtin_elementwise_fma(-b, q, a);
    q = __builtin_elementwise_fma(r0, y, q);
    const v2f_s r1 = __builtin_elementwise_fma(-b, q, a);
    return __builtin_elementwise_fma(r1, y, q);
}

template <bool FASTDIV>
__device__ __forceinline__ void strict_pair2(const float4 sj, const float4 sk, const float4 pi, v2f_s& fxy)
{
    const v2f_s p = {pi.x, pi.y};
    const v2f_s dj = v2f_s{sj.x, sj.y} - p;                              // nbody.rs:174-175 for source j
    const v2f_s dk = v2f_s{sk.x, sk.y} - p;                              //                  for source j + 1
    const v2f_s qj = dj * dj, qk = dk * dk;
    const v2f_s dist = {__fadd_rn(qj.x, qj.y), __fadd_rn(qk.x, qk.y)};   // :176
    const v2f_s den = dist + v2f_s{kEps, kEps};                          // :180
    const v2f_s num = v2f_s{pi.w, pi.w} * v2f_s{sj.w, sk.w};             // m1 * m2
    const v2f_s f = ieee_div2<FASTDIV>(num, den);
    fxy = fxy + v2f_s{f.x, f.x} * dj;                                    // :183 then :141-142, source j first
    fxy = fxy + v2f_s{f.y, f.y} * dk;                                    //                    then source j + 1
}

// |coordinate| bound under which the short division is exact for every pair (with the mass bounds of strict_fastdiv_ok):
// d^2 + EPS <= 8 * 1e5^2 + EPS < 1e11
constexpr unsigned kFastDivCoordBits = 0x47C35000u;   // 1.0e5f

// One thread per body.  (The compile unit is built with -amdgpu-sched-strategy=max-ilp, see Makefile: the default scheduler
// runs the unrolled divisions one after the other, each instruction waiting ~9 cycles for the previous one.)
template <bool kCheck, bool FASTDIV>
__device__ __forceinline__ void strict_one(const float4 sj, const int j, const int i, const int n, const float4 pi,
                                           float& fx, float& fy)
{
    float tx, ty;
    ref_force<FASTDIV>(pi.x, pi.y, pi.w, sj.x, sj.y, sj.w, tx, ty);     // nbody.rs:140
    if (kCheck && (j == i || j >= n)) { tx = 0.0f; ty = 0.0f; }         // :136; +0 leaves a sum that started at +0 unchanged
    fx = __fadd_rn(fx, tx);                                             // :141
    fy = __fadd_rn(fy, ty);                                             // :142
}

template <bool FASTDIV>
__device__ __forceinline__ void strict_sweep(const float4* __restrict__ posm, const int n, const int lo, const int n_targets,
                                             float2* __restrict__ force_out, float4 (*tile)[kTile])
{
    const int tid = threadIdx.x;
    const int it = blockIdx.x * kTile + tid;                   // target index within the slab
    const int i = lo + (it < n_targets ? it : n_targets - 1);  // global body index (clamped)
    const float4 pi = posm[i];
    const int wg_first = lo + blockIdx.x * kTile;              // this workgroup's targets: [wg_first, wg_first + kTile)
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
        if (jbase + kTile <= n && (jbase + kTile <= wg_first || jbase >= wg_first + kTile)) {
            v2f_s fxy = {fx, fy};
#pragma unroll 4
            for (int k = 0; k < kTile; k += 2) strict_pair2<FASTDIV>(tile[buf][k], tile[buf][k + 1], pi, fxy);
            fx = fxy.x; fy = fxy.y;
        } else {
#pragma unroll 2
            for (int k = 0; k < kTile; k++) strict_one<true, FASTDIV>(tile[buf][k], jbase + k, i, n, pi, fx, fy);
        }
        buf ^= 1;
    }
    if (it < n_targets) force_out[it] = make_float2(fx, fy);
}

// guard: device word with max|coordinate| of the sources as float bits (k_max_coord, same stream), or null = always the
// compiler's IEEE division.  The choice is uniform for the whole launch.
__global__ __launch_bounds__(kTile) void k_force_strict(const float4* __restrict__ posm, const int n, const int lo,
                                                        const int n_targets, float2* __restrict__ force_out,
                                                        const unsigned* __restrict__ guard)
{
    __shared__ float4 tile[2][kTile];
    if (guard && guard[0] <= kFastDivCoordBits)
        strict_sweep<true>(posm, n, lo, n_targets, force_out, tile);
    else
        strict_sweep<false>(posm, n, lo, n_targets, force_out, tile);
}

// ---- term producers + one summing wave ---------------------------------------------------------------------------------
// The order of the ADDS is the reference's and cannot be split, but the TERMS are independent: of the ~12 VALU instructions a
// pair costs, only the two adds belong to the serial chain.  A workgroup of W waves serves 64 targets (lane = target in every
// wave): waves 1..W-1 evaluate force(i, j) for the sources of a chunk -- two consecutive sources per step with packed fp32,
// SHARE consecutive sources per producer and chunk, taken from the scalar cache as SGPR operands -- and park the terms in LDS;
// wave 0 folds the previous chunk's terms into the running sums in ascending j while they work on the next one.  Every lane
// executes exactly the operations of nbody.rs:174-183 per term and one IEEE add per term and component, in source order: the
// same bits as one thread per body, with W times the waves per target and the adds off the producers' critical path.
//   <16 waves, 4 sources per producer>: chunk of 60 sources, 60 KB of LDS, one workgroup per CU (4 waves per SIMD)
//   < 8 waves, 8 sources per producer>: chunk of 56 sources, 56 KB of LDS, two workgroups per CU
typedef float v16f_s __attribute__((ext_vector_type(16)));

template <int SHARE> struct PcSources;
template <> struct PcSources<4> { v16f_s a; };                  // 4 source records {x, y, z, m} in 16 SGPRs
template <> struct PcSources<8> { v16f_s a, b; };               // 8 in 32

// The sources of one producer for one chunk: wave-uniform addresses, so they come through the scalar cache into SGPRs (the
// VALU takes them as scalar operands: no LDS traffic, no VGPRs) as 64-byte requests.  Written as asm because the prefetch has to
// be exactly this: ISSUE once the current records have been consumed (a chunk's first 3*SHARE/2 instructions), ARRIVE (s_waitcnt)
// after the chunk's arithmetic.  Left to the compiler the loads become narrow requests (it drops the unused z; the scalar
// cache's request rate then bounds the kernel) and any double buffer is scheduled away into load-wait-use.  j0 and npad are
// multiples of SHARE: a share that would cross the end of the padded array lies entirely beyond n (all its terms are zeroed), so
// it may read any SHARE records.
__device__ __forceinline__ void pc_issue(const float4* __restrict__ q, PcSources<4>& r)
{
    asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=&s"(r.a) : "s"(q));
}
__device__ __forceinline__ void pc_issue(const float4* __restrict__ q, PcSources<8>& r)
{
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40" : "=&s"(r.a), "=&s"(r.b) : "s"(q));
}
// every later use of the records depends on this statement, so none can be scheduled above the wait (which also covers this
// wave's LDS writes: the two share lgkmcnt)
__device__ __forceinline__ void pc_arrive(PcSources<4>& r) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r.a)); }
__device__ __forceinline__ void pc_arrive(PcSources<8>& r) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r.a), "+s"(r.b)); }
__device__ __forceinline__ float pc_field(const PcSources<4>& r, const int k, const int f) { return r.a[4 * k + f]; }
__device__ __forceinline__ float pc_field(const PcSources<8>& r, const int k, const int f)
{
    return k < 4 ? r.a[4 * k + f] : r.b[4 * (k - 4) + f];
}

template <int WAVES, int SHARE, bool FASTDIV>
__device__ __forceinline__ void strict_pc_sweep(const float4* __restrict__ posm, const int n, const int lo, const int n_targets,
                                                float2* __restrict__ force_out, float4* __restrict__ terms_raw)
{
    constexpr int kChunk = (WAVES - 1) * SHARE;    // sources per hand-over
    constexpr bool kLate = WAVES <= 8;             // see the producers' hand-over
    constexpr int kPairs = kChunk / 2;
    float4 (*terms)[kPairs][64] = reinterpret_cast<float4 (*)[kPairs][64]>(terms_raw);   // [buffer][source pair][target]
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int it = blockIdx.x * 64 + lane;
    const int i = lo + (it < n_targets ? it : n_targets - 1);
    const int npad = ((n + kTile - 1) / kTile) * kTile;      // posm is padded with zero-mass records to a multiple of kTile
    const int chunks = (n + kChunk - 1) / kChunk;
    if (w != 0) {
        const float4 pi = posm[i];
        const v2f_s p = {pi.x, pi.y}, pm = {pi.w, pi.w};
        const int off = (w - 1) * SHARE;
        // chunks whose sources include one of this workgroup's targets, and the last one if it is ragged, take the index tests
        const int wg_first = lo + blockIdx.x * 64;
        const int c_self_lo = wg_first / kChunk, c_self_hi = min((wg_first + 63) / kChunk, chunks - 1);
        const int c_ragged = (n % kChunk) ? chunks - 1 : chunks;
        const float4* const q_last = posm + (npad - SHARE);
        const float4* q = posm + off;              // this producer's records of the chunk after the current one
        if (q > q_last) q = q_last;
        PcSources<SHARE> src;
        pc_issue(q, src);
        pc_arrive(src);
        int j0 = off;
        unsigned buf = 0;                          // byte offset of the current buffer
        char* const out0 = reinterpret_cast<char*>(terms_raw + (w - 1) * (SHARE / 2) * 64 + lane);
        // one loop body, compiled with and without the index tests (a test inside would split the share's independent
        // divisions into separate basic blocks and serialise them)
        auto sweep = [&](const int c_end, auto tag) __attribute__((always_inline)) {
            constexpr bool kCheck = decltype(tag)::value;
            for (; j0 < c_end * kChunk; j0 += kChunk) {
                // nbody.rs:174-175 and the numerator of :180 for the whole share: the records are dead after these
                v2f_s d[SHARE], num[SHARE / 2];
#pragma unroll
                for (int k = 0; k < SHARE; k++) d[k] = v2f_s{pc_field(src, k, 0), pc_field(src, k, 1)} - p;
#pragma unroll
                for (int k = 0; k < SHARE; k += 2) num[k / 2] = pm * v2f_s{pc_field(src, k, 3), pc_field(src, k + 1, 3)};
                __builtin_amdgcn_sched_barrier(0);
                q += kChunk;
                if (q > q_last) q = q_last;
                pc_issue(q, src);                  // the next chunk's records: in flight during the rest of this chunk
                __builtin_amdgcn_sched_barrier(0);
                float4 t[SHARE / 2];
#pragma unroll
                for (int k = 0; k < SHARE; k += 2) {
                    const v2f_s qj = d[k] * d[k], qk = d[k + 1] * d[k + 1];
                    const v2f_s dist = {__fadd_rn(qj.x, qj.y), __fadd_rn(qk.x, qk.y)};  // :176
                    const v2f_s den = dist + v2f_s{kEps, kEps};                         // :180
                    const v2f_s f = ieee_div2<FASTDIV>(num[k / 2], den);
                    v2f_s tj = v2f_s{f.x, f.x} * d[k];                                  // :183
                    v2f_s tk = v2f_s{f.y, f.y} * d[k + 1];
                    if (kCheck) {                                                       // :136; +0 leaves a sum that started at +0 unchanged
                        const int j = j0 + k;
                        if (j == i || j >= n) tj = v2f_s{0.0f, 0.0f};
                        if (j + 1 == i || j + 1 >= n) tk = v2f_s{0.0f, 0.0f};
                    }
                    t[k / 2] = make_float4(tj.x, tj.y, tk.x, tk.y);
                }
                // Hand-over.  Two workgroups per CU (kLate): wait, meet the other waves, and only THEN write this chunk's terms --
                // they drain while the next chunk is computed (the wait covers the previous chunk's writes, issued a whole chunk
                // ago), and the summing wave reads a chunk two barriers after its arithmetic.  One workgroup per CU: write, wait,
                // meet -- measured 3 % faster there (profiles/r02_strict_kernel_sweep.txt; the late writes land on the summing
                // wave's first reads of the round and nothing else on the CU fills the gap).
                float4* const out = reinterpret_cast<float4*>(out0 + buf);
                if (!kLate) {
#pragma unroll
                    for (int k = 0; k < SHARE / 2; k++) out[k * 64] = t[k];
                }
                pc_arrive(src);
                __syncthreads();
                if (kLate) {
#pragma unroll
                    for (int k = 0; k < SHARE / 2; k++) out[k * 64] = t[k];
                }
                buf ^= (unsigned)(kPairs * 64 * sizeof(float4));
            }
        };
        // j0 runs off, off + kChunk, ...: `j0 < c_end * kChunk` holds exactly for the chunks below c_end (off < kChunk)
        sweep(min(c_self_lo, c_ragged), std::false_type{});
        sweep(min(c_self_hi + 1, c_ragged), std::true_type{});
        sweep(c_ragged, std::false_type{});
        sweep(chunks, std::true_type{});
        if (kLate) {
            pc_arrive(src);                        // the last chunk's writes
            __syncthreads();
        }
        __syncthreads();                           // (the summing wave's loop ends with one)
    } else {
        // The adds are one dependent chain per target (9.5 cycles per v_pk_add_f32, tools/ubench_chain.hip), so the only thing
        // the summing wave can do about its pace is never to wait for LDS: a chunk's term records are read in groups, two
        // groups ahead of the adds, and the last two groups of a chunk are folded in at the START of the next round, under the
        // latency of that round's first reads.
        constexpr int kG = (kPairs % 4 == 0) ? 4 : 5;          // records per group: 28 = 7 x 4, 30 = 6 x 5
        constexpr int kGroups = kPairs / kG;
        static_assert(kGroups * kG == kPairs && kGroups >= 3, "chunk does not split into groups");
        v2f_s fxy = {0.0f, 0.0f};                  // nbody.rs:130
        float4 ring[2][kG], held[2][kG];
#define NBX_PC_READ(G, DST) _Pragma("unroll") for (int u = 0; u < kG; u++) DST[u] = in[(G) * kG + u][lane];
#define NBX_PC_FOLD(SRC)    _Pragma("unroll") for (int u = 0; u < kG; u++) {                                             \
            fxy = fxy + v2f_s{SRC[u].x, SRC[u].y};     /* nbody.rs:141-142, source j     */                               \
            fxy = fxy + v2f_s{SRC[u].z, SRC[u].w}; }   /*                   source j + 1 */
        __syncthreads();                           // chunk 0 is being computed,
        if (kLate) __syncthreads();                // ... written (see the producers' hand-over)
        for (int c = 0; c < chunks; c++) {
            const float4 (*in)[64] = terms[c & 1];
            NBX_PC_READ(0, ring[0])
            NBX_PC_READ(1, ring[1])
            __builtin_amdgcn_sched_barrier(0);
            if (c > 0) {
                NBX_PC_FOLD(held[0])
                NBX_PC_FOLD(held[1])
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < kGroups - 2; g++) {
                float4 cur4[kG];
#pragma unroll
                for (int u = 0; u < kG; u++) cur4[u] = ring[g & 1][u];
                NBX_PC_READ(g + 2, ring[g & 1])
                __builtin_amdgcn_sched_barrier(0);
                NBX_PC_FOLD(cur4)
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int u = 0; u < kG; u++) {         // the last two groups, in order
                held[0][u] = ring[(kGroups - 2) & 1][u];
                held[1][u] = ring[(kGroups - 1) & 1][u];
            }
            __syncthreads();
        }
        if (chunks > 0) {
            NBX_PC_FOLD(held[0])
            NBX_PC_FOLD(held[1])
        }
#undef NBX_PC_READ
#undef NBX_PC_FOLD
        if (it < n_targets) force_out[it] = make_float2(fxy.x, fxy.y);
    }
}

template <int WAVES, int SHARE>
__global__ __launch_bounds__(WAVES * 64) void k_force_strict_pc(const float4* __restrict__ posm, const int n, const int lo,
                                                                const int n_targets, float2* __restrict__ force_out,
                                                                const unsigned* __restrict__ guard)
{
    __shared__ float4 terms[2 * ((WAVES - 1) * SHARE / 2) * 64];
    if (guard && guard[0] <= kFastDivCoordBits)
        strict_pc_sweep<WAVES, SHARE, true>(posm, n, lo, n_targets, force_out, terms);
    else
        strict_pc_sweep<WAVES, SHARE, false>(posm, n, lo, n_targets, force_out, terms);
}

// Which kernel, by targets per GPU (profiles/r02_strict_kernel_sweep.txt): 16 waves per workgroup while the 64-target
// workgroups do not outnumber the CUs (one workgroup per CU: the more producers the better), 8 waves (two workgroups per CU)
// up to ~120 000 targets, one thread per body beyond -- unless its waves would fill the 1024 SIMDs unevenly (163 840 targets
// = 2.5 waves per SIMD: the half-empty round costs more than the producer/consumer hand-overs).
int strict_kernel_choice(int n_targets)
{
    if (n_targets <= 16384) return 16;
    if (n_targets < 122880) return 8;
    const int waves = (n_targets + 63) / 64;
    const int rounds = (waves + 1023) / 1024;
    return 10 * waves >= 9 * 1024 * rounds ? 1 : 8;
}

// Masses for which m_i * m_j is a normal float in [1e-20, 1e20] for every pair; together with |coordinates| <= 1e5
// (checked on the device) no pair's division needs the scaling / fix-up steps of the IEEE expansion.
bool strict_fastdiv_ok(float mass_min, float mass_max) { return mass_min >= 1.0e-10f && mass_max <= 1.0e10f; }

hipError_t launch_force_strict(const float4* posm, int n, int lo, int n_targets, float2* force_out,
                               hipStream_t stream, ForceLaunch* info, unsigned* guard, int kernel)
{
    if (n_targets <= 0) return hipSuccess;
    if (guard) {   // refresh max|coordinate| of the current sources (padding records are zeros)
        const hipError_t e = launch_max_coord(posm, ((n + kTile - 1) / kTile) * kTile, guard, stream);
        if (e != hipSuccess) return e;
    }
    if (kernel != 1 && kernel != 8 && kernel != 16) kernel = strict_kernel_choice(n_targets);
    // ForceLaunch of the bit-exact kernels: jsplit = 1 (the source loop is never split); variant = -(waves per workgroup of
    // 64 targets), -1 = one thread per body
    if (kernel == 1) {
        const dim3 grid((n_targets + kTile - 1) / kTile);
        hipLaunchKernelGGL(k_force_strict, grid, dim3(kTile), 0, stream, posm, n, lo, n_targets, force_out, guard);
        if (info) *info = ForceLaunch{(int)grid.x, kTile, 1, 1, 2, -1};
    } else {
        const dim3 grid((n_targets + 63) / 64);
        if (kernel == 16)
            hipLaunchKernelGGL((k_force_strict_pc<16, 4>), grid, dim3(1024), 0, stream, posm, n, lo, n_targets, force_out, guard);
        else
            hipLaunchKernelGGL((k_force_strict_pc<8, 8>), grid, dim3(512), 0, stream, posm, n, lo, n_targets, force_out, guard);
        if (info) *info = ForceLaunch{(int)grid.x, kernel * 64, 1, 1, 2, -kernel};
    }
    return hipGetLastError();
}

}  // namespace nbx
