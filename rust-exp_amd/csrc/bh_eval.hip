// bh_eval.hip -- K3: Barnes-Hut force evaluation on gfx950.  COMPILED WITH -ffp-contract=off.
//
// Replaces Node::compute_force (nbody.rs:333-377), evaluated per body by the reference's worker
// threads (nbody.rs:443-447).  The quadtree itself is built on the device (bh_build.hip: the fast mode's default from 1 024 bodies
// on; up to 65 536 bodies bit-identical to the host tree) or on the host exactly as the reference builds it (host_tree.cpp;
// nbody.rs:388-415: always in the bit-exact mode), and flattened in PRE-ORDER (children UL,UR,LL,LR,
// empty exterior nodes dropped) with a skip pointer per node, so the recursive descent becomes a
// stackless walk:  open a node -> next index;  accept / leaf -> skip[index].
//
//   interior:  s = x2-x1 ; d = sqrt(dx^2+dy^2) ; if s/d < theta -> force(body, COM) else open   (:341-360)
//   exterior:  skip if position bit-equal to the body (:365), else force(body, particle)        (:371)
//
// mode 0 (fast):   test as q < theta^2 * d^2 (q = s*s or -1 for a leaf: no sqrt, no node-type branch), v_rcp_f32 pair law,
//                  contributions accumulated in walk order.  Output = acceleration.
// mode 1 (strict): IEEE sqrt and divide, force() in the reference's expression order, and the
//                  reference's HIERARCHICAL summation (every opened node returns the left-to-right
//                  sum of its four children, :354-360) reproduced with an explicit frame stack
//                  (depth <= 52, the reference panics beyond depth 50).  Output = force, bit-exact.
//
// One lane per body.  k_bh_eval_fast / k_bh_eval_strict: every lane walks on its own (bodies in particle-index
// order, or in a Morton order when `perm` is given).  k_bh_eval_fast_wave: the 64 lanes of a wave share one walk
// (see there) -- used whenever a spatial order of the bodies is available (device-built tree, or host tree with
// n >= 65536); results are bit-identical to the per-lane walk.  Tree nodes are read-only 32-B records.
#include "kernels.h"
#include "bh_gate.h"

namespace nbx {

constexpr int kMaxFrames = 56;

// The fast kernels decide "take or open" without the square root and without a node-type branch:
//     take = q < theta^2 * d^2,   q = s*s (interior)  or  -1 (leaf)            (BhNode::q, set when the tree is flattened)
// interior: s/d < theta <=> s*s < theta^2 d^2 for s, d >= 0 (d = 0 -> open, as in the reference where s/0 = +inf; a
// non-positive theta accepts nothing).  leaf: always taken; the reference's self-skip (position bit-equal,
// nbody.rs:365) needs no test here because a coincident leaf contributes m * 0 / (0 + EPS) = exactly 0.
// Decisions within 1e-5 of the boundary are re-made with the reference's own arithmetic (take_node), so the fast walks
// open exactly the nodes the reference opens; they differ from it by rcp-vs-divide, FMA and summation order only.
// Both walks below add a node's term only in the lanes that take it, in walk order: identical bits.

// take (accept an interior node / evaluate a leaf) or open?  q < theta^2 d^2 decides everything outside a 1e-5-wide band
// around the boundary; inside the band (or with NaNs) the reference's own test runs -- unfused d^2, correctly rounded
// sqrt and divide (nbody.rs:341-345) -- so every decision is the one the reference makes.  The band is far wider than
// the rounding of either form (a few 1e-7), so the cheap comparison can never contradict the exact one outside it.
// the reference's test, kept out of line: it runs for a few decisions per million and must not cost the walk registers
__device__ __attribute__((noinline)) bool take_node_exact(const float s, const float dx, const float dy, const float theta)
{
    const float dist_sq = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));         // :344
    return s / sqrtf(dist_sq) < theta;                                             // :345
}

__device__ __forceinline__ bool take_node(const float q, const float s, const float d2, const float dx, const float dy,
                                          const float th2_lo, const float th2_hi, const float theta)
{
    bool take = q < th2_lo * d2;
    const bool below_hi = q <= th2_hi * d2;
    // lane masks combined on the scalar unit (one s_andn2 instead of a third vector compare for "not take")
    if (__builtin_expect((__ballot(below_hi) & ~__ballot(take)) != 0ull, 0)) {     // somebody in the band: a few decisions per million
        if (below_hi && !take) take = take_node_exact(s, dx, dy, theta);
    }
    return take;                                                                   // (NaN d^2: not taken, like s/NaN < theta)
}

__global__ __launch_bounds__(kTile) void k_bh_eval_fast(const float4* __restrict__ posm, const int lo,
                                                        const int n_targets, const BhNode* __restrict__ nodes,
                                                        int n_nodes, const float theta,
                                                        float2* __restrict__ out, const unsigned* __restrict__ perm, const BuildGate gate)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;   // 64 threads per workgroup for small systems, kTile otherwise
    if (t >= n_targets || !gate_open(gate, n_nodes)) return;
    // perm (optional): a spatial (Morton) order of the bodies, so the 64 lanes of a wave walk nearly the same
    // nodes; it only changes which thread handles which body, never a result
    const int it = perm ? (int)perm[t] - lo : t;
    const float4 pi = posm[lo + it];
    const float th2 = theta > 0.0f ? theta * theta : 0.0f;
    const float th2_lo = th2 * (1.0f - 1.0e-5f), th2_hi = th2 * (1.0f + 1.0e-5f);
    float ax = 0.0f, ay = 0.0f;
    int i = 0;
    while (i < n_nodes) {
        const float4 a = *reinterpret_cast<const float4*>(&nodes[i]);         // px,py,m,s
        const float4 c = *reinterpret_cast<const float4*>(&nodes[i].skip);    // skip, interior, q, -
        const int skip = __float_as_int(c.x);
        const float q = c.z;
        const float dx = a.x - pi.x;
        const float dy = a.y - pi.y;
        const float d2 = __builtin_fmaf(dy, dy, dx * dx);
        const bool take = take_node(q, a.w, d2, dx, dy, th2_lo, th2_hi, theta);
        if (take) {   // under the exec mask: lanes that open the node execute nothing here
            const float s = a.z * __builtin_amdgcn_rcpf(d2 + kEps);
            ax = __builtin_fmaf(s, dx, ax);
            ay = __builtin_fmaf(s, dy, ay);
        }
        i = take ? skip : i + 1;
    }
    out[it] = make_float2(ax, ay);
}

// Wave-uniform variant of the fast walk (used when the bodies arrive in a spatial order, i.e. with the device-built
// tree's Morton permutation): the 64 lanes of a wave walk ONE node sequence -- the union of what their bodies
// need -- so the node record comes through the scalar cache (one s_load_dwordx8, no per-lane address divergence) and
// the walk index and the loop branch live on the scalar unit.  Per-lane semantics are unchanged: a lane that took a
// node parks until the walk leaves that subtree (resume index r = skip), so every body still makes exactly the
// decisions of k_bh_eval_fast and accumulates its contributions in the same (walk) order => bit-identical results.
// (At least one lane is active at every visited node: lanes that open a node stay active at i+1, and when nobody
// opens, the active lanes all resume at skip = the next i.)
// One wave per workgroup: walks differ in length (dense core vs outskirts).
// BPW = bodies per wave.  A small system is latency-bound with most SIMDs idle (10 000 bodies are 157 full waves for
// 1024 SIMDs): giving each wave only 8..32 consecutive (Morton-ordered) bodies multiplies the number of walks in
// flight AND shortens each of them, since the union of what 8 neighbours need is smaller than what 64 need.  The
// unused lanes never take part (resume index = "never"); per-body results do not depend on BPW.
constexpr int kWaveBlock = 64;
template <int BPW>
__global__ __launch_bounds__(kWaveBlock) void k_bh_eval_fast_wave(const float4* __restrict__ posm, const int lo,
                                                                  const int n_targets, const BhNode* __restrict__ nodes,
                                                                  int n_nodes, const float theta,
                                                                  float2* __restrict__ out, const unsigned* __restrict__ perm,
                                                                  const int xcd_order, const BuildGate gate)
{
    if (!gate_open(gate, n_nodes)) return;
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs (block b -> XCD b % 8), each with its own L2.  Handing
    // XCD k the k-th CONTIGUOUS eighth of the Morton-ordered bodies keeps the part of the tree an L2 sees to that region's
    // subtrees instead of all of it (gridDim.x is a multiple of 8; see launch_bh_eval).
    const int blk = xcd_order ? (int)(blockIdx.x & 7u) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int t = blk * BPW + threadIdx.x;
    const bool valid = (int)threadIdx.x < BPW && t < n_targets;
    if (__ballot(valid) == 0ull) return;
    const int it = valid ? (perm ? (int)perm[t] - lo : t) : 0;
    const float4 pi = posm[lo + it];
    const float th2 = theta > 0.0f ? theta * theta : 0.0f;
    const float th2_lo = th2 * (1.0f - 1.0e-5f), th2_hi = th2 * (1.0f + 1.0e-5f);
    float ax = 0.0f, ay = 0.0f;
    int r = valid ? 0 : 0x7FFFFFFF;   // resume index: the lane takes part in node i iff r <= i
    int i = 0;                        // wave-uniform
    // (fetching node i+1 ahead of the decision was tried and is slower; so is giving every wave two independent walks to
    //  overlap their latencies -- 0.755 vs 0.658 ms at 1 M bodies: the loop is bound by instruction issue, the scalar unit of a
    //  CU being shared by its four SIMDs, not by the latency of the node load.  Round 2 also cut the vector work per visit
    //  from 19 to 14 instructions -- exec-masked take block, lane masks combined on the scalar unit -- for no change in time,
    //  and tried the node record through the vector memory path (every lane loads the same 32 bytes, record kept in VGPRs, only
    //  the skip pointer read back to the scalar unit): 0.86 vs 0.66 ms; and requested node[skip] speculatively as soon as node i
    //  had arrived, so that a wave leaving the subtree finds its next record waiting: 0.71 vs 0.62 ms -- every extra scalar load
    //  costs more than the latency it hides.  PMC of the shipped walk, profiles/r02_bh_walk_pmc_summary.json:
    //  13.4 VALU + 12.9 SALU instructions and 4.9 branches per visit, VALU 55 % busy, 66 % of the wave-cycles waiting on the
    //  scalar load: a dependent chain load -> decide -> next index, 800 cycles per visit with 8 waves per SIMD.
    //  Last experiment of round 2: the whole loop hand-written -- 14 scalar instructions per visit instead of the compiler's 22
    //  (SMEM with a 32-bit register offset, SCC straight from the mask arithmetic, exec set and restored around the take block
    //  without execz skips), bit-identical results: 0.632 vs 0.618 ms at 1 M bodies, 0.227 vs 0.252 at 262 144, 0.099 vs 0.111 at
    //  10 000.  Where the tree outgrows the caches the visit is bound by the latency of the dependent scalar load, not by issue; the
    //  small-tree gain did not justify 60 lines of assembly with hand-placed wait states.  Not shipped.)
    while (i < n_nodes) {
        typedef float f8 __attribute__((ext_vector_type(8)));
        const f8 rec = *reinterpret_cast<const f8*>(&nodes[(unsigned)__builtin_amdgcn_readfirstlane(i)]);
        const float nx = rec[0], ny = rec[1], nm = rec[2];
        const int skip = __float_as_int(rec[4]);
        const float q = rec[6];
        const float dx = nx - pi.x;
        const float dy = ny - pi.y;
        const float d2 = __builtin_fmaf(dy, dy, dx * dx);
        const bool take = (r <= i) && take_node(q, rec[3], d2, dx, dy, th2_lo, th2_hi, theta);   // parked lanes take nothing
        if (take) {   // under the exec mask (no select instructions; lanes that open or are parked execute nothing here)
            const float s = nm * __builtin_amdgcn_rcpf(d2 + kEps);
            ax = __builtin_fmaf(s, dx, ax);
            ay = __builtin_fmaf(s, dy, ay);
            r = skip;                                   // done with this subtree
        }
        // lanes still at or before i are the active ones that did not take the node: they want it opened
        i = (__ballot(r <= i) != 0ull) ? i + 1 : skip;
    }
    if (valid) out[it] = make_float2(ax, ay);
}

__global__ __launch_bounds__(kTile) void k_bh_eval_strict(const float4* __restrict__ posm, const int lo,
                                                          const int n_targets, const BhNode* __restrict__ nodes,
                                                          const int n_nodes, const float theta,
                                                          float2* __restrict__ out, const unsigned* __restrict__ perm)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;   // 64 threads per workgroup for small systems, kTile otherwise
    if (t >= n_targets) return;
    // perm (optional): Morton order of the bodies -- neighbouring lanes walk nearly the same nodes (coherent loads,
    // little divergence). It only decides which thread evaluates which body: every body's result is unchanged.
    const int it = perm ? (int)perm[t] - lo : t;
    const float4 pi = posm[lo + it];
    const float th_lo = theta - fabsf(theta) * 1.0e-5f, th_hi = theta + fabsf(theta) * 1.0e-5f;
    // frame stack: partial sums of the enclosing opened nodes and where each subtree ends.  The end of the innermost
    // open subtree lives in a register (cur_end); the stack keeps the outer ones, so a node visit touches the stack
    // (scratch memory) only when a node is opened or a subtree finishes.
    float sfx[kMaxFrames], sfy[kMaxFrames];
    int send[kMaxFrames];
    int sp = 0;
    int cur_end = -1;             // no open frame
    float fx = 0.0f, fy = 0.0f;   // running sum of the innermost open frame (nbody.rs:336-337)
    int i = 0;
    for (;;) {
        while (i == cur_end) {    // subtree finished: return (fx,fy) to the parent's sum
            sp--;
            fx = __fadd_rn(sfx[sp], fx);        // nbody.rs:358  fx += fx_add
            fy = __fadd_rn(sfy[sp], fy);
            cur_end = send[sp];
        }
        if (i >= n_nodes) break;
        const float4 a = *reinterpret_cast<const float4*>(&nodes[i]);
        const int2 b = *reinterpret_cast<const int2*>(&nodes[i].skip);
        if (b.y) {
            const float dx = __fsub_rn(a.x, pi.x);                               // :342
            const float dy = __fsub_rn(a.y, pi.y);                               // :343
            const float dist_sq = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
            // The reference's test is fl(s / fl(sqrt(dist_sq))) < theta (:344-345), two correctly rounded operations:
            // within 1.2e-7 of the real s/sqrt(dist_sq).  s * rsq(dist_sq) is within 1.8e-7 of it, so whenever that
            // estimate is further than 1e-5 (relative) from theta, both land on the same side and the exact sqrt and
            // divide (~30 instructions) can be skipped without changing a single decision.  NaN (d = 0 with s = 0)
            // fails both comparisons and takes the exact path.
            const float est = a.w * __builtin_amdgcn_rsqf(dist_sq);
            bool accept;
            if (est < th_lo) {
                accept = true;
            } else if (est > th_hi) {
                accept = false;
            } else {
                // sqrtf, NOT __fsqrt_rn: the latter lowers to the 1-ulp native v_sqrt_f32; sqrtf is correctly
                // rounded (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt) like Rust's f32::sqrt
                const float d = sqrtf(dist_sq);                                  // :344
                accept = a.w / d < theta;                                        // :345
            }
            if (accept) {
                // force(px,py,m, self.px,self.py,self.m)  :348  (same dx, dy, dist_sq as above: same operations)
                const float f = __fmul_rn(pi.w, a.z) / __fadd_rn(dist_sq, kEps);
                fx = __fadd_rn(fx, __fmul_rn(f, dx));
                fy = __fadd_rn(fy, __fmul_rn(f, dy));
                i = b.x;
            } else if (sp < kMaxFrames) {
                sfx[sp] = fx; sfy[sp] = fy; send[sp] = cur_end; sp++;           // open: children sum from 0
                cur_end = b.x;
                fx = 0.0f; fy = 0.0f;
                i = i + 1;
            } else {
                i = b.x;  // unreachable: the host build rejects depth > 50
            }
        } else {
            if (!(a.x == pi.x && a.y == pi.y)) {                                 // :365
                const float ddx = __fsub_rn(a.x, pi.x), ddy = __fsub_rn(a.y, pi.y);
                const float dist_sq = __fadd_rn(__fmul_rn(ddx, ddx), __fmul_rn(ddy, ddy));
                const float f = __fmul_rn(pi.w, a.z) / __fadd_rn(dist_sq, kEps);
                fx = __fadd_rn(fx, __fmul_rn(f, ddx));                           // :371-373 + :358
                fy = __fadd_rn(fy, __fmul_rn(f, ddy));
            }
            i = b.x;
        }
    }
    out[it] = make_float2(fx, fy);
}

// Wave-uniform form of the bit-exact walk (bodies in Morton order): the 64 lanes share one walk like
// k_bh_eval_fast_wave -- scalar node loads, uniform control flow, lanes parked on subtrees they accepted -- and keep the
// reference's hierarchical sums.  The frames become WAVE frames: when any lane opens a node the wave pushes one frame
// (its end index lives in lane d of a register, read back with v_readlane), and exactly the lanes that opened it save
// their running sums at depth d (bit d of `mine`); when the walk reaches the frame's end those lanes fold the child sum
// back, parent + children, like nbody.rs:358.  The per-lane arrays are indexed by the uniform depth, so the scratch
// accesses are coalesced.  Every lane performs the same decisions and the same additions in the same order as in
// k_bh_eval_strict: same bits.
template <int BPW>
__global__ __launch_bounds__(kWaveBlock) void k_bh_eval_strict_wave(const float4* __restrict__ posm, const int lo,
                                                                    const int n_targets, const BhNode* __restrict__ nodes,
                                                                    const int n_nodes, const float theta,
                                                                    float2* __restrict__ out, const unsigned* __restrict__ perm)
{
    const int t = blockIdx.x * BPW + threadIdx.x;
    const bool valid = (int)threadIdx.x < BPW && t < n_targets;
    if (__ballot(valid) == 0ull) return;
    const int it = valid ? (perm ? (int)perm[t] - lo : t) : 0;
    const float4 pi = posm[lo + it];
    const float th_lo = theta - fabsf(theta) * 1.0e-5f, th_hi = theta + fabsf(theta) * 1.0e-5f;
    float sfx[kMaxFrames], sfy[kMaxFrames];   // this lane's saved sums, by wave frame depth
    unsigned long long mine = 0ull;           // bit d: this lane opened wave frame d
    int ends = 0;                             // lane d holds the end index of wave frame d
    int d = 0;                                // wave frames open (uniform)
    int cur_end = -1;                         // end of the innermost wave frame (uniform)
    float fx = 0.0f, fy = 0.0f;
    int r = valid ? 0 : 0x7FFFFFFF;           // resume index: the lane takes part in node i iff r <= i
    int i = 0;                                // wave-uniform
    for (;;) {
        while (i == cur_end) {                // the wave leaves a subtree: its openers return the child sum to the parent's
            d--;
            if ((mine >> d) & 1ull) {
                fx = __fadd_rn(sfx[d], fx);   // nbody.rs:358  fx += fx_add
                fy = __fadd_rn(sfy[d], fy);
                mine &= ~(1ull << d);
            }
            cur_end = d > 0 ? __builtin_amdgcn_readlane(ends, d - 1) : -1;
        }
        if (i >= n_nodes) break;
        typedef float f8 __attribute__((ext_vector_type(8)));
        const f8 rec = *reinterpret_cast<const f8*>(&nodes[(unsigned)__builtin_amdgcn_readfirstlane(i)]);
        const float nx = rec[0], ny = rec[1], nm = rec[2], ns = rec[3];
        const int skip = __float_as_int(rec[4]);
        const bool interior = __float_as_int(rec[5]) != 0;
        const bool active = r <= i;
        const float dx = __fsub_rn(nx, pi.x);                                    // :342 / :174
        const float dy = __fsub_rn(ny, pi.y);
        const float dist_sq = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
        bool take;                                                               // evaluate force(body, node) now
        if (interior) {
            const float est = ns * __builtin_amdgcn_rsqf(dist_sq);               // see k_bh_eval_strict
            bool accept = est < th_lo;
            if (active && !accept && !(est > th_hi)) {                           // near the boundary (or NaN): exact test
                const float dd = sqrtf(dist_sq);                                 // :344
                accept = ns / dd < theta;                                        // :345
            }
            take = active && accept;
        } else {
            take = active && !(nx == pi.x && ny == pi.y);                        // :365
        }
        if (take) {
            const float f = __fmul_rn(pi.w, nm) / __fadd_rn(dist_sq, kEps);      // :180
            fx = __fadd_rn(fx, __fmul_rn(f, dx));
            fy = __fadd_rn(fy, __fmul_rn(f, dy));
        }
        const bool open = interior && active && !take;
        if (active && !open) r = skip;                                           // leaf, accepted node: done with this subtree
        if (__ballot(open) != 0ull && d < kMaxFrames) {
            if (open) {                                                          // children sum from 0 (:336-337)
                sfx[d] = fx; sfy[d] = fy;
                fx = 0.0f; fy = 0.0f;
                mine |= 1ull << d;
            }
            if ((int)threadIdx.x == d) ends = skip;                              // lane d keeps frame d's end
            cur_end = skip;
            d++;
            i = i + 1;
        } else {
            i = skip;
        }
    }
    if (valid) out[it] = make_float2(fx, fy);
}

// Kick-drift from a per-body force (divide by m, nbody.rs:453-454) or acceleration (is_accel),
// optional velocity kill (nbody.rs:466-471).
__global__ __launch_bounds__(kTile) void k_integrate_f2(float4* __restrict__ posm, const int lo, const int n_targets,
                                                        float4* __restrict__ vel, const float2* __restrict__ force,
                                                        const float dt, const int is_accel, const int killbox, const BuildGate gate)
{
    const int i = blockIdx.x * kTile + threadIdx.x;
    int unused = 0;
    // (every thread reads the flag BEFORE thread 0 may raise it?  no: a thread that runs later would see the flag and leave, which
    //  is the same outcome -- refused or poisoned, nobody moves a body)
    if (!gate_open(gate, unused, i == 0) || i >= n_targets) return;
    const float2 f = force[i];
    float4 v = vel[i];
    float4 p = posm[lo + i];
    if (is_accel) {
        v.x = __fadd_rn(v.x, __fmul_rn(dt, f.x));
        v.y = __fadd_rn(v.y, __fmul_rn(dt, f.y));
    } else {
        v.x = __fadd_rn(v.x, __fmul_rn(dt, f.x) / p.w);
        v.y = __fadd_rn(v.y, __fmul_rn(dt, f.y) / p.w);
    }
    p.x = __fadd_rn(p.x, __fmul_rn(dt, v.x));
    p.y = __fadd_rn(p.y, __fmul_rn(dt, v.y));
    if (killbox) {
        const float lim = __fmul_rn(100.0f, 0.55f);
        if (fabsf(__fsub_rn(0.0f, p.x)) > lim || fabsf(__fsub_rn(0.0f, p.y)) > lim) {
            v.x = 0.0f;
            v.y = 0.0f;
        }
    }
    vel[i] = v;
    posm[lo + i] = p;
}

// Work counter for the roofline note in DESIGN.md: nodes visited and pair evaluations per launch
__global__ __launch_bounds__(kTile) void k_bh_count(const float4* __restrict__ posm, const int lo, const int n_targets,
                                                    const BhNode* __restrict__ nodes, const int n_nodes, const float theta,
                                                    unsigned long long* __restrict__ totals)
{
    const int it = blockIdx.x * kTile + threadIdx.x;
    unsigned visits = 0, pairs = 0, tests = 0;
    if (it < n_targets) {
        const float4 pi = posm[lo + it];
        const float th2 = theta > 0.0f ? theta * theta : 0.0f;
        int i = 0;
        while (i < n_nodes) {
            const float4 a = *reinterpret_cast<const float4*>(&nodes[i]);
            const float4 c = *reinterpret_cast<const float4*>(&nodes[i].skip);
            const float dx = a.x - pi.x, dy = a.y - pi.y;
            const float d2 = __builtin_fmaf(dy, dy, dx * dx);
            const bool take = take_node(c.z, a.w, d2, dx, dy, th2 * (1.0f - 1.0e-5f), th2 * (1.0f + 1.0e-5f), theta);
            visits++;
            pairs += take ? 1u : 0u;
            tests += c.z >= 0.0f ? 1u : 0u;   // interior (q = s*s; a leaf has q = -1): the reference's opening test runs
            i = take ? __float_as_int(c.x) : i + 1;
        }
    }
    unsigned long long v = visits, q = pairs, w = tests;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        v += __shfl_xor(v, off);
        q += __shfl_xor(q, off);
        w += __shfl_xor(w, off);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&totals[0], v);
        atomicAdd(&totals[1], q);
        atomicAdd(&totals[2], w);
    }
}

// (x, y) of every body as two planar arrays, written straight into pinned host memory (zero-copy, coalesced 256-B
// segments): all the host quadtree build needs from the device each step -- 8 of the 16 bytes per body, no host-side
// de-interleave.
__global__ __launch_bounds__(kTile) void k_split_xy(const float4* __restrict__ posm, const int n, float* __restrict__ xs,
                                                    float* __restrict__ ys)
{
    const int i = blockIdx.x * kTile + threadIdx.x;
    if (i >= n) return;
    const float4 p = posm[i];
    xs[i] = p.x;
    ys[i] = p.y;
}

hipError_t launch_split_xy(const float4* posm, int n, float* xs_host_pinned, float* ys_host_pinned, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_split_xy, dim3((n + kTile - 1) / kTile), dim3(kTile), 0, stream, posm, n, xs_host_pinned, ys_host_pinned);
    return hipGetLastError();
}

hipError_t launch_bh_count(const float4* posm, int lo, int n_targets, const BhNode* nodes, int n_nodes, float theta,
                           unsigned long long* totals, hipStream_t stream)
{
    if (n_targets <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_bh_count, dim3((n_targets + kTile - 1) / kTile), dim3(kTile), 0, stream, posm, lo, n_targets, nodes,
                       n_nodes, theta, totals);
    return hipGetLastError();
}

hipError_t launch_bh_eval(const float4* posm, int lo, int n_targets, const BhNode* nodes, int n_nodes, float theta,
                          int mode, float2* force_out, hipStream_t stream, const unsigned* perm, int* gate_counters,
                          int gate_node_cap, int gate_crowd_limit, int gate_queue_limit)
{
    if (n_targets <= 0) return hipSuccess;
    const BuildGate gate{gate_counters, gate_node_cap, gate_crowd_limit, gate_queue_limit, nullptr};
    if (gate_counters && !(mode == 0 || (mode == 2 && perm))) return hipErrorInvalidValue;   // only the fast walks are gated
    // per-lane walks: one wave per workgroup while the system is too small to fill the chip (spreads the waves over the CUs)
    const int block = n_targets <= 65536 ? 64 : kTile;
    const dim3 grid((n_targets + block - 1) / block);
    if (mode == 3 && perm) {
        int bpw = 64;
        while (bpw > 8 && (n_targets + bpw - 1) / bpw < 4096) bpw >>= 1;
        const dim3 g((n_targets + bpw - 1) / bpw);
        auto go = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, g, dim3(kWaveBlock), 0, stream, posm, lo, n_targets, nodes, n_nodes, theta, force_out, perm);
        };
        if (bpw == 64) go(k_bh_eval_strict_wave<64>);
        else if (bpw == 32) go(k_bh_eval_strict_wave<32>);
        else if (bpw == 16) go(k_bh_eval_strict_wave<16>);
        else go(k_bh_eval_strict_wave<8>);
    } else if (mode == 1 || mode == 3)
        hipLaunchKernelGGL(k_bh_eval_strict, grid, dim3(block), 0, stream, posm, lo, n_targets, nodes, n_nodes, theta,
                           force_out, perm);
    else if (mode == 2 && perm) {
        // bodies per wave: aim at >= 4 walks per SIMD (4096 waves), between 4 and 64 bodies each
        int bpw = 64;
        while (bpw > 4 && (n_targets + bpw - 1) / bpw < 4096) bpw >>= 1;   // (4 instead of 8 from 16 384 bodies down: 0.123 -> 0.120 ms per step at 10 000, 0.098 -> 0.094 at 2 000; 2 adds nothing)
        // XCD-aware block order (see the kernel): eval 0.634 -> 0.620 ms at 1 M bodies, 0.295 -> 0.252 at 262 144, 0.117 -> 0.111 at 10 000
        const int xcd_order = 1;
        const int nblk = (n_targets + bpw - 1) / bpw;
        const dim3 g(xcd_order ? (unsigned)((nblk + 7) / 8 * 8) : (unsigned)nblk);
        auto go = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, g, dim3(kWaveBlock), 0, stream, posm, lo, n_targets, nodes, n_nodes, theta, force_out, perm,
                               xcd_order, gate);
        };
        if (bpw == 64) go(k_bh_eval_fast_wave<64>);
        else if (bpw == 32) go(k_bh_eval_fast_wave<32>);
        else if (bpw == 16) go(k_bh_eval_fast_wave<16>);
        else if (bpw == 8) go(k_bh_eval_fast_wave<8>);
        else go(k_bh_eval_fast_wave<4>);
    }
    else
        hipLaunchKernelGGL(k_bh_eval_fast, grid, dim3(block), 0, stream, posm, lo, n_targets, nodes, n_nodes, theta,
                           force_out, perm, gate);
    return hipGetLastError();
}

hipError_t launch_integrate_f2(float4* posm, int lo, int n_targets, float4* vel, const float2* force, float dt,
                               int is_accel, int killbox, hipStream_t stream, int* gate_counters, int gate_node_cap,
                               int gate_crowd_limit, int gate_queue_limit, int* gate_host_out)
{
    if (n_targets <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_integrate_f2, dim3((n_targets + kTile - 1) / kTile), dim3(kTile), 0, stream, posm, lo,
                       n_targets, vel, force, dt, is_accel, killbox,
                       BuildGate{gate_counters, gate_node_cap, gate_crowd_limit, gate_queue_limit, gate_host_out});
    return hipGetLastError();
}

}  // namespace nbx
