// bh_walk.hip -- K3, round 4: the fast Barnes-Hut walk over CHILD GROUPS.  COMPILED WITH -ffp-contract=off.
//
// Replaces Node::compute_force (nbody.rs:333-377) for the fast mode, like bh_eval.hip's node walk did in rounds 1-3, and makes
// the same decisions -- the reference's, for every body and node.  What bounded that walk, measured
// (profiles/r04_bh_walk_pmc_n1048576.json): one wave-visit = 13.3 VALU + 12.8 SALU + 4.8 branch + 1 SMEM instructions, and a
// SIMD issues at most ONE scalar-side instruction (SALU, branch, SMEM) every 4 cycles: 18.7 x 4 = 75 of the 96 cycles a visit
// took.  The walk was bound by SCALAR issue (0.58 busy) and VALU issue (0.59 busy) together, behind a dependent load per visit.
// This file is organised around cutting both:
//
//  * the unit of the walk is an OPENED node, not a visited one.  When a node is opened all (<= 4) children are needed
//    (nbody.rs:354-360), so the tree is re-laid as one record per interior node holding its children's (x, y, m, T) -- one
//    s_load_dwordx16, one scalar-cache line -- plus four child words: 950 visits per wave at a million bodies, 287 loads.
//  * the opening test is ONE compare: dist_sq > T with the per-node threshold of bh_threshold.h (the reference's
//    s/sqrt(dist_sq) < theta, exactly): no theta^2 d^2 products, no band, no second compare, no mask test.
//  * (x, y) arithmetic is packed: d = (nx,ny) - (px,py) is one v_pk_add_f32 with the record's SGPR pair as operand,
//    (dx^2, dy^2) one v_pk_mul_f32, the accumulation one v_pk_fma_f32.  dist_sq = dx^2 + dy^2 unfused, as the reference.
//  * who is inside a subtree is a 64-bit scalar mask M that travels with the walk: EXEC = M for the whole group, the compare is a
//    v_cmpx (EXEC = the lanes that take the child: the pair law runs under it, no select), M & ~EXEC = the lanes that want the
//    child opened, and the SCC that s_andn2 leaves says whether anybody does.  Opened children go on a wave-level stack
//    (lanes of three VGPRs: group, mask lo, mask hi).
//  * the loop is written in assembly (k_bh_walk_groups): 8 VALU + 3 scalar instructions per child, ~12 scalar per group.  The
//    compiler's version of the same walk (NBX_OPT_BH_WALK = 2, walk_groups_compiled below; also what a wave falls back to if
//    its stack outgrows the 64 lanes) spends 49 SALU + 20 branches per group on the same work: 0.55 ms against the node walk's
//    0.62 at a million bodies -- scalar-issue bound like its predecessor.
//  * the walk ends the step: it applies the kick-drift and the velocity kill (nbody.rs:453-471) to its own bodies as soon as their
//    acceleration is complete (BhKick; no other walk reads a body's position from the particle array -- the records hold copies).
//  * eight waves per SIMD are part of the design: 16 384 walks at a million bodies are exactly two rounds of the chip's 8 192
//    wave slots.  The loop owns s20-s56; what lives across it must keep the kernel at <= 80 SGPRs (tests/test_kernel_resources.py).
//
// Order of accumulation (all three forms: assembly, compiled, per-lane): a group's present children in DESCENDING slot order,
// every child's pair law added in the lanes that take it; then the subtrees of the opened children in ASCENDING slot order,
// depth first.  A lane adds the terms of exactly the groups it is inside, in that order, whichever bodies share its wave:
// bit-identical results.  (Rounds 1-3 accumulated in pre-order of the nodes; the fast mode never promised a summation order --
// the bit-exact mode keeps the reference's hierarchical sums in bh_eval.hip.)
#include <cstdlib>

#include "bh_gate.h"
#include "bh_threshold.h"
#include "kernels.h"

namespace nbx {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

// ---- tree -> groups ------------------------------------------------------------------------------------------------------
// Group 0 = the root alone (the reference tests the root like any other node, nbody.rs:338-345: a tall root box can be accepted
// from its far ends); group r + 1 = the children of the r-th interior node in pre-order.  The device build hands every interior
// node its r (BhNode::pad1: its pre-order slot minus the leaves before it, both known where the node is emitted), so this is a
// map without a scan AND the records lie in the depth-first order the walk visits them in: descending into a node's first opened
// child is a step to the next 80 bytes.  Host-built trees (no r): r = the node's pre-order index -- gaps that are never touched.
// kid[c]: BYTE offset of child c's own group record (interior child), kGroupLeaf, or kGroupAbsent (present children first).
constexpr int kGroupLeaf = -1, kGroupAbsent = -2;

__device__ __forceinline__ void group_slot(const BhNode nd, const int index, const float theta, const bool compact, float4& rec,
                                           int& kid)
{
    if (nd.interior) {
        rec = make_float4(nd.px, nd.py, nd.m, bh_take_threshold(nd.s, theta));
        kid = ((compact ? nd.pad1 : index) + 1) * (int)sizeof(BhGroup);
    } else {
        // a leaf is always evaluated (nbody.rs:371); the body's own leaf adds m * 0 / (0 + EPS) = exactly 0 (nbody.rs:365)
        rec = make_float4(nd.px, nd.py, nd.m, -1.0f);
        kid = kGroupLeaf;
    }
}

__global__ __launch_bounds__(kTile) void k_bh_groups(const BhNode* __restrict__ nodes, int n_nodes, const float theta,
                                                     BhGroup* __restrict__ groups, const int compact, const BuildGate gate)
{
    if (!gate_open(gate, n_nodes)) return;
    const int k = blockIdx.x * kTile + threadIdx.x;
    const float4 none = make_float4(0.0f, 0.0f, 0.0f, __builtin_inff());   // never taken (and never visited: its kid says absent)
    if (k == 0) {
        // an empty tree: one massless leaf, so that the walk has something harmless to evaluate
        float4 r0 = make_float4(0.0f, 0.0f, 0.0f, -1.0f);
        int kid0 = kGroupLeaf;
        if (n_nodes > 0) group_slot(nodes[0], 0, theta, compact != 0, r0, kid0);
        groups[0].c[0] = r0; groups[0].c[1] = none; groups[0].c[2] = none; groups[0].c[3] = none;
        groups[0].kid = make_int4(kid0, kGroupAbsent, kGroupAbsent, kGroupAbsent);
    }
    if (k >= n_nodes) return;
    const int4 hdr = *reinterpret_cast<const int4*>(&nodes[k].skip);   // skip, interior, q, pad1
    if (!hdr.y) return;
    float4 rec[4] = {none, none, none, none};
    int kid[4] = {kGroupAbsent, kGroupAbsent, kGroupAbsent, kGroupAbsent};
    int c = k + 1;
#pragma unroll
    for (int slot = 0; slot < 4; slot++) {
        if (c < hdr.x) {
            const BhNode nd = nodes[c];
            group_slot(nd, c, theta, compact != 0, rec[slot], kid[slot]);
            c = nd.skip;
        }
    }
    if (kid[0] == kGroupAbsent) {
        // an interior node without a single child (no builder makes one): the walks assume a group has a present child in slot 0 --
        // give it a massless leaf, which adds nothing, instead of a slot they would open
        rec[0] = make_float4(0.0f, 0.0f, 0.0f, -1.0f);
        kid[0] = kGroupLeaf;
    }
    BhGroup* g = &groups[(compact ? hdr.w : k) + 1];
    g->c[0] = rec[0]; g->c[1] = rec[1]; g->c[2] = rec[2]; g->c[3] = rec[3];
    g->kid = make_int4(kid[0], kid[1], kid[2], kid[3]);
}

// ---- the shared walk, compiler-generated form ------------------------------------------------------------------------------
// Wave-level stack of pending groups: entry e lives in lane e of three VGPRs (group offset, mask lo, mask hi); entries beyond
// the 64th go to LDS.  Every opened child is pushed (last slot first, so the first slot's subtree is walked first); at most 4
// per level are pending, 52 levels at most (the builds refuse deeper trees like the reference's panic, nbody.rs:230).
constexpr int kSpill = 4 * 52 - 64;

__device__ __forceinline__ int writelane(const int value, const int lane, int reg)
{
    // (VOP3 takes one SGPR: the lane select goes through m0)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(reg) : "s"(value), "s"(lane) : "m0");
    return reg;
}

struct WaveStack {
    int g = 0, lo = 0, hi = 0;   // lanes = entries
    int sp = 0;                  // wave-uniform
    __device__ __forceinline__ void push(const int group, const u64 mask, volatile int* spill)
    {
        if (sp < 64) {
            g = writelane(group, sp, g);
            lo = writelane((int)(unsigned)mask, sp, lo);
            hi = writelane((int)(unsigned)(mask >> 32), sp, hi);
        } else if (sp - 64 < kSpill) {
            if (threadIdx.x == 0) {
                spill[3 * (sp - 64) + 0] = group;
                spill[3 * (sp - 64) + 1] = (int)(unsigned)mask;
                spill[3 * (sp - 64) + 2] = (int)(unsigned)(mask >> 32);
            }
        }
        sp++;
    }
    __device__ __forceinline__ void pop(int& group, u64& mask, volatile int* spill)
    {
        sp--;
        int a, b, c;
        if (sp < 64) {
            a = __builtin_amdgcn_readlane(g, sp);
            b = __builtin_amdgcn_readlane(lo, sp);
            c = __builtin_amdgcn_readlane(hi, sp);
        } else if (sp - 64 < kSpill) {
            a = __builtin_amdgcn_readfirstlane(spill[3 * (sp - 64) + 0]);
            b = __builtin_amdgcn_readfirstlane(spill[3 * (sp - 64) + 1]);
            c = __builtin_amdgcn_readfirstlane(spill[3 * (sp - 64) + 2]);
        } else {   // unreachable (see kSpill): an empty mask -- the group is loaded and nobody is inside
            a = 0; b = 0; c = 0;
        }
        group = a;
        mask = (u64)(unsigned)b | ((u64)(unsigned)c << 32);
    }
};

// One child of the group in (G, K): d = (nx,ny) - p, dist_sq as the reference forms it (nbody.rs:342-344); the lanes of M with
// "not (dist_sq <= T)" take it -- a NaN takes (and poisons the sum, as a NaN does in the reference) instead of opening, so a leaf
// (T = -1) is taken by ALL of M and never opened, without a test of its kind; the pair law a += m d / (dist_sq + EPS)
// (nbody.rs:174-183, rcp for /) runs under the exec mask of the takers (the empty asm keeps the block a branch: if-converted it
// costs two v_cndmask per child); the other lanes of M open the child: its group goes on the stack with their mask.
#define NBX_GROUP_CHILD(c)                                                                 \
    {                                                                                      \
        const v2f nxy = {G[4 * (c) + 0], G[4 * (c) + 1]};                                  \
        const v2f d = nxy - p;                                                             \
        const v2f sq = d * d;                                                              \
        const float d2 = sq.x + sq.y;                                                      \
        const u64 tm = __ballot(!(d2 <= G[4 * (c) + 3])) & M;                              \
        if (__builtin_amdgcn_inverse_ballot_w64(tm)) {                                     \
            asm volatile("");                                                              \
            const float s = G[4 * (c) + 2] * __builtin_amdgcn_rcpf(d2 + kEps);             \
            const v2f ss = {s, s};                                                         \
            acc = __builtin_elementwise_fma(ss, d, acc);                                   \
        }                                                                                  \
        const u64 om = M & ~tm;                                                            \
        if (om != 0ull) st.push(K[c], om, spill);                                          \
    }

__device__ __forceinline__ v2f walk_groups_compiled(const BhGroup* __restrict__ groups, const v2f p, u64 M, volatile int* spill)
{
    v2f acc = {0.0f, 0.0f};
    WaveStack st;
    unsigned g = 0u;   // byte offset of the group record
    for (;;) {
        // (constant address space: a uniform load from it is a scalar load whatever else the kernel does -- the records were written
        //  by an earlier kernel)
        typedef const v16f __attribute__((address_space(4))) * rec16_t;
        typedef const v4i __attribute__((address_space(4))) * rec4_t;
        const uintptr_t rec = reinterpret_cast<uintptr_t>(groups) + (uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((int)g);
        const v16f G = *reinterpret_cast<rec16_t>(rec);
        const v4i K = *reinterpret_cast<rec4_t>(rec + 64);
        if (K[2] != kGroupAbsent) {
            if (K[3] != kGroupAbsent) NBX_GROUP_CHILD(3)
            NBX_GROUP_CHILD(2)
            NBX_GROUP_CHILD(1)
        } else if (K[1] != kGroupAbsent) {
            NBX_GROUP_CHILD(1)
        }
        NBX_GROUP_CHILD(0)
        if (st.sp == 0) break;
        int ng;
        st.pop(ng, M, spill);
        g = (unsigned)ng;
    }
    return acc;
}

// ---- the shared walk, hand-scheduled ---------------------------------------------------------------------------------------
// The same walk with the scalar side written by hand.  Registers (fixed; bound through the asm constraints):
//   s[20:35] group record (x,y,m,T) x 4    s[36:39] child words        s[40:41] M        s[42:43] lanes that open the child
//   s44 stack pointer   s45 byte offset of the group   s[46:47] EXEC at entry   s[48:49] groups   s[50:51] groups + 64   s52 overflow
//   s53, s[54:55] newest stack entry (group, mask; s53 < 0: none)   s56 groups loaded (the walk's cost)
//   v[10:11] p   v[12:13] sum   v[14:15] d   v[16:17] (dx^2, dy^2)   v18 dist_sq   v[20:21] m/(dist_sq+EPS)   v22 v23 v24 stack
// (The register numbers are low on purpose: with 68 numbered SGPRs the kernel runs 8 waves per SIMD, with 82 it ran 7 -- and
//  16 384 walks on 7 168 slots are two rounds plus a third of 256 walks per XCD, visible in tools/bh_walk_trace.py's timeline.)
// Per child: 8 VALU (v_pk_add, v_pk_mul, v_add, v_cmpx, v_add, v_rcp, v_mul, v_pk_fma; the last three skipped when no lane takes
// the child) + s_andn2 + s_mov exec + 2 branches.  The s_nop 0 after each packed op and after v_rcp are the wait states gfx950
// asks for (packed-result forwarding; transcendental result).  A wave whose stack would pass 64 entries leaves with s52 = 1 and redoes its
// walk in the compiled form (LDS spill) -- the same sums in the same order.
#define NBX_ASM_CHILD(x, m, T, K, c)                                                                       \
    "Lc" #c "_%=:\n"                                                                                       \
    " v_pk_add_f32 v[14:15], " x ", v[10:11] neg_lo:[0,1] neg_hi:[0,1]\n"                                  \
    " s_nop 0\n"                                                                                           \
    " v_pk_mul_f32 v[16:17], v[14:15], v[14:15]\n"                                                         \
    " s_nop 0\n"                                                                                           \
    " v_add_f32 v18, v16, v17\n"                                                                           \
    " v_cmpx_nge_f32 vcc, " T ", v18\n"               /* EXEC = lanes of M with not (T >= dist_sq): they take the child */ \
    " v_add_f32 v20, 0x38d1b717, v18\n"               /* dist_sq + EPS (nbody.rs:180) */                   \
    " s_andn2_b64 s[42:43], s[40:41], exec\n"         /* the lanes of M that open it; SCC = anybody */      \
    " s_cbranch_execz Lskip" #c "_%=\n"               /* nobody takes it (the top of every walk) */        \
    " v_rcp_f32 v20, v20\n"                                                                                \
    " s_nop 0\n"                                                                                           \
    " v_mul_f32 v20, " m ", v20\n"                                                                         \
    " v_pk_fma_f32 v[12:13], v[20:21], v[14:15], v[12:13] op_sel_hi:[0,1,1]\n"                             \
    "Lskip" #c "_%=:\n"                                                                                    \
    " s_mov_b64 exec, s[40:41]\n"                                                                          \
    " s_cbranch_scc1 Lpush" #c "_%=\n"                                                                     \
    "Lback" #c "_%=:\n"
// push (group K, mask s[42:43]): the newest entry stays in scalar registers (s53, s[54:55]; s53 < 0 = none) -- it is what the next
// pop wants whenever this group opens anything -- and only an entry it displaces goes to lane s44 of the stack registers
#define NBX_ASM_PUSH(K, c)                                                                                 \
    "Lpush" #c "_%=:\n"                                                                                    \
    " s_cmp_lt_i32 s53, 0\n"                                                                               \
    " s_cbranch_scc1 Lfill" #c "_%=\n"                                                                     \
    " s_cmp_ge_u32 s44, 64\n"                                                                              \
    " s_cbranch_scc1 Lovf_%=\n"                                                                            \
    " s_mov_b32 m0, s44\n"                                                                                 \
    " s_add_u32 s44, s44, 1\n"                                                                             \
    " v_writelane_b32 v22, s53, m0\n"                                                                      \
    " v_writelane_b32 v23, s54, m0\n"                                                                      \
    " v_writelane_b32 v24, s55, m0\n"                                                                      \
    "Lfill" #c "_%=:\n"                                                                                    \
    " s_mov_b32 s53, " K "\n"                                                                              \
    " s_mov_b64 s[54:55], s[42:43]\n"                                                                      \
    " s_branch Lback" #c "_%=\n"

__device__ __forceinline__ v2f walk_groups_asm(const BhGroup* __restrict__ groups, const v2f p, u64 M, int& overflow, int& turns)
{
    v2f acc = {0.0f, 0.0f};
    const char* base = reinterpret_cast<const char*>(groups);
    asm volatile(
        " s_mov_b64 s[46:47], exec\n"
        " s_mov_b32 s44, 0\n"
        " s_mov_b32 s45, 0\n"
        " s_mov_b32 s52, 0\n"
        " s_mov_b32 s53, -1\n"
        " s_mov_b32 s56, 0\n"
        " s_branch Lload_%=\n"
        "Lpop_%=:\n"
        " s_cmp_lt_i32 s53, 0\n"
        " s_cbranch_scc1 Lpopv_%=\n"
        " s_mov_b32 s45, s53\n"                   // the entry in scalar registers
        " s_mov_b64 s[40:41], s[54:55]\n"
        " s_mov_b32 s53, -1\n"
        " s_branch Lload_%=\n"
        "Lpopv_%=:\n"
        " s_cmp_eq_u32 s44, 0\n"
        " s_cbranch_scc1 Ldone_%=\n"
        " s_add_u32 s44, s44, -1\n"
        " v_readlane_b32 s45, v22, s44\n"
        " v_readlane_b32 s40, v23, s44\n"
        " v_readlane_b32 s41, v24, s44\n"
        "Lload_%=:\n"
        " s_add_u32 s56, s56, 1\n"                // groups loaded so far: the walk's cost (next step's launch order)
        " s_load_dwordx16 s[20:35], s[48:49], s45\n"
        " s_load_dwordx4 s[36:39], s[50:51], s45\n"
        " s_mov_b64 exec, s[40:41]\n"
        " s_waitcnt lgkmcnt(0)\n"
        " s_cmp_eq_u32 s38, -2\n"                 // slot 2 absent: one or two children
        " s_cbranch_scc1 Lle2_%=\n"
        " s_cmp_eq_u32 s39, -2\n"
        " s_cbranch_scc1 Lc2_%=\n"
        NBX_ASM_CHILD("s[32:33]", "s34", "s35", "s39", 3)
        NBX_ASM_CHILD("s[28:29]", "s30", "s31", "s38", 2)
        NBX_ASM_CHILD("s[24:25]", "s26", "s27", "s37", 1)
        NBX_ASM_CHILD("s[20:21]", "s22", "s23", "s36", 0)
        " s_branch Lpop_%=\n"
        "Lle2_%=:\n"
        " s_cmp_eq_u32 s37, -2\n"
        " s_cbranch_scc0 Lc1_%=\n"
        " s_branch Lc0_%=\n"
        NBX_ASM_PUSH("s39", 3)
        NBX_ASM_PUSH("s38", 2)
        NBX_ASM_PUSH("s37", 1)
        NBX_ASM_PUSH("s36", 0)
        "Lovf_%=:\n"
        " s_mov_b32 s52, 1\n"
        "Ldone_%=:\n"
        " s_mov_b64 exec, s[46:47]\n"
        : "+{v[12:13]}"(acc), "={s52}"(overflow), "+{s[40:41]}"(M), "={s56}"(turns)
        : "{s[48:49]}"(base), "{s[50:51]}"(base + 64), "{v[10:11]}"(p)
        : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35", "s36", "s37",
          "s38", "s39", "s42", "s43", "s44", "s45", "s46", "s47", "s54", "s55", "s53", "v14", "v15", "v16", "v17", "v18", "v20", "v21", "v22", "v23", "v24",
          "vcc", "scc", "m0");
    return acc;
}

// ---- the same loop with the distance of the NEXT child formed while the current one is tested (round 6) ---------
// v_pk_add -> v_pk_mul -> v_add are each other's operands, and gfx950 wants a wait state behind a packed result (the two s_nop 0 of
// NBX_ASM_CHILD): here the next child's v_pk_add / v_pk_mul sit in those slots -- two register sets, A = v[14:17] for children 3 and 1,
// B = v[26:29] for 2 and 0 -- so a group of k children spends 1 wait state on distances instead of 2 k.  A group enters at its first
// present child through a prologue (one child: the plain block); the order of the sums is the same, the results bit for bit.
// Measured (profiles/r06_bh_walk_pipelined_ab.txt, three alternating runs at 1 M bodies): traversal 0.4378 / 0.4374 / 0.4401 ->
// 0.4335 / 0.4333 / 0.4343 ms, -1.0 %; 10 000 bodies 0.0417 -> 0.0412.  Shipped (NBX_BH_WALK_PIPE=0: the loop of rounds 4-5); the 4 %
// the instruction count promised is not there -- 2.5 issue slots of a turn's ~62 are not what a resident walk waits for.
#define NBX_ASM_PRO(x, D, SQ)                                                                              \
    " v_pk_add_f32 " D ", " x ", v[10:11] neg_lo:[0,1] neg_hi:[0,1]\n"                                     \
    " s_nop 0\n"                                                                                           \
    " v_pk_mul_f32 " SQ ", " D ", " D "\n"
#define NBX_ASM_TAIL(m, D, c)                                                                              \
    " v_add_f32 v20, 0x38d1b717, v18\n"                                                                    \
    " s_andn2_b64 s[42:43], s[40:41], exec\n"                                                              \
    " s_cbranch_execz Lskip" #c "_%=\n"                                                                    \
    " v_rcp_f32 v20, v20\n"                                                                                \
    " s_nop 0\n"                                                                                           \
    " v_mul_f32 v20, " m ", v20\n"                                                                         \
    " v_pk_fma_f32 v[12:13], v[20:21], " D ", v[12:13] op_sel_hi:[0,1,1]\n"                                \
    "Lskip" #c "_%=:\n"                                                                                    \
    " s_mov_b64 exec, s[40:41]\n"                                                                          \
    " s_cbranch_scc1 Lpush" #c "_%=\n"                                                                     \
    "Lback" #c "_%=:\n"
// child c (distance in D / SQlo, SQhi), the next child's distance into Dn / SQn
#define NBX_ASM_BODY(m, T, D, SQlo, SQhi, xn, Dn, SQn, c)                                                  \
    "Lb" #c "_%=:\n"                                                                                       \
    " v_pk_add_f32 " Dn ", " xn ", v[10:11] neg_lo:[0,1] neg_hi:[0,1]\n"                                   \
    " v_add_f32 v18, " SQlo ", " SQhi "\n"                                                                 \
    " v_pk_mul_f32 " SQn ", " Dn ", " Dn "\n"                                                              \
    " v_cmpx_nge_f32 vcc, " T ", v18\n"                                                                    \
    NBX_ASM_TAIL(m, D, c)
#define NBX_ASM_BODY_LAST(m, T, D, SQlo, SQhi, c)                                                          \
    "Lb" #c "_%=:\n"                                                                                       \
    " v_add_f32 v18, " SQlo ", " SQhi "\n"                                                                 \
    " v_cmpx_nge_f32 vcc, " T ", v18\n"                                                                    \
    NBX_ASM_TAIL(m, D, c)

__device__ __forceinline__ v2f walk_groups_asm_pipelined(const BhGroup* __restrict__ groups, const v2f p, u64 M, int& overflow, int& turns)
{
    v2f acc = {0.0f, 0.0f};
    const char* base = reinterpret_cast<const char*>(groups);
    asm volatile(
        " s_mov_b64 s[46:47], exec\n"
        " s_mov_b32 s44, 0\n"
        " s_mov_b32 s45, 0\n"
        " s_mov_b32 s52, 0\n"
        " s_mov_b32 s53, -1\n"
        " s_mov_b32 s56, 0\n"
        " s_branch Lload_%=\n"
        "Lpop_%=:\n"
        " s_cmp_lt_i32 s53, 0\n"
        " s_cbranch_scc1 Lpopv_%=\n"
        " s_mov_b32 s45, s53\n"
        " s_mov_b64 s[40:41], s[54:55]\n"
        " s_mov_b32 s53, -1\n"
        " s_branch Lload_%=\n"
        "Lpopv_%=:\n"
        " s_cmp_eq_u32 s44, 0\n"
        " s_cbranch_scc1 Ldone_%=\n"
        " s_add_u32 s44, s44, -1\n"
        " v_readlane_b32 s45, v22, s44\n"
        " v_readlane_b32 s40, v23, s44\n"
        " v_readlane_b32 s41, v24, s44\n"
        "Lload_%=:\n"
        " s_add_u32 s56, s56, 1\n"
        " s_load_dwordx16 s[20:35], s[48:49], s45\n"
        " s_load_dwordx4 s[36:39], s[50:51], s45\n"
        " s_mov_b64 exec, s[40:41]\n"
        " s_waitcnt lgkmcnt(0)\n"
        " s_cmp_eq_u32 s38, -2\n"                 // slot 2 absent: one or two children
        " s_cbranch_scc1 Lle2_%=\n"
        " s_cmp_eq_u32 s39, -2\n"
        " s_cbranch_scc1 Le2_%=\n"
        NBX_ASM_PRO("s[32:33]", "v[14:15]", "v[16:17]")
        NBX_ASM_BODY("s34", "s35", "v[14:15]", "v16", "v17", "s[28:29]", "v[26:27]", "v[28:29]", 3)
        NBX_ASM_BODY("s30", "s31", "v[26:27]", "v28", "v29", "s[24:25]", "v[14:15]", "v[16:17]", 2)
        NBX_ASM_BODY("s26", "s27", "v[14:15]", "v16", "v17", "s[20:21]", "v[26:27]", "v[28:29]", 1)
        NBX_ASM_BODY_LAST("s22", "s23", "v[26:27]", "v28", "v29", 0)
        " s_branch Lpop_%=\n"
        "Le2_%=:\n"                               // three children: in at child 2
        NBX_ASM_PRO("s[28:29]", "v[26:27]", "v[28:29]")
        " s_branch Lb2_%=\n"
        "Lle2_%=:\n"
        " s_cmp_eq_u32 s37, -2\n"
        " s_cbranch_scc1 Lc0_%=\n"
        NBX_ASM_PRO("s[24:25]", "v[14:15]", "v[16:17]")   // two children: in at child 1
        " s_branch Lb1_%=\n"
        "Lc0_%=:\n"                               // one child: the plain block (nothing to overlap with)
        " v_pk_add_f32 v[26:27], s[20:21], v[10:11] neg_lo:[0,1] neg_hi:[0,1]\n"
        " s_nop 0\n"
        " v_pk_mul_f32 v[28:29], v[26:27], v[26:27]\n"
        " s_nop 0\n"
        " s_branch Lb0_%=\n"
        NBX_ASM_PUSH("s39", 3)
        NBX_ASM_PUSH("s38", 2)
        NBX_ASM_PUSH("s37", 1)
        NBX_ASM_PUSH("s36", 0)
        "Lovf_%=:\n"
        " s_mov_b32 s52, 1\n"
        "Ldone_%=:\n"
        " s_mov_b64 exec, s[46:47]\n"
        : "+{v[12:13]}"(acc), "={s52}"(overflow), "+{s[40:41]}"(M), "={s56}"(turns)
        : "{s[48:49]}"(base), "{s[50:51]}"(base + 64), "{v[10:11]}"(p)
        : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35", "s36", "s37",
          "s38", "s39", "s42", "s43", "s44", "s45", "s46", "s47", "s54", "s55", "s53", "v14", "v15", "v16", "v17", "v18", "v20", "v21", "v22", "v23", "v24",
          "v26", "v27", "v28", "v29", "vcc", "scc", "m0");
    return acc;
}

// The walks differ in length (a dense core's bodies sit deep in the tree), and 16 384 of them on 8 192 wave slots are two rounds and
// a tail whose length is the spread of those lengths.  Measured and removed: launching them longest first (round 4: any departure
// from Morton order costs the L2 more than the tail gains) and running the costliest p % as two halves of 32 bodies (round 5:
// traversal 0.434 -> 0.482 / 0.499 / 0.543 / 0.635 ms with p = 10 / 25 / 50 / 100 at 1 M bodies -- half a walk costs 0.7 of a whole
// one; docs/rounds/r05.md, profiles/r05_bh_walk_split_ab.jsonl).
// TRACE: the timeline instrumentation (tools/bh_walk_trace.py) is a kernel of its own -- its pointer and the start time are four
// scalar registers across the loop, which the plain kernel spends on `sorted` instead (BhKick: the new positions in walk order)
template <int BPW, int ASM, bool TRACE>   // ASM: 0 compiled loop, 1 hand-scheduled, 2 hand-scheduled with the next child's distance overlapped
__global__ __launch_bounds__(64) void k_bh_walk_groups(const float4* posm, const int lo, const int n_targets,
                                                       const BhGroup* __restrict__ groups, void* __restrict__ sink,
                                                       const unsigned* __restrict__ perm, const int xcd_order, const BuildGate gate,
                                                       unsigned long long* __restrict__ trace, float4* kick_posm, const float kick_dt,
                                                       float4* __restrict__ sorted)
{
    // sink: float2 out[] (accelerations) -- or, with the kick folded in (kick_posm != nullptr), float4 vel[]: ONE pointer, because
    // every scalar register that lives across the walk loop counts (80 SGPRs + the trap handler's 16 = 96 is the last allocation
    // that leaves eight waves per SIMD; two pointers, dt and a flag made it 82 -> seven)
    __shared__ int spill_mem[3 * kSpill];
    int n_nodes_unused = 0;
    // (with the kick folded in, this kernel is the last of a gated step: its first thread hands the build's counters to the host
    //  and raises the poison flag of a refused step, as k_integrate_f2 does otherwise)
    if (!gate_open(gate, n_nodes_unused, kick_posm != nullptr && blockIdx.x == 0 && threadIdx.x == 0)) return;
    const unsigned long long t_start = (TRACE && trace) ? __builtin_amdgcn_s_memrealtime() : 0ull;   // the 100 MHz clock all XCDs share
    // XCD-aware order (as the node walk, bh_eval.hip): XCD k walks the k-th contiguous eighth of the Morton-ordered bodies
    const int blk = xcd_order ? (int)(blockIdx.x & 7u) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int t = blk * BPW + threadIdx.x;
    const bool valid = (int)threadIdx.x < BPW && t < n_targets;
    const u64 M = __ballot(valid);
    if (M == 0ull) return;
    const int it = valid ? (perm ? (int)perm[t] - lo : t) : 0;
    const float4 pi = posm[lo + it];
    const v2f p = {pi.x, pi.y};
    v2f acc;
    int overflow = ASM ? 0 : 1, turns = 0;
    if (ASM == 2) acc = walk_groups_asm_pipelined(groups, p, M, overflow, turns);
    else if (ASM) acc = walk_groups_asm(groups, p, M, overflow, turns);
    if (__builtin_amdgcn_readfirstlane(overflow)) acc = walk_groups_compiled(groups, p, M, spill_mem);   // (uniform: the asm's output is an SGPR)
    if (valid) {
        if (kick_posm) {   // kick-drift with the acceleration just found: the operations and order of k_integrate_f2 (is_accel, killbox)
            float4* const vel = static_cast<float4*>(sink);
            float4 v = vel[it];
            float4 q = pi;
            v.x = __fadd_rn(v.x, __fmul_rn(kick_dt, acc.x));
            v.y = __fadd_rn(v.y, __fmul_rn(kick_dt, acc.y));
            q.x = __fadd_rn(q.x, __fmul_rn(kick_dt, v.x));
            q.y = __fadd_rn(q.y, __fmul_rn(kick_dt, v.y));
            const float lim = __fmul_rn(100.0f, 0.55f);
            if (fabsf(__fsub_rn(0.0f, q.x)) > lim || fabsf(__fsub_rn(0.0f, q.y)) > lim) {
                v.x = 0.0f;
                v.y = 0.0f;
            }
            vel[it] = v;
            kick_posm[lo + it] = q;
            if (!TRACE && sorted) sorted[blk * BPW + (int)threadIdx.x] = q;
        } else {
            static_cast<float2*>(sink)[it] = make_float2(acc.x, acc.y);
        }
    }
    if (TRACE && trace && threadIdx.x == 0) {   // tools/bh_walk_trace.py: when and where this walk ran (s_memrealtime: 10 ns ticks; HW_ID, XCC_ID)
        trace[4 * (size_t)blockIdx.x + 0] = t_start;
        trace[4 * (size_t)blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        trace[4 * (size_t)blockIdx.x + 2] = (unsigned long long)((unsigned)__builtin_amdgcn_readfirstlane(turns) & 0x7FFFFFFFu) |
                                            ((ASM && __builtin_amdgcn_readfirstlane(overflow)) ? 0x80000000ull : 0ull) |   // redone with the LDS spill
                                            ((unsigned long long)(unsigned)blk << 32);
        trace[4 * (size_t)blockIdx.x + 3] = (unsigned long long)__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)) |
                                            ((unsigned long long)__builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)) << 32);
    }
}

// ---- the same walk, private to a lane (bodies in particle-index order: host tree below 65 536 bodies, NBX_OPT_BH_WAVE = 0) ---
constexpr int kLaneStack = 4 * 52;

__global__ __launch_bounds__(kTile) void k_bh_walk_groups_lane(const float4* __restrict__ posm, const int lo, const int n_targets,
                                                               const BhGroup* __restrict__ groups, float2* __restrict__ out,
                                                               const unsigned* __restrict__ perm, const BuildGate gate)
{
    int n_nodes_unused = 0;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_targets || !gate_open(gate, n_nodes_unused)) return;
    const int it = perm ? (int)perm[t] - lo : t;
    const float4 pi = posm[lo + it];
    const v2f p = {pi.x, pi.y};
    v2f acc = {0.0f, 0.0f};
    int stack[kLaneStack];
    int sp = 0;
    unsigned g = 0u;
    for (;;) {
        const BhGroup G = *reinterpret_cast<const BhGroup*>(reinterpret_cast<const char*>(groups) + g);
        const int K[4] = {G.kid.x, G.kid.y, G.kid.z, G.kid.w};
#pragma unroll
        for (int c = 3; c >= 0; c--) {
            if (K[c] == kGroupAbsent) continue;
            const v2f nxy = {G.c[c].x, G.c[c].y};
            const v2f d = nxy - p;
            const v2f sq = d * d;
            const float d2 = sq.x + sq.y;
            if (!(d2 <= G.c[c].w)) {
                const float s = G.c[c].z * __builtin_amdgcn_rcpf(d2 + kEps);
                const v2f ss = {s, s};
                acc = __builtin_elementwise_fma(ss, d, acc);
            } else if (sp < kLaneStack) {
                stack[sp++] = K[c];
            }
        }
        if (sp == 0) break;
        g = (unsigned)stack[--sp];
    }
    out[it] = make_float2(acc.x, acc.y);
}

// Work counter over the groups (nbx_bh_work with NBX_OPT_BH_WALK != 0): children visited, pair laws evaluated, opening tests and
// groups loaded, per launch.  The same visits / pair laws / tests as bh_eval.hip's k_bh_count over the nodes if -- and only if --
// both walks make the same decisions.
__global__ __launch_bounds__(kTile) void k_bh_count_groups(const float4* __restrict__ posm, const int lo, const int n_targets,
                                                           const BhGroup* __restrict__ groups, unsigned long long* __restrict__ totals)
{
    const int it = blockIdx.x * kTile + threadIdx.x;
    unsigned visits = 0, pairs = 0, tests = 0, loads = 0;
    if (it < n_targets) {
        const float4 pi = posm[lo + it];
        const v2f p = {pi.x, pi.y};
        int stack[kLaneStack];
        int sp = 0;
        unsigned g = 0u;
        for (;;) {
            const BhGroup G = *reinterpret_cast<const BhGroup*>(reinterpret_cast<const char*>(groups) + g);
            const int K[4] = {G.kid.x, G.kid.y, G.kid.z, G.kid.w};
            loads++;
#pragma unroll
            for (int c = 3; c >= 0; c--) {
                if (K[c] == kGroupAbsent) continue;
                const v2f nxy = {G.c[c].x, G.c[c].y};
                const v2f d = nxy - p;
                const v2f sq = d * d;
                const float d2 = sq.x + sq.y;
                visits++;
                if (K[c] >= 0) tests++;   // an interior node: the reference runs its opening test (nbody.rs:338-345)
                if (!(d2 <= G.c[c].w)) pairs++;
                else if (sp < kLaneStack) stack[sp++] = K[c];
            }
            if (sp == 0) break;
            g = (unsigned)stack[--sp];
        }
    }
    unsigned long long v = visits, q = pairs, w = tests, l = loads;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        v += __shfl_xor(v, off);
        q += __shfl_xor(q, off);
        w += __shfl_xor(w, off);
        l += __shfl_xor(l, off);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&totals[0], v);
        atomicAdd(&totals[1], q);
        atomicAdd(&totals[2], w);
        atomicAdd(&totals[3], l);
    }
}

hipError_t launch_bh_count_groups(const float4* posm, int lo, int n_targets, const BhGroup* groups, unsigned long long* totals,
                                  hipStream_t stream)
{
    if (n_targets <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_bh_count_groups, dim3((unsigned)((n_targets + kTile - 1) / kTile)), dim3(kTile), 0, stream, posm, lo, n_targets,
                       groups, totals);
    return hipGetLastError();
}

// ---- launchers ---------------------------------------------------------------------------------------------------------------
size_t bh_groups_count(int node_cap) { return (size_t)node_cap + 1; }
bool bh_groups_addressable(int node_cap) { return ((size_t)node_cap + 1) * sizeof(BhGroup) < ((size_t)1 << 31); }

hipError_t launch_bh_groups(const BhNode* nodes, int n_nodes_or_cap, float theta, BhGroup* groups, bool compact, hipStream_t stream,
                            int* gate_counters, int gate_node_cap, int gate_crowd_limit, int gate_queue_limit)
{
    const BuildGate gate{gate_counters, gate_node_cap, gate_crowd_limit, gate_queue_limit, nullptr};
    const int threads = n_nodes_or_cap > 0 ? n_nodes_or_cap : 1;   // an empty tree still gets its group 0
    hipLaunchKernelGGL(k_bh_groups, dim3((unsigned)((threads + kTile - 1) / kTile)), dim3(kTile), 0, stream, nodes, n_nodes_or_cap,
                       theta, groups, compact ? 1 : 0, gate);
    return hipGetLastError();
}

// how many walks (workgroups) launch_bh_walk_groups starts for n_targets bodies in the wave form, and with how many bodies each
int bh_walk_count(int n_targets, int* bodies_per_walk)
{
    int bpw = 64;
    // at least 4 096 walks, between 2 and 64 bodies each: a small system's step is one walk's chain of dependent loads, and the
    // fewer bodies share a walk the shorter it is (10 000 bodies: traversal 0.0413 / 0.0316 / 0.0344 / 0.0381 ms with 1 / 2 / 4 / 8)
    while (bpw > 2 && (n_targets + bpw - 1) / bpw < 4096) bpw >>= 1;
    static const int forced = [] { const char* v = std::getenv("NBX_BH_BPW"); return v ? std::atoi(v) : 0; }();   // (A/B knob)
    if (forced == 1 || forced == 2 || forced == 4 || forced == 8 || forced == 16 || forced == 32 || forced == 64) bpw = forced;
    if (bodies_per_walk) *bodies_per_walk = bpw;
    const int nblk = (n_targets + bpw - 1) / bpw;
    return (nblk + 7) / 8 * 8;
}

template <int ASM>
static void launch_wave_walk(int bpw, dim3 g, hipStream_t stream, const float4* posm, int lo, int n_targets, const BhGroup* groups,
                             float2* out, const unsigned* perm, BuildGate gate, unsigned long long* trace, BhKick kick)
{
    auto go = [&](auto kernel) {
        void* sink = kick.vel ? static_cast<void*>(kick.vel) : static_cast<void*>(out);
        hipLaunchKernelGGL(kernel, g, dim3(64), 0, stream, posm, lo, n_targets, groups, sink, perm, 1, gate, trace,
                           kick.vel ? kick.posm : nullptr, kick.dt, kick.vel ? kick.sorted : nullptr);
    };
#define NBX_WALK(B) do { if (trace) go(k_bh_walk_groups<B, ASM, true>); else go(k_bh_walk_groups<B, ASM, false>); } while (0)
    if (bpw == 64) NBX_WALK(64);
    else if (bpw == 32) NBX_WALK(32);
    else if (bpw == 16) NBX_WALK(16);
    else if (bpw == 8) NBX_WALK(8);
    else if (bpw == 4) NBX_WALK(4);
    else if (bpw == 2) NBX_WALK(2);
    else NBX_WALK(1);
#undef NBX_WALK
}

hipError_t launch_bh_walk_groups(const float4* posm, int lo, int n_targets, const BhGroup* groups, float2* out, hipStream_t stream,
                                 const unsigned* perm, bool wave, bool hand_scheduled, int* gate_counters, int gate_node_cap,
                                 int gate_crowd_limit, int gate_queue_limit, unsigned long long* trace, const BhKick* kick)
{
    if (n_targets <= 0) return hipSuccess;
    const BhKick kd = kick ? *kick : BhKick{nullptr, nullptr, 0.0f, nullptr, nullptr};
    if (kd.vel && !(wave && perm && kd.posm)) return hipErrorInvalidValue;   // (the per-lane form has no kick)
    const BuildGate gate{gate_counters, gate_node_cap, gate_crowd_limit, gate_queue_limit, kd.vel ? kd.host_out : nullptr};
    if (wave && perm) {
        // bodies per wave: aim at >= 4 walks per SIMD (4096 waves), between 4 and 64 bodies each (as the node walk)
        int bpw = 64;
        const dim3 g((unsigned)bh_walk_count(n_targets, &bpw));
        static const int pipe = [] { const char* v = std::getenv("NBX_BH_WALK_PIPE"); return v ? std::atoi(v) : 1; }();   // (0: the loop of rounds 4-5, for the A/B)
        if (hand_scheduled && pipe) launch_wave_walk<2>(bpw, g, stream, posm, lo, n_targets, groups, out, perm, gate, trace, kd);
        else if (hand_scheduled) launch_wave_walk<1>(bpw, g, stream, posm, lo, n_targets, groups, out, perm, gate, trace, kd);
        else launch_wave_walk<0>(bpw, g, stream, posm, lo, n_targets, groups, out, perm, gate, trace, kd);
    } else {
        const int block = n_targets <= 65536 ? 64 : kTile;
        hipLaunchKernelGGL(k_bh_walk_groups_lane, dim3((unsigned)((n_targets + block - 1) / block)), dim3(block), 0, stream, posm, lo,
                           n_targets, groups, out, perm, gate);
    }
    return hipGetLastError();
}

// test hook: T of bh_threshold.h evaluated on the device
__global__ void k_bh_thresholds(const float* __restrict__ s, const float* __restrict__ theta, float* __restrict__ out, const int count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = bh_take_threshold(s[i], theta[i]);
}
hipError_t launch_bh_thresholds(const float* s, const float* theta, float* out, int count, hipStream_t stream)
{
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_bh_thresholds, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, s, theta, out, count);
    return hipGetLastError();
}

}  // namespace nbx
