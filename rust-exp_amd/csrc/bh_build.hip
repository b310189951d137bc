// bh_build.hip -- quadtree build ON THE DEVICE.  The fast mode's DEFAULT from 1 024 bodies on
// (NBX_OPT_BH_TREE; 512 with exactly summed nodes); the bit-exact mode builds on the host unless asked (reference fold only).
//
// The host build (host_ops.cpp) inserts the bodies one by one exactly as the reference does (nbody.rs:388-415); at 1 M bodies that
// is 13-16 ms per step, at the reference's 10 000 bodies 0.5 ms.  This file builds the SAME flattened tree without leaving the GPU:
//
//   1. root AABB = min/max of positions (exact; nbody.rs:388-398)
//   2. per body: the path of quadrant choices, replaying quadrant_from_point / create_children with the
//      reference's own f32 midpoint arithmetic (cx = (x1+x2)*0.5, nbody.rs:289-290, :324-331) for 31 levels
//      -> 62-bit key, 2 bits per level, quadrant order [UL,UR,LL,LR] = 0..3 like the reference's child array
//   3. radix sort (rocPRIM) of (key, body index)
//   3b. the reference's EPS merge (nbody.rs:249-260) -- reference fold: whole clusters of close bodies replayed in arrival order
//       (k_cells / k_blobs / k_place, section 3c); exact-sum class: close PAIRS decided from the sorted keys and the arrival
//       order (k_merge_links / k_merge_keys, section 3b)
//   4. nodes straight from the sorted keys: every node is (first body a, depth l); how many nodes start at each body
//      follows from the digits it shares with its two neighbours, an exclusive scan of those counts gives every node's
//      PRE-ORDER slot, and a node's skip pointer is the slot of the first node after its bodies (see "the tree from
//      the sorted keys" below) -- no level-by-level sweep, no host round trips, one read-back of the node count
//   5. interior masses and centres of mass, two classes (NBX_OPT_BH_FOLD):
//      fold = 1 (default up to 65 536 bodies): the reference's f32 running fold over the node's bodies in ARRIVAL order
//               (nbody.rs:303-320) -- small nodes in k_emit, the others in k_fold_big (one pair of waves per node: m chain, IEEE
//               reciprocals, p chain), the root on a side stream from the start of the build.  The flattened tree then equals the
//               host tree BIT FOR BIT; what the cluster replay cannot reproduce node for node (a blob whose successive centres
//               part ways above its leaf, a merge that hinges on another cluster, ...: 3c) is detected and the step goes to the
//               host build.
//      fold = 0 (above 65 536 bodies): deterministic fp64 prefix sums over the sorted bodies, one rounding to f32 per node.
//      Node sizes: the first body's path replayed with the reference's f32 midpoints.
//
// fold = 0 is its own tolerance class.  Same node set, same s = x2-x1 per node, same leaf records as the
// host build + flatten, INCLUDING the reference's EPS merge for pairs (nbody.rs:249-260); what differs there:
//   * interior centres of mass are the f32 rounding of the exact weighted mean instead of the reference's
//     particle-by-particle f32 running fold (which drifts by up to ~6e-4 relative at 100 k bodies);
//   * a merged pair's leaf sits on the path of its FIRST-arrived member, the reference's on the path of the blob's centre
//     (different only when a third body shares the pair's last common cell, <= EPS-sized);
//   * clusters of three or more bodies within EPS: the reference folds arrivals into one blob while each stays within EPS
//     of the blob's current centre; here only the first two of a run of mutually-close sorted neighbours merge (bodies whose
//     62-bit keys are identical -- the same level-31 cell, 4.7e-8 of the box -- always share one leaf, any number of them);
//     more than max(16, n/2000) such bodies send the step to the host build;
//   * no depth-50 panic (nbody.rs:230-232): keys stop at level 31.
// A pair within EPS whose members are not neighbours in key order (a third body of their common cell between them) merges in the
// reference when every body between them arrived later: the neighbours-only merge misses it.  fold = 1 replays it like every
// other cluster (3c); fold = 0 lives with it (its own tolerance class).
#include <atomic>
#include <cstring>   // rocPRIM's texture_cache_iterator.hpp calls memset() without including it

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>

#include "kernels.h"

namespace nbx {

// rocPRIM's (key, index) sort for systems above kSmallFrontMax bodies: its merge-sort path (the library's choice up to 2^20 pairs)
// with first-level blocks of 512 x 8 pairs instead of 256 x 4 -- two merge passes fewer: 183 vs 206 us at 1 048 576 pairs,
// 109 vs 111 at 262 144 (tools/ubench_sort_cfg.hip, profiles/r04_ubench_sort_cfg.txt)
constexpr int kBigSortFrom = 262144;   // (below: the library's own shape -- 65 536 pairs lose 10 us to the bigger blocks, too few of them)
using BuildSortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::merge_sort_config<512, 512, 8, 128, 128, 4>,
                                                   rocprim::default_config, (size_t)1 << 20>;

constexpr int kLevels = 31;   // 62-bit keys

__device__ __forceinline__ unsigned enc_f32(float f)
{
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);   // monotonic: float order == unsigned order
}
__device__ __forceinline__ float dec_f32(unsigned u)
{
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}

// box[0..3] = enc(min x), enc(min y), enc(max x), enc(max y).  One launch, no initialisation kernel: every workgroup leaves its
// partial box in part[], takes a ticket, and the LAST one to finish folds the partials (fixed order) and publishes the box
// (round 2: k_init_box + atomicMin/Max into a pre-initialised word).  The ticket word must be zero at launch: the last
// workgroup clears it again (the engine zeroes it once when the workspace is allocated).
// clear_*: words the kernels BEHIND this one add to (warm sort: the splitter candidates' ranks, the buckets' counts, the build's
// counters -- which k_keys clears in the cold path), cleared here to save a launch
__global__ __launch_bounds__(kTile) void k_bbox(const float4* __restrict__ posm, const int n, unsigned* __restrict__ box,
                                                float4* __restrict__ part, int* __restrict__ ticket, int* __restrict__ clear_a,
                                                const int count_a, int* __restrict__ clear_b, const int count_b, int* __restrict__ clear_c,
                                                const int count_c)
{
    for (int i = blockIdx.x * kTile + threadIdx.x; i < count_a; i += (int)gridDim.x * kTile) clear_a[i] = 0;
    for (int i = blockIdx.x * kTile + threadIdx.x; i < count_b; i += (int)gridDim.x * kTile) clear_b[i] = 0;
    for (int i = blockIdx.x * kTile + threadIdx.x; i < count_c; i += (int)gridDim.x * kTile) clear_c[i] = 0;
    float x1 = 3.40282347e+38f, y1 = 3.40282347e+38f, x2 = -3.40282347e+38f, y2 = -3.40282347e+38f;
    {   // eight independent loads in flight per thread (round 5: one at a time, a million bodies took 12 us -- sixteen dependent
        // round trips per thread; min and max are exact in any order)
        constexpr int kFlight = 8;
        const int stride = (int)gridDim.x * kTile;
        for (int i0 = blockIdx.x * kTile + threadIdx.x; i0 < n; i0 += kFlight * stride) {
            float4 q[kFlight];
#pragma unroll
            for (int u = 0; u < kFlight; u++) {
                const int i = i0 + u * stride;
                q[u] = posm[i < n ? i : i0];
            }
#pragma unroll
            for (int u = 0; u < kFlight; u++) {
                x1 = fminf(x1, q[u].x); y1 = fminf(y1, q[u].y); x2 = fmaxf(x2, q[u].x); y2 = fmaxf(y2, q[u].y);
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        x1 = fminf(x1, __shfl_xor(x1, off)); y1 = fminf(y1, __shfl_xor(y1, off));
        x2 = fmaxf(x2, __shfl_xor(x2, off)); y2 = fmaxf(y2, __shfl_xor(y2, off));
    }
    __shared__ float red[4][4];
    __shared__ int last;
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wave][0] = x1; red[wave][1] = y1; red[wave][2] = x2; red[wave][3] = y2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) {
            x1 = fminf(x1, red[w][0]); y1 = fminf(y1, red[w][1]); x2 = fmaxf(x2, red[w][2]); y2 = fmaxf(y2, red[w][3]);
        }
        part[blockIdx.x] = make_float4(x1, y1, x2, y2);
        __threadfence();
        last = atomicAdd(ticket, 1) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    x1 = 3.40282347e+38f; y1 = 3.40282347e+38f; x2 = -3.40282347e+38f; y2 = -3.40282347e+38f;
    for (int b = threadIdx.x; b < (int)gridDim.x; b += kTile) {     // at most 256 partials (the launcher caps the grid)
        const float4 q = part[b];
        x1 = fminf(x1, q.x); y1 = fminf(y1, q.y); x2 = fmaxf(x2, q.z); y2 = fmaxf(y2, q.w);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        x1 = fminf(x1, __shfl_xor(x1, off)); y1 = fminf(y1, __shfl_xor(y1, off));
        x2 = fmaxf(x2, __shfl_xor(x2, off)); y2 = fmaxf(y2, __shfl_xor(y2, off));
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[wave][0] = x1; red[wave][1] = y1; red[wave][2] = x2; red[wave][3] = y2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) {
            x1 = fminf(x1, red[w][0]); y1 = fminf(y1, red[w][1]); x2 = fmaxf(x2, red[w][2]); y2 = fmaxf(y2, red[w][3]);
        }
        box[0] = enc_f32(x1); box[1] = enc_f32(y1); box[2] = enc_f32(x2); box[3] = enc_f32(y2);
        *ticket = 0;
    }
}

// one step of quadrant_from_point + the child's AABB from create_children (unfused f32, nbody.rs:289-300,:324-331)
__device__ __forceinline__ int descend(float& x1, float& y1, float& x2, float& y2, const float x, const float y)
{
    const float cx = __fmul_rn(__fadd_rn(x1, x2), 0.5f);
    const float cy = __fmul_rn(__fadd_rn(y1, y2), 0.5f);
    int q;
    if (y < cy) { q = 2; y2 = cy; } else { q = 0; y1 = cy; }
    if (x < cx) { x2 = cx; } else { q += 1; x1 = cx; }
    return q;
}

__device__ __forceinline__ void fold_mass(float& px, float& py, float& m, const float qx, const float qy, const float qm)
{
    if (m == 0.0f) { px = qx; py = qy; m = qm; return; }                 // nbody.rs:305-311
    const float inv = 1.0f / __fadd_rn(m, qm);                            // :315
    px = __fmul_rn(__fadd_rn(__fmul_rn(px, m), __fmul_rn(qx, qm)), inv);  // :316
    py = __fmul_rn(__fadd_rn(__fmul_rn(py, m), __fmul_rn(qy, qm)), inv);  // :317
    m = __fadd_rn(m, qm);                                                 // :318
}

// one level down by a recorded quadrant choice: the child's AABB as create_children makes it (nbody.rs:289-300)
__device__ __forceinline__ void descend_digit(float& x1, float& y1, float& x2, float& y2, const int q)
{
    const float cx = __fmul_rn(__fadd_rn(x1, x2), 0.5f);
    const float cy = __fmul_rn(__fadd_rn(y1, y2), 0.5f);
    if (q & 2) y2 = cy; else y1 = cy;
    if (q & 1) x1 = cx; else x2 = cx;
}

// the path of an arbitrary point (a blob's centre): the same 31 quadrant choices k_keys records for a body
__device__ __forceinline__ unsigned long long path_key(const unsigned* __restrict__ box, const float x, const float y)
{
    float x1 = dec_f32(box[0]), y1 = dec_f32(box[1]), x2 = dec_f32(box[2]), y2 = dec_f32(box[3]);
    unsigned long long key = 0;
#pragma unroll 1
    for (int l = 0; l < kLevels; l++) key = (key << 2) | (unsigned long long)descend(x1, y1, x2, y2, x, y);
    return key;
}

__global__ __launch_bounds__(kTile) void k_keys(const float4* __restrict__ posm, const int n,
                                                const unsigned* __restrict__ box, unsigned long long* __restrict__ keys,
                                                unsigned* __restrict__ idx, int* __restrict__ counters,
                                                unsigned long long* __restrict__ cell_table, const int cell_slots)
{
    const int i = blockIdx.x * kTile + threadIdx.x;
    // this build's counters and tickets (see Workspace): cleared here instead of by a memset of their own
    if (i < 8) counters[i] = 0;
    // ... and the table of occupied grid cells that k_cells fills after the sort (reference fold only; at most 4 slots per body)
    for (int t = i; t < cell_slots; t += (int)gridDim.x * kTile) cell_table[t] = 0ull;
    if (i >= n) return;
    float x1 = dec_f32(box[0]), y1 = dec_f32(box[1]), x2 = dec_f32(box[2]), y2 = dec_f32(box[3]);
    const float4 p = posm[i];
    unsigned long long key = 0;
#pragma unroll 1
    for (int l = 0; l < kLevels; l++) key = (key << 2) | (unsigned long long)descend(x1, y1, x2, y2, p.x, p.y);
    keys[i] = key;
    idx[i] = (unsigned)i;
}

// ---- the tree from the sorted keys, without a level-by-level sweep --------------------------------------------------
//
// With the keys sorted, every tree node is a pair (a, l): the bodies that share the first l digits of key[a], where a is
// the FIRST body of that group.  Let c(j) = number of leading digits key[j-1] and key[j] have in common (c(0) = c(n) = -1).
//   * the deepest node starting at a is a's leaf, at depth leaf(a) = min(31, 1 + max(c(a), c(a+1))): one level below the
//     depth at which a still shares a node with a neighbour (the reference splits a node as soon as it holds two bodies,
//     nbody.rs:262-283, so the leaf sits exactly there);
//   * the shallowest node starting at a has depth c(a)+1 (one digit deeper than what a shares with its left neighbour);
//   * every depth in between starts at a too (single-child chain nodes included, as in the reference's tree).
// So body a contributes cnt(a) = leaf(a) - c(a) nodes (0 for a body whose key equals its left neighbour's: it lives in
// that neighbour's level-31 leaf), and in PRE-ORDER all nodes starting at a precede all nodes starting at a+1, shallow
// to deep.  An exclusive scan of cnt therefore gives every node's pre-order slot, and a node's skip pointer -- the slot
// after its subtree -- is simply base[b], b = first body outside the node (found by galloping over the sorted keys).
// Centres of mass come from fp64 prefix sums over the sorted bodies (direct fp64 sums for nodes of <= 8 bodies).

__device__ __forceinline__ int common_digits(const unsigned long long x, const unsigned long long y)
{
    const unsigned long long d = x ^ y;
    if (d == 0ull) return kLevels;              // identical down to level 31
    return (__clzll((long long)d) - 2) >> 1;    // keys occupy the low 62 bits, digit l = bits 61-2l, 60-2l
}

struct ScanItem {
    double m, mx, my;
    int cnt;     // nodes starting at the body
    int ent;     // 1 if the body starts an entity (a leaf), i.e. if it starts any node at all
};
__device__ __forceinline__ ScanItem scan_add(const ScanItem& a, const ScanItem& b)
{
    return ScanItem{a.m + b.m, a.mx + b.mx, a.my + b.my, a.cnt + b.cnt, a.ent + b.ent};
}
constexpr int kScanPerThread = 4;
constexpr int kScanBlock = kTile * kScanPerThread;

// first index > j whose key differs from keys[j] (n if none): bodies with identical (merged) keys form one leaf
__device__ __forceinline__ int run_end(const unsigned long long* __restrict__ keys, const int j, const int n)
{
    const unsigned long long k = keys[j];
    int lo = j, step = 1;                   // keys[lo] == k
    while (lo + step < n && keys[lo + step] == k) { lo += step; step <<= 1; }
    int hi = lo + step < n ? lo + step : n; // first known mismatch (n = past the end)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (keys[mid] == k) lo = mid; else hi = mid;
    }
    return hi;
}

// number of tree nodes that start at sorted body j.  Bodies with identical keys (the same level-31 cell, or an EPS-merged
// pair after k_merge_keys) are ONE leaf: only the first of them starts nodes, and its leaf sits one level below the depth
// it shares with its nearest DIFFERENT neighbours -- exactly where the reference leaves a merged blob (nbody.rs:249-260).
__device__ __forceinline__ int nodes_starting_at(const unsigned long long* __restrict__ keys, const int j, const int n)
{
    const unsigned long long k = keys[j];
    const int cl = j == 0 ? -1 : common_digits(keys[j - 1], k);
    if (j > 0 && cl >= kLevels) return 0;
    const int e = run_end(keys, j, n);
    const int cr = e == n ? -1 : common_digits(k, keys[e]);
    int leaf = 1 + (cl > cr ? cl : cr);
    if (leaf > kLevels) leaf = kLevels;
    return leaf - cl;
}

// ---- 3b. the reference's EPS merge, for pairs ----------------------------------------------------------------------------
// nbody.rs:249-260: a body B arriving at a non-empty exterior node merges into it when the node's content A is closer than
// EPS in both axes.  B arrives at A's leaf iff that leaf -- one level below the deepest cell A shares with any body inserted
// BEFORE B -- still contains B, i.e. iff no earlier body C shares at least as many leading digits with A as B does:
//     merge(A, B)  <=>  |dx| < EPS and |dy| < EPS  and  there is no C with idx(C) < idx(B), C != A, common(A, C) >= common(A, B)
// (A = the earlier of the two).  Candidates for C are contiguous around the pair in the sorted order (everything sharing
// >= common(A, B) digits with A), so the test is a short outward scan from the pair.
// The unit of all this is an ENTITY: a maximal run of bodies with identical keys (one level-31 cell: they always end up in one
// leaf, in index order thanks to the stable sort) -- usually a single body.  close[j] = 1 marks a boundary j between two
// different entities (the one ending at j-1 and the one starting at j) that the reference merges.
constexpr int kMergeScanCap = 4096;   // per side; undecided after that many neighbours -> merge (needs an early, crowded pair)
constexpr int kRunCap = 4096;         // longest identical-key run walked back to its start (longer: treated as starting there)

__device__ __forceinline__ int run_start(const unsigned long long* __restrict__ keys, const int j)
{
    const unsigned long long k = keys[j];
    int r = j;
    for (int t = 0; r > 0 && t < kRunCap && keys[r - 1] == k; t++) r--;
    return r;
}

// Also gathers the bodies into sorted order (sb[j] = posm[idx[j]]: round 2 had a kernel of its own for that).
__device__ __forceinline__ void merge_links_body(const int j, const float4* __restrict__ posm, float4* __restrict__ sb,
                                                 const unsigned long long* __restrict__ keys, const unsigned* __restrict__ idx,
                                                 const int n, unsigned char* __restrict__ close, int* __restrict__ crowded)
{
    const float4 b = posm[idx[j]];
    sb[j] = b;
    unsigned char out = 0;
    if (j + 1 < n && keys[j + 1] == keys[j] && !(j > 0 && keys[j - 1] == keys[j])) {
        // The first of several bodies of one level-31 cell: one leaf as long as every arrival is within EPS of the centre the
        // earlier ones have folded to.  Where an ulp of the coordinates is no longer small against EPS (|x| in the thousands) the
        // folded centre of even identical positions can sit more than EPS away (nbody.rs:315-317 round three times) and the
        // reference splits: such bodies are counted as left behind.
        float cx = 0.0f, cy = 0.0f, cm = 0.0f;
        int left = 0;
        for (int t = j; t < n && keys[t] == keys[j]; t++) {        // (the stable sort left them in index order)
            const float4 q = posm[idx[t]];
            if (t > j && !(fabsf(__fsub_rn(cx, q.x)) < kEps && fabsf(__fsub_rn(cy, q.y)) < kEps)) left++;
            fold_mass(cx, cy, cm, q.x, q.y, q.w);
        }
        if (left) atomicAdd(crowded, left);
    }
    if (j > 0 && keys[j - 1] != keys[j]) {
        // the entity's position is its first arrival's (later arrivals of the same cell are < 5e-8 of the box away)
        const int r = run_start(keys, j - 1);
        const float4 a = posm[idx[r]];
        if (fabsf(__fsub_rn(a.x, b.x)) < kEps && fabsf(__fsub_rn(a.y, b.y)) < kEps) {   // nbody.rs:249
            const int c = common_digits(keys[j - 1], keys[j]);
            const unsigned ia = idx[r], ib = idx[j];           // first arrival of either entity (stable sort: run start)
            const unsigned second = ia > ib ? ia : ib;
            const unsigned long long kf = ia < ib ? keys[j - 1] : keys[j];   // the earlier entity's path
            bool earlier_rival = false;
            for (int x = r - 1, t = 0; x >= 0 && t < kMergeScanCap && !earlier_rival; x--, t++) {
                if (common_digits(kf, keys[x]) < c) break;
                earlier_rival = idx[x] < second;
            }
            const unsigned long long kj = keys[j];
            for (int x = j + 1, t = 0; x < n && t < kMergeScanCap && !earlier_rival; x++, t++) {
                if (keys[x] == kj) continue;                   // the right entity's own later arrivals
                if (common_digits(kf, keys[x]) < c) break;
                earlier_rival = idx[x] < second;
            }
            out = earlier_rival ? 0 : 1;
        }
    }
    close[j] = out;
}

__global__ __launch_bounds__(kTile) void k_merge_links(const float4* __restrict__ posm, float4* __restrict__ sb,
                                                       const unsigned long long* __restrict__ keys,
                                                       const unsigned* __restrict__ idx, const int n,
                                                       unsigned char* __restrict__ close, int* __restrict__ crowded)
{
    const int j = blockIdx.x * kTile + threadIdx.x;
    if (j < n) merge_links_body(j, posm, sb, keys, idx, n, close, crowded);
}

// ---- 3c. the reference's EPS merge in full (reference fold: that class promises the reference's tree node for node) ---------
//
// Sequential insertion (nbody.rs:226-284) decides a body B's fate when it ARRIVES: among the entities in the tree at that moment
// -- single bodies and blobs of merged bodies -- B walks down to the leaf of the one entity A that shares the most leading path
// digits with it (a tie, or none: B opens a leaf of its own); if A's current centre is closer than EPS in both axes B is folded
// into A (add_mass), else the leaf splits and B becomes an entity.  A blob's centre moves with every member, later arrivals are
// tested against the moved centre, and the blob travels down by its centre whenever its leaf splits (nbody.rs:271-281).
// All of this involves only bodies within 2 EPS of one another: B within EPS of a centre is within 2 EPS of one of the members.
// So:
//   * k_cells   a hash table of the occupied cells of a grid (the quadtree level whose cells are >= 2.5 EPS wide; a cell is a
//               contiguous range of the sorted keys) -> "who is within 2 EPS of this point" is nine probes, not a search;
//   * k_blobs   every entity-by-key (a run of identical keys; usually one body) looks around; one with company that arrived
//               before all of its neighbours collects its connected component (chains of < 2 EPS links; usually 2-5 bodies) and,
//               if it is the component's first arrival, REPLAYS the component's arrivals in index order, by the rule above:
//               entities in LDS, exact f32 folds, the nearest-entity rule from the keys, outside bodies that arrived earlier and
//               share the cell taken into account (k_merge_links' rival scan).  A blob's path is its centre's; its members take
//               the path of its last centre as their key ("ghosts" when that is not their own: listed for k_place);
//   * k_place   the bodies in the order of their ENTITY keys (a ghost moves next to its entity: usually by a slot or two, but by
//               any distance when a coarse cell boundary runs between the two) -- keys, indices, records, out of place.
// The tree files a blob under the path of its LAST centre.  In the reference the path is made of stretches, each laid down by the
// centre the blob had while its leaf went from one depth to the next (the first by the opener's own position); the last centre's
// path is that path down to the blob's final leaf iff every centre the blob ever had shares it that far, and the opener's
// position shares it as far as the opener's leaf went before it took in its first body (at most the digits those two share).
// What that, or the replay, cannot reproduce soundly is not guessed: the step then goes to the host build (counted in
// counters[1], the reasons in counters[5]):
//   * a blob whose successive centres do not share one path down to its final leaf (k_emit compares pmin with the leaf depth;
//     blobs of three or more bodies with an unmerged body a fraction of EPS away, mostly).  Telling WHICH centre laid down which
//     stretch was built too: every entity's depth over time then hinges on its nearest earlier-arrived neighbours in key order,
//     and in the dense cores where blobs form those are members of OTHER components more often than not (the 10 000-body
//     nb_random_disk: 695 of 1 000 steps refused, against none like this);
//   * a body of ANOTHER component among the rivals of a merge (its entity may sit elsewhere),
//   * any body outside the component within EPS of any centre a blob ever had (the 2 EPS argument holds for exact arithmetic;
//     this checks the computed centres),
//   * two entities in one level-31 cell that do not merge (the reference goes deeper than the keys do),
//   * components of more than kBlobRuns entities / kBlobBodies bodies, more than kGhostCap ghosts, crowded neighbourhoods.
constexpr int kCloseScanCap = 512;     // entities looked at around one point
constexpr int kSideStreamsFrom = 4096;
constexpr int kBlobRuns = 48;
constexpr int kBlobBodies = 96;
constexpr int kGhostCap = 4096;
constexpr int kRivalScanCap = 1024;

// why a build of the reference-fold class refused: counters[1] counts, counters[5] collects these bits (NBX_LOG prints them)
enum : int {
    kWhyCrowdedScan = 1,      // more than kCloseScanCap entities around one point
    kWhyBigComponent = 4,     // more than kBlobRuns entities / kBlobBodies bodies in one component
    kWhyRival = 8,            // a merge hinges on a body of another component (or on too long a scan)
    kWhyOutsider = 16,        // somebody outside the component within EPS of a blob's centre
    kWhyLevel31 = 32,         // two entities in one level-31 cell that do not merge
    kWhyGhosts = 64,          // more than kGhostCap bodies to move
    kWhyCentrePath = 128,     // a blob's centres do not share one path down to the blob's leaf
    kWhyBigLeaf = 256,        // a leaf of more bodies than the leaf fold orders
    kWhyDepthPanic = 512,     // (bit-exact mode only) a leaf deeper than 25 levels: the reference may panic on its depth counter
};
__device__ __forceinline__ void refuse(int* __restrict__ counters, const int why)
{
    atomicAdd(&counters[1], 1);
    atomicOr(&counters[5], why);
}

struct CellGrid {
    int D, sh;                         // D digits of a key name a grid cell; key >> sh = the cell's prefix
    const unsigned long long* hk;      // open addressing: prefix + 1 (0 = free) ...
    const int* hv;                     // ... -> the first sorted slot of the cell
    unsigned mask;
};

__device__ __forceinline__ CellGrid make_grid(const unsigned* __restrict__ box, const unsigned long long* hk, const int* hv,
                                              const unsigned mask)
{
    // cells at least 2.5 EPS wide in both axes (widths halve per level; the f32 midpoints move them by rounding only)
    float wx = dec_f32(box[2]) - dec_f32(box[0]), wy = dec_f32(box[3]) - dec_f32(box[1]);
    int D = 0;
    while (D < kLevels && wx * 0.5f >= 2.5f * kEps && wy * 0.5f >= 2.5f * kEps) { wx *= 0.5f; wy *= 0.5f; D++; }
    return CellGrid{D, 2 * (kLevels - D), hk, hv, mask};
}

__device__ __forceinline__ unsigned hash_cell(unsigned long long c)
{
    c ^= c >> 33; c *= 0xff51afd7ed558ccdull;
    c ^= c >> 33; c *= 0xc4ceb9fe1a85ec53ull;
    c ^= c >> 33;
    return (unsigned)c;
}

// Gathers the bodies into sorted order (sb[j] = posm[idx[j]]), starts every body as its own entity, and enters the first body
// of every grid cell into the table.
__global__ __launch_bounds__(kTile) void k_cells(const float4* __restrict__ posm, float4* __restrict__ sb,
                                                 const unsigned long long* __restrict__ keys, const unsigned* __restrict__ idx,
                                                 const unsigned* __restrict__ box, const int n, unsigned long long* __restrict__ hk,
                                                 int* __restrict__ hv, const unsigned mask, unsigned long long* __restrict__ ekey,
                                                 unsigned char* __restrict__ pmin)
{
    const int j = blockIdx.x * kTile + threadIdx.x;
    if (j >= n) return;
    sb[j] = posm[idx[j]];
    const unsigned long long k = keys[j];
    ekey[j] = k;
    pmin[j] = (unsigned char)kLevels;
    const CellGrid g = make_grid(box, hk, hv, mask);
    const unsigned long long prefix = k >> g.sh;
    if (j > 0 && (keys[j - 1] >> g.sh) == prefix) return;
    unsigned h = hash_cell(prefix) & mask;
    for (;;) {
        const unsigned long long old = atomicCAS(&hk[h], 0ull, prefix + 1ull);   // (a prefix is entered once: by its first slot)
        if (old == 0ull) { hv[h] = j; return; }
        h = (h + 1u) & mask;
    }
}

// every second bit of a word: bit b of v -> bit 2b (and back)
__device__ __forceinline__ unsigned long long spread_bits(const unsigned v)
{
    unsigned long long x = v;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x << 2)) & 0x3333333333333333ull;
    x = (x | (x << 1)) & 0x5555555555555555ull;
    return x;
}
__device__ __forceinline__ unsigned compact_bits(unsigned long long x)
{
    x &= 0x5555555555555555ull;
    x = (x | (x >> 1)) & 0x3333333333333333ull;
    x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
    x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
    x = (x | (x >> 16)) & 0x00000000FFFFFFFFull;
    return (unsigned)x;
}

// f(first slot of an entity-by-key) for every one in the 3 x 3 block of grid cells around the cell of `at` (a path key), until f
// returns false.  0: all visited; 1: stopped by f; 2: more than kCloseScanCap of them (a collinear or collapsed system).
// own >= 0: `at` is the key of the body in sorted slot `own` -- its own cell is then found by walking left from that slot (the
// neighbouring keys are in the cache of the wave's coalesced loads) instead of through the table.  The table probes of the other
// cells are issued together, then the hits' slots, before anything is looked at: one round trip each instead of nine in a row
// (nearly every probe finds an empty cell).
template <class F>
__device__ __forceinline__ int visit_entities_near(const CellGrid& g, const unsigned long long at,
                                                   const unsigned long long* __restrict__ keys, const int n, const int own, F&& f)
{
    const unsigned long long centre = at >> g.sh;     // digit = (lower << 1) | right, one per level
    const unsigned ix = compact_bits(centre), iy = compact_bits(centre >> 1);
    const long long lim = 1ll << g.D;
    unsigned long long prefix[9], found[9];
    unsigned h[9];
#pragma unroll
    for (int c9 = 0; c9 < 9; c9++) {
        const long long cx = (long long)ix + (c9 % 3 - 1), cy = (long long)iy + (c9 / 3 - 1);
        const bool inside = !(cx < 0 || cy < 0 || cx >= lim || cy >= lim);
        const unsigned long long p = (spread_bits((unsigned)cy) << 1) | spread_bits((unsigned)cx);
        prefix[c9] = p;
        h[c9] = hash_cell(p) & g.mask;
        found[c9] = !inside ? 0ull : (c9 == 4 && own >= 0) ? p + 1ull : g.hk[h[c9]];
    }
    int start[9];
#pragma unroll
    for (int c9 = 0; c9 < 9; c9++) start[c9] = (found[c9] == prefix[c9] + 1ull && !(c9 == 4 && own >= 0)) ? g.hv[h[c9]] : -1;
    if (own >= 0) {
        int t = own, steps = 0;
        while (t > 0 && (keys[t - 1] >> g.sh) == centre && ++steps <= 64) t--;
        start[4] = steps > 64 ? g.hv[h[4]] : t;       // (a crowded cell: the table knows where it starts -- it holds every cell)
        if (steps > 64) found[4] = g.hk[h[4]];
    }
    int seen = 0;
#pragma unroll 1
    for (int c9 = 0; c9 < 9; c9++) {
        if (found[c9] == 0ull) continue;              // outside the grid, or nobody there
        int t = start[c9];
        if (found[c9] != prefix[c9] + 1ull) {         // the slot held another cell: probe on
            unsigned hh = h[c9];
            unsigned long long kk = found[c9];
            while (kk != 0ull && kk != prefix[c9] + 1ull) { hh = (hh + 1u) & g.mask; kk = g.hk[hh]; }
            if (kk == 0ull) continue;
            t = g.hv[hh];
        }
        const unsigned long long pc = prefix[c9];
        while (t < n && (keys[t] >> g.sh) == pc) {
            if (++seen > kCloseScanCap) return 2;
            if (!f(t)) return 1;
            t = run_end(keys, t, n);
        }
    }
    return 0;
}

__device__ __forceinline__ bool within(const float4 a, const float4 b, const float r)
{
    return fabsf(__fsub_rn(a.x, b.x)) < r && fabsf(__fsub_rn(a.y, b.y)) < r;
}

struct BlobShared {
    int run_first[kBlobRuns], run_last[kBlobRuns];   // the component: entities by key, as ranges of sorted slots
    int mem_slot[kBlobBodies];                       // its bodies in arrival order ...
    unsigned mem_idx[kBlobBodies];
    unsigned char mem_ent[kBlobBodies];              // ... and the entity each of them ended in
    unsigned char ent_pmin[kBlobBodies];             // entities of the replay: fewest digits two successive centres' paths shared
    unsigned char ent_c1[kBlobBodies];               //   digits the opener shared with the first body it took in (kLevels + 1: none yet)
    unsigned long long ent_key[kBlobBodies];         //   path: the opener's key, then the path of the current centre
    float ent_x[kBlobBodies], ent_y[kBlobBodies], ent_m[kBlobBodies];
    int ent_first[kBlobBodies];                      //   the opener's sorted slot
};

__device__ __forceinline__ bool in_component(const BlobShared& s, const int nruns, const int slot)
{
    for (int u = 0; u < nruns; u++)
        if (slot >= s.run_first[u] && slot < s.run_last[u]) return true;
    return false;
}

struct ReplayView {
    const CellGrid& g;
    const float4* __restrict__ sb;
    const unsigned long long* __restrict__ keys;
    const unsigned* __restrict__ idx;
    const unsigned* __restrict__ box;
    int n;
};

// Is a body that does not belong to the component, arrived before body (slot, ib) and shares at least c digits with it in the
// tree when that body arrives?  0 no, 1 yes (the body then never reaches the component's entity), 2 cannot tell -> host build
__device__ __forceinline__ int outside_rival(const BlobShared& s, const int nruns, const ReplayView& v, const int slot,
                                             const unsigned long long kb, const unsigned ib, const int c)
{
    int steps = 0;
    for (int dir = -1; dir <= 1; dir += 2) {
        for (int x = slot + dir; x >= 0 && x < v.n; x += dir) {
            if (common_digits(kb, v.keys[x]) < c) break;
            if (++steps > kRivalScanCap) return 2;
            if (v.idx[x] >= ib || in_component(s, nruns, x)) continue;   // arrives later / the replay knows it
            // an outsider that was there first.  It is an entity under its own key unless it belongs to a component of its own
            // (then its entity may carry another member's key): anybody within 2 EPS of it?
            const float4 px = v.sb[x];
            const unsigned long long kx = v.keys[x];
            bool company = false;
            const int st = visit_entities_near(v.g, kx, v.keys, v.n, x, [&](const int t) {
                if (v.keys[t] == kx) return true;
                if (within(px, v.sb[t], 2.0f * kEps)) { company = true; return false; }
                return true;
            });
            return (st == 2 || company) ? 2 : 1;
        }
    }
    return 0;
}

// The component's arrivals replayed in index order by ONE lane.  0, or why the host build has to do this step.
__device__ int replay_component(BlobShared& s, const int nruns, const ReplayView& v, unsigned long long* __restrict__ ekey,
                                unsigned char* __restrict__ pmin, int* __restrict__ ghosts, int* __restrict__ counters)
{
    int k = 0;
    for (int r = 0; r < nruns; r++)
        for (int slot = s.run_first[r]; slot < s.run_last[r]; slot++) {   // (the caller made sure they fit)
            const unsigned a = v.idx[slot];
            int pos = k++;
            while (pos > 0 && s.mem_idx[pos - 1] > a) { s.mem_idx[pos] = s.mem_idx[pos - 1]; s.mem_slot[pos] = s.mem_slot[pos - 1]; pos--; }
            s.mem_idx[pos] = a;
            s.mem_slot[pos] = slot;
        }
    int ne = 0;
    for (int t = 0; t < k; t++) {
        const int slot = s.mem_slot[t];
        const unsigned long long kb = v.keys[slot];
        const float4 pb = v.sb[slot];
        int best = -1, cbest = -1;
        bool tie = false;
        for (int e = 0; e < ne; e++) {
            const int c = common_digits(kb, s.ent_key[e]);
            if (c > cbest) { cbest = c; best = e; tie = false; }
            else if (c == cbest) tie = true;
        }
        bool fresh = best < 0 || tie;   // no entity of the component yet / two equally near: a leaf of its own (nbody.rs:234-240)
        if (!fresh) {
            const int st = outside_rival(s, nruns, v, slot, kb, s.mem_idx[t], cbest);
            if (st == 2) return kWhyRival;
            fresh = st == 1;
        }
        if (!fresh) {
            // arrives at entity `best`'s leaf (nbody.rs:249-260)
            if (fabsf(__fsub_rn(s.ent_x[best], pb.x)) < kEps && fabsf(__fsub_rn(s.ent_y[best], pb.y)) < kEps) {
                float x = s.ent_x[best], y = s.ent_y[best], m = s.ent_m[best];
                fold_mass(x, y, m, pb.x, pb.y, pb.w);
                s.ent_x[best] = x; s.ent_y[best] = y; s.ent_m[best] = m;
                s.mem_ent[t] = (unsigned char)best;
                // the blob's path from here on is its centre's (nbody.rs:271-281: a split re-inserts it by its position)
                const unsigned long long kc = path_key(v.box, x, y);
                if ((int)s.ent_c1[best] > kLevels) s.ent_c1[best] = (unsigned char)cbest;   // the leaf was at most this deep
                else {
                    const int c = common_digits(kc, s.ent_key[best]);
                    if (c < (int)s.ent_pmin[best]) s.ent_pmin[best] = (unsigned char)c;
                }
                s.ent_key[best] = kc;
                // nobody outside the component may ever be within EPS of this centre
                const float4 centre = make_float4(x, y, 0.0f, 0.0f);
                const int st = visit_entities_near(v.g, kc, v.keys, v.n, -1, [&](const int u) {
                    return !within(centre, v.sb[u], kEps) || in_component(s, nruns, u);
                });
                if (st != 0) return kWhyOutsider;
                continue;
            }
            if (cbest >= kLevels) return kWhyLevel31;   // the same level-31 cell and not close: the reference splits deeper than the keys go
        }
        s.ent_key[ne] = kb;
        s.ent_x[ne] = pb.x; s.ent_y[ne] = pb.y; s.ent_m[ne] = pb.w;
        s.ent_first[ne] = slot;
        s.ent_pmin[ne] = (unsigned char)kLevels;
        s.ent_c1[ne] = (unsigned char)(kLevels + 1);
        s.mem_ent[t] = (unsigned char)ne;
        ne++;
    }
    // A blob is filed under the path of its LAST centre.  That is its path in the reference's tree down to its final leaf iff
    // every centre it ever had shares that path that far (each stretch of the path was laid down by the centre of its time; the
    // fewest digits two successive centres share is the fewest any shares with the last) and the opener's own position shares
    // it down to the depth its leaf had when it took in its first body -- at most the digits the two shared.  k_emit knows the
    // final leaf depth and compares (pmin); two entities that end on one 62-bit path would need a deeper tree than the keys hold.
    // (Bounding every centre's stretch like the opener's -- it ends above the digits the NEXT body shared with the path -- was
    //  tried: the same 167 of 300 steps of the collapsing 65 536-body disc refused, 3 of 600 fuzz cases more kept.  Not kept.)
    for (int e = 0; e < ne; e++) {
        if ((int)s.ent_c1[e] > kLevels) continue;            // never took anybody in: its own key, nothing to check
        const int ca = common_digits(v.keys[s.ent_first[e]], s.ent_key[e]);
        if (ca < (int)s.ent_c1[e] && ca < (int)s.ent_pmin[e]) s.ent_pmin[e] = (unsigned char)ca;
        for (int o = 0; o < ne; o++)
            if (o != e && s.ent_key[o] == s.ent_key[e]) return kWhyLevel31;
    }
    int why = 0;
    for (int t = 0; t < k; t++) {
        const int slot = s.mem_slot[t];
        const int e = s.mem_ent[t];
        pmin[slot] = s.ent_pmin[e];
        const unsigned long long ke = s.ent_key[e];
        if (ke == v.keys[slot]) continue;
        ekey[slot] = ke;                                     // a ghost: filed under its entity's path
        const int gi = atomicAdd(&counters[4], 1);
        if (gi < kGhostCap) ghosts[gi] = slot; else why = kWhyGhosts;
    }
    return why;
}

__global__ __launch_bounds__(kTile) void k_blobs(const float4* __restrict__ sb, const unsigned long long* __restrict__ keys,
                                                 const unsigned* __restrict__ idx, const unsigned* __restrict__ box, const int n,
                                                 const unsigned long long* __restrict__ hk, const int* __restrict__ hv,
                                                 const unsigned mask, unsigned long long* __restrict__ ekey,
                                                 unsigned char* __restrict__ pmin, int* __restrict__ ghosts, int* __restrict__ counters)
{
    __shared__ BlobShared bs[kTile / 64];             // one component at a time per wave
    const int j = blockIdx.x * kTile + threadIdx.x;
    const CellGrid g = make_grid(box, hk, hv, mask);
    bool root = false;
    unsigned mine = 0;
    if (j < n && !(j > 0 && keys[j - 1] == keys[j])) {   // the first body of an entity-by-key speaks for it
        const unsigned long long kj = keys[j];
        const float4 p = sb[j];
        mine = idx[j];
        bool company = false, later = true;
        const int st = visit_entities_near(g, kj, keys, n, j, [&](const int t) {
            if (keys[t] == kj || !within(p, sb[t], 2.0f * kEps)) return true;
            company = true;
            if (idx[t] < mine) later = false;
            return true;
        });
        if (st == 2) refuse(counters, kWhyCrowdedScan);
        // alone, or a neighbour arrived first (the component's first arrival replays it): nothing to do
        root = st != 2 && company && later;
        const int last = run_end(keys, j, n);
        if (st != 2 && !company && last - j > 1) {
            // Several bodies of one level-31 cell and nobody else around: one leaf -- as long as every arrival is within EPS of the
            // centre the earlier ones have folded to.  Where an ulp of the coordinates is no longer small against EPS (|x| in the
            // thousands) the folded centre of even IDENTICAL positions can sit more than EPS away (nbody.rs:315-317 round three
            // times): the reference then splits, 31 levels are not enough, and the host build has to do it.
            float cx = 0.0f, cy = 0.0f, cm = 0.0f;
            bool one_leaf = true;
            for (int t = j; t < last && one_leaf; t++) {          // (the stable sort left them in index order)
                const float4 q = sb[t];
                if (t > j && !(fabsf(__fsub_rn(cx, q.x)) < kEps && fabsf(__fsub_rn(cy, q.y)) < kEps)) one_leaf = false;
                fold_mass(cx, cy, cm, q.x, q.y, q.w);
            }
            if (!one_leaf) refuse(counters, kWhyLevel31);
        }
    }
    // the wave's candidates one after the other (they share the wave's LDS record; a lane cannot wait for another lane)
    BlobShared& s = bs[threadIdx.x >> 6];
    unsigned long long todo = __ballot(root);
    while (todo) {
        const int lane = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        if ((int)(threadIdx.x & 63) != lane) continue;
        int nruns = 1;
        s.run_first[0] = j;
        s.run_last[0] = run_end(keys, j, n);
        int bodies = s.run_last[0] - j;
        bool first = true, fits = bodies <= kBlobBodies;
        for (int r = 0; r < nruns && first && fits; r++) {
            const int fr = s.run_first[r];
            const float4 pr = sb[fr];
            const unsigned long long kr = keys[fr];
            const int st = visit_entities_near(g, kr, keys, n, fr, [&](const int t) {
                if (keys[t] == kr || !within(pr, sb[t], 2.0f * kEps)) return true;
                for (int u = 0; u < nruns; u++)
                    if (s.run_first[u] == t) return true;
                if (idx[t] < mine) { first = false; return false; }
                const int e = run_end(keys, t, n);
                bodies += e - t;
                if (nruns == kBlobRuns || bodies > kBlobBodies) { fits = false; return false; }
                s.run_first[nruns] = t; s.run_last[nruns] = e;
                nruns++;
                return true;
            });
            if (st == 2) fits = false;
        }
        if (first) {
            const ReplayView v{g, sb, keys, idx, box, n};
            const int why = fits ? replay_component(s, nruns, v, ekey, pmin, ghosts, counters) : kWhyBigComponent;
            if (why) refuse(counters, why);
        }
    }
}

// The bodies in the order of their entity keys.  Everybody but the ghosts keeps its relative order (their keys are sorted); a
// ghost goes behind the bodies that carry its entity's key themselves.  Keys, indices, records and pmin, out of place.
__global__ __launch_bounds__(kTile) void k_place(const unsigned long long* __restrict__ keys, const unsigned long long* __restrict__ ekey,
                                                 const unsigned* __restrict__ idx, const float4* __restrict__ sb,
                                                 const unsigned char* __restrict__ pmin, const int* __restrict__ ghosts,
                                                 const int* __restrict__ counters, const int n, unsigned long long* __restrict__ keys2,
                                                 unsigned* __restrict__ idx2, float4* __restrict__ sb2, unsigned char* __restrict__ pmin2)
{
    __shared__ int gslot[kGhostCap];
    __shared__ unsigned long long gkey[kGhostCap];
    int G = counters[4];
    const bool overflow = G > kGhostCap;              // the step is refused then: everybody stays where it is, under its own
    if (overflow) G = 0;                              // key (the arrays below must hold a permutation whatever happens)
    for (int t = threadIdx.x; t < G; t += kTile) {
        const int sl = ghosts[t];
        gslot[t] = sl;
        gkey[t] = ekey[sl];
    }
    __syncthreads();
    const int j = blockIdx.x * kTile + threadIdx.x;
    if (j >= n) return;
    const unsigned long long own = keys[j], ek = overflow ? own : ekey[j];
    int pos = j;
    if (G > 0) {
        if (ek == own) {
            int before = 0, ahead = 0;
            for (int t = 0; t < G; t++) { before += gslot[t] < j ? 1 : 0; ahead += gkey[t] < own ? 1 : 0; }
            pos = j - before + ahead;
        } else {
            int lo = 0, hi = n;                       // first slot whose key is above the entity's
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (keys[mid] <= ek) lo = mid + 1; else hi = mid;
            }
            int before = 0, ahead = 0;
            for (int t = 0; t < G; t++) {
                before += gslot[t] < lo ? 1 : 0;
                ahead += (gkey[t] < ek || (gkey[t] == ek && gslot[t] < j)) ? 1 : 0;
            }
            pos = lo - before + ahead;
        }
    }
    if (pos < 0 || pos >= n) return;
    keys2[pos] = ek;
    idx2[pos] = idx[j];
    sb2[pos] = sb[j];
    pmin2[pos] = pmin[j];
}

// Pairs of entities only: of a chain of close boundaries every other one is dropped by the local rule "a boundary merges iff
// the boundary at the start of its left entity does not" (deterministic, no scan, merges stay disjoint).  All members of a
// merged pair of entities take the key of the entity that arrived first; the array stays sorted (the new key lies between the
// old ones).  Bodies left behind by the rule -- third and later entities of a chain, where the reference would have grown a
// bigger blob -- are counted in *crowded.
__device__ __forceinline__ void merge_keys_body(const int j, const unsigned long long* __restrict__ keys,
                                                const unsigned* __restrict__ idx, const unsigned char* __restrict__ close,
                                                const int n, unsigned long long* __restrict__ out, int* __restrict__ crowded)
{
    const int r = run_start(keys, j);             // this body's entity is [r, e)
    const int e = run_end(keys, j, n);
    unsigned long long k = keys[j];
    if (r > 0 && close[r]) {
        const int rl = run_start(keys, r - 1);    // left neighbour entity [rl, r)
        if (!(rl > 0 && close[rl])) k = idx[rl] < idx[r] ? keys[rl] : keys[r];   // merges with it
        else atomicAdd(crowded, 1);               // its left neighbour is already taken
    } else if (e < n && close[e]) {
        k = idx[r] < idx[e] ? keys[r] : keys[e];  // the entity starting at e merges with this one (close[r] is 0 here)
    }
    out[j] = k;
}

__global__ __launch_bounds__(kTile) void k_merge_keys(const unsigned long long* __restrict__ keys, const unsigned* __restrict__ idx,
                                                      const unsigned char* __restrict__ close, const int n,
                                                      unsigned long long* __restrict__ out, int* __restrict__ crowded)
{
    const int j = blockIdx.x * kTile + threadIdx.x;
    if (j < n) merge_keys_body(j, keys, idx, close, n, out, crowded);
}

__device__ __forceinline__ ScanItem scan_item(const float4* __restrict__ sb, const unsigned long long* __restrict__ keys,
                                              const int j, const int n, const unsigned char* __restrict__ cached = nullptr,
                                              unsigned char* __restrict__ cache = nullptr)
{
    if (j >= n) return ScanItem{0.0, 0.0, 0.0, 0, 0};
    const float4 p = sb[j];
    const int cnt = cached ? (int)cached[j] : nodes_starting_at(keys, j, n);
    if (cache) cache[j] = (unsigned char)cnt;   // (at most 32)
    return ScanItem{(double)p.w, (double)p.w * (double)p.x, (double)p.w * (double)p.y, cnt, cnt > 0 ? 1 : 0};
}

// Deterministic three-kernel exclusive scan (fixed summation tree: the same inputs give the same bits on every run,
// which a decoupled-look-back scan does not guarantee for floating point).
__device__ __forceinline__ ScanItem block_exclusive(const ScanItem mine, ScanItem* total)
{
    __shared__ ScanItem wsum[kTile / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    ScanItem inc = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        ScanItem o;
        o.m = __shfl_up(inc.m, off); o.mx = __shfl_up(inc.mx, off); o.my = __shfl_up(inc.my, off); o.cnt = __shfl_up(inc.cnt, off);
        o.ent = __shfl_up(inc.ent, off);
        if (lane >= off) inc = scan_add(o, inc);
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    ScanItem before{0.0, 0.0, 0.0, 0, 0};
    ScanItem all{0.0, 0.0, 0.0, 0, 0};
#pragma unroll
    for (int w = 0; w < kTile / 64; w++) {
        if (w < wave) before = scan_add(before, wsum[w]);
        all = scan_add(all, wsum[w]);
    }
    __syncthreads();
    if (total) *total = all;
    // exclusive = everything before this wave + the wave-inclusive value minus this thread's own item
    ScanItem ex;
    ex.m = __shfl_up(inc.m, 1); ex.mx = __shfl_up(inc.mx, 1); ex.my = __shfl_up(inc.my, 1); ex.cnt = __shfl_up(inc.cnt, 1);
    ex.ent = __shfl_up(inc.ent, 1);
    if (lane == 0) ex = ScanItem{0.0, 0.0, 0.0, 0, 0};
    return scan_add(before, ex);
}

// exclusive scan of the block sums in place (one workgroup); block_sums[nb] = grand total
__device__ __forceinline__ void scan_blocks(ScanItem* __restrict__ block_sums, const int nb);

// Block sums, then -- in the LAST workgroup to finish (ticket) -- their exclusive scan: the fixed summation tree of round 2's
// separate k_scan_blocks launch (same bits on every run), without the launch.
__global__ __launch_bounds__(kTile) void k_scan_reduce(const float4* __restrict__ sb, const unsigned long long* __restrict__ keys,
                                                       const int n, ScanItem* __restrict__ block_sums, int* __restrict__ ticket,
                                                       unsigned char* __restrict__ cnt_cache)
{
    const int j0 = blockIdx.x * kScanBlock + threadIdx.x * kScanPerThread;
    ScanItem s = scan_item(sb, keys, j0, n, nullptr, cnt_cache);
#pragma unroll
    for (int u = 1; u < kScanPerThread; u++) s = scan_add(s, scan_item(sb, keys, j0 + u, n, nullptr, cnt_cache));
    ScanItem total;
    (void)block_exclusive(s, &total);
    __shared__ int last;
    if (threadIdx.x == 0) {
        block_sums[blockIdx.x] = total;
        __threadfence();
        last = atomicAdd(ticket, 1) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    scan_blocks(block_sums, (int)gridDim.x);
}

__device__ __forceinline__ void scan_blocks(ScanItem* __restrict__ block_sums, const int nb)
{
    const int chunk = (nb + kTile - 1) / kTile;
    const int a = threadIdx.x * chunk, b = min(a + chunk, nb);
    ScanItem s{0.0, 0.0, 0.0, 0, 0};
    for (int i = a; i < b; i++) s = scan_add(s, block_sums[i]);
    ScanItem total;
    ScanItem run = block_exclusive(s, &total);
    for (int i = a; i < b; i++) {
        const ScanItem v = block_sums[i];
        block_sums[i] = run;
        run = scan_add(run, v);
    }
    if (threadIdx.x == 0) block_sums[nb] = total;
}

struct Prefix {
    double* m;    // [n+1] exclusive prefix sums over the sorted bodies
    double* mx;
    double* my;
    int* base;    // [n+1] pre-order slot of the first node starting at body j; base[n] = number of nodes
    int* ent;     // [n+1] entities (leaves) that start before body j: a node at slot k that starts at body a has ent[a] leaves and
                  //       k - ent[a] interior nodes before it in pre-order (its own leaf is the last node starting at a)
    unsigned char* cnt;   // [n] nodes starting at body j: found by k_scan_reduce, reused by k_scan_write (round 4)
    int* owner;           // [node_cap] the body at which the node of pre-order slot k starts: written by k_scan_write, so that
                          //            k_emit need not search base[] (20 dependent loads per node at a million bodies)
    int owner_cap;
};

__global__ __launch_bounds__(kTile) void k_scan_write(const float4* __restrict__ sb, const unsigned long long* __restrict__ keys,
                                                      const int n, const ScanItem* __restrict__ block_sums, Prefix p,
                                                      int* __restrict__ counters)
{
    const int j0 = blockIdx.x * kScanBlock + threadIdx.x * kScanPerThread;
    ScanItem it[kScanPerThread];
    ScanItem s{0.0, 0.0, 0.0, 0, 0};
#pragma unroll
    for (int u = 0; u < kScanPerThread; u++) {
        it[u] = scan_item(sb, keys, j0 + u, n, p.cnt);
        s = scan_add(s, it[u]);
    }
    ScanItem run = scan_add(block_sums[blockIdx.x], block_exclusive(s, nullptr));
#pragma unroll
    for (int u = 0; u < kScanPerThread; u++) {
        const int j = j0 + u;
        if (j < n) {
            p.m[j] = run.m; p.mx[j] = run.mx; p.my[j] = run.my; p.base[j] = run.cnt; p.ent[j] = run.ent;
            for (int t = 0; t < it[u].cnt; t++)          // the (at most 32) nodes that start here, shallow to deep
                if (run.cnt + t < p.owner_cap) p.owner[run.cnt + t] = j;
        }
        run = scan_add(run, it[u]);
        if (j == n - 1) {
            p.m[n] = run.m; p.mx[n] = run.mx; p.my[n] = run.my; p.base[n] = run.cnt; p.ent[n] = run.ent;
            counters[0] = run.cnt;
        }
    }
}

// first body j >= b that does NOT share its first `level` digits with body a (bodies [a, b) are known to)
__device__ __forceinline__ int group_end(const unsigned long long* __restrict__ keys, const unsigned long long ka, int b,
                                         const int n, const int level)
{
    const int sh = 2 * (kLevels - level);   // level 0: shift 62 -> every key matches
    const unsigned long long pa = ka >> sh;
    if (b >= n || (keys[b] >> sh) != pa) return b;
    int lo = b, step = 1;                   // keys[lo] matches
    while (lo + step < n && (keys[lo + step] >> sh) == pa) { lo += step; step <<= 1; }
    int hi = lo + step < n ? lo + step : n; // first known mismatch (n = past the end)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((keys[mid] >> sh) == pa) lo = mid; else hi = mid;
    }
    return hi;
}

// One thread per NODE (pre-order slot k): its first body a is the last one with base[a] <= k (binary search over the scan),
// its depth follows from k - base[a], its body range from a gallop over the sorted keys, and the whole 32-byte record is
// written at once.  (Round 2's first version looped per BODY over the chain of nodes that start at it -- up to 31 for a body
// that opens a deep chain, one for most: 134 us at 1 M bodies, against 28 us like this.)
__device__ __forceinline__ void emit_node(const float4* __restrict__ sb, const unsigned long long* __restrict__ keys,
                                          const unsigned* __restrict__ idx, const unsigned* __restrict__ box, const Prefix pre,
                                          const int n, BhNode* __restrict__ out, const int fold, int4* __restrict__ big,
                                          const int big_cap, int* __restrict__ counters, const int root_aside, const int k,
                                          const unsigned char* __restrict__ pmin);
//
// fold (round 3): how an interior node's mass and centre are obtained.
//   0 = exact: fp64 sums over the node's bodies, rounded once (round 2; systems above kFoldFaithfulMax bodies)
//   1 = faithful: the reference's own f32 running fold (nbody.rs:303-320) over the node's bodies in ARRIVAL (index) order --
//       what sequential insertion leaves in every node, bit for bit.  Nodes of at most kFoldSmall bodies are folded right
//       here (selection of the next index among <= kFoldSmall); bigger ones are queued for k_fold_big (one wave per node).
constexpr int kFoldSmall = 8;

__global__ __launch_bounds__(kTile) void k_emit(const float4* __restrict__ sb, const unsigned long long* __restrict__ keys,
                                                const unsigned* __restrict__ idx, const unsigned* __restrict__ box, const Prefix pre,
                                                const int n, const int node_cap, BhNode* __restrict__ out, const int fold,
                                                int4* __restrict__ big, const int big_cap, int* __restrict__ counters,
                                                const int root_aside, const unsigned char* __restrict__ pmin)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = pre.base[n];
    if (k < total && total <= node_cap) {
        emit_node(sb, keys, idx, box, pre, n, out, fold, big, big_cap, counters, root_aside, k, pmin);
    }
}

__device__ __forceinline__ void emit_node(const float4* __restrict__ sb, const unsigned long long* __restrict__ keys,
                                          const unsigned* __restrict__ idx, const unsigned* __restrict__ box, const Prefix pre,
                                          const int n, BhNode* __restrict__ out, const int fold, int4* __restrict__ big,
                                          const int big_cap, int* __restrict__ counters, const int root_aside, const int k,
                                          const unsigned char* __restrict__ pmin)
{
    const int a = pre.owner[k];                     // the body this node starts at: base[a] <= k < base[a] + cnt(a)
    const int first = pre.base[a];
    const int count = pre.base[a + 1] - first;      // > 0: bodies that start no node share base[] with their successor
    const unsigned long long ka = keys[a];
    const int top = a == 0 ? 0 : common_digits(keys[a - 1], ka) + 1;   // depth of the shallowest node starting here
    const int leaf = top + count - 1;
    const int l = top + (k - first);
    const float4 p = sb[a];
    // node size: the path replayed with the reference's f32 midpoints (nbody.rs:289-300), by the key's digits -- the quadrant
    // choices of the body that opened the entity
    float x1 = dec_f32(box[0]), y1 = dec_f32(box[1]), x2 = dec_f32(box[2]), y2 = dec_f32(box[3]);
#pragma unroll 1
    for (int d = 0; d < l; d++) descend_digit(x1, y1, x2, y2, (int)(ka >> (2 * (kLevels - 1 - d))) & 3);
    BhNode o;
    o.s = __fsub_rn(x2, x1);                        // nbody.rs:341
    // interior nodes before this one in pre-order (meaningful for an interior node: where the fast walk files its child group,
    // bh_walk.hip): every node before slot k is interior except the leaves of the entities that start before body a
    o.pad1 = k - pre.ent[a];
    if (l == leaf) {
        // the leaf: this body, or the bodies that share its key (same level-31 cell / EPS-merged pair of entities), folded in
        // ARRIVAL order like the reference's add_mass (nbody.rs:303-320).  Equal keys come out of the stable sort in index order;
        // a merged pair of entities is two such ascending segments back to back: fold them as a two-way merge by index.
        const int b = run_end(keys, a, n);
        int split = b, segments = 1;                // start of the second ascending segment, if any
        for (int j = a + 1; j < b; j++)
            if (idx[j] < idx[j - 1]) {
                if (segments == 1) split = j;
                segments++;
            }
        float px = 0.0f, py = 0.0f, m = 0.0f;
        if (segments <= 2) {
            int u = a, v = split;
            while (u < split || v < b) {
                const bool take_u = v >= b || (u < split && idx[u] < idx[v]);
                const float4 q = sb[take_u ? u : v];
                if (take_u) u++; else v++;
                fold_mass(px, py, m, q.x, q.y, q.w);    // the first one is copied exactly (m == 0 branch)
            }
        } else if (b - a <= kBlobBodies) {
            // a blob of several entities (k_place files the ghosts behind the entity's own run, in slot order): the next
            // smallest index, b - a times
            unsigned last = 0;
            for (int t = 0; t < b - a; t++) {
                unsigned best = 0xFFFFFFFFu;
                int bj = a;
                for (int j = a; j < b; j++) {
                    const unsigned v = idx[j];
                    if ((t == 0 || v > last) && v < best) { best = v; bj = j; }
                }
                const float4 q = sb[bj];
                fold_mass(px, py, m, q.x, q.y, q.w);
                last = best;
            }
        } else {
            refuse(counters, kWhyBigLeaf);          // (the replay admits no blob this big)
        }
        o.px = px; o.py = py; o.m = m;
        o.skip = first + count;
        o.interior = 0; o.q = -1.0f;
        // The reference panics when its depth COUNTER passes 50 (nbody.rs:230-232), and that counter grows by two per level while
        // a leaf is being split down (the re-insert of nbody.rs:278-281 starts one above the node it descends from): a body that
        // ends at level d can have driven it to 2 d.  No leaf deeper than 25 levels -> no panic; deeper ones (two bodies a few
        // 1e-6 of the box apart) are left to the host build, which counts like the reference -- asked for by the bit-exact
        // mode, whose contract includes the panics (the fast mode documents that it has none).
        if ((root_aside & 2) && l > 25) refuse(counters, kWhyDepthPanic);
        // a blob's centres must all have travelled down the path it is filed under as far as this leaf (k_blobs)
        if (pmin && (int)pmin[a] < l) refuse(counters, kWhyCentrePath);
        if (fold == 1 && b - a > 1 && (px != p.x || py != p.y) && keys[a] == path_key(box, p.x, p.y)) {
            // A merged blob travels by its OWN centre in the reference (the split re-inserts (px, py), nbody.rs:271-281).  The
            // replay files a blob under its centre's path; one it never saw -- bodies of ONE level-31 cell without company --
            // sits on the path of its first member.  The same leaf unless the centre left that cell:
            float u1 = dec_f32(box[0]), v1 = dec_f32(box[1]), u2 = dec_f32(box[2]), v2 = dec_f32(box[3]);
#pragma unroll 1
            for (int d = 0; d < l; d++) descend(u1, v1, u2, v2, px, py);
            if (u1 != x1 || v1 != y1 || u2 != x2 || v2 != y2) refuse(counters, kWhyCentrePath);   // counted as "crowded": host build
        }
    } else if (fold == 1) {
        const int b = group_end(keys, ka, a + 1, n, l);
        o.skip = pre.base[b];
        o.interior = 1; o.q = __fmul_rn(o.s, o.s);
        o.px = p.x; o.py = p.y; o.m = 0.0f;
        if (b - a <= kFoldSmall) {
            // the node's bodies in index order: pick the smallest index above the last one, b - a times
            // (requesting all <= 8 indices and records up front was tried: k_emit 18 -> 24 us at 10 000 bodies)
            float px = 0.0f, py = 0.0f, m = 0.0f;
            unsigned last = 0;
            for (int t = 0; t < b - a; t++) {
                unsigned best = 0xFFFFFFFFu;
                int bj = a;
                for (int j = a; j < b; j++) {
                    const unsigned v = idx[j];
                    if ((t == 0 || v > last) && v < best) { best = v; bj = j; }
                }
                const float4 q = sb[bj];
                fold_mass(px, py, m, q.x, q.y, q.w);
                last = best;
            }
            o.px = px; o.py = py; o.m = m;
        } else if (k == 0 && (root_aside & 1)) {
            // the root's fold runs on the side stream (k_fold_root) and writes (px, py, m) of this record itself
            float4* dst = reinterpret_cast<float4*>(&out[0]);
            reinterpret_cast<float*>(dst)[3] = o.s;
            dst[1] = make_float4(__int_as_float(o.skip), __int_as_float(o.interior), o.q, __int_as_float(o.pad1));
            return;
        } else {
            const int slot = atomicAdd(&counters[2], 1);
            if (slot < big_cap) big[slot] = make_int4(k, a, b, 0);
        }
    } else {
        const int b = group_end(keys, ka, a + 1, n, l);
        double m, mx, my;
        if (b - a <= 8) {
            m = 0.0; mx = 0.0; my = 0.0;
            for (int j = a; j < b; j++) {
                const float4 q = sb[j];
                m += (double)q.w; mx += (double)q.w * (double)q.x; my += (double)q.w * (double)q.y;
            }
        } else {
            m = pre.m[b] - pre.m[a]; mx = pre.mx[b] - pre.mx[a]; my = pre.my[b] - pre.my[a];
        }
        if (m != 0.0) { o.px = (float)(mx / m); o.py = (float)(my / m); }
        else          { o.px = p.x; o.py = p.y; }       // massless group: any position, zero contribution
        o.m = (float)m;
        o.skip = pre.base[b];
        o.interior = 1; o.q = __fmul_rn(o.s, o.s);
    }
    float4* dst = reinterpret_cast<float4*>(&out[k]);
    dst[0] = make_float4(o.px, o.py, o.m, o.s);
    dst[1] = make_float4(__int_as_float(o.skip), __int_as_float(o.interior), o.q, __int_as_float(o.pad1));
}

// The reference's running fold for ONE queued node per workgroup of two waves (fold = 1).  The node's bodies are the sorted
// range [a, b); the fold needs them in index order:
//   * the root (b - a == n): every body, in the order of posm itself
//   * up to kFoldRank bodies: every lane ranks its bodies' indices against all others (LDS broadcast) -> ordered list
//   * more: the bodies are marked in an LDS bitmap over a window of 65 536 body indices and the bitmap is walked 2 048 indices
//     at a time, compacting the set bits into an ordered list
// and folds them 64 at a time, as a pipeline of the two waves (one __syncthreads per chunk):
//   wave 0   gathers the chunk's records (two chunks ahead), runs the m chain  m_t = m_(t-1) + mass_t  (serial: f32 addition
//            does not associate), then in parallel  inv_t = 1 / m_t (IEEE), (x m)_t, (y m)_t  -> rec[chunk parity]
//   wave 1   runs the p chain of the PREVIOUS chunk:  p_t = (p_(t-1) * m_(t-1) + (x m)_t) * inv_t , three packed (x, y)
//            operations per member, operands broadcast out of LDS sixteen members ahead of their use
// exactly the operations and the order of add_mass (nbody.rs:315-318); the first member is copied (:305-311).  The root's p
// chain -- n members, ~3 dependent packed operations each -- is the critical path of the whole build; every other node runs
// beside it on its own pair of waves.
typedef float fold_v2 __attribute__((ext_vector_type(2)));
constexpr int kFoldRank = 256;

struct FoldShared {
    unsigned bitmap[2048];        // 65 536 body indices per window          (rank path: the indices being ranked)
    unsigned short lst[2048];     // the set bits of 64 bitmap words, in order (rank path: sorted positions in index order)
    float4 rec[2][64];            // per member of a chunk: m_(t-1), 1 / m_t, x m, y m
    alignas(16) float mass_in[64];
    alignas(16) float mass_run[64];
    float2 first_xy;
    int cnt[2];
};

// p chain over rec[t0 .. cnt): operands fetched kFoldAhead members ahead of the dependent chain; `first` = rec[0 .. kFoldAhead)
// already in registers (read right behind the barrier that published the chunk, together with its size).
// (Per member the wave issues one broadcast ds_read_b128 -- 12 cycles -- beside the three dependent packed operations -- 9.5
//  cycles each: 16.3 ns measured against a 12 ns chain.  Taking the operands out of the lanes with four v_readlane_b32 per member
//  instead, lane t holding member t, was built and is SLOWER: 18.2 ns -- SGPR writes by the VALU do not hide behind the chain.)
constexpr int kFoldAhead = 16;
__device__ __forceinline__ fold_v2 fold_p_chain(const float4* __restrict__ rec, const float4 (&first)[kFoldAhead], const int t0,
                                                const int cnt, fold_v2 pc)
{
    if (t0 == 0 && cnt == 64) {
        float4 r[kFoldAhead], nx[kFoldAhead];
#pragma unroll
        for (int u = 0; u < kFoldAhead; u++) r[u] = first[u];
#pragma unroll
        for (int t = 0; t < 64; t += kFoldAhead) {
            if (t + kFoldAhead < 64) {
#pragma unroll
                for (int u = 0; u < kFoldAhead; u++) nx[u] = rec[t + kFoldAhead + u];
            }
#pragma unroll
            for (int u = 0; u < kFoldAhead; u++) pc = ((pc * fold_v2{r[u].x, r[u].x}) + fold_v2{r[u].z, r[u].w}) * fold_v2{r[u].y, r[u].y};
#pragma unroll
            for (int u = 0; u < kFoldAhead; u++) r[u] = nx[u];
        }
        return pc;
    }
#pragma unroll 4
    for (int t = t0; t < cnt; t++) {
        const float4 r = rec[t];
        pc = ((pc * fold_v2{r.x, r.x}) + fold_v2{r.z, r.w}) * fold_v2{r.y, r.y};
    }
    return pc;
}

// fold of the sorted range [a, b) (the whole workgroup of two waves takes part); o[0..2] = px, py, m
__device__ __forceinline__ void fold_one(FoldShared& sh, const float4* __restrict__ posm, const float4* __restrict__ sb,
                                         const unsigned* __restrict__ idx, const int a, const int b, const int n, float* __restrict__ o)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int size = b - a;
    const int kind = size == n ? 0 : (size <= kFoldRank ? 1 : 2);     // member source: identity / rank / bitmap
    // ---- producer state (wave 0) ----
    float m = 0.0f;
    bool any = false;
    int next_pos = 0;                              // identity / rank path: next member
    int base = 0, g = 0, total = 0, c0 = 0;        // bitmap path: window, next group, members listed, next member
    bool window_ready = false;
    // next chunk of (at most 64) members in index order: every lane's record and the chunk's size (0 = no more)
    auto fetch = [&](float4& r, int& cnt) {
        r = make_float4(0.f, 0.f, 0.f, 0.f);
        cnt = 0;
        if (kind == 0) {
            cnt = n - next_pos < 64 ? n - next_pos : 64;
            if (lane < cnt) r = posm[next_pos + lane];
            next_pos += cnt;
        } else if (kind == 1) {
            cnt = size - next_pos < 64 ? size - next_pos : 64;
            if (lane < cnt) r = sb[a + (int)sh.lst[next_pos + lane]];
            next_pos += cnt;
        } else {
            while (c0 >= total) {                  // the list is used up: next group of 64 bitmap words / next window
                if (!window_ready) {
                    if (base >= n) return;
                    for (int t = lane; t < 2048; t += 64) sh.bitmap[t] = 0u;
                    // (eight index loads in flight per lane)
                    for (int j = a + lane; j < b; j += 64 * 8) {
                        unsigned v[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) v[u] = j + 64 * u < b ? idx[j + 64 * u] - (unsigned)base : 0xFFFFFFFFu;
#pragma unroll
                        for (int u = 0; u < 8; u++)
                            if (v[u] < 65536u) atomicOr(&sh.bitmap[v[u] >> 5], 1u << (v[u] & 31u));
                    }
                    window_ready = true;
                    g = 0;
                }
                // groups of 2 048 indices: only those below n exist, and an empty one costs a ballot, not a prefix sum
                // (k_fold_big at 10 000 bodies: 78 -> 68 us)
                const int groups = n - base >= 65536 ? 32 : (n - base + 2047) >> 11;
                if (g >= groups) { window_ready = false; base += 65536; continue; }
                unsigned word = sh.bitmap[g * 64 + lane];
                if (__ballot(word != 0u) == 0ull) { g++; continue; }
                const int c = __popc(word);
                int incl = c;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int o = __shfl_up(incl, off);
                    if (lane >= off) incl += o;
                }
                total = __shfl(incl, 63);
                c0 = 0;
                int pos = incl - c;
                while (word) {
                    const int bit = __ffs((int)word) - 1;
                    word &= word - 1u;
                    sh.lst[pos++] = (unsigned short)(lane * 32 + bit);
                }
                g++;
            }
            cnt = total - c0 < 64 ? total - c0 : 64;
            if (lane < cnt) r = posm[base + (g - 1) * 2048 + (int)sh.lst[c0 + lane]];
            c0 += 64;
        }
    };
    // ---- consumer state (wave 1) ----
    fold_v2 pc = {0.0f, 0.0f};
    bool started = false;
    float4 first[kFoldAhead];                      // the first records of the chunk to consume next, and its size
    int cnt_c = 0;
#pragma unroll
    for (int u = 0; u < kFoldAhead; u++) first[u] = make_float4(0.f, 0.f, 0.f, 0.f);

    float4 q_cur = make_float4(0.f, 0.f, 0.f, 0.f), q_nxt = q_cur;
    int cnt_cur = 0, cnt_nxt = 0;
    if (wave == 0) {
        if (kind == 1) {
            // rank path: every lane ranks up to four of the node's indices against all of them
            unsigned mine[4];
            int rank[4] = {0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int j = a + u * 64 + lane;
                mine[u] = j < b ? idx[j] : 0xFFFFFFFFu;
                sh.bitmap[u * 64 + lane] = mine[u];
            }
            for (int t = 0; t < size; t++) {
                const unsigned v = sh.bitmap[t];
#pragma unroll
                for (int u = 0; u < 4; u++) rank[u] += v < mine[u] ? 1 : 0;
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (a + u * 64 + lane < b) sh.lst[rank[u]] = (unsigned short)(u * 64 + lane);   // position inside [a, b)
        }
        fetch(q_cur, cnt_cur);
        fetch(q_nxt, cnt_nxt);
    }
    // One round = wave 0 produces chunk i while wave 1 consumes chunk i - 1; the first empty chunk ends the loop.
    for (int i = 0;; i++) {
        const int buf = i & 1;
        int produced = 0;
        if (wave == 0) {
            float4 q2;
            int cnt2;
            fetch(q2, cnt2);                       // chunk i + 2: in flight during this round
            if (cnt_cur > 0) {
                sh.mass_in[lane] = q_cur.w;
                if (cnt_cur == 64) {               // m chain, operands read at once
                    float4 mi[16], mo[16];
                    const float4* in4 = reinterpret_cast<const float4*>(sh.mass_in);
                    float4* out4 = reinterpret_cast<float4*>(sh.mass_run);
#pragma unroll
                    for (int u = 0; u < 16; u++) mi[u] = in4[u];
                    float mr = m;                  // 0 + mass = mass exactly: the copy of the first member (nbody.rs:305-311)
#pragma unroll
                    for (int u = 0; u < 16; u++) {
                        mo[u].x = __fadd_rn(mr, mi[u].x);
                        mo[u].y = __fadd_rn(mo[u].x, mi[u].y);
                        mo[u].z = __fadd_rn(mo[u].y, mi[u].z);
                        mo[u].w = __fadd_rn(mo[u].z, mi[u].w);
                        mr = mo[u].w;
                    }
#pragma unroll
                    for (int u = 0; u < 16; u++) out4[u] = mo[u];
                } else {
                    float mr = m;
                    for (int t = 0; t < cnt_cur; t++) {
                        mr = __fadd_rn(mr, sh.mass_in[t]);
                        sh.mass_run[t] = mr;
                    }
                }
                if (lane < cnt_cur) {
                    const float mt = sh.mass_run[lane];
                    const float mp = lane == 0 ? m : sh.mass_run[lane - 1];
                    sh.rec[buf][lane] = make_float4(mp, 1.0f / mt, __fmul_rn(q_cur.x, q_cur.w), __fmul_rn(q_cur.y, q_cur.w));
                }
                if (!any) {
                    if (lane == 0) sh.first_xy = make_float2(q_cur.x, q_cur.y);
                    any = true;
                }
                m = sh.mass_run[cnt_cur - 1];
            }
            if (lane == 0) sh.cnt[buf] = cnt_cur;
            produced = cnt_cur;
            q_cur = q_nxt; cnt_cur = cnt_nxt;
            q_nxt = q2; cnt_nxt = cnt2;
        } else if (i > 0) {
            const int pb = (i - 1) & 1;
            int t0 = 0;
            if (!started) {                        // nbody.rs:305-311: the first body is copied, not folded
                const float2 f = sh.first_xy;
                pc = fold_v2{f.x, f.y};
                started = true;
                t0 = 1;
            }
            pc = fold_p_chain(sh.rec[pb], first, t0, cnt_c, pc);
        }
        __syncthreads();
        if (wave == 0) {
            if (produced == 0) break;
        } else {
            // the chunk just published: its size and its first records in one LDS round trip, off the next round's chain
            cnt_c = sh.cnt[buf];
#pragma unroll
            for (int u = 0; u < kFoldAhead; u++) first[u] = sh.rec[buf][u];
            if (cnt_c == 0) break;
        }
    }
    if (wave == 1 && lane == 0) { o[0] = pc.x; o[1] = pc.y; }
    if (wave == 0 && lane == 0) o[2] = m;
    __syncthreads();                               // the next node reuses the LDS
}

// The ROOT's fold needs nothing but the bodies in index order -- not the keys, not the sort -- and is the longest chain of
// the build (n members): it runs on a side stream from the very start of the build, beside everything else -- the other
// nodes' folds included -- and writes (px, py, m) of the root's record itself (k_emit leaves those three words alone).
__global__ __launch_bounds__(128) void k_fold_root(const float4* __restrict__ posm, const int n, BhNode* __restrict__ out)
{
    __shared__ FoldShared sh;
    fold_one(sh, posm, nullptr, nullptr, 0, n, n, reinterpret_cast<float*>(&out[0]));
}

__global__ __launch_bounds__(128) void k_fold_big(const float4* __restrict__ posm, const float4* __restrict__ sb,
                                                  const unsigned* __restrict__ idx, const int4* __restrict__ big, const int big_cap,
                                                  const int* __restrict__ counters, const int n, BhNode* __restrict__ out)
{
    __shared__ FoldShared sh;
    int count = counters[2];
    if (count > big_cap) count = big_cap;
    for (int w = blockIdx.x; w < count; w += gridDim.x) {
        const int4 nd = big[w];
        fold_one(sh, posm, sb, idx, nd.y, nd.z, n, reinterpret_cast<float*>(&out[nd.x]));
    }
}

// Workspace header (the first 4 KiB + 256 B): ints [0] node count, [1] bodies the pairs-only EPS merge left behind (or blobs whose
// centre left their first member's cell), [2] nodes queued for k_fold_big, [3] ticket of k_scan_reduce -- all cleared by k_keys at every build -- [9] the "poison" flag of the gated steps (kernels.h), [8] ticket of k_bbox (self-clearing; zeroed once by device_tree_workspace_init),
// [12..15] the root box (encoded); then 256 float4 partial boxes of k_bbox.
constexpr size_t kHeaderBytes = 256 + 256 * sizeof(float4);

constexpr int kSmallFrontMax = 16384;   // up to here: box, keys and sort in two launches (below: "small systems")
// the warm sort of bigger systems (below: "the sort starts from last step's order")
constexpr int kBucketCap = 4096;                 // pairs a bucket can hold (its fixed slots; what one workgroup sorts)
constexpr int kBucketTarget = 640;               // bodies per bucket aimed at
constexpr int kMaxBuckets = 4096;
constexpr int kOversample = 4;                   // splitter candidates per bucket: bucket sizes of a system reshuffled at bucket scale are
                                                 // Erlang-4 around the target, P(size > kBucketCap = 6.4 x target) = 2e-8 per bucket
constexpr int kMaxSamples = kOversample * kMaxBuckets;
constexpr int kIncMaxBodies = kMaxBuckets * kBucketTarget;
constexpr int kWhySortOverflow = 1 << 20;

int inc_buckets(int n)
{
    int b = (n + kBucketTarget - 1) / kBucketTarget;
    return b < 1 ? 1 : (b > kMaxBuckets ? kMaxBuckets : b);
}
bool inc_sort_enabled(int n)
{
    static const int on = [] { const char* v = std::getenv("NBX_INC_SORT"); return v ? std::atoi(v) : 1; }();   // 0: A/B against the library sort
    return on != 0 && n > kSmallFrontMax && n <= kIncMaxBodies;
}

// slots of the grid-cell table: a power of two, at least two per body (one cell per body at most)
static size_t cell_table_slots(int n)
{
    size_t h = 1024;
    while (h < 2 * (size_t)n) h <<= 1;
    return h;
}

size_t device_tree_workspace_bytes(int n, int node_cap, size_t* sort_tmp_bytes)
{
    size_t tmp = 0;
    size_t tmp_small = 0;   // (the workspace serves either shape: which one runs depends on n alone, but n may shrink below the switch)
    (void)rocprim::radix_sort_pairs<BuildSortConfig>(nullptr, tmp, (unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                                     (unsigned*)nullptr, (unsigned*)nullptr, (size_t)n, 0, 2 * kLevels, (hipStream_t)0);
    (void)rocprim::radix_sort_pairs(nullptr, tmp_small, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (unsigned*)nullptr,
                                    (unsigned*)nullptr, (size_t)n, 0, 2 * kLevels, (hipStream_t)0);
    if (tmp_small > tmp) tmp = tmp_small;
    if (sort_tmp_bytes) *sort_tmp_bytes = tmp;
    const size_t nb = ((size_t)n + kScanBlock - 1) / kScanBlock;
    size_t bytes = 0;
    auto add = [&](size_t b) { bytes += (b + 255) & ~(size_t)255; };
    add(kHeaderBytes);                                 // counters, tickets, box, partial boxes
    add(sizeof(unsigned long long) * (size_t)n * 2);   // keys in/out
    add(sizeof(unsigned) * (size_t)n * 2);             // idx in/out
    add(tmp);
    add(sizeof(float4) * (size_t)n);                   // sorted bodies
    add(sizeof(double) * ((size_t)n + 1) * 3);         // prefix sums m, m*x, m*y
    add(sizeof(int) * ((size_t)n + 1));                // pre-order base
    add(sizeof(int) * ((size_t)n + 1));                // entities before every body
    add((size_t)n);                                    // nodes starting at every body (cache between the two scan kernels)
    add(sizeof(int) * (size_t)node_cap);               // owner body of every node slot
    add(sizeof(ScanItem) * (nb + 1));                  // block sums
    add((size_t)n);                                    // EPS-merge links / pmin
    add(sizeof(int4) * (size_t)n);                     // nodes queued for k_fold_big (more than n of them -> host build)
    // reference fold: the EPS blobs (k_cells / k_blobs / k_place)
    add(sizeof(float4) * (size_t)n);                   // bodies in entity order
    add(sizeof(unsigned long long) * (size_t)n);       // entity keys
    add((size_t)n);                                    // pmin in entity order
    add((sizeof(unsigned long long) + sizeof(int)) * cell_table_slots(n));
    add(sizeof(int) * kGhostCap);
    add(sizeof(unsigned long long) * kMaxBuckets);     // warm sort (round 5): splitters
    add(sizeof(int) * kMaxBuckets);                    // ... pairs per bucket
    add(sizeof(unsigned long long) * kMaxSamples);     // ... splitter candidates
    add(sizeof(int) * (kMaxSamples + 64));             // ... their ranks, one ticket per row of the ranking
    const size_t slots_inc = inc_sort_enabled(n) ? (size_t)inc_buckets(n) * kBucketCap : 0;
    add(sizeof(ulonglong2) * slots_inc);               // ... the buckets' slots
    return bytes;
}

// once per (re)allocation of the workspace: the self-clearing ticket of k_bbox starts at zero
hipError_t device_tree_workspace_init(void* workspace, hipStream_t stream)
{
    return hipMemsetAsync(workspace, 0, 256, stream);
}

// ---- small systems: box, keys and sort in two launches ---------------------------------------------------------------------
// Up to kSmallFrontMax bodies (the reference's default scene has 10 000, RustNBodyExperiment.hs:42-47) the build is a chain of
// short kernels and pays for every launch: k_bbox, k_keys and rocPRIM's radix sort (merge-sort path: 5 launches, 36 us at 10 000
// bodies) are 7 of them.  Here:
//   k_front_chunks  one workgroup per chunk of 256 bodies, a body per thread: the root AABB (every workgroup folds ALL positions
//                   itself -- min and max are exact in any order, 160 KB of L2-resident reads cost less than a grid-wide hand-off),
//                   the body's path key, and a bitonic sort of the chunk's (key, index) pairs IN REGISTERS: partners inside a
//                   wave trade through ds_bpermute (33 of the 36 stages), across waves through LDS (3);
//   k_front_rank    one thread per body: its place in the whole order = its place in its chunk + the number of smaller pairs in
//                   every other chunk -- a 9-probe search each, over a copy of all chunk keys in LDS (8 bytes x n <= 128 KB of
//                   gfx950's 160) -- written straight there.
// (key, index) pairs are distinct and the order total: the result is exactly what the stable radix sort of the keys delivers.
// (Tried first: ONE 1024-thread workgroup holding all pairs in LDS, 105 bitonic stages: 0.29 ms per build at 10 000 bodies against
//  0.076 -- every stage moves all 147 KB through one CU's 128 B/clk of LDS; then chunks of 1 024 sorted in LDS and ranked by
//  binary searches in global memory: 0.099 -- 100 dependent L2 round trips per body.)
constexpr int kChunk = kTile;          // bodies per workgroup of k_front_chunks = threads
constexpr unsigned long long kPadKey = ~0ull;   // (real keys occupy 62 bits)

__device__ __forceinline__ bool pair_less(const unsigned long long ka, const unsigned ia, const unsigned long long kb, const unsigned ib)
{
    return ka < kb || (ka == kb && ia < ib);
}

__global__ __launch_bounds__(kTile) void k_front_chunks(const float4* __restrict__ posm, const int n, unsigned* __restrict__ box,
                                                        unsigned long long* __restrict__ keys_out, unsigned* __restrict__ idx_out,
                                                        int* __restrict__ counters, unsigned long long* __restrict__ cell_table,
                                                        const int cell_slots)
{
    __shared__ unsigned long long skey[kChunk];
    __shared__ unsigned sidx[kChunk];
    __shared__ float red[kTile / 64][4];
    const int tid = threadIdx.x;
    // this build's counters and tickets, and the table of occupied grid cells (reference fold), as k_keys clears them
    if (blockIdx.x == 0 && tid < 8) counters[tid] = 0;
    for (int t = blockIdx.x * kTile + tid; t < cell_slots; t += (int)gridDim.x * kTile) cell_table[t] = 0ull;
    // 1. root AABB (nbody.rs:388-398); eight independent loads in flight per thread (16 / 20 / 32: no difference at 2 000 ... 16 384)
    constexpr int kBoxFlight = 8;
    float x1 = 3.40282347e+38f, y1 = 3.40282347e+38f, x2 = -3.40282347e+38f, y2 = -3.40282347e+38f;
    for (int i0 = tid; i0 < n; i0 += kBoxFlight * kTile) {
        float4 q[kBoxFlight];
#pragma unroll
        for (int u = 0; u < kBoxFlight; u++) {
            const int i = i0 + u * kTile;
            q[u] = posm[i < n ? i : i0];
        }
#pragma unroll
        for (int u = 0; u < kBoxFlight; u++) {
            x1 = fminf(x1, q[u].x); y1 = fminf(y1, q[u].y); x2 = fmaxf(x2, q[u].x); y2 = fmaxf(y2, q[u].y);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        x1 = fminf(x1, __shfl_xor(x1, off)); y1 = fminf(y1, __shfl_xor(y1, off));
        x2 = fmaxf(x2, __shfl_xor(x2, off)); y2 = fmaxf(y2, __shfl_xor(y2, off));
    }
    if ((tid & 63) == 0) { red[tid >> 6][0] = x1; red[tid >> 6][1] = y1; red[tid >> 6][2] = x2; red[tid >> 6][3] = y2; }
    __syncthreads();
    x1 = red[0][0]; y1 = red[0][1]; x2 = red[0][2]; y2 = red[0][3];
#pragma unroll
    for (int w = 1; w < kTile / 64; w++) {
        x1 = fminf(x1, red[w][0]); y1 = fminf(y1, red[w][1]); x2 = fmaxf(x2, red[w][2]); y2 = fmaxf(y2, red[w][3]);
    }
    if (blockIdx.x == 0 && tid == 0) { box[0] = enc_f32(x1); box[1] = enc_f32(y1); box[2] = enc_f32(x2); box[3] = enc_f32(y2); }
    // 2. this thread's body: its path of quadrant choices (k_keys; the box goes through the same encode / decode as there)
    const int body = blockIdx.x * kChunk + tid;
    unsigned long long key = kPadKey;
    unsigned id = 0xFFFFFFFFu;
    if (body < n) {
        float ax = dec_f32(enc_f32(x1)), ay = dec_f32(enc_f32(y1)), cx = dec_f32(enc_f32(x2)), cy = dec_f32(enc_f32(y2));
        const float4 p = posm[body];
        key = 0;
#pragma unroll 1
        for (int l = 0; l < kLevels; l++) key = (key << 2) | (unsigned long long)descend(ax, ay, cx, cy, p.x, p.y);
        id = (unsigned)body;
    }
    // 3. bitonic sort of the chunk's 256 pairs, one per thread (padding pairs are larger than every real one)
#pragma unroll 1
    for (int k = 2; k <= kChunk; k <<= 1) {
#pragma unroll 1
        for (int j = k >> 1; j > 0; j >>= 1) {
            unsigned long long ok;
            unsigned oi;
            if (j < 64) {                         // the partner is a lane of this wave
                ok = (unsigned long long)(unsigned)__shfl_xor((int)(unsigned)key, j) |
                     ((unsigned long long)(unsigned)__shfl_xor((int)(unsigned)(key >> 32), j) << 32);
                oi = (unsigned)__shfl_xor((int)id, j);
            } else {
                __syncthreads();
                skey[tid] = key; sidx[tid] = id;
                __syncthreads();
                ok = skey[tid ^ j]; oi = sidx[tid ^ j];
            }
            const bool up = (tid & k) == 0;       // this block of k sorts ascending
            const bool lower = (tid & j) == 0;    // this thread holds the pair's lower position
            const bool mine_less = pair_less(key, id, ok, oi);
            if (mine_less != (up == lower)) { key = ok; id = oi; }
        }
    }
    if (body < n) {                                // padding sorted to the end of the (last) chunk
        keys_out[body] = key;
        idx_out[body] = id;
    }
}

// (threads x copy-in loads in flight: 256 x 16 -> build 0.0580 ms at 10 000 bodies / 0.0679 at 16 384; 128 x 20 0.0614 / 0.0759;
//  128 x 32 0.0567 / 0.0638; 64 x 32 0.0592 / 0.0686: twice the CUs share the probing, each copy-in stays two batches deep)
constexpr int kRankThreads = 128;
__global__ __launch_bounds__(kRankThreads) void k_front_rank(const unsigned long long* __restrict__ ckeys, const unsigned* __restrict__ cidx,
                                                      const int n, unsigned long long* __restrict__ keys_out,
                                                      unsigned* __restrict__ idx_out)
{
    extern __shared__ unsigned long long rkeys[];          // every chunk's sorted keys, padded to whole chunks
    const int chunks = (n + kChunk - 1) / kChunk;
    {   // copy in: two keys per 16-byte load, thirty-two loads in flight per thread (the chunks were written by the kernel before:
        // every first touch goes past the L2, a microsecond or two each)
        constexpr int kFlight = 32;
        const int pairs = chunks * kChunk / 2;
        const ulonglong2* src = reinterpret_cast<const ulonglong2*>(ckeys);   // (the workspace arrays are 256-byte aligned)
        ulonglong2* dst = reinterpret_cast<ulonglong2*>(rkeys);
        for (int t0 = threadIdx.x; t0 < pairs; t0 += kFlight * kRankThreads) {
            ulonglong2 q[kFlight];
#pragma unroll
            for (int u = 0; u < kFlight; u++) {
                const int t = t0 + u * kRankThreads;
                q[u] = make_ulonglong2(kPadKey, kPadKey);
                if (2 * t + 1 < n) q[u] = src[t];
                else if (2 * t < n) q[u].x = ckeys[2 * t];
            }
#pragma unroll
            for (int u = 0; u < kFlight; u++) {
                const int t = t0 + u * kRankThreads;
                if (t < pairs) dst[t] = q[u];
            }
        }
    }
    __syncthreads();
    const int j = blockIdx.x * kRankThreads + threadIdx.x;
    if (j >= n) return;
    const unsigned long long key = rkeys[j];
    const unsigned id = cidx[j];
    const int mine = j / kChunk;
    int pos = j - mine * kChunk;
    // pairs of chunk c smaller than mine: the keys below mine -- lower bound over the chunk's 256 keys, 8 halvings and a last
    // probe, eight chunks abreast, no branch in sight -- plus, among keys EQUAL to mine (bodies of one level-31 cell that landed
    // in different chunks: rare), those with a smaller index
    constexpr int kAbreast = 8;
    for (int c0 = 0; c0 < chunks; c0 += kAbreast) {
        int lo[kAbreast];
#pragma unroll
        for (int u = 0; u < kAbreast; u++) lo[u] = 0;
#pragma unroll
        for (int step = kChunk / 2; step >= 1; step >>= 1) {
#pragma unroll
            for (int u = 0; u < kAbreast; u++) {
                const int c = c0 + u < chunks ? c0 + u : c0;
                lo[u] += rkeys[c * kChunk + lo[u] + step - 1] < key ? step : 0;
            }
        }
#pragma unroll
        for (int u = 0; u < kAbreast; u++) {
            const int c = c0 + u < chunks ? c0 + u : c0;
            const unsigned long long at = rkeys[c * kChunk + lo[u]];
            lo[u] += at < key ? 1 : 0;
            const bool counted = c0 + u < chunks && c != mine;
            pos += counted ? lo[u] : 0;
            if (counted && at == key) {           // (lo[u] did not move: it names the first key equal to mine)
                for (int t = lo[u]; t < kChunk && rkeys[c * kChunk + t] == key; t++) pos += cidx[c * kChunk + t] < id ? 1 : 0;
            }
        }
    }
    keys_out[pos] = key;
    idx_out[pos] = id;
}

bool small_front_enabled(int n)
{
    static const int limit = [] {
        const char* v = std::getenv("NBX_SMALL_FRONT_MAX");   // 0 turns the two-launch front off (A/B against rocPRIM's sort)
        const int x = v ? std::atoi(v) : kSmallFrontMax;
        return x < 0 ? 0 : (x > kSmallFrontMax ? kSmallFrontMax : x);
    }();
    return n <= limit;
}

hipError_t launch_front_small(const float4* posm, int n, unsigned* box, unsigned long long* chunk_keys, unsigned* chunk_idx,
                              unsigned long long* keys_out, unsigned* idx_out, int* counters, unsigned long long* cell_table,
                              int cell_slots, hipStream_t stream)
{
    // more than 64 KB of dynamic LDS is an opt-in per DEVICE (a single-process group drives several): once for each
    static std::atomic<unsigned long long> opted{0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !((opted.load(std::memory_order_acquire) >> dev) & 1ull)) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_front_rank), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * kSmallFrontMax);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) opted.fetch_or(1ull << dev, std::memory_order_release);
    }
    const int chunks = (n + kChunk - 1) / kChunk;
    hipLaunchKernelGGL(k_front_chunks, dim3((unsigned)chunks), dim3(kTile), 0, stream, posm, n, box, chunk_keys, chunk_idx, counters,
                       cell_table, cell_slots);
    hipLaunchKernelGGL(k_front_rank, dim3((unsigned)((n + kRankThreads - 1) / kRankThreads)), dim3(kRankThreads),
                       (size_t)8 * (size_t)chunks * kChunk, stream, chunk_keys, chunk_idx, n, keys_out, idx_out);
    return hipGetLastError();
}

// ---- big systems: the sort starts from last step's order (round 5) ----------------------------------------------------------
// Above kSmallFrontMax bodies rounds 2-4 handed (key, index) to rocPRIM every step: block sort + 8 merge passes, 17 launches,
// 178 us of a 0.79 ms step at 1 M bodies -- a sort FROM SCRATCH of a sequence that, in last step's order, is sorted but for one
// step's motion.  Round 5, when the previous build left its order in idx1 ("warm"), a sample sort with last step's order as the
// sampling frame, four launches:
//   k_sample_keys   S = 4 B candidates (B = n / 640 buckets): the NEW keys of the bodies that stood at the S-quantiles of last
//                   step's order.  They have moved with everybody else, so they still sample the distribution evenly; four per
//                   bucket keep the bucket sizes within reach of their slots even when a step reshuffles the system at bucket
//                   scale (a collapsing core: sizes are then Erlang-4 around the target, not all equal).
//   k_sample_rank   every candidate's rank among the S by counting (S^2 = 4e7 compares over the whole chip, 256 x 256 per
//                   workgroup out of LDS); the last workgroup to finish files every 4th of them as a splitter.
//   k_keys_scatter  walks the bodies in LAST step's order (coalesced index loads, positions gathered): path key as k_keys (four
//                   bodies per thread, their descents interleaved), bucket = number of splitters <= key (branch-free search over
//                   an LDS copy), place inside the bucket from one LDS counter per bucket and ONE global atomic per (workgroup,
//                   touched bucket) -- neighbours in the old order mostly share a bucket -- into fixed slots of kBucketCap pairs.
//   k_bucket_sort   one workgroup per bucket.  A bucket's keys are ~640 neighbours on the Z-curve: between its smallest and
//                   largest key they lie about evenly, so (key - min) >> shift spreads them over 2 048 sub-buckets (0.3 pairs
//                   each); LDS counters + one scan give every pair its sub-bucket's start, and its place inside = the number of
//                   smaller (key, index) pairs among the sub-bucket's members (1-3 LDS reads).  Written to start(bucket) + place,
//                   start = sum of the counts before (every workgroup sums the <= 4 096 counts itself).  Buckets whose keys clump
//                   (sum of squared sub-bucket counts > 48 per pair: EPS clusters, identical positions) take a bitonic network over
//                   (key, index) instead -- pairs blocked over the threads, partners in the same thread / the same wave (lane
//                   exchange) / other waves (3 stages through LDS).  (tools/ubench_bucket_sort.hip: 21-26 us against 44-89 for the
//                   network alone at 1 M pairs; one wave per bucket with DPP / swizzle exchanges: 70-176.)
// (key, index) pairs are distinct and the order total: the output is exactly the stable radix sort's, so the tree is
// bit-identical.  A bucket that outgrows its slots is REFUSED like an exhausted node pool -- counters[1], kWhySortOverflow -- the
// step is redone on the host tree and the next build sorts from scratch.  Cold builds (first step, new particles, after a
// refusal) and systems above kIncMaxBodies keep the library sort.
__device__ __forceinline__ unsigned long long body_key(const unsigned* __restrict__ box, const float4 p)
{
    float x1 = dec_f32(box[0]), y1 = dec_f32(box[1]), x2 = dec_f32(box[2]), y2 = dec_f32(box[3]);
    unsigned long long key = 0;
#pragma unroll 1
    for (int l = 0; l < kLevels; l++) key = (key << 2) | (unsigned long long)descend(x1, y1, x2, y2, p.x, p.y);
    return key;
}

// where splitter candidate i stands in last step's order: the middle of the i-th of S equal stretches
__device__ __forceinline__ int sample_pos(const int i, const int n, const int samples)
{
    return (int)(((2ll * i + 1ll) * (long long)n) / (2ll * samples));
}

// grid = sb x sb workgroups (sb = blocks of 256 candidates): workgroup (a, c) counts, for each candidate of block a, the
// candidates of block c that precede it ((key, candidate number) order: the ranks are a permutation of 0 .. S-1).  Every workgroup
// computes the keys of both its blocks itself (two descents per thread: cheaper than a launch of its own in front); the c = 0
// column leaves block a's keys for k_keys_scatter.  No hand-off inside the kernel: a device-wide fence writes the L2 back on
// this chip (k_sample_rank with a ticket and a last workgroup took 48 us, 40 of them fences).
__global__ __launch_bounds__(kTile) void k_sample_rank(const float4* __restrict__ posm, const int n, const unsigned* __restrict__ box,
                                                       const unsigned* __restrict__ perm, const int samples,
                                                       unsigned long long* __restrict__ skeys, int* __restrict__ srank)
{
    __shared__ __attribute__((aligned(16))) unsigned long long other[kTile];
    const int sb = (samples + kTile - 1) / kTile;
    const int a = blockIdx.x / sb, c = blockIdx.x - a * sb;
    const int tid = threadIdx.x;
    const int oc = c * kTile + tid, mine_i = a * kTile + tid;
    unsigned long long ok = kPadKey, mine = kPadKey;
    if (a == c) {
        if (mine_i < samples) mine = body_key(box, posm[perm[sample_pos(mine_i, n, samples)]]);
        ok = mine;
    } else {
        // two descents, interleaved
        const float4 p0 = posm[perm[sample_pos(mine_i < samples ? mine_i : 0, n, samples)]];
        const float4 p1 = posm[perm[sample_pos(oc < samples ? oc : 0, n, samples)]];
        float ax1 = dec_f32(box[0]), ay1 = dec_f32(box[1]), ax2 = dec_f32(box[2]), ay2 = dec_f32(box[3]);
        float bx1 = ax1, by1 = ay1, bx2 = ax2, by2 = ay2;
        unsigned long long k0 = 0, k1 = 0;
#pragma unroll 1
        for (int l = 0; l < kLevels; l++) {
            k0 = (k0 << 2) | (unsigned long long)descend(ax1, ay1, ax2, ay2, p0.x, p0.y);
            k1 = (k1 << 2) | (unsigned long long)descend(bx1, by1, bx2, by2, p1.x, p1.y);
        }
        if (mine_i < samples) mine = k0;
        if (oc < samples) ok = k1;
    }
    other[tid] = ok;
    if (c == 0 && mine_i < samples) skeys[mine_i] = mine;
    __syncthreads();
    int before = 0;
    if (a != c) {
        // another block: every one of its candidates has a smaller (c < a) or larger number than mine -> ties go one way
        const ulonglong2* o2 = reinterpret_cast<const ulonglong2*>(other);   // two keys per LDS read (every lane reads the same address)
        if (c < a) {
#pragma unroll 8
            for (int t = 0; t < kTile / 2; t++) { const ulonglong2 q = o2[t]; before += (q.x <= mine ? 1 : 0) + (q.y <= mine ? 1 : 0); }
        } else {
#pragma unroll 8
            for (int t = 0; t < kTile / 2; t++) { const ulonglong2 q = o2[t]; before += (q.x < mine ? 1 : 0) + (q.y < mine ? 1 : 0); }
        }
    } else {
#pragma unroll 8
        for (int t = 0; t < kTile; t++) before += (other[t] < mine || (other[t] == mine && t < tid)) ? 1 : 0;
    }
    // (padding candidates beyond S carry the largest key and the largest numbers: they precede no real candidate)
    if (mine_i < samples && before) atomicAdd(&srank[mine_i], before);
}

// bucket of a key: the number of splitters <= key (0 .. ns), branch-free over the LDS copy; P2 = power of two > ns
__device__ __forceinline__ int bucket_of(const unsigned long long* __restrict__ s, const int ns, const int P2, const unsigned long long key)
{
    int lo = 0;
    for (int step = P2 >> 1; step > 0; step >>= 1) {
        const int t = lo + step;
        lo = (t <= ns && s[t - 1] <= key) ? t : lo;
    }
    return lo;
}

template <int EA>
__global__ __launch_bounds__(kTile) void k_keys_scatter(const float4* __restrict__ posm, const int n, const unsigned* __restrict__ box,
                                                        const unsigned* __restrict__ perm, const unsigned long long* __restrict__ skeys,
                                                        const int* __restrict__ srank, const int samples,
                                                        const int buckets, int* __restrict__ gcount,
                                                        ulonglong2* __restrict__ slots,
                                                        unsigned long long* __restrict__ cell_table, const int cell_slots)
{
    extern __shared__ unsigned long long sm[];
    const int ns = buckets - 1;
    int P2 = 1;
    while (P2 <= ns) P2 <<= 1;
    unsigned long long* s = sm;                                      // splitters [ns]
    int* hist = reinterpret_cast<int*>(sm + (ns > 0 ? ns : 1));      // [buckets]: pairs of this workgroup per bucket, then their first slot
    const int tid = threadIdx.x;
    for (int t = blockIdx.x * kTile + tid; t < cell_slots; t += (int)gridDim.x * kTile) cell_table[t] = 0ull;   // (reference fold, as k_keys)
    // the bodies first (their loads fly while the splitters come in): EA per thread, their descents interleaved
    unsigned id[EA];
    float px[EA], py[EA];
    const int t0 = blockIdx.x * (kTile * EA) + tid;
#pragma unroll
    for (int r = 0; r < EA; r++) {
        const int t = t0 + r * kTile;
        id[r] = perm[t < n ? t : n - 1];
    }
#pragma unroll
    for (int r = 0; r < EA; r++) { const float4 p = posm[id[r]]; px[r] = p.x; py[r] = p.y; }
    {   // the splitters: of the S ranked candidates, those of rank q * (S / B) - 1, q = 1 .. B - 1, ascending
        const int per = samples / buckets;     // = kOversample (samples = kOversample * buckets)
        constexpr int kFlight = 8;             // ranks in flight per thread (one at a time: 26 dependent round trips, 11 us of this kernel)
        for (int i0 = tid; i0 < samples; i0 += kFlight * kTile) {
            int r[kFlight];
#pragma unroll
            for (int u = 0; u < kFlight; u++) { const int i = i0 + u * kTile; r[u] = i < samples ? srank[i] + 1 : 1; }
            unsigned long long key[kFlight];
#pragma unroll
            for (int u = 0; u < kFlight; u++) {
                const int i = i0 + u * kTile;
                key[u] = skeys[i < samples ? i : 0];       // (unconditional: the load does not wait for the rank)
            }
#pragma unroll
            for (int u = 0; u < kFlight; u++)
                if (i0 + u * kTile < samples && r[u] % per == 0 && r[u] / per < buckets) s[r[u] / per - 1] = key[u];
        }
    }
    for (int b = tid; b < buckets; b += kTile) hist[b] = 0;
    unsigned long long key[EA];
    {
        const float bx1 = dec_f32(box[0]), by1 = dec_f32(box[1]), bx2 = dec_f32(box[2]), by2 = dec_f32(box[3]);
        float x1[EA], y1[EA], x2[EA], y2[EA];
#pragma unroll
        for (int r = 0; r < EA; r++) { x1[r] = bx1; y1[r] = by1; x2[r] = bx2; y2[r] = by2; key[r] = 0; }
#pragma unroll 1
        for (int l = 0; l < kLevels; l++) {
#pragma unroll
            for (int r = 0; r < EA; r++) key[r] = (key[r] << 2) | (unsigned long long)descend(x1[r], y1[r], x2[r], y2[r], px[r], py[r]);
        }
    }
    __syncthreads();
    int bk[EA], off[EA];
#pragma unroll
    for (int r = 0; r < EA; r++) {
        bk[r] = -1;
        if (t0 + r * kTile < n) {
            bk[r] = bucket_of(s, ns, P2, key[r]);
            off[r] = atomicAdd(&hist[bk[r]], 1);
        }
    }
    __syncthreads();
    for (int b = tid; b < buckets; b += kTile) {
        const int c = hist[b];
        if (c > 0) hist[b] = atomicAdd(&gcount[b], c);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < EA; r++) {
        if (bk[r] < 0) continue;
        const int slot = hist[bk[r]] + off[r];
        if (slot < kBucketCap) {       // (a pair beyond the bucket's slots is dropped: gcount says so, k_bucket_sort refuses the build)
            slots[(size_t)bk[r] * kBucketCap + (size_t)slot] = make_ulonglong2(key[r], (unsigned long long)id[r]);   // one 16-byte store
        }
    }
}

__device__ __forceinline__ unsigned long long shfl_xor_u64(const unsigned long long v, const int mask)
{
    return (unsigned long long)(unsigned)__shfl_xor((int)(unsigned)v, mask) |
           ((unsigned long long)(unsigned)__shfl_xor((int)(unsigned)(v >> 32), mask) << 32);
}

// the network: bitonic sort of P = 256 * E pairs by one workgroup; pair e lives in thread e / E, register e % E
template <int E>
__device__ __forceinline__ void bucket_network(const ulonglong2* __restrict__ ps, const int cnt,
                                               unsigned long long* __restrict__ keys_out, unsigned* __restrict__ idx_out,
                                               const int out_base, unsigned* __restrict__ lds)
{
    constexpr int P = kTile * E;
    const int tid = threadIdx.x;
    unsigned long long k[E];
    unsigned id[E];
#pragma unroll
    for (int r = 0; r < E; r++) {
        const int e = tid * E + r;
        k[r] = kPadKey; id[r] = 0xFFFFFFFFu;
        if (e < cnt) { const ulonglong2 q = ps[e]; k[r] = q.x; id[r] = (unsigned)q.y; }
    }
#pragma unroll 1
    for (int kk = 2; kk <= P; kk <<= 1) {
        // partners in other threads: j = kk/2 ... E (runtime j, the register index r stays a constant)
#pragma unroll 1
        for (int j = kk >> 1; j >= E; j >>= 1) {
            const int tj = j / E;                       // partner thread = tid ^ tj
            const bool up = ((tid * E) & kk) == 0;      // (kk > j >= E: bit kk of e = tid*E + r does not depend on r)
            const bool keep_min = up == ((tid & tj) == 0);
            if (tj < 64) {
#pragma unroll
                for (int r = 0; r < E; r++) {
                    const unsigned long long ok = shfl_xor_u64(k[r], tj);
                    const unsigned oi = (unsigned)__shfl_xor((int)id[r], tj);
                    const bool mine_less = pair_less(k[r], id[r], ok, oi);
                    if (mine_less != keep_min) { k[r] = ok; id[r] = oi; }
                }
            } else {
                __syncthreads();
#pragma unroll
                for (int r = 0; r < E; r++) {
                    lds[r * kTile + tid] = (unsigned)k[r];
                    lds[P + r * kTile + tid] = (unsigned)(k[r] >> 32);
                    lds[2 * P + r * kTile + tid] = id[r];
                }
                __syncthreads();
                const int pt = tid ^ tj;
#pragma unroll
                for (int r = 0; r < E; r++) {
                    const unsigned long long ok = (unsigned long long)lds[r * kTile + pt] | ((unsigned long long)lds[P + r * kTile + pt] << 32);
                    const unsigned oi = lds[2 * P + r * kTile + pt];
                    const bool mine_less = pair_less(k[r], id[r], ok, oi);
                    if (mine_less != keep_min) { k[r] = ok; id[r] = oi; }
                }
            }
        }
        // partners in this thread's registers: j = min(kk/2, E/2) ... 1 (compile-time j)
#pragma unroll
        for (int j = E / 2; j > 0; j >>= 1) {
            if (j > (kk >> 1)) continue;
#pragma unroll
            for (int a = 0; a < E; a++) {
                if (a & j) continue;
                const int b = a | j;
                const bool up = ((tid * E + a) & kk) == 0;
                const bool b_less = pair_less(k[b], id[b], k[a], id[a]);
                if (b_less == up) {
                    const unsigned long long tk = k[a]; k[a] = k[b]; k[b] = tk;
                    const unsigned ti = id[a]; id[a] = id[b]; id[b] = ti;
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < E; r++) {
        const int e = tid * E + r;
        if (e < cnt) { keys_out[out_base + e] = k[r]; idx_out[out_base + e] = id[r]; }
    }
}

// the common case: sub-buckets by interpolation, place inside a sub-bucket by counting.  false: the keys clump, nothing was written
// (and start() has not been called).  start() -- collective, called once -- returns where the bucket's pairs go in the output.
constexpr int kSub = 2048;          // sub-buckets per bucket
constexpr int kClumpPerPair = 48;   // sum of squared sub-bucket counts per pair beyond which the network is cheaper
template <int E, class StartFn>
__device__ __forceinline__ bool bucket_by_counting(const ulonglong2* __restrict__ ps, const int cnt,
                                                   unsigned long long* __restrict__ keys_out, unsigned* __restrict__ idx_out,
                                                   unsigned* __restrict__ lds, StartFn start)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int* hist = reinterpret_cast<int*>(lds);                                              // [kSub + 1]
    unsigned long long* skey = reinterpret_cast<unsigned long long*>(lds + kSub + 2);     // [kBucketCap]
    unsigned* sidx = reinterpret_cast<unsigned*>(skey + kBucketCap);                      // [kBucketCap]
    __shared__ unsigned long long red[2][kTile / 64];
    __shared__ int wsum[kTile / 64];
    __shared__ unsigned long long wsq[kTile / 64];
    unsigned long long k[E];
    unsigned id[E];
    unsigned long long mn = kPadKey, mx = 0;
#pragma unroll
    for (int r = 0; r < E; r++) {
        const int e = r * kTile + tid;
        k[r] = kPadKey; id[r] = 0xFFFFFFFFu;
        if (e < cnt) { const ulonglong2 q = ps[e]; k[r] = q.x; id[r] = (unsigned)q.y; mn = k[r] < mn ? k[r] : mn; mx = k[r] > mx ? k[r] : mx; }
    }
    for (int t = tid; t <= kSub; t += kTile) hist[t] = 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long a = shfl_xor_u64(mn, o), b = shfl_xor_u64(mx, o);
        mn = a < mn ? a : mn; mx = b > mx ? b : mx;
    }
    if (lane == 0) { red[0][wave] = mn; red[1][wave] = mx; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < kTile / 64; w++) { mn = red[0][w] < mn ? red[0][w] : mn; mx = red[1][w] > mx ? red[1][w] : mx; }
    const unsigned long long width = mx - mn;                        // sub-bucket = (key - mn) >> shift, 0 .. kSub - 1
    int shift = 0;
    if (width >= (unsigned long long)kSub) shift = 64 - __clzll((long long)width) - 11;   // bit length of width - log2(kSub)
    int dg[E], off[E];
#pragma unroll
    for (int r = 0; r < E; r++) {
        dg[r] = -1;
        if (r * kTile + tid < cnt) { dg[r] = (int)((k[r] - mn) >> shift); off[r] = atomicAdd(&hist[dg[r]], 1); }
    }
    __syncthreads();
    {   // exclusive scan of the counters (8 per thread); the sum of their squares on the way
        constexpr int kPer = kSub / kTile;
        int c[kPer], sum = 0;
        unsigned long long sq = 0;
#pragma unroll
        for (int u = 0; u < kPer; u++) { c[u] = hist[tid * kPer + u]; sum += c[u]; sq += (unsigned long long)c[u] * (unsigned long long)c[u]; }
        int incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sq += shfl_xor_u64(sq, o);
        if (lane == 63) wsum[wave] = incl;
        if (lane == 0) wsq[wave] = sq;
        __syncthreads();
        int before = incl - sum;
        for (int w = 0; w < wave; w++) before += wsum[w];
        sq = 0;
#pragma unroll
        for (int w = 0; w < kTile / 64; w++) sq += wsq[w];
        if (sq > (unsigned long long)kClumpPerPair * (unsigned long long)cnt) return false;   // (uniform: every thread has the same sum)
#pragma unroll
        for (int u = 0; u < kPer; u++) { hist[tid * kPer + u] = before; before += c[u]; }
        if (tid == kTile - 1) hist[kSub] = before;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < E; r++)
        if (dg[r] >= 0) { const int p = hist[dg[r]] + off[r]; skey[p] = k[r]; sidx[p] = id[r]; }
    const int out_base = start();                                    // (ends on a barrier: the sub-buckets are complete behind it)
    int pos[E];
#pragma unroll
    for (int r = 0; r < E; r++) {
        pos[r] = -1;
        if (dg[r] < 0) continue;
        const int s0 = hist[dg[r]], s1 = hist[dg[r] + 1];
        int p = s0;
        for (int t = s0; t < s1; t++) p += pair_less(skey[t], sidx[t], k[r], id[r]) ? 1 : 0;
        pos[r] = p;
    }
    __syncthreads();                                                 // every place is known: the staging arrays become the sorted bucket
#pragma unroll
    for (int r = 0; r < E; r++)
        if (pos[r] >= 0) { skey[pos[r]] = k[r]; sidx[pos[r]] = id[r]; }
    __syncthreads();
    for (int t = tid; t < cnt; t += kTile) { keys_out[out_base + t] = skey[t]; idx_out[out_base + t] = sidx[t]; }   // coalesced
    return true;
}

constexpr size_t kBucketSortLds = sizeof(unsigned) * (kSub + 2) + 12 * (size_t)kBucketCap;
__global__ __launch_bounds__(kTile) void k_bucket_sort(const ulonglong2* __restrict__ slots, const int* __restrict__ gcount, const int buckets,
                                                       const int n, unsigned long long* __restrict__ keys_out, unsigned* __restrict__ idx_out,
                                                       int* __restrict__ counters)
{
    extern __shared__ unsigned lds_sort[];
    __shared__ int red[2][kTile / 64];
    const int tid = threadIdx.x, b = blockIdx.x;
    // pairs before this bucket = the sum of the counts before it, each clamped to what its slots hold; the sum over ALL buckets says
    // whether any bucket overflowed.  Collective; its loads fly beside the LDS work of the caller.
    auto start = [&]() -> int {
        int before = 0, total = 0;
        for (int j = tid; j < buckets; j += kTile) {
            int c = gcount[j];
            c = c > kBucketCap ? kBucketCap : c;
            total += c;
            before += j < b ? c : 0;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { before += __shfl_xor(before, o); total += __shfl_xor(total, o); }
        if ((tid & 63) == 0) { red[0][tid >> 6] = before; red[1][tid >> 6] = total; }
        __syncthreads();
        before = 0; total = 0;
#pragma unroll
        for (int w = 0; w < kTile / 64; w++) { before += red[0][w]; total += red[1][w]; }
        if (total != n && b == buckets - 1) {
            // some bucket outgrew its slots: refuse the build (gate / device_tree_build_end read counters[1]) and leave a well-formed
            // tail -- largest key, a valid index -- so that the kernels behind this one stay inside their arrays
            if (tid == 0) { atomicAdd(&counters[1], 0x20000000); atomicOr(&counters[5], kWhySortOverflow); }
            for (int t = total + tid; t < n; t += kTile) { keys_out[t] = (1ull << (2 * kLevels)) - 1ull; idx_out[t] = 0u; }
        }
        return before;
    };
    int cnt = gcount[b];
    cnt = cnt > kBucketCap ? kBucketCap : cnt;
    if (cnt == 0) { (void)start(); return; }
    const ulonglong2* ps = slots + (size_t)b * kBucketCap;
    bool done;
    if (cnt <= kTile) done = bucket_by_counting<1>(ps, cnt, keys_out, idx_out, lds_sort, start);
    else if (cnt <= 2 * kTile) done = bucket_by_counting<2>(ps, cnt, keys_out, idx_out, lds_sort, start);
    else if (cnt <= 4 * kTile) done = bucket_by_counting<4>(ps, cnt, keys_out, idx_out, lds_sort, start);
    else if (cnt <= 8 * kTile) done = bucket_by_counting<8>(ps, cnt, keys_out, idx_out, lds_sort, start);
    else done = bucket_by_counting<16>(ps, cnt, keys_out, idx_out, lds_sort, start);
    if (done) return;
    __syncthreads();
    const int before = start();
    __syncthreads();
    if (cnt <= kTile) bucket_network<1>(ps, cnt, keys_out, idx_out, before, lds_sort);
    else if (cnt <= 2 * kTile) bucket_network<2>(ps, cnt, keys_out, idx_out, before, lds_sort);
    else if (cnt <= 4 * kTile) bucket_network<4>(ps, cnt, keys_out, idx_out, before, lds_sort);
    else if (cnt <= 8 * kTile) bucket_network<8>(ps, cnt, keys_out, idx_out, before, lds_sort);
    else bucket_network<16>(ps, cnt, keys_out, idx_out, before, lds_sort);
}

// the sort of a warm build: bodies in last step's order (perm) -> sorted (key, body) pairs in keys_out / idx_out (perm == idx_out is fine:
// it is read by the first three kernels and written by the last)
hipError_t launch_inc_sort(const float4* posm, int n, const unsigned* box, const unsigned* perm, unsigned long long* spl, int* gcount,
                           unsigned long long* skeys, int* srank, ulonglong2* slots, unsigned long long* keys_out,
                           unsigned* idx_out, int* counters, unsigned long long* cell_table, int cell_slots, hipStream_t stream)
{
    static_assert(kBucketSortLds <= 64 * 1024, "k_bucket_sort's LDS must stay within the default limit (no per-device opt-in)");
    const int buckets = inc_buckets(n);
    const int samples = kOversample * buckets;
    const int sb = (samples + kTile - 1) / kTile;
    hipLaunchKernelGGL(k_sample_rank, dim3((unsigned)(sb * sb)), dim3(kTile), 0, stream, posm, n, box, perm, samples, skeys, srank);
    const size_t shm = sizeof(unsigned long long) * (size_t)(buckets > 1 ? buckets - 1 : 1) + sizeof(int) * (size_t)buckets;
    if (n >= 262144)
        hipLaunchKernelGGL(k_keys_scatter<4>, dim3((unsigned)((n + 4 * kTile - 1) / (4 * kTile))), dim3(kTile), shm, stream, posm, n, box, perm,
                           skeys, srank, samples, buckets, gcount, slots, cell_table, cell_slots);
    else
        hipLaunchKernelGGL(k_keys_scatter<1>, dim3((unsigned)((n + kTile - 1) / kTile)), dim3(kTile), shm, stream, posm, n, box, perm, skeys,
                           srank, samples, buckets, gcount, slots, cell_table, cell_slots);
    hipLaunchKernelGGL(k_bucket_sort, dim3((unsigned)buckets), dim3(kTile), kBucketSortLds, stream, slots, gcount, buckets, n,
                       keys_out, idx_out, counters);
    return hipGetLastError();
}

namespace {
struct Workspace {
    int* counters;
    unsigned* box;
    float4* part;
    unsigned long long *keys0, *keys1;
    unsigned *idx0, *idx1;
    void* sort_tmp;
    float4* sb;
    Prefix pre;
    ScanItem* block_sums;
    unsigned char* link;
    int4* big;
    float4* sb2;
    unsigned long long* ekey;
    unsigned char* pmin2;
    unsigned long long* hk;
    int* hv;
    unsigned hmask;
    int* ghosts;
    unsigned long long* spl;      // round 5, warm sort: splitters, per-bucket counts, the buckets' slots
    int* gcount;
    ulonglong2* slots;            // kBucketCap (key, index) slots per bucket
    unsigned long long* skeys;
    int* srank;                   // [kMaxSamples] ranks of the splitter candidates
};
Workspace carve(void* workspace, int n, size_t sort_tmp, int node_cap)
{
    char* w = static_cast<char*>(workspace);
    auto take = [&](size_t b) { char* p = w; w += (b + 255) & ~(size_t)255; return p; };
    const size_t nb = ((size_t)n + kScanBlock - 1) / kScanBlock;
    Workspace k;
    char* header = take(kHeaderBytes);
    k.counters = reinterpret_cast<int*>(header);
    k.box = reinterpret_cast<unsigned*>(k.counters + 12);
    k.part = reinterpret_cast<float4*>(header + 256);
    k.keys0 = reinterpret_cast<unsigned long long*>(take(sizeof(unsigned long long) * (size_t)n * 2));
    k.keys1 = k.keys0 + n;
    k.idx0 = reinterpret_cast<unsigned*>(take(sizeof(unsigned) * (size_t)n * 2));
    k.idx1 = k.idx0 + n;
    k.sort_tmp = take(sort_tmp);
    k.sb = reinterpret_cast<float4*>(take(sizeof(float4) * (size_t)n));
    double* d = reinterpret_cast<double*>(take(sizeof(double) * ((size_t)n + 1) * 3));
    k.pre.m = d; k.pre.mx = d + (size_t)n + 1; k.pre.my = d + 2 * ((size_t)n + 1);
    k.pre.base = reinterpret_cast<int*>(take(sizeof(int) * ((size_t)n + 1)));
    k.pre.ent = reinterpret_cast<int*>(take(sizeof(int) * ((size_t)n + 1)));
    k.pre.cnt = reinterpret_cast<unsigned char*>(take((size_t)n));
    k.pre.owner = reinterpret_cast<int*>(take(sizeof(int) * (size_t)node_cap));
    k.pre.owner_cap = node_cap;
    k.block_sums = reinterpret_cast<ScanItem*>(take(sizeof(ScanItem) * (nb + 1)));
    k.link = reinterpret_cast<unsigned char*>(take((size_t)n));
    k.big = reinterpret_cast<int4*>(take(sizeof(int4) * (size_t)n));
    k.sb2 = reinterpret_cast<float4*>(take(sizeof(float4) * (size_t)n));
    k.ekey = reinterpret_cast<unsigned long long*>(take(sizeof(unsigned long long) * (size_t)n));
    k.pmin2 = reinterpret_cast<unsigned char*>(take((size_t)n));
    const size_t slots = cell_table_slots(n);
    char* table = take((sizeof(unsigned long long) + sizeof(int)) * slots);
    k.hk = reinterpret_cast<unsigned long long*>(table);
    k.hv = reinterpret_cast<int*>(table + sizeof(unsigned long long) * slots);
    k.hmask = (unsigned)(slots - 1);
    k.ghosts = reinterpret_cast<int*>(take(sizeof(int) * kGhostCap));
    k.spl = reinterpret_cast<unsigned long long*>(take(sizeof(unsigned long long) * kMaxBuckets));
    k.gcount = reinterpret_cast<int*>(take(sizeof(int) * kMaxBuckets));
    k.skeys = reinterpret_cast<unsigned long long*>(take(sizeof(unsigned long long) * kMaxSamples));
    k.srank = reinterpret_cast<int*>(take(sizeof(int) * (kMaxSamples + 64)));
    const size_t slots_inc = inc_sort_enabled(n) ? (size_t)inc_buckets(n) * kBucketCap : 0;
    k.slots = reinterpret_cast<ulonglong2*>(take(sizeof(ulonglong2) * slots_inc));
    return k;
}

// root AABB -> path keys -> sorted (key, body) pairs in keys1 / idx1
hipError_t sort_bodies(const float4* posm, int n, const Workspace& k, size_t sort_tmp, hipStream_t stream, bool cell_table, bool warm)
{
    if (small_front_enabled(n)) {   // a small system (the reference's own 10 000 bodies): two launches instead of seven, no library sort
        const hipError_t e = launch_front_small(posm, n, k.box, k.keys0, k.idx0, k.keys1, k.idx1, k.counters, k.hk,
                                                cell_table ? (int)(k.hmask + 1u) : 0, stream);
        if (e == hipSuccess) return e;
        (void)hipGetLastError();    // the LDS opt-in or a launch was refused (nothing ran): the general path below serves any size
    }
    const int nb = (n + kTile - 1) / kTile;
    if (warm && inc_sort_enabled(n)) {   // idx1 holds last step's order: sort from there (round 5)
        hipLaunchKernelGGL(k_bbox, dim3(nb < 256 ? nb : 256), dim3(kTile), 0, stream, posm, n, k.box, k.part, k.counters + 8, k.srank,
                           kOversample * inc_buckets(n), k.gcount, inc_buckets(n), k.counters, 8);
        return launch_inc_sort(posm, n, k.box, k.idx1, k.spl, k.gcount, k.skeys, k.srank, k.slots, k.keys1, k.idx1, k.counters, k.hk,
                               cell_table ? (int)(k.hmask + 1u) : 0, stream);
    }
    hipLaunchKernelGGL(k_bbox, dim3(nb < 256 ? nb : 256), dim3(kTile), 0, stream, posm, n, k.box, k.part, k.counters + 8, (int*)nullptr, 0,
                       (int*)nullptr, 0, (int*)nullptr, 0);
    hipLaunchKernelGGL(k_keys, dim3(nb), dim3(kTile), 0, stream, posm, n, k.box, k.keys0, k.idx0, k.counters, k.hk,
                       cell_table ? (int)(k.hmask + 1u) : 0);
    if (n >= kBigSortFrom)
        return rocprim::radix_sort_pairs<BuildSortConfig>(k.sort_tmp, sort_tmp, k.keys0, k.keys1, k.idx0, k.idx1, (size_t)n, 0, 2 * kLevels, stream);
    return rocprim::radix_sort_pairs(k.sort_tmp, sort_tmp, k.keys0, k.keys1, k.idx0, k.idx1, (size_t)n, 0, 2 * kLevels, stream);
}
}  // namespace

// Spatial (Morton, reference quadrant order) permutation of the bodies only: bbox + path keys + radix sort.
// Used to make the traversal of a HOST-built tree wave-coherent. *perm_dev points into the workspace.
hipError_t device_spatial_order(const float4* posm, int n, void* workspace, size_t workspace_bytes, const unsigned** perm_dev,
                                hipStream_t stream, bool warm)
{
    *perm_dev = nullptr;
    if (n <= 0) return hipSuccess;
    size_t sort_tmp = 0;
    if (device_tree_workspace_bytes(n, 1, &sort_tmp) > workspace_bytes) return hipErrorInvalidValue;
    const Workspace k = carve(workspace, n, sort_tmp, 1);
    const hipError_t e = sort_bodies(posm, n, k, sort_tmp, stream, false, warm);
    if (e != hipSuccess) return e;
    *perm_dev = k.idx1;
    return hipGetLastError();
}

// ---- help for the HOST quadtree build: routing + stable scatter on the device ------------------------------------------
//
// The threaded host build (host_ops.cpp) inserts the first `warm` bodies sequentially, freezes the top levels, and then
// needs every remaining body's bucket (the top-tree leaf it falls into, found by the reference's own quadrant test,
// nbody.rs:322-331) and the bodies grouped by bucket in index order.  Both are data-parallel, the positions already
// live here, and the host has better things to do with its 10 ms: the device descends the (uploaded, few-thousand-node)
// top tree per body, sorts (bucket, index) pairs with a STABLE radix sort on the bucket bits only -- so index order
// survives inside every bucket -- and writes the insert events and the bucket offsets straight into pinned host memory.
struct TopNodeDev {
    float x1, y1, x2, y2;
    int32_t first_child;   // children first_child .. +3 in the order [UL, UR, LL, LR]
    int32_t bucket;        // >= 0: this node is a bucket root; -1: pass-through
};

__global__ __launch_bounds__(kTile) void k_route(const float4* __restrict__ posm, const int warm, const int rest,
                                                 const TopNodeDev* __restrict__ top, unsigned* __restrict__ keys,
                                                 unsigned* __restrict__ idx, int* __restrict__ pbucket_host)
{
    const int i = blockIdx.x * kTile + threadIdx.x;
    if (i >= rest) return;
    const float4 p = posm[warm + i];
    int k = 0;
    TopNodeDev nd = top[0];
    while (nd.bucket < 0) {
        const float cx = __fmul_rn(__fadd_rn(nd.x1, nd.x2), 0.5f);   // quadrant_from_point, unfused like the host
        const float cy = __fmul_rn(__fadd_rn(nd.y1, nd.y2), 0.5f);
        k = nd.first_child + (p.y < cy ? 2 : 0) + (p.x < cx ? 0 : 1);
        nd = top[k];
    }
    keys[i] = (unsigned)nd.bucket;
    idx[i] = (unsigned)i;
    pbucket_host[i] = nd.bucket;
}

struct HostEvent { float x, y, m; unsigned depth; };   // == QuadTree::Event

__global__ __launch_bounds__(kTile) void k_gather_events(const float4* __restrict__ posm, const int warm, const int rest,
                                                         const unsigned* __restrict__ keys_sorted,
                                                         const unsigned* __restrict__ idx_sorted,
                                                         const int* __restrict__ bucket_depth, HostEvent* __restrict__ events_host)
{
    const int p = blockIdx.x * kTile + threadIdx.x;
    if (p >= rest) return;
    const float4 b = posm[warm + (int)idx_sorted[p]];
    events_host[p] = HostEvent{b.x, b.y, b.w, (unsigned)bucket_depth[keys_sorted[p]]};
}

// offset[b] = first sorted position whose bucket is >= b (b = 0 .. nb); one thread per bucket
__global__ __launch_bounds__(kTile) void k_bucket_offsets(const unsigned* __restrict__ keys_sorted, const int rest, const int nb,
                                                          unsigned long long* __restrict__ offset_host)
{
    const int b = blockIdx.x * kTile + threadIdx.x;
    if (b > nb) return;
    int lo = 0, hi = rest;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (keys_sorted[mid] < (unsigned)b) lo = mid + 1; else hi = mid;
    }
    offset_host[b] = (unsigned long long)lo;
}

size_t device_route_workspace_bytes(int rest, int ntop, int nb)
{
    size_t tmp = 0;
    (void)rocprim::radix_sort_pairs(nullptr, tmp, (unsigned*)nullptr, (unsigned*)nullptr, (unsigned*)nullptr, (unsigned*)nullptr,
                                    (size_t)rest, 0, 32, (hipStream_t)0);
    size_t bytes = 0;
    auto add = [&](size_t b) { bytes += (b + 255) & ~(size_t)255; };
    add(sizeof(unsigned) * (size_t)rest * 4);     // keys in/out, idx in/out
    add(tmp);
    add(sizeof(TopNodeDev) * (size_t)ntop);
    add(sizeof(int) * (size_t)nb);
    return bytes;
}

// top_host: ntop records of 6 x 4 bytes (x1, y1, x2, y2, first_child, bucket) ; all *_host outputs are pinned, device-visible
hipError_t device_route_and_scatter(const float4* posm, int warm, int rest, const void* top_host, int ntop,
                                    const int* bucket_depth_host, int nb, void* workspace, size_t workspace_bytes,
                                    int* pbucket_host, void* events_host, unsigned long long* offset_host, hipStream_t stream)
{
    if (rest <= 0) return hipSuccess;
    if (device_route_workspace_bytes(rest, ntop, nb) > workspace_bytes) return hipErrorInvalidValue;
    size_t tmp = 0;
    (void)rocprim::radix_sort_pairs(nullptr, tmp, (unsigned*)nullptr, (unsigned*)nullptr, (unsigned*)nullptr, (unsigned*)nullptr,
                                    (size_t)rest, 0, 32, (hipStream_t)0);
    char* w = static_cast<char*>(workspace);
    auto take = [&](size_t b) { char* p = w; w += (b + 255) & ~(size_t)255; return p; };
    unsigned* keys0 = reinterpret_cast<unsigned*>(take(sizeof(unsigned) * (size_t)rest * 4));
    unsigned* keys1 = keys0 + rest;
    unsigned* idx0 = keys1 + rest;
    unsigned* idx1 = idx0 + rest;
    void* sort_tmp = take(tmp);
    TopNodeDev* top = reinterpret_cast<TopNodeDev*>(take(sizeof(TopNodeDev) * (size_t)ntop));
    int* depth = reinterpret_cast<int*>(take(sizeof(int) * (size_t)nb));
    hipError_t e = hipMemcpyAsync(top, top_host, sizeof(TopNodeDev) * (size_t)ntop, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(depth, bucket_depth_host, sizeof(int) * (size_t)nb, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return e;
    const int blocks = (rest + kTile - 1) / kTile;
    hipLaunchKernelGGL(k_route, dim3(blocks), dim3(kTile), 0, stream, posm, warm, rest, top, keys0, idx0, pbucket_host);
    int bits = 1;
    while ((1 << bits) < nb) bits++;
    e = rocprim::radix_sort_pairs(sort_tmp, tmp, keys0, keys1, idx0, idx1, (size_t)rest, 0, bits, stream);   // stable
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_gather_events, dim3(blocks), dim3(kTile), 0, stream, posm, warm, rest, keys1, idx1, depth,
                       reinterpret_cast<HostEvent*>(events_host));
    hipLaunchKernelGGL(k_bucket_offsets, dim3((nb + 1 + kTile - 1) / kTile), dim3(kTile), 0, stream, keys1, rest, nb, offset_host);
    return hipGetLastError();
}

// The sorted body order restricted to one slab of targets [lo, hi) (multi-GPU: every device walks the same tree for
// its own slab): the entries of `perm` that fall in the slab, relative order kept, so that the slab's bodies are
// still handed to consecutive lanes in Morton order and the wave-uniform walk applies.
namespace {
struct InSlab {
    unsigned lo, hi;
    __device__ bool operator()(const unsigned v) const { return v >= lo && v < hi; }
};
}  // namespace

size_t device_slab_order_workspace_bytes(int n)
{
    size_t tmp = 0;
    (void)rocprim::select(nullptr, tmp, (const unsigned*)nullptr, (unsigned*)nullptr, (unsigned*)nullptr, (size_t)n, InSlab{0u, 0u},
                          (hipStream_t)0);
    return ((tmp + 255) & ~(size_t)255) + (((size_t)n * sizeof(unsigned) + 255) & ~(size_t)255) + 256;
}

hipError_t device_slab_order(const unsigned* perm, int n, int lo, int hi, void* workspace, size_t workspace_bytes,
                             const unsigned** slab_perm, hipStream_t stream)
{
    *slab_perm = nullptr;
    if (n <= 0 || hi <= lo) return hipSuccess;
    if (device_slab_order_workspace_bytes(n) > workspace_bytes) return hipErrorInvalidValue;
    size_t tmp = 0;
    (void)rocprim::select(nullptr, tmp, (const unsigned*)nullptr, (unsigned*)nullptr, (unsigned*)nullptr, (size_t)n, InSlab{0u, 0u},
                          (hipStream_t)0);
    char* w = static_cast<char*>(workspace);
    void* sel_tmp = w;
    w += (tmp + 255) & ~(size_t)255;
    unsigned* out = reinterpret_cast<unsigned*>(w);
    w += ((size_t)n * sizeof(unsigned) + 255) & ~(size_t)255;
    unsigned* count = reinterpret_cast<unsigned*>(w);
    const hipError_t e = rocprim::select(sel_tmp, tmp, perm, out, count, (size_t)n, InSlab{(unsigned)lo, (unsigned)hi}, stream);
    if (e != hipSuccess) return e;
    *slab_perm = out;
    return hipGetLastError();
}

// Builds the flattened tree for posm[0..n) into `out` (capacity node_cap records), in two halves so that a host
// driving several devices can start every build before it waits for any of them:
//   begin: enqueues everything on `stream`, including the copy of the node count into the pinned host_counters;
//          *perm_dev = the sorted body order (device pointer inside the workspace: thread t handles body perm[t])
//   end:   waits for the stream; *n_nodes_host = node count; *status = 1 when the tree needs more than node_cap
//          nodes (nothing usable was written), 2 when more than max(16, n/2000) bodies sit in clusters of >= 3 within EPS
//          (the caller should build on the host: the reference's multi-body merges are not reproduced here)
hipError_t device_tree_build_begin(const float4* posm, int n, void* workspace, size_t workspace_bytes, int node_cap, BhNode* out,
                                   int* host_counters /* pinned, >= 4 ints; null: the caller's gated kick-drift publishes them */,
                                   const unsigned** perm_dev, hipStream_t stream, int fold,
                                   hipStream_t side, hipEvent_t ev_go, hipEvent_t ev_done, bool depth_panic_guard, bool warm)
{
    *perm_dev = nullptr;
    if (n <= 0) return hipSuccess;
    size_t sort_tmp = 0;
    if (device_tree_workspace_bytes(n, node_cap, &sort_tmp) > workspace_bytes) return hipErrorInvalidValue;
    const Workspace k = carve(workspace, n, sort_tmp, node_cap);
    // side streams pay from a few thousand bodies on: a join costs ~10 us, the root's chain 16 ns per body
    // (NBX_SIDE_STREAMS_FROM overrides the measured crossover: profiles/r03_bh_side_stream_crossover.txt)
    static const int side_from = [] { const char* v = std::getenv("NBX_SIDE_STREAMS_FROM"); return v ? std::atoi(v) : kSideStreamsFrom; }();
    const bool root_aside = fold == 1 && side && ev_go && ev_done && n >= side_from;
    hipError_t e;
    if (root_aside) {
        // the root's fold -- n serial steps, needs only the bodies in index order -- starts NOW on the side stream, beside
        // the sort, the scans and the other nodes' folds; the main stream picks its result up before k_fold_big
        if ((e = hipEventRecord(ev_go, stream)) != hipSuccess) return e;           // the positions are final on `stream` here
        if ((e = hipStreamWaitEvent(side, ev_go, 0)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_fold_root, dim3(1), dim3(128), 0, side, posm, n, out);
        if ((e = hipEventRecord(ev_done, side)) != hipSuccess) return e;
    }
    e = sort_bodies(posm, n, k, sort_tmp, stream, fold == 1, warm);
    if (e != hipSuccess) return e;
    *perm_dev = k.idx1;
    const int nb = (n + kTile - 1) / kTile;
    const int sb = (n + kScanBlock - 1) / kScanBlock;
    // the bodies in sorted order and the reference's EPS merge; everything below works on ENTITY keys (every member of a
    // blob carries the key of the blob's first arrival), the bodies' indices and records in that order
    const unsigned long long* mk = k.keys0;      // (keys0 / idx0 are free again after the sort)
    const unsigned* mi = k.idx1;
    const float4* ms = k.sb;
    const unsigned char* pmin = nullptr;
    if (fold == 1) {
        // blobs of any size, replayed (3c): the tree is then the reference's, node for node -- or the step is refused
        hipLaunchKernelGGL(k_cells, dim3(nb), dim3(kTile), 0, stream, posm, k.sb, k.keys1, k.idx1, k.box, n, k.hk, k.hv, k.hmask, k.ekey, k.link);
        hipLaunchKernelGGL(k_blobs, dim3(nb), dim3(kTile), 0, stream, k.sb, k.keys1, k.idx1, k.box, n, k.hk, k.hv, k.hmask, k.ekey, k.link,
                           k.ghosts, k.counters);
        hipLaunchKernelGGL(k_place, dim3(nb), dim3(kTile), 0, stream, k.keys1, k.ekey, k.idx1, k.sb, k.link, k.ghosts, k.counters, n,
                           k.keys0, k.idx0, k.sb2, k.pmin2);
        mi = k.idx0; ms = k.sb2; pmin = k.pmin2;
    } else {
        // pairs of neighbouring entities only (3b): links from the sorted keys + arrival order (this kernel also gathers the
        // bodies into sorted order), then both members of a pair share one key
        hipLaunchKernelGGL(k_merge_links, dim3(nb), dim3(kTile), 0, stream, posm, k.sb, k.keys1, k.idx1, n, k.link, k.counters + 1);
        hipLaunchKernelGGL(k_merge_keys, dim3(nb), dim3(kTile), 0, stream, k.keys1, k.idx1, k.link, n, k.keys0, k.counters + 1);
    }
    // (for small systems the pair merge and the scan were tried as phases of ONE 1024-thread workgroup: 90 us against 22 for the
    //  four launches at 10 000 bodies -- per-body work here is chains of dependent loads that miss the L2 after every kernel
    //  boundary, and one CU hides far less of that than forty)
    hipLaunchKernelGGL(k_scan_reduce, dim3(sb), dim3(kTile), 0, stream, ms, mk, n, k.block_sums, k.counters + 3, k.pre.cnt);
    hipLaunchKernelGGL(k_scan_write, dim3(sb), dim3(kTile), 0, stream, ms, mk, n, k.block_sums, k.pre, k.counters);
    // one thread per node; the node count is only known on the device, so the grid covers the whole pool (threads beyond
    // base[n] leave at once; the pool check is inside)
    const int eb = n <= 65536 ? 64 : kTile;   // spread a small system's few waves over the CUs
    hipLaunchKernelGGL(k_emit, dim3((unsigned)((node_cap + eb - 1) / eb)), dim3(eb), 0, stream, ms, mk, mi, k.box, k.pre, n, node_cap, out,
                       fold, k.big, n, k.counters, (root_aside ? 1 : 0) | (depth_panic_guard ? 2 : 0), pmin);
    if (fold == 1) {
        // one pair of waves per queued node; the count lives on the device: enough workgroups for every plausible queue
        // (a uniform system queues ~n/5 nodes), they loop when there are more
        const int fb = n / 4 + 64;
        hipLaunchKernelGGL(k_fold_big, dim3((unsigned)(fb < 8192 ? fb : 8192)), dim3(128), 0, stream, posm, ms, mi, k.big, n, k.counters, n, out);
        if (root_aside && (e = hipStreamWaitEvent(stream, ev_done, 0)) != hipSuccess) return e;   // the tree is complete on `stream` from here
    }
    if (host_counters && (e = hipMemcpyAsync(host_counters, k.counters, 8 * sizeof(int), hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
    return hipGetLastError();
}

int* device_tree_counters(void* workspace) { return static_cast<int*>(workspace); }   // the header comes first (carve)

void device_tree_limits(int n, int fold, int* crowd_limit, int* queue_limit)
{
    // fold = 1 promises the reference's tree node for node: ANY body the pairs-only merge left behind (or a blob whose centre
    // left its first member's cell) sends the step to the host build; fold = 0 tolerates a few (its own tolerance class)
    *crowd_limit = fold == 1 ? 0 : (n / 2000 > 16 ? n / 2000 : 16);
    *queue_limit = fold == 1 ? n : 0x7FFFFFFF;
}

hipError_t device_tree_build_end(int n, int node_cap, const int* host_counters, int* n_nodes_host, int* status, hipStream_t stream, int fold)
{
    *status = 0;
    *n_nodes_host = 0;
    if (n <= 0) return hipSuccess;
    const hipError_t e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
    if (host_counters[0] > node_cap) { *status = 1; return hipSuccess; }   // node pool exhausted (pathological input)
    // Many bodies in clusters of three or more within EPS: the reference grows multi-body blobs there (nbody.rs:249-260) that
    // the pairs-only merge does not reproduce -- leave such systems to the reference-faithful host build.
    int crowd_limit = 0, queue_limit = 0;
    device_tree_limits(n, fold, &crowd_limit, &queue_limit);
    if (host_counters[1] > crowd_limit) { *status = 2; return hipSuccess; }
    if (host_counters[2] > queue_limit) { *status = 1; return hipSuccess; }   // fold queue overflow
    *n_nodes_host = host_counters[0];
    return hipSuccess;
}

}  // namespace nbx
