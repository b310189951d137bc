// bh_build.hip -- quadtree build ON THE DEVICE.  The fast mode's DEFAULT from 1 024 bodies on
// (NBX_OPT_BH_TREE; 512 with exactly summed nodes); the bit-exact mode builds on the host unless asked (reference fold only).
//
// The host build (host_tree.cpp) inserts the bodies one by one exactly as the reference does (nbody.rs:388-415); at 1 M bodies that
// is 13-16 ms per step, at the reference's 10 000 bodies 0.5 ms.  The build here -- five units around bh_build_internal.h -- makes
// the SAME flattened tree without leaving the GPU:
//
//   1. root AABB = min/max of positions (exact; nbody.rs:388-398)                                                      [bh_front.hip]
//   2. per body: the path of quadrant choices, replaying quadrant_from_point / create_children with the
//      reference's own f32 midpoint arithmetic (cx = (x1+x2)*0.5, nbody.rs:289-290, :324-331) for 31 levels
//      -> 62-bit key, 2 bits per level, quadrant order [UL,UR,LL,LR] = 0..3 like the reference's child array
//   3. sort of (key, body index): from last step's order when there is one (bh_sort.hip, round 5: four launches instead of the
//      library's seventeen), in two launches up to 16 384 bodies (bh_front.hip), rocPRIM's radix sort on a cold build
//   3b. the reference's EPS merge (nbody.rs:249-260) -- reference fold: whole clusters of close bodies replayed in arrival order
//       (k_cells / k_blobs / k_place: bh_cluster.hip); exact-sum class: close PAIRS decided from the sorted keys and the arrival
//       order (k_merge_links / k_merge_keys, section 3b below)
//   4. nodes straight from the sorted keys: every node is (first body a, depth l); how many nodes start at each body
//      follows from the digits it shares with its two neighbours, an exclusive scan of those counts gives every node's
//      PRE-ORDER slot, and a node's skip pointer is the slot of the first node after its bodies (see "the tree from
//      the sorted keys" below) -- no level-by-level sweep, no host round trips, one read-back of the node count
//   5. interior masses and centres of mass, two classes (NBX_OPT_BH_FOLD):
//      fold = 1 (default up to 65 536 bodies): the reference's f32 running fold over the node's bodies in ARRIVAL order
//               (nbody.rs:303-320) -- small nodes in k_emit, the others in k_fold_big (bh_fold.hip; one pair of waves per node: m chain, IEEE
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
#include <cstdlib>

#include "bh_build_internal.h"

namespace nbx {

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

// Also gathers the bodies into sorted order (sb[j] = posm[idx[j]]: round 2 had a kernel of its own for that) -- unless the sort
// delivered them already (sb_ready: the warm sort carries the records along, bh_sort.hip; the two gathers here fetched 239 MB for
// 32 MB of records at a million bodies, a 128-byte line per 16-byte record, and made this kernel the build's most HBM-bound).
__device__ __forceinline__ void merge_links_body(const int j, const float4* __restrict__ posm, float4* __restrict__ sb,
                                                 const unsigned long long* __restrict__ keys, const unsigned* __restrict__ idx,
                                                 const int n, unsigned char* __restrict__ close, int* __restrict__ crowded,
                                                 const bool sb_ready)
{
    const float4 b = sb_ready ? sb[j] : posm[idx[j]];
    if (!sb_ready) sb[j] = b;
    unsigned char out = 0;
    if (j + 1 < n && keys[j + 1] == keys[j] && !(j > 0 && keys[j - 1] == keys[j])) {
        // The first of several bodies of one level-31 cell: one leaf as long as every arrival is within EPS of the centre the
        // earlier ones have folded to.  Where an ulp of the coordinates is no longer small against EPS (|x| in the thousands) the
        // folded centre of even identical positions can sit more than EPS away (nbody.rs:315-317 round three times) and the
        // reference splits: such bodies are counted as left behind.
        float cx = 0.0f, cy = 0.0f, cm = 0.0f;
        int left = 0;
        for (int t = j; t < n && keys[t] == keys[j]; t++) {        // (the stable sort left them in index order)
            const float4 q = sb_ready ? sb[t] : posm[idx[t]];
            if (t > j && !(fabsf(__fsub_rn(cx, q.x)) < kEps && fabsf(__fsub_rn(cy, q.y)) < kEps)) left++;
            fold_mass(cx, cy, cm, q.x, q.y, q.w);
        }
        if (left) atomicAdd(crowded, left);
    }
    if (j > 0 && keys[j - 1] != keys[j]) {
        // the entity's position is its first arrival's (later arrivals of the same cell are < 5e-8 of the box away)
        const int r = run_start(keys, j - 1);
        const float4 a = sb_ready ? sb[r] : posm[idx[r]];
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
                                                       unsigned char* __restrict__ close, int* __restrict__ crowded, const int sb_ready)
{
    const int j = blockIdx.x * kTile + threadIdx.x;
    if (j < n) merge_links_body(j, posm, sb, keys, idx, n, close, crowded, sb_ready != 0);
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
// separate k_scan_blocks launch (same bits on every run), without the launch.  ticket == nullptr (round 5, big systems): no
// hand-off in here, k_scan_blocks follows as a launch of its own -- a device-wide fence writes the L2 back on this chip, and a
// thousand workgroups each paying one cost this kernel 15 of its 36 us at a million bodies; a launch costs 3.
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
        if (ticket) {
            __threadfence();
            last = atomicAdd(ticket, 1) == (int)gridDim.x - 1;
        } else {
            last = 0;
        }
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    scan_blocks(block_sums, (int)gridDim.x);
}

__global__ __launch_bounds__(kTile) void k_scan_blocks(ScanItem* __restrict__ block_sums, const int nb)
{
    scan_blocks(block_sums, nb);
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
    static_assert(kScanPerThread == 4, "the vector stores below write four consecutive prefixes");
    ScanItem at[kScanPerThread];
#pragma unroll
    for (int u = 0; u < kScanPerThread; u++) { at[u] = run; run = scan_add(run, it[u]); }
    if (j0 + kScanPerThread <= n) {
        // four consecutive bodies per thread: 16-byte stores (two per double array, one per int array) instead of four narrow ones
        // each -- a lane's four 8-byte stores sat 32 bytes apart from its neighbour's (round 5: 77 MB written for 39 MB of prefixes)
        reinterpret_cast<double2*>(p.m + j0)[0] = make_double2(at[0].m, at[1].m);
        reinterpret_cast<double2*>(p.m + j0)[1] = make_double2(at[2].m, at[3].m);
        reinterpret_cast<double2*>(p.mx + j0)[0] = make_double2(at[0].mx, at[1].mx);
        reinterpret_cast<double2*>(p.mx + j0)[1] = make_double2(at[2].mx, at[3].mx);
        reinterpret_cast<double2*>(p.my + j0)[0] = make_double2(at[0].my, at[1].my);
        reinterpret_cast<double2*>(p.my + j0)[1] = make_double2(at[2].my, at[3].my);
        *reinterpret_cast<int4*>(p.base + j0) = make_int4(at[0].cnt, at[1].cnt, at[2].cnt, at[3].cnt);
        *reinterpret_cast<int4*>(p.ent + j0) = make_int4(at[0].ent, at[1].ent, at[2].ent, at[3].ent);
    } else {
#pragma unroll
        for (int u = 0; u < kScanPerThread; u++) {
            const int j = j0 + u;
            if (j < n) { p.m[j] = at[u].m; p.mx[j] = at[u].mx; p.my[j] = at[u].my; p.base[j] = at[u].cnt; p.ent[j] = at[u].ent; }
        }
    }
#pragma unroll
    for (int u = 0; u < kScanPerThread; u++) {
        const int j = j0 + u;
        if (j < n)
            for (int t = 0; t < it[u].cnt; t++)          // the (at most 32) nodes that start here, shallow to deep
                if (at[u].cnt + t < p.owner_cap) p.owner[at[u].cnt + t] = j;
        if (j == n - 1) {
            const ScanItem end = scan_add(at[u], it[u]);
            p.m[n] = end.m; p.mx[n] = end.mx; p.my[n] = end.my; p.base[n] = end.cnt; p.ent[n] = end.ent;
            counters[0] = end.cnt;
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
constexpr int kTicketScanMax = 65536;   // up to here the scan of the block sums rides in k_scan_reduce's last workgroup (one launch less)

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
    // choices of the body that opened the entity.  Only the x extent goes into the record (nbody.rs:341), and the two axes halve
    // independently: the y half of the replay is left to the one check below that needs it (round 5: a sixth of this kernel's
    // vector instructions, profiles/r05_bh_step_issue_counters.json)
    float x1 = dec_f32(box[0]), x2 = dec_f32(box[2]);
#pragma unroll 1
    for (int d = 0; d < l; d++) {
        const float cx = __fmul_rn(__fadd_rn(x1, x2), 0.5f);
        if ((ka >> (2 * (kLevels - 1 - d))) & 1ull) x1 = cx; else x2 = cx;
    }
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
            float y1 = v1, y2 = v2;                 // (the y half of this leaf's own path)
#pragma unroll 1
            for (int d = 0; d < l; d++) {
                descend(u1, v1, u2, v2, px, py);
                const float cy = __fmul_rn(__fadd_rn(y1, y2), 0.5f);
                if ((ka >> (2 * (kLevels - 1 - d))) & 2ull) y2 = cy; else y1 = cy;
            }
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
                                   hipStream_t side, hipEvent_t ev_go, hipEvent_t ev_done, bool depth_panic_guard, bool warm, const float4* sorted_pos)
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
        launch_fold_root(posm, n, out, side);
        if ((e = hipEventRecord(ev_done, side)) != hipSuccess) return e;
    }
    bool sb_ready = false;   // did the sort deliver the bodies' records in sorted order (k.sb) already?
    e = sort_bodies(posm, n, k, sort_tmp, stream, fold == 1, warm, sorted_pos, /*want_sb=*/true, &sb_ready);
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
        if ((e = launch_cluster_replay(posm, n, k, stream, sb_ready)) != hipSuccess) return e;
        mi = k.idx0; ms = k.sb2; pmin = k.pmin2;
    } else {
        // pairs of neighbouring entities only (3b): links from the sorted keys + arrival order (this kernel also gathers the
        // bodies into sorted order), then both members of a pair share one key
        hipLaunchKernelGGL(k_merge_links, dim3(nb), dim3(kTile), 0, stream, posm, k.sb, k.keys1, k.idx1, n, k.link, k.counters + 1, sb_ready ? 1 : 0);
        hipLaunchKernelGGL(k_merge_keys, dim3(nb), dim3(kTile), 0, stream, k.keys1, k.idx1, k.link, n, k.keys0, k.counters + 1);
    }
    // (for small systems the pair merge and the scan were tried as phases of ONE 1024-thread workgroup: 90 us against 22 for the
    //  four launches at 10 000 bodies -- per-body work here is chains of dependent loads that miss the L2 after every kernel
    //  boundary, and one CU hides far less of that than forty)
    if (n > kTicketScanMax) {   // big systems: the block sums' scan as a launch of its own (see k_scan_reduce)
        hipLaunchKernelGGL(k_scan_reduce, dim3(sb), dim3(kTile), 0, stream, ms, mk, n, k.block_sums, (int*)nullptr, k.pre.cnt);
        hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(kTile), 0, stream, k.block_sums, sb);
    } else {
        hipLaunchKernelGGL(k_scan_reduce, dim3(sb), dim3(kTile), 0, stream, ms, mk, n, k.block_sums, k.counters + 3, k.pre.cnt);
    }
    hipLaunchKernelGGL(k_scan_write, dim3(sb), dim3(kTile), 0, stream, ms, mk, n, k.block_sums, k.pre, k.counters);
    // one thread per node; the node count is only known on the device, so the grid covers the whole pool (threads beyond
    // base[n] leave at once; the pool check is inside)
    const int eb = n <= 65536 ? 64 : kTile;   // spread a small system's few waves over the CUs
    hipLaunchKernelGGL(k_emit, dim3((unsigned)((node_cap + eb - 1) / eb)), dim3(eb), 0, stream, ms, mk, mi, k.box, k.pre, n, node_cap, out,
                       fold, k.big, n, k.counters, (root_aside ? 1 : 0) | (depth_panic_guard ? 2 : 0), pmin);
    if (fold == 1) {
        launch_fold_big(posm, ms, mi, k.big, n, k.counters, n, out, stream);
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
