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
//       (k_cells / k_blobs / k_place: bh_cluster.hip); exact-sum class (round 6): every CHAIN of close sorted neighbours replayed in
//       arrival order by one wave (k_chain_links / k_chain_heads / k_chain, section 3b below); only chains of more than 60 bodies,
//       beyond a limit, are refused
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
// host build + flatten, INCLUDING the reference's EPS merge (nbody.rs:249-260) of chains of any length; what differs there:
//   * interior centres of mass are the f32 rounding of the exact weighted mean instead of the reference's
//     particle-by-particle f32 running fold (which drifts by up to ~6e-4 relative at 100 k bodies);
//   * bodies the reference merges although more than two others lie between them in key order (close in space, far apart on the
//     Z-curve: a cluster astride a coarse cell boundary) stay apart -- 10-65 of 40 000 blobs of the collapsed 2 M-body model;
//   * a merged blob whose centre's path lies outside its chain's place in the sorted order is filed under the member's path that
//     follows the centre's deepest;
//   * a chain of more than 60 linked bodies is replayed in pieces (cuts at multiples of 32 sorted places); the bodies of a blob that ends
//     at a cut count as left behind: more than max(16, n/2000) of them send the step to the host build (as chains of three did before);
//   * no depth-50 panic (nbody.rs:230-232): keys stop at level 31.
// fold = 1 replays whole clusters through a grid of cells (3c) and refuses what it cannot reproduce.
#include <cstdlib>

#include "bh_build_internal.h"

namespace nbx {

// ---- 3b. the reference's EPS merge, exact-sum class: chains of close bodies replayed in arrival order (round 6) ------------
// nbody.rs:249-260: a body B arriving at a non-empty exterior node merges into it when the node's content -- a body, or the
// running centre of the bodies merged there so far -- is closer than EPS in both axes; otherwise the node splits until the two
// part ways (the blob travels by its CENTRE, :271-281).  B arrives at the leaf of an earlier entity X iff X is the ONE earlier
// entity whose path shares the most leading digits with B's: a tie means B's cell below their common node is still empty.
//     merge(X, B)  <=>  X = the unique argmax of common(path(X's centre now), key(B)) over the entities that arrived before B
//                       and  |dx| < EPS and |dy| < EPS against X's centre
// Rounds 2-5 decided that for PAIRS of sorted neighbours and left longer chains alone (counted; beyond max(16, n/2000) bodies the
// step went to the host build: every few steps of a collapsing system from 1.5 M bodies on, 50-85 ms each).  Now every CHAIN is
// replayed:
//   k_chain_links   boundary b (between sorted bodies b-1 and b) is linked when some pair (j, t), j < b <= t <= j + 3, lies within
//                   2 EPS in both axes -- sorted neighbours that could meet in one leaf, with up to two strangers between them in
//                   key order.  (2 EPS, three ahead: measured on chains of 2-6 bodies in random arrival order and on the 2 M-body
//                   model of the benchmarks 5 to 40 steps into its collapse -- 1 EPS / neighbours only misses 2 % / 0.3 % of the
//                   reference's blobs, this none of the injected ones and 10-65 of 40 000 in the collapsed model: pairs that are
//                   close in space and far apart on the Z-curve.)  Bodies of one level-31 cell beyond kRunLink of them stay out
//                   (they share a leaf anyway, any number of them).
//   k_chain_heads   lists the bodies at which a maximal run of linked boundaries starts (a SEGMENT: at most 60 bodies -- a longer
//                   chain is cut at multiples of 32 that have 14 linked boundaries on either side)
//   k_chain         one wave per segment that holds a pair within EPS, one lane per body.  The wave replays the reference's insertion
//                   in arrival order: the live entities compare their centre's path with the newcomer's key (the deepest of them:
//                   five ballots), the unique winner tests EPS against its centre, and nobody OUTSIDE the segment who arrived
//                   earlier may sit at least as deep in the newcomer's path (the 64 sorted neighbours on either side, held in
//                   registers; up to 256 where a cell is that crowded) -- then the winner folds the newcomer in (add_mass, f32) and
//                   follows its new centre down from the segment's common cell until it has parted from everybody still to come.
//                   Output: the segment's bodies regrouped blob by blob, each blob in arrival order (what emit_node's leaf fold
//                   wants), every member under ONE key -- the path of the blob's centre, where the reference files it -- written to
//                   a second set of arrays (keys0 / idx0 / sb2) so that neighbouring waves' probes read the sort's own output;
//                   everybody else is as k_chain_links copied them.
// Same leaves, same node set as the host tree wherever a blob's bodies are that close on the Z-curve (tests: trees with injected
// chains, the 2 M-body model before and during its collapse: chains of at most 36 bodies there).  What the replay only approximates
// -- in a chain of more than 60 the blobs that end at a cut which a pair within EPS spans (measured: forces up to 1.7e-2 of max|F|
// off the reference's tree on 777 bodies 0.26 EPS apart with masses over six decades), a merge behind a probe that hit its cap -- is
// counted (counters[6]) and counts as left behind like the chains of rounds 2-5 did: beyond max(16, n/2000) such bodies the step
// goes to the host build, so the class's bounds hold for every tree it serves.  counters[7]: bodies merged.
constexpr int kRunLink = 8;          // bodies of one level-31 cell that still take part in a replay as individuals
constexpr int kLinkLook = 3;         // a body links boundaries up to this many sorted places ahead
constexpr float kLinkEps = 2.0e-4f;  // 2 EPS
constexpr float kTightEps = 1.0001e-4f;   // EPS, and a rounding of the difference (the merge test itself is exact, in k_chain)
constexpr int kCutEvery = 32, kCutSpan = 14;   // segments stay <= 2 * kCutSpan + kCutEvery = 60 bodies
constexpr int kRivalProbes = 4;      // 64 neighbours each, per side
constexpr int kTallySlots = 256;     // slots the replay's tallies are spread over (2 words each: approximate, merged),
constexpr int kTallyStride = 16;     // 64 bytes apart (kTallySlots * kTallyStride = kGhostCap words: the other class's list)
static_assert(kTallySlots * kTallyStride <= kGhostCap, "the tallies live in the ghost list");

__device__ __forceinline__ bool in_long_run(const unsigned long long* __restrict__ keys, const int j, const int n)
{
    const unsigned long long k = keys[j];
    const bool l = j > 0 && keys[j - 1] == k, r = j + 1 < n && keys[j + 1] == k;
    if (!l && !r) return false;
    int len = 1;
    for (int t = j - 1; t >= 0 && len <= kRunLink && keys[t] == k; t--) len++;
    for (int t = j + 1; t < n && len <= kRunLink && keys[t] == k; t++) len++;
    return len > kRunLink;
}

// Also gathers the bodies into sorted order (sb[j] = posm[idx[j]]: round 2 had a kernel of its own for that) -- unless the sort
// delivered them already (sb_ready: the warm sort carries the records along, bh_sort.hip; the two gathers of round 4 fetched
// 239 MB for 32 MB of records at a million bodies, a 128-byte line per 16-byte record).  The neighbours' records come out of LDS.
__global__ __launch_bounds__(kTile) void k_chain_links(const float4* __restrict__ posm, float4* __restrict__ sb,
                                                       const unsigned long long* __restrict__ keys,
                                                       const unsigned* __restrict__ idx, const int n,
                                                       unsigned char* __restrict__ link, int* __restrict__ crowded, const int sb_ready,
                                                       unsigned long long* __restrict__ out_keys, unsigned* __restrict__ out_idx,
                                                       float4* __restrict__ out_sb, int* __restrict__ tally)
{
    __shared__ float2 tile[kTile + 2 * kLinkLook];
    static_assert(kTallySlots <= kTile, "one thread per slot");
    if (blockIdx.x == 0 && threadIdx.x < kTallySlots) { tally[kTallyStride * threadIdx.x] = 0; tally[kTallyStride * threadIdx.x + 1] = 0; }   // (the replay's, two launches on)
    const int j0 = blockIdx.x * kTile - kLinkLook;
    for (int t = threadIdx.x; t < kTile + 2 * kLinkLook; t += kTile) {
        const int j = j0 + t;
        float4 b = make_float4(3.0e38f, 3.0e38f, 0.0f, 0.0f);    // (outside the array: close to nobody)
        if (j >= 0 && j < n) {
            const unsigned i = idx[j];
            b = sb_ready ? sb[j] : posm[i];
            if (t >= kLinkLook && t < kTile + kLinkLook) {
                // everybody as the sort left them, into the arrays the scans and k_emit read: k_chain rewrites the segments it replays
                // (this kernel has the records in hand; as k_chain's own first act the copy ran at 1 TB/s, half-empty waves of 32)
                if (!sb_ready) sb[j] = b;
                out_keys[j] = keys[j]; out_idx[j] = i; out_sb[j] = b;
            }
        }
        tile[t] = make_float2(b.x, b.y);
    }
    __syncthreads();
    const int j = blockIdx.x * kTile + threadIdx.x;          // boundary j: between bodies j - 1 and j
    if (j >= n) return;
    const int c = threadIdx.x + kLinkLook;                   // body j in the tile
    bool covered = false, tight = false;
#pragma unroll
    for (int u = 1; u <= kLinkLook; u++)                     // left end j - u, right ends j .. j - u + kLinkLook
#pragma unroll
        for (int v = 0; v + u <= kLinkLook; v++) {
            const float2 p = tile[c - u], q = tile[c + v];
            const float dx = fabsf(p.x - q.x), dy = fabsf(p.y - q.y);
            covered |= dx < kLinkEps && dy < kLinkEps;
            tight |= dx < kTightEps && dy < kTightEps;
        }
    // bit 0: linked; bit 1: a pair within EPS spans this boundary -- a segment without one cannot merge anybody (the first merge of a
    // replay is between two BODIES) and is not replayed: two thirds of the segments of the collapsing 1 M-body model
    unsigned char out = 0;
    if (j > 0 && covered && !in_long_run(keys, j - 1, n) && !in_long_run(keys, j, n)) out = tight ? 3 : 1;
    link[j] = out;
    if (j + 1 < n && keys[j + 1] == keys[j] && !(j > 0 && keys[j - 1] == keys[j]) && in_long_run(keys, j, n)) {
        // The first of MANY bodies of one level-31 cell: one leaf as long as every arrival is within EPS of the centre the
        // earlier ones have folded to.  Where an ulp of the coordinates is no longer small against EPS (|x| in the thousands) the
        // folded centre of even identical positions can sit more than EPS away (nbody.rs:315-317 round three times) and the
        // reference splits: such bodies are counted as left behind (counters[1]; shorter runs are replayed body by body).
        float cx = 0.0f, cy = 0.0f, cm = 0.0f;
        int left = 0;
        for (int t = j; t < n && keys[t] == keys[j]; t++) {        // (the stable sort left them in index order)
            const float4 q = sb_ready ? sb[t] : posm[idx[t]];
            if (t > j && !(fabsf(__fsub_rn(cx, q.x)) < kEps && fabsf(__fsub_rn(cy, q.y)) < kEps)) left++;
            fold_mass(cx, cy, cm, q.x, q.y, q.w);
        }
        if (left) atomicAdd(crowded, left);
    }
}

// is boundary b (between sorted bodies b - 1 and b) linked, cuts of long chains taken out.  The 14 links on either side of a multiple
// of 32 are two aligned 16-byte loads (link[] holds 0 / 1 and is 256-byte aligned): as 28 dependent byte loads this test WAS the two
// kernels below wherever the system is dense.
__device__ __forceinline__ bool chain_linked(const unsigned char* __restrict__ link, const int b, const int n)
{
    if (b <= 0 || b >= n || !(link[b] & 1)) return false;
    if (b % kCutEvery != 0 || b - kCutSpan < 1 || b + kCutSpan > n - 1) return true;
    static_assert(kCutEvery % 16 == 0 && kCutSpan == 14, "the two loads below cover link[b - 16 .. b + 15]");
    uint4 lo = *reinterpret_cast<const uint4*>(link + b - 16);   // (b + 15 <= n: the array is padded, carve)
    uint4 hi = *reinterpret_cast<const uint4*>(link + b);
    lo.x &= 0x01010101u; lo.y &= 0x01010101u; lo.z &= 0x01010101u; lo.w &= 0x01010101u;   // (bit 0 of every byte: linked)
    hi.x &= 0x01010101u; hi.y &= 0x01010101u; hi.z &= 0x01010101u; hi.w &= 0x01010101u;
    const bool all = (lo.x & 0xFFFF0000u) == 0x01010000u && lo.y == 0x01010101u && lo.z == 0x01010101u && lo.w == 0x01010101u &&
                     (hi.x & 0xFFFFFF00u) == 0x01010100u && hi.y == 0x01010101u && hi.z == 0x01010101u && (hi.w & 0x00FFFFFFu) == 0x00010101u;
    return !all;   // deep inside a long chain: cut here
}

__device__ __forceinline__ int wave_max_i32(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(v, off); v = o > v ? o : v; }
    return v;
}
// lane `src`'s value, src the same in every lane (a ballot's first bit, a loop counter): v_readlane, a few cycles -- __shfl goes
// through the LDS crossbar (ds_bpermute) whatever the index is, and the replay's serial loop is made of these
__device__ __forceinline__ int bcast_i32(const int v, const int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ unsigned bcast_u32(const unsigned v, const int src) { return (unsigned)__builtin_amdgcn_readlane((int)v, src); }
__device__ __forceinline__ float bcast_f32(const float v, const int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }
__device__ __forceinline__ unsigned long long shfl_u64(const unsigned long long v, const int src)
{
    const unsigned lo = bcast_u32((unsigned)v, src), hi = bcast_u32((unsigned)(v >> 32), src);
    return ((unsigned long long)hi << 32) | lo;
}

// earlier arrivals OUTSIDE the segment [p0, p0 + t) that sit at least `depth` digits deep in the path kb: 1 = there is one,
// 0 = none, -1 = none among the neighbours probed but the probes ran out (counted as approximate by the caller).  The first 64
// neighbours on either side are in registers (nk / ni: lane l holds the l-th neighbour to the left and to the right, loaded once
// per segment -- a probe from memory inside the replay's serial loop made one long segment the kernel's duration: 56 us at a
// million bodies for 4 500 merges); only a cell more crowded than that goes back to memory.
struct ChainNeighbours {
    unsigned long long kl, kr;
    unsigned il, ir;      // 0xFFFFFFFF: outside the array
};
__device__ __forceinline__ int chain_rival(const unsigned long long* __restrict__ keys, const unsigned* __restrict__ idx, const int n,
                                           const int p0, const int t, const unsigned long long kb, const unsigned ib, const int depth,
                                           const int lane, const ChainNeighbours nb)
{
    const bool deep_l = nb.il != 0xFFFFFFFFu && common_digits(nb.kl, kb) >= depth;
    const bool deep_r = nb.ir != 0xFFFFFFFFu && common_digits(nb.kr, kb) >= depth;
    if (__ballot((deep_l && nb.il < ib) || (deep_r && nb.ir < ib)) != 0ull) return 1;
    int out = 0;
    for (int side = 0; side < 2; side++) {
        if (__ballot(side == 0 ? deep_l : deep_r) != ~0ull) continue;   // the end of that cell (or of the array) was in sight
        for (int probe = 1; probe < kRivalProbes; probe++) {
            const int q = side == 0 ? p0 - 1 - probe * 64 - lane : p0 + t + probe * 64 + lane;
            const bool in = q >= 0 && q < n;
            const bool deep = in && common_digits(keys[in ? q : 0], kb) >= depth;
            const bool rival = deep && idx[q] < ib;
            if (__ballot(rival) != 0ull) return 1;
            if (__ballot(deep) != ~0ull) break;
            if (probe == kRivalProbes - 1) out = -1;
        }
    }
    return out;
}

// the bodies that start a segment, in any order (head_list; their number in *head_count).  One atomic per 1 024 bodies: one per wave
// -- 16 384 on one address at a million bodies once the system is dense -- took 6 .. 90 us.
constexpr int kHeadsBlock = 1024;
__global__ __launch_bounds__(kHeadsBlock) void k_chain_heads(const unsigned char* __restrict__ link, const int n, int* __restrict__ head_list,
                                                             int* __restrict__ head_count)
{
    __shared__ int wave_base[kHeadsBlock / 64 + 1];
    const int p = blockIdx.x * kHeadsBlock + threadIdx.x;
    const bool head = p < n && !chain_linked(link, p, n) && chain_linked(link, p + 1, n);
    const unsigned long long hm = __ballot(head);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_base[wave + 1] = __popcll(hm);
    __syncthreads();
    if (threadIdx.x == 0) {
        int total = 0;
        for (int w = 0; w < kHeadsBlock / 64; w++) { const int c = wave_base[w + 1]; wave_base[w + 1] = total; total += c; }
        wave_base[0] = total ? atomicAdd(head_count, total) : 0;
    }
    __syncthreads();
    if (head) head_list[wave_base[0] + wave_base[wave + 1] + __popcll(hm & ((1ull << lane) - 1ull))] = p;
}

// One wave per segment, taken off the list in strides of the grid: segments sit side by side in the sorted order wherever the system is
// dense (30 000 of them in the core of the collapsing 1 M-body model), and a wave per 32 sorted places that replays the ones starting
// there one after the other leaves those stretches to a few waves.
// the replay of the segment that starts at sorted body p0 (the whole wave; lane = threadIdx.x)
__device__ __forceinline__ void chain_segment(const int p0, const int lane, const unsigned long long* __restrict__ keys,
                                              const unsigned* __restrict__ idx, const float4* __restrict__ sb,
                                              const unsigned char* __restrict__ link, const int n, const unsigned* __restrict__ box,
                                              unsigned long long* __restrict__ out_keys, unsigned* __restrict__ out_idx,
                                              float4* __restrict__ out_sb, int& merged_total, int& approx_total)
{
    {
        // the segment: lanes 0 .. t - 1 <-> sorted bodies p0 .. p0 + t - 1
        const unsigned long long lw = __ballot(lane >= 1 && chain_linked(link, p0 + lane, n)) | 1ull;
        const int t = ~lw == 0ull ? 64 : __builtin_ctzll(~lw);               // (<= 60 by the cut rule)
        if (__ballot(lane >= 1 && lane < t && (link[p0 + lane] & 2)) == 0ull) return;   // nobody within EPS of anybody: as copied
        const bool member = lane < t;
        const int p = member ? p0 + lane : p0;
        const unsigned long long key = keys[p];
        const unsigned my_idx = idx[p];
        const float4 rec = sb[p];
        ChainNeighbours nb;                                  // the 64 sorted neighbours on either side of the segment
        {
            const int ql = p0 - 1 - lane, qr = p0 + t + lane;
            nb.kl = ql >= 0 ? keys[ql] : 0ull; nb.il = ql >= 0 ? idx[ql] : 0xFFFFFFFFu;
            nb.kr = qr < n ? keys[qr] : 0ull; nb.ir = qr < n ? idx[qr] : 0xFFFFFFFFu;
        }
        bool live = false;
        int blob = lane;                                     // the lane that heads my blob
        float cx = rec.x, cy = rec.y, cm = rec.w;            // (head lanes) the blob's running centre and mass, nbody.rs:303-320
        unsigned long long rep = key;                        // (head lanes) the path of that centre
        // The cell all members share (their keys' common digits: the first and the last suffice, the keys are sorted), replayed once:
        // a blob's centre lies in it, so its path is these digits + the rest of the descent -- a third of path_key's 31 levels
        const unsigned long long k_first = shfl_u64(key, 0), k_last = shfl_u64(key, t - 1);
        const int d0 = common_digits(k_first, k_last);
        float bx1 = dec_f32(box[0]), by1 = dec_f32(box[1]), bx2 = dec_f32(box[2]), by2 = dec_f32(box[3]);
#pragma unroll 1
        for (int l = 0; l < d0; l++) descend_digit(bx1, by1, bx2, by2, (int)((k_first >> (2 * (kLevels - 1 - l))) & 3ull));
        // arrival order: a body's place among the segment's indices (one pass of broadcasts; a wave minimum per step -- six dependent
        // cross-lane exchanges -- was a third of the loop)
        int place = 0;
        for (int l = 0; l < t; l++) place += (bcast_u32(my_idx, l) < my_idx) ? 1 : 0;
        // (exact: is the path in `rep` the centre's own down to level 31?  Inside the loop a path is only followed until it has parted
        //  from the keys of everybody still to come -- all that common(rep, key of an arrival) can ever see, a third of the levels;
        //  the blobs' full paths, which file them in the tree, are made once after the loop, all of them side by side)
        bool exact = true;
        for (int step = 0; step < t; step++) {
            const int j = __builtin_ctzll(__ballot(member && place == step));
            const float xb = bcast_f32(rec.x, j), yb = bcast_f32(rec.y, j);
            // nbody.rs:249 against every live entity's centre: nobody that close -> a new entity, whatever leaf it arrives at
            const bool close = live && fabsf(__fsub_rn(cx, xb)) < kEps && fabsf(__fsub_rn(cy, yb)) < kEps;
            const unsigned long long closem = __ballot(close);
            bool merged = false;
            if (closem) {
                const unsigned ib = bcast_u32(my_idx, j);
                const unsigned long long kb = shfl_u64(key, j);
                const int c = live ? common_digits(rep, kb) : -1;
                // the live entities deepest in the newcomer's path: the maximum of c (0 .. 31) bit by bit, five ballots
                unsigned long long who = __ballot(live);
                int cmax = 0;
#pragma unroll
                for (int bit = 4; bit >= 0; bit--) {
                    const unsigned long long m = __ballot(live && ((c >> bit) & 1)) & who;
                    if (m) { who = m; cmax |= 1 << bit; }
                }
                // the one entity whose leaf the newcomer arrives at, if there is one, and if it is close
                const int rival = __popcll(who) == 1 && (who & closem) ? chain_rival(keys, idx, n, p0, t, kb, ib, cmax, lane, nb) : 1;
                if (rival <= 0) {
                    if (rival < 0) approx_total++;
                    const int x = __builtin_ctzll(who);
                    const float mb = bcast_f32(rec.w, j);
                    if (lane == x) { fold_mass(cx, cy, cm, xb, yb, mb); exact = false; }
                    if (lane == j) blob = x;
                    merged = true;
                    merged_total++;
                    // the new centre's path from the segment's common cell down, every lane alongside: lane l holds it against its own
                    // key while its body is still to come (place > step), and the descent ends when nobody does any more
                    const float ncx = bcast_f32(cx, x), ncy = bcast_f32(cy, x);
                    unsigned long long kc;
                    if (d0 < kLevels && ncx >= bx1 && ncx < bx2 && ncy >= by1 && ncy < by2) {   // (half-open like quadrant_from_point)
                        float x1 = bx1, y1 = by1, x2 = bx2, y2 = by2;
                        kc = d0 ? k_first >> (2 * (kLevels - d0)) : 0ull;
                        bool follows = member && place > step;
                        int l = d0;
#pragma unroll 1
                        for (; l < kLevels && __ballot(follows) != 0ull; l++) {
                            const int q = descend(x1, y1, x2, y2, ncx, ncy);
                            kc = (kc << 2) | (unsigned long long)q;
                            follows = follows && (int)((key >> (2 * (kLevels - 1 - l))) & 3ull) == q;
                        }
                        kc <<= 2 * (kLevels - l);
                    } else {
                        kc = path_key(box, ncx, ncy);   // (a centre rounded out of the common cell, or one level-31 cell: from the root)
                    }
                    if (lane == x) rep = kc;
                }
            }
            if (lane == j) live = !merged;
        }
        // A piece of a longer chain: the blob at a cut that a pair within EPS spans may be one of the reference's cut in two -- its bodies
        // count as approximate (and, through counters[6], as left behind: beyond the class's limit the step goes to the host build)
        if (p0 > 0 && (link[p0] & 2)) approx_total += __popcll(__ballot(member && blob == bcast_i32(blob, 0)));
        if (p0 + t < n && (link[p0 + t] & 2)) approx_total += __popcll(__ballot(member && blob == bcast_i32(blob, t - 1)));
        // the blobs' exact paths (heads whose centre moved), side by side
        {
            const bool redo = member && live && !exact;
            if (__ballot(redo)) {
                const bool inside = redo && d0 < kLevels && cx >= bx1 && cx < bx2 && cy >= by1 && cy < by2;
                if (__ballot(redo && !inside)) {
                    if (redo) rep = path_key(box, cx, cy);
                } else if (redo) {
                    float x1 = bx1, y1 = by1, x2 = bx2, y2 = by2;
                    unsigned long long kc = d0 ? k_first >> (2 * (kLevels - d0)) : 0ull;
#pragma unroll 1
                    for (int l = d0; l < kLevels; l++) kc = (kc << 2) | (unsigned long long)descend(x1, y1, x2, y2, cx, cy);
                    rep = kc;
                }
            }
        }
        // One key per blob: the path of its CENTRE -- the reference files a blob where its centre is (nbody.rs:271-281); every centre
        // since the blob's last split lies in its leaf's cell, so the last one's path is as good as any -- unless that path lies
        // outside the segment's place in the sorted order (the centre has left the cells of all members, and their neighbours'):
        // then the member's whose path follows the centre's deepest (first of them in key order).
        const unsigned long long k_prev = p0 > 0 ? keys[p0 - 1] : 0ull, k_next = p0 + t < n ? keys[p0 + t] : ~0ull;
        unsigned long long out_key = key;
        unsigned long long multi = __ballot(member && blob != lane);
        while (multi) {
            const int x = bcast_i32(blob, __builtin_ctzll(multi));
            const bool mine = member && blob == x;
            multi &= ~__ballot(mine);
            const unsigned long long rx = shfl_u64(rep, x);
            unsigned long long k = rx;
            if (!((p0 == 0 || k_prev < rx) && (p0 + t >= n || rx < k_next))) {
                const int c = mine ? common_digits(rx, key) : -1;
                const int best = wave_max_i32(c);
                k = shfl_u64(key, __builtin_ctzll(__ballot(mine && c == best)));
            }
            if (mine) out_key = k;
        }
        // blobs in key order, a blob's bodies in arrival order.  Two blobs under ONE key -- both filed under a member's path, and the
        // members share a level-31 cell: bodies of one position whose folded centre drifted an ulp, which where an ulp is no longer
        // small against EPS (|x| in the thousands) makes the reference split them, nbody.rs:315-317 -- become one leaf here: counted
        // as approximate
        int rank = 0;
        bool collides = false;
        for (int l = 0; l < t; l++) {
            const unsigned long long ko = shfl_u64(out_key, l);
            const unsigned io = bcast_u32(my_idx, l);
            rank += (ko < out_key || (ko == out_key && io < my_idx)) ? 1 : 0;
            collides = collides || (ko == out_key && bcast_i32(blob, l) != blob);
        }
        approx_total += __popcll(__ballot(member && collides));
        if (member) { out_keys[p0 + rank] = out_key; out_idx[p0 + rank] = my_idx; out_sb[p0 + rank] = rec; }
    }
}

// the tallies: over kTallySlots words each, summed into counters[6] / [7] by k_scan_write's last workgroup -- as one atomic per wave on
// ONE word they were k_chain: 37 us for 2 600 of them at a million bodies (6 700 replay steps in all), 100 us for 8 000
__device__ __forceinline__ void chain_tally(int* __restrict__ tally, const int lane, const int merged_total, const int approx_total)
{
    if (lane == 0) {
        if (merged_total) atomicAdd(&tally[(blockIdx.x % kTallySlots) * kTallyStride + 1], merged_total);
        if (approx_total) atomicAdd(&tally[(blockIdx.x % kTallySlots) * kTallyStride], approx_total);
    }
}

__global__ __launch_bounds__(64) void k_chain(const unsigned long long* __restrict__ keys, const unsigned* __restrict__ idx,
                                              const float4* __restrict__ sb, const unsigned char* __restrict__ link, const int n,
                                              const unsigned* __restrict__ box, unsigned long long* __restrict__ out_keys,
                                              unsigned* __restrict__ out_idx, float4* __restrict__ out_sb,
                                              const int* __restrict__ head_list, const int* __restrict__ head_count, int* __restrict__ tally)
{
    const int lane = threadIdx.x;
    const int count = *head_count;
    int merged_total = 0, approx_total = 0;
    for (int h = blockIdx.x; h < count; h += (int)gridDim.x)
        chain_segment(head_list[h], lane, keys, idx, sb, link, n, box, out_keys, out_idx, out_sb, merged_total, approx_total);
    chain_tally(tally, lane, merged_total, approx_total);
}

// Small systems (the two-launch front's, up to kSmallFrontMax bodies): no list -- a wave finds the segments that start in its 16 sorted
// places itself and replays them one after the other; one launch less on a step that is ten launches of 4-25 us (the reference's own
// scene: 10 000 bodies, 0.10 ms).
constexpr int kSmallBlock = 16;   // sorted places per wave of k_chain_small (64: the 10 000-body disc's core left 157 waves 0.17 ms per step, from 0.14)
__global__ __launch_bounds__(64) void k_chain_small(const unsigned long long* __restrict__ keys, const unsigned* __restrict__ idx,
                                                    const float4* __restrict__ sb, const unsigned char* __restrict__ link, const int n,
                                                    const unsigned* __restrict__ box, unsigned long long* __restrict__ out_keys,
                                                    unsigned* __restrict__ out_idx, float4* __restrict__ out_sb, int* __restrict__ tally)
{
    const int lane = threadIdx.x;
    const int first = blockIdx.x * kSmallBlock;
    // bit l: boundary first + l (lanes 0 .. kSmallBlock); a body starts a segment when its own boundary is not linked and the next one is
    const unsigned long long la = __ballot(lane <= kSmallBlock && chain_linked(link, first + lane, n));
    unsigned long long heads = ~la & (la >> 1) & ((1ull << kSmallBlock) - 1ull);
    int merged_total = 0, approx_total = 0;
    while (heads) {
        const int p0 = first + __builtin_ctzll(heads);
        heads &= heads - 1;
        chain_segment(p0, lane, keys, idx, sb, link, n, box, out_keys, out_idx, out_sb, merged_total, approx_total);
    }
    chain_tally(tally, lane, merged_total, approx_total);
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
                                                      int* __restrict__ counters, const int* __restrict__ tally)
{
    if (tally && blockIdx.x == gridDim.x - 1) {   // the chain replay's tallies (k_chain), one slot per thread -> counters[6], [7]
        static_assert(kTallySlots == kTile, "one slot per thread");
        __shared__ int tsum[2][kTile / 64];
        int a = tally[kTallyStride * threadIdx.x], m = tally[kTallyStride * threadIdx.x + 1];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); m += __shfl_xor(m, off); }
        if ((threadIdx.x & 63) == 0) { tsum[0][threadIdx.x >> 6] = a; tsum[1][threadIdx.x >> 6] = m; }
        __syncthreads();
        if (threadIdx.x == 0) {
            int ta = 0, tm = 0;
            for (int w = 0; w < kTile / 64; w++) { ta += tsum[0][w]; tm += tsum[1][w]; }
            counters[6] = ta; counters[7] = tm;
            if (ta) atomicAdd(&counters[1], ta);   // what the replay only approximated counts as left behind: beyond the class's limit -> host build
        }
    }
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
    const int* tally = nullptr;
    if (fold == 1) {
        // blobs of any size, replayed (3c): the tree is then the reference's, node for node -- or the step is refused
        if ((e = launch_cluster_replay(posm, n, k, stream, sb_ready)) != hipSuccess) return e;
        mi = k.idx0; ms = k.sb2; pmin = k.pmin2;
    } else {
        // chains of close bodies replayed in arrival order (3b): links from the sorted bodies (this kernel also gathers them into
        // sorted order), then one wave per 32 sorted places regroups what the reference merges
        hipLaunchKernelGGL(k_chain_links, dim3(nb), dim3(kTile), 0, stream, posm, k.sb, k.keys1, k.idx1, n, k.link, k.counters + 1, sb_ready ? 1 : 0,
                           k.keys0, k.idx0, k.sb2, k.ghosts);
        if (n <= kSmallFrontMax) {
            hipLaunchKernelGGL(k_chain_small, dim3((unsigned)((n + kSmallBlock - 1) / kSmallBlock)), dim3(64), 0, stream, k.keys1, k.idx1, k.sb, k.link, n, k.box, k.keys0,
                               k.idx0, k.sb2, k.ghosts);
        } else {
            int* const head_list = reinterpret_cast<int*>(k.big);          // (k_fold_big's queue: the other class's)
            hipLaunchKernelGGL(k_chain_heads, dim3((unsigned)((n + kHeadsBlock - 1) / kHeadsBlock)), dim3(kHeadsBlock), 0, stream, k.link, n, head_list, k.counters + 4);
            // (a wave per 16 bodies, at most 16 384: a wave without a segment reads the count and leaves -- with a wave per 128 bodies the 5 000
            //  segments of nb_random_disk(65536)'s core had 512 waves to share: k_chain 80 us of a 0.26 ms step)
            const int waves = n / 16 < 256 ? 256 : (n / 16 > 16384 ? 16384 : n / 16);
            hipLaunchKernelGGL(k_chain, dim3((unsigned)waves), dim3(64), 0, stream, k.keys1, k.idx1, k.sb, k.link, n, k.box, k.keys0, k.idx0, k.sb2,
                               head_list, k.counters + 4, k.ghosts);
        }
        tally = k.ghosts;
        mi = k.idx0; ms = k.sb2;
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
    hipLaunchKernelGGL(k_scan_write, dim3(sb), dim3(kTile), 0, stream, ms, mk, n, k.block_sums, k.pre, k.counters, tally);
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
