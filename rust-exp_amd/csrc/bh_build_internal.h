// bh_build_internal.h -- what the units of the device tree build share (internal; see bh_build.hip for the whole picture):
//   bh_front.hip    root box, path keys, the sorts' dispatch (small systems: two launches; cold: library sort), workspace, helpers for
//                   the host build and for sharded engines
//   bh_sort.hip     the sort that starts from last step's order (round 5)
//   bh_build.hip    EPS merge of pairs, scans, node records (k_emit), the build's entry points
//   bh_cluster.hip  the reference's EPS merge in full: clusters replayed in arrival order (reference fold)
//   bh_fold.hip     the reference's running fold of big nodes and of the root
#pragma once
#include "kernels.h"

namespace nbx {

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

// The root box from k_bbox's partial boxes (at most 256: the launcher caps its grid), folded by EVERY workgroup of the first kernel
// that needs it (k_keys cold, k_sample_rank warm) -- collective over a workgroup of kTile threads.  Round 5: k_bbox used to fold
// them itself in its last workgroup, behind a ticket and two device-wide fences (each writes the L2 back on this chip): 11 us for
// a pass over 16 MB; min and max are exact in any order, so who folds changes no bit.  publish: this workgroup also files the box
// (encoded) for the kernels behind.
__device__ __forceinline__ void fold_box_partials(const float4* __restrict__ part, const int parts, unsigned* __restrict__ box,
                                                  const bool publish, float& x1, float& y1, float& x2, float& y2)
{
    __shared__ float red_box[kTile / 64][4];
    x1 = 3.40282347e+38f; y1 = 3.40282347e+38f; x2 = -3.40282347e+38f; y2 = -3.40282347e+38f;
    for (int b = threadIdx.x; b < parts; b += kTile) {
        const float4 q = part[b];
        x1 = fminf(x1, q.x); y1 = fminf(y1, q.y); x2 = fmaxf(x2, q.z); y2 = fmaxf(y2, q.w);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        x1 = fminf(x1, __shfl_xor(x1, off)); y1 = fminf(y1, __shfl_xor(y1, off));
        x2 = fmaxf(x2, __shfl_xor(x2, off)); y2 = fmaxf(y2, __shfl_xor(y2, off));
    }
    if ((threadIdx.x & 63) == 0) { red_box[threadIdx.x >> 6][0] = x1; red_box[threadIdx.x >> 6][1] = y1; red_box[threadIdx.x >> 6][2] = x2; red_box[threadIdx.x >> 6][3] = y2; }
    __syncthreads();
    x1 = red_box[0][0]; y1 = red_box[0][1]; x2 = red_box[0][2]; y2 = red_box[0][3];
#pragma unroll
    for (int w = 1; w < kTile / 64; w++) {
        x1 = fminf(x1, red_box[w][0]); y1 = fminf(y1, red_box[w][1]); x2 = fmaxf(x2, red_box[w][2]); y2 = fmaxf(y2, red_box[w][3]);
    }
    if (publish && threadIdx.x == 0) { box[0] = enc_f32(x1); box[1] = enc_f32(y1); box[2] = enc_f32(x2); box[3] = enc_f32(y2); }
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


int inc_buckets(int n);             // bh_sort.hip: buckets of the warm sort, and whether n bodies take it
bool inc_sort_enabled(int n);

// limits of the cluster replay (bh_cluster.hip) and why a build refuses
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

// Workspace header (the first 4 KiB + 256 B): ints [0] node count, [1] bodies the pairs-only EPS merge left behind (or blobs whose
// centre left their first member's cell), [2] nodes queued for k_fold_big, [3] ticket of k_scan_reduce (systems up to kTicketScanMax bodies) -- all cleared by k_keys / k_bbox at every build -- [9] the "poison" flag of the gated steps (kernels.h; zeroed once by device_tree_workspace_init),
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
// (kWhySortOverflow = 1 << 20: kernels.h -- the host layer reads it too)

constexpr unsigned long long kPadKey = ~0ull;   // (real keys occupy 62 bits)
__device__ __forceinline__ bool pair_less(const unsigned long long ka, const unsigned ia, const unsigned long long kb, const unsigned ib)
{
    return ka < kb || (ka == kb && ia < ib);
}

// slots of the grid-cell table: a power of two, at least two per body (one cell per body at most)
inline size_t cell_table_slots(int n)
{
    size_t h = 1024;
    while (h < 2 * (size_t)n) h <<= 1;
    return h;
}

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
    int* gcount;                  // round 5, warm sort: per-bucket counts, the buckets' slots, the splitter candidates and their ranks
    ulonglong2* slots;            // kBucketCap (key, index) slots per bucket
    float4* slot_recs;            // ... and the bodies' records beside them (delivered to sb in sorted order by k_bucket_sort)
    unsigned long long* skeys;
    int* srank;                   // [kMaxSamples] ranks of the splitter candidates
};
inline Workspace carve(void* workspace, int n, size_t sort_tmp, int node_cap)
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
    const size_t pstride = ((size_t)n + 2) & ~(size_t)1;   // (even: every prefix array starts 16-byte aligned for k_scan_write's vector stores)
    double* d = reinterpret_cast<double*>(take(sizeof(double) * pstride * 3));
    k.pre.m = d; k.pre.mx = d + pstride; k.pre.my = d + 2 * pstride;
    k.pre.base = reinterpret_cast<int*>(take(sizeof(int) * ((size_t)n + 1)));
    k.pre.ent = reinterpret_cast<int*>(take(sizeof(int) * ((size_t)n + 1)));
    k.pre.cnt = reinterpret_cast<unsigned char*>(take((size_t)n));
    k.pre.owner = reinterpret_cast<int*>(take(sizeof(int) * (size_t)node_cap));
    k.pre.owner_cap = node_cap;
    k.block_sums = reinterpret_cast<ScanItem*>(take(sizeof(ScanItem) * (nb + 1)));
    k.link = reinterpret_cast<unsigned char*>(take((size_t)n + 16));   // (+ 16: chain_linked reads 16 bytes at a time, bh_build.hip)
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
    k.gcount = reinterpret_cast<int*>(take(sizeof(int) * kMaxBuckets));
    k.skeys = reinterpret_cast<unsigned long long*>(take(sizeof(unsigned long long) * kMaxSamples));
    k.srank = reinterpret_cast<int*>(take(sizeof(int) * (kMaxSamples + 64)));
    const size_t slots_inc = inc_sort_enabled(n) ? (size_t)inc_buckets(n) * kBucketCap : 0;
    k.slots = reinterpret_cast<ulonglong2*>(take(sizeof(ulonglong2) * slots_inc));
    k.slot_recs = reinterpret_cast<float4*>(take(sizeof(float4) * slots_inc));
    return k;
}


// ---- between the units ------------------------------------------------------------------------------------------------------
hipError_t launch_inc_sort(const float4* posm, const float4* sorted_pos, int n, unsigned* box, const float4* part, int parts, const unsigned* perm, int* gcount,
                           unsigned long long* skeys, int* srank, ulonglong2* slots, float4* slot_recs, unsigned long long* keys_out,
                           unsigned* idx_out, float4* sb_out, int* counters, unsigned long long* cell_table, int cell_slots, hipStream_t stream);
// root AABB -> path keys -> sorted (key, body) pairs in keys1 / idx1 (bh_front.hip)
// want_sb: also deliver the bodies' records in sorted order (k.sb[j] = posm[idx1[j]]) if the sort can do that on its way (the warm
// sort can); *sb_ready says whether it did
hipError_t sort_bodies(const float4* posm, int n, const Workspace& k, size_t sort_tmp, hipStream_t stream, bool cell_table, bool warm,
                       const float4* sorted_pos = nullptr, bool want_sb = false, bool* sb_ready = nullptr);
// the reference's EPS merge in full (bh_cluster.hip): entities in k.keys0 / k.idx0 / k.sb2 / k.pmin2
hipError_t launch_cluster_replay(const float4* posm, int n, const Workspace& k, hipStream_t stream, bool sb_ready);
// the reference's running fold (bh_fold.hip): the root on its own stream, the queued nodes behind k_emit
void launch_fold_root(const float4* posm, int n, BhNode* out, hipStream_t side);
void launch_fold_big(const float4* posm, const float4* sb, const unsigned* idx, const int4* big, int big_cap, const int* counters, int n,
                     BhNode* out, hipStream_t stream);

}  // namespace nbx
