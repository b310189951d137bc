// bh_front.hip -- the front of the device tree build (bh_build.hip has the whole picture): root AABB, path keys, the dispatch
// between the three sorts (small systems: box, keys and sort in two launches; warm: bh_sort.hip; cold: the library's), the
// workspace, the Morton order alone (host-built trees, sharded engines) and the routing help for the host build.
#include <atomic>
#include <cstring>   // rocPRIM's texture_cache_iterator.hpp calls memset() without including it

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>

#include "bh_build_internal.h"

namespace nbx {

// rocPRIM's (key, index) sort for systems above kSmallFrontMax bodies: its merge-sort path (the library's choice up to 2^20 pairs)
// with first-level blocks of 512 x 8 pairs instead of 256 x 4 -- two merge passes fewer: 183 vs 206 us at 1 048 576 pairs,
// 109 vs 111 at 262 144 (tools/ubench_sort_cfg.hip, profiles/r04_ubench_sort_cfg.txt)
constexpr int kBigSortFrom = 262144;   // (below: the library's own shape -- 65 536 pairs lose 10 us to the bigger blocks, too few of them)
using BuildSortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::merge_sort_config<512, 512, 8, 128, 128, 4>,
                                                   rocprim::default_config, (size_t)1 << 20>;

// Partial boxes: workgroup b leaves (min x, min y, max x, max y) of its share of the bodies in part[b]; the first kernel behind
// folds them (fold_box_partials) and files box[0..3] = enc(min x), enc(min y), enc(max x), enc(max y).
// clear_*: words the kernels BEHIND this one add to (warm sort: the splitter candidates' ranks, the buckets' counts, the build's
// counters -- which k_keys clears in the cold path), cleared here to save a launch
__global__ __launch_bounds__(kTile) void k_bbox(const float4* __restrict__ posm, const int n, float4* __restrict__ part,
                                                int* __restrict__ clear_a, const int count_a, int* __restrict__ clear_b, const int count_b,
                                                int* __restrict__ clear_c, const int count_c)
{
    for (int i = blockIdx.x * kTile + threadIdx.x; i < count_a; i += (int)gridDim.x * kTile) clear_a[i] = 0;
    for (int i = blockIdx.x * kTile + threadIdx.x; i < count_b; i += (int)gridDim.x * kTile) clear_b[i] = 0;
    for (int i = blockIdx.x * kTile + threadIdx.x; i < count_c; i += (int)gridDim.x * kTile) clear_c[i] = 0;
    float x1 = 3.40282347e+38f, y1 = 3.40282347e+38f, x2 = -3.40282347e+38f, y2 = -3.40282347e+38f;
    {   // eight independent loads in flight per thread (one at a time, a million bodies took 12 us -- sixteen dependent round
        // trips per thread; min and max are exact in any order)
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
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wave][0] = x1; red[wave][1] = y1; red[wave][2] = x2; red[wave][3] = y2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) {
            x1 = fminf(x1, red[w][0]); y1 = fminf(y1, red[w][1]); x2 = fmaxf(x2, red[w][2]); y2 = fmaxf(y2, red[w][3]);
        }
        part[blockIdx.x] = make_float4(x1, y1, x2, y2);
    }
}

__global__ __launch_bounds__(kTile) void k_keys(const float4* __restrict__ posm, const int n, const float4* __restrict__ part,
                                                const int parts, unsigned* __restrict__ box, unsigned long long* __restrict__ keys,
                                                unsigned* __restrict__ idx, int* __restrict__ counters,
                                                unsigned long long* __restrict__ cell_table, const int cell_slots)
{
    const int i = blockIdx.x * kTile + threadIdx.x;
    // this build's counters and tickets (see Workspace): cleared here instead of by a memset of their own
    if (i < 8) counters[i] = 0;
    // ... and the table of occupied grid cells that k_cells fills after the sort (reference fold only; at most 4 slots per body)
    for (int t = i; t < cell_slots; t += (int)gridDim.x * kTile) cell_table[t] = 0ull;
    float x1, y1, x2, y2;
    fold_box_partials(part, parts, box, blockIdx.x == 0, x1, y1, x2, y2);
    if (i >= n) return;
    const float4 p = posm[i];
    unsigned long long key = 0;
#pragma unroll 1
    for (int l = 0; l < kLevels; l++) key = (key << 2) | (unsigned long long)descend(x1, y1, x2, y2, p.x, p.y);
    keys[i] = key;
    idx[i] = (unsigned)i;
}

// the library sort's temporary storage for n pairs (both shapes: which one runs depends on n alone, but n may shrink below the
// switch).  The two size queries are host work inside the library: asked once per n, not on every step (a thread-local memo of
// the last answer -- a step is ~0.1 ms of enqueueing)
static size_t library_sort_tmp_bytes(int n)
{
    thread_local int memo_n = -1;
    thread_local size_t memo_bytes = 0;
    if (n == memo_n) return memo_bytes;
    size_t tmp = 0, tmp_small = 0;
    (void)rocprim::radix_sort_pairs<BuildSortConfig>(nullptr, tmp, (unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                                     (unsigned*)nullptr, (unsigned*)nullptr, (size_t)n, 0, 2 * kLevels, (hipStream_t)0);
    (void)rocprim::radix_sort_pairs(nullptr, tmp_small, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (unsigned*)nullptr,
                                    (unsigned*)nullptr, (size_t)n, 0, 2 * kLevels, (hipStream_t)0);
    memo_n = n;
    memo_bytes = tmp_small > tmp ? tmp_small : tmp;
    return memo_bytes;
}

size_t device_tree_workspace_bytes(int n, int node_cap, size_t* sort_tmp_bytes)
{
    const size_t tmp = library_sort_tmp_bytes(n);   // (small systems too: the library sort is what a refused two-launch front falls back to)
    if (sort_tmp_bytes) *sort_tmp_bytes = tmp;
    const size_t nb = ((size_t)n + kScanBlock - 1) / kScanBlock;
    size_t bytes = 0;
    auto add = [&](size_t b) { bytes += (b + 255) & ~(size_t)255; };
    add(kHeaderBytes);                                 // counters, tickets, box, partial boxes
    add(sizeof(unsigned long long) * (size_t)n * 2);   // keys in/out
    add(sizeof(unsigned) * (size_t)n * 2);             // idx in/out
    add(tmp);
    add(sizeof(float4) * (size_t)n);                   // sorted bodies
    add(sizeof(double) * (((size_t)n + 2) & ~(size_t)1) * 3);   // prefix sums m, m*x, m*y (each 16-byte aligned)
    add(sizeof(int) * ((size_t)n + 1));                // pre-order base
    add(sizeof(int) * ((size_t)n + 1));                // entities before every body
    add((size_t)n);                                    // nodes starting at every body (cache between the two scan kernels)
    add(sizeof(int) * (size_t)node_cap);               // owner body of every node slot
    add(sizeof(ScanItem) * (nb + 1));                  // block sums
    add((size_t)n + 16);                               // EPS-merge links / pmin
    add(sizeof(int4) * (size_t)n);                     // nodes queued for k_fold_big (more than n of them -> host build)
    // reference fold: the EPS blobs (k_cells / k_blobs / k_place)
    add(sizeof(float4) * (size_t)n);                   // bodies in entity order
    add(sizeof(unsigned long long) * (size_t)n);       // entity keys
    add((size_t)n);                                    // pmin in entity order
    add((sizeof(unsigned long long) + sizeof(int)) * cell_table_slots(n));
    add(sizeof(int) * kGhostCap);
    add(sizeof(int) * kMaxBuckets);                    // warm sort (round 5): pairs per bucket
    add(sizeof(unsigned long long) * kMaxSamples);     // ... splitter candidates
    add(sizeof(int) * (kMaxSamples + 64));             // ... their ranks, one ticket per row of the ranking
    const size_t slots_inc = inc_sort_enabled(n) ? (size_t)inc_buckets(n) * kBucketCap : 0;
    add(sizeof(ulonglong2) * slots_inc);               // ... the buckets' slots
    add(sizeof(float4) * slots_inc);                   // ... the records beside them
    return bytes;
}

// once per (re)allocation of the workspace: the poison flag of the gated steps and the scan's ticket start at zero
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

// root AABB -> path keys -> sorted (key, body) pairs in keys1 / idx1
hipError_t sort_bodies(const float4* posm, int n, const Workspace& k, size_t sort_tmp, hipStream_t stream, bool cell_table, bool warm,
                       const float4* sorted_pos, bool want_sb, bool* sb_ready)
{
    if (sb_ready) *sb_ready = false;
    if (small_front_enabled(n)) {   // a small system (the reference's own 10 000 bodies): two launches instead of seven, no library sort
        const hipError_t e = launch_front_small(posm, n, k.box, k.keys0, k.idx0, k.keys1, k.idx1, k.counters, k.hk,
                                                cell_table ? (int)(k.hmask + 1u) : 0, stream);
        if (e == hipSuccess) return e;
        (void)hipGetLastError();    // the LDS opt-in or a launch was refused (nothing ran): the general path below serves any size
    }
    const int nb = (n + kTile - 1) / kTile;
    const int parts = nb < 256 ? nb : 256;
    if (warm && inc_sort_enabled(n)) {   // idx1 holds last step's order: sort from there (round 5)
        hipLaunchKernelGGL(k_bbox, dim3(parts), dim3(kTile), 0, stream, posm, n, k.part, k.srank, kOversample * inc_buckets(n), k.gcount,
                           inc_buckets(n), k.counters, 8);
        if (sb_ready) *sb_ready = want_sb;
        return launch_inc_sort(posm, sorted_pos, n, k.box, k.part, parts, k.idx1, k.gcount, k.skeys, k.srank, k.slots, k.slot_recs, k.keys1, k.idx1,
                               want_sb ? k.sb : nullptr, k.counters, k.hk, cell_table ? (int)(k.hmask + 1u) : 0, stream);
    }
    hipLaunchKernelGGL(k_bbox, dim3(parts), dim3(kTile), 0, stream, posm, n, k.part, (int*)nullptr, 0, (int*)nullptr, 0, (int*)nullptr, 0);
    hipLaunchKernelGGL(k_keys, dim3(nb), dim3(kTile), 0, stream, posm, n, k.part, parts, k.box, k.keys0, k.idx0, k.counters, k.hk,
                       cell_table ? (int)(k.hmask + 1u) : 0);
    if (n >= kBigSortFrom)
        return rocprim::radix_sort_pairs<BuildSortConfig>(k.sort_tmp, sort_tmp, k.keys0, k.keys1, k.idx0, k.idx1, (size_t)n, 0, 2 * kLevels, stream);
    return rocprim::radix_sort_pairs(k.sort_tmp, sort_tmp, k.keys0, k.keys1, k.idx0, k.idx1, (size_t)n, 0, 2 * kLevels, stream);
}

// Spatial (Morton, reference quadrant order) permutation of the bodies only: bbox + path keys + radix sort.
// Used to make the traversal of a HOST-built tree wave-coherent. *perm_dev points into the workspace.
hipError_t device_spatial_order(const float4* posm, int n, void* workspace, size_t workspace_bytes, const unsigned** perm_dev,
                                hipStream_t stream, bool warm, const float4* sorted_pos)
{
    *perm_dev = nullptr;
    if (n <= 0) return hipSuccess;
    size_t sort_tmp = 0;
    if (device_tree_workspace_bytes(n, 1, &sort_tmp) > workspace_bytes) return hipErrorInvalidValue;
    const Workspace k = carve(workspace, n, sort_tmp, 1);
    const hipError_t e = sort_bodies(posm, n, k, sort_tmp, stream, false, warm, sorted_pos);
    if (e != hipSuccess) return e;
    *perm_dev = k.idx1;
    return hipGetLastError();
}

// ---- help for the HOST quadtree build: routing + stable scatter on the device ------------------------------------------
//
// The threaded host build (host_tree.cpp) inserts the first `warm` bodies sequentially, freezes the top levels, and then
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

}  // namespace nbx
