// bh_build.hip -- quadtree build ON THE DEVICE (opt-in, NBX_OPT_BH_TREE = 1).
//
// The default Barnes-Hut path builds the tree on the host exactly as the reference does (nbody.rs:388-415:
// sequential insertion, running centre of mass) -- bit-faithful but ~60 ms per step at 1 M bodies.  This file
// builds the SAME tree shape without leaving the GPU:
//
//   1. root AABB = min/max of positions (exact; nbody.rs:388-398)
//   2. per body: the path of quadrant choices, replaying quadrant_from_point / create_children with the
//      reference's own f32 midpoint arithmetic (cx = (x1+x2)*0.5, nbody.rs:289-290, :324-331) for 31 levels
//      -> 62-bit key, 2 bits per level, quadrant order [UL,UR,LL,LR] = 0..3 like the reference's child array
//   3. radix sort (rocPRIM) of (key, body index)
//   4. breadth-first: a node = a range of the sorted bodies sharing a key prefix; a node with >= 2 bodies is
//      interior (like the reference: it splits as soon as a second, non-merged particle arrives) and its
//      non-empty children are found by binary search on the next 2-bit digit
//   5. bottom-up: subtree sizes and centres of mass (children folded left to right with the reference's
//      running formula (p*m + p'*m') * (1/(m+m')), nbody.rs:315-318)
//   6. top-down: pre-order offsets;  7. emit the flattened BhNode array the traversal kernels already walk.
//
// Same node set, same s = x2-x1 per node (box replayed with the same f32 arithmetic), same leaf records as the
// host build + flatten.  What differs (its own tolerance class): interior centres of
// mass are folded child by child instead of particle by particle in index order (rounding-level), bodies
// closer than EPS are NOT merged (nbody.rs:249-260 merges them in arrival order; here they get their own
// leaves a few levels deeper), bodies identical down to level 31 share one leaf, and there is no depth-50
// panic.  Hence: fast mode only; the bit-exact mode keeps the host build.
#include <cstring>   // rocPRIM's texture_cache_iterator.hpp calls memset() without including it

#include <rocprim/device/device_radix_sort.hpp>

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

// box[0..3] = enc(min x), enc(min y), enc(max x), enc(max y); initialised to {~0,~0,0,0} by the launcher
__global__ __launch_bounds__(kTile) void k_bbox(const float4* __restrict__ posm, const int n, unsigned* box)
{
    float x1 = 3.40282347e+38f, y1 = 3.40282347e+38f, x2 = -3.40282347e+38f, y2 = -3.40282347e+38f;
    for (int i = blockIdx.x * kTile + threadIdx.x; i < n; i += gridDim.x * kTile) {
        const float4 p = posm[i];
        x1 = fminf(x1, p.x); y1 = fminf(y1, p.y); x2 = fmaxf(x2, p.x); y2 = fmaxf(y2, p.y);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        x1 = fminf(x1, __shfl_xor(x1, off)); y1 = fminf(y1, __shfl_xor(y1, off));
        x2 = fmaxf(x2, __shfl_xor(x2, off)); y2 = fmaxf(y2, __shfl_xor(y2, off));
    }
    // one set of atomics per WORKGROUP (the launcher caps the grid at 256 blocks): per-wave atomics on four
    // words cost 0.19 ms at 1 M bodies
    __shared__ float red[4][4];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wave][0] = x1; red[wave][1] = y1; red[wave][2] = x2; red[wave][3] = y2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) {
            x1 = fminf(x1, red[w][0]); y1 = fminf(y1, red[w][1]); x2 = fmaxf(x2, red[w][2]); y2 = fmaxf(y2, red[w][3]);
        }
        atomicMin(&box[0], enc_f32(x1)); atomicMin(&box[1], enc_f32(y1));
        atomicMax(&box[2], enc_f32(x2)); atomicMax(&box[3], enc_f32(y2));
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

__global__ __launch_bounds__(kTile) void k_keys(const float4* __restrict__ posm, const int n,
                                                const unsigned* __restrict__ box, unsigned long long* __restrict__ keys,
                                                unsigned* __restrict__ idx)
{
    const int i = blockIdx.x * kTile + threadIdx.x;
    if (i >= n) return;
    float x1 = dec_f32(box[0]), y1 = dec_f32(box[1]), x2 = dec_f32(box[2]), y2 = dec_f32(box[3]);
    const float4 p = posm[i];
    unsigned long long key = 0;
#pragma unroll 1
    for (int l = 0; l < kLevels; l++) key = (key << 2) | (unsigned long long)descend(x1, y1, x2, y2, p.x, p.y);
    keys[i] = key;
    idx[i] = (unsigned)i;
}

struct TreeArrays {
    int* lo;       // first sorted body of the node
    int* hi;       // one past the last
    int* level;
    int* child0;   // first child node id (children are consecutive, quadrant order) or -1
    int* nchild;
    int* size;     // nodes in the subtree (pre-order span)
    int* offset;   // pre-order position
    float4* com;   // px, py, m, -
};

__device__ __forceinline__ int lower_bound_digit(const unsigned long long* keys, int lo, int hi, int shift, unsigned d)
{
    while (lo < hi) {   // first index whose digit >= d
        const int mid = (lo + hi) >> 1;
        if (((unsigned)(keys[mid] >> shift) & 3u) < d) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// split the nodes [first, last) of one level; children are appended at *node_count
__global__ __launch_bounds__(kTile) void k_split_level(const unsigned long long* __restrict__ keys, TreeArrays t,
                                                       const int first, const int last, int* node_count, const int cap,
                                                       int* overflow)
{
    const int k = first + blockIdx.x * kTile + threadIdx.x;
    if (k >= last) return;
    const int lo = t.lo[k], hi = t.hi[k], lvl = t.level[k];
    t.child0[k] = -1;
    t.nchild[k] = 0;
    if (hi - lo < 2 || lvl >= kLevels) return;   // exterior (single body, or bodies identical to 31 levels)
    const int shift = 2 * (kLevels - 1 - lvl);
    int b[5];
    b[0] = lo; b[4] = hi;
    b[1] = lower_bound_digit(keys, lo, hi, shift, 1u);
    b[2] = lower_bound_digit(keys, b[1], hi, shift, 2u);
    b[3] = lower_bound_digit(keys, b[2], hi, shift, 3u);
    int cnt = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) cnt += b[q + 1] > b[q];
    const int base = atomicAdd(node_count, cnt);
    if (base + cnt > cap) { *overflow = 1; return; }
    t.child0[k] = base;
    t.nchild[k] = cnt;
    int c = base;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        if (b[q + 1] > b[q]) {
            t.lo[c] = b[q]; t.hi[c] = b[q + 1]; t.level[c] = lvl + 1;
            c++;
        }
    }
}

__device__ __forceinline__ void fold_mass(float& px, float& py, float& m, const float qx, const float qy, const float qm)
{
    if (m == 0.0f) { px = qx; py = qy; m = qm; return; }                 // nbody.rs:305-311
    const float inv = 1.0f / __fadd_rn(m, qm);                            // :315
    px = __fmul_rn(__fadd_rn(__fmul_rn(px, m), __fmul_rn(qx, qm)), inv);  // :316
    py = __fmul_rn(__fadd_rn(__fmul_rn(py, m), __fmul_rn(qy, qm)), inv);  // :317
    m = __fadd_rn(m, qm);                                                 // :318
}

// bottom-up over one level: subtree size + centre of mass
__global__ __launch_bounds__(kTile) void k_up_level(const float4* __restrict__ posm, const unsigned* __restrict__ idx,
                                                    TreeArrays t, const int first, const int last)
{
    const int k = first + blockIdx.x * kTile + threadIdx.x;
    if (k >= last) return;
    float px = 0.f, py = 0.f, m = 0.f;
    int size = 1;
    const int nc = t.nchild[k];
    if (nc == 0) {
        for (int i = t.lo[k]; i < t.hi[k]; i++) {   // one body, or several identical to 31 levels
            const float4 p = posm[idx[i]];
            fold_mass(px, py, m, p.x, p.y, p.w);
        }
    } else {
        const int c0 = t.child0[k];
        for (int c = c0; c < c0 + nc; c++) {
            const float4 q = t.com[c];
            fold_mass(px, py, m, q.x, q.y, q.z);
            size += t.size[c];
        }
    }
    t.com[k] = make_float4(px, py, m, 0.f);
    t.size[k] = size;
}

// top-down over one level: pre-order offsets of the children
__global__ __launch_bounds__(kTile) void k_down_level(TreeArrays t, const int first, const int last)
{
    const int k = first + blockIdx.x * kTile + threadIdx.x;
    if (k >= last) return;
    const int nc = t.nchild[k];
    if (nc == 0) return;
    int run = t.offset[k] + 1;
    const int c0 = t.child0[k];
    for (int c = c0; c < c0 + nc; c++) {
        t.offset[c] = run;
        run += t.size[c];
    }
}

__global__ __launch_bounds__(kTile) void k_emit(const float4* __restrict__ posm, const unsigned* __restrict__ idx,
                                                const unsigned* __restrict__ box, TreeArrays t, const int n_nodes,
                                                BhNode* __restrict__ out)
{
    const int k = blockIdx.x * kTile + threadIdx.x;
    if (k >= n_nodes) return;
    // the node's AABB: replay the first body's path for `level` steps (same arithmetic as the host tree)
    float x1 = dec_f32(box[0]), y1 = dec_f32(box[1]), x2 = dec_f32(box[2]), y2 = dec_f32(box[3]);
    const float4 p = posm[idx[t.lo[k]]];
    const int lvl = t.level[k];
#pragma unroll 1
    for (int l = 0; l < lvl; l++) descend(x1, y1, x2, y2, p.x, p.y);
    const float4 c = t.com[k];
    BhNode b;
    b.px = c.x; b.py = c.y; b.m = c.z; b.s = __fsub_rn(x2, x1);   // nbody.rs:341
    const int off = t.offset[k];
    b.skip = off + t.size[k];
    b.interior = t.nchild[k] > 0 ? 1 : 0;
    b.pad0 = 0; b.pad1 = 0;
    out[off] = b;
}

__global__ void k_init_root(TreeArrays t, const int n, int* node_count, int* overflow, unsigned* box)
{
    t.lo[0] = 0; t.hi[0] = n; t.level[0] = 0; t.offset[0] = 0;
    *node_count = 1;
    *overflow = 0;
    box[0] = 0xFFFFFFFFu; box[1] = 0xFFFFFFFFu; box[2] = 0u; box[3] = 0u;
}

size_t device_tree_workspace_bytes(int n, int node_cap, size_t* sort_tmp_bytes)
{
    size_t tmp = 0;
    (void)rocprim::radix_sort_pairs(nullptr, tmp, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (unsigned*)nullptr,
                              (unsigned*)nullptr, (size_t)n, 0, 2 * kLevels, (hipStream_t)0);
    if (sort_tmp_bytes) *sort_tmp_bytes = tmp;
    size_t bytes = 0;
    auto add = [&](size_t b) { bytes += (b + 255) & ~(size_t)255; };
    add(sizeof(unsigned long long) * (size_t)n * 2);   // keys in/out
    add(sizeof(unsigned) * (size_t)n * 2);             // idx in/out
    add(tmp);
    add(sizeof(int) * (size_t)node_cap * 7);           // lo hi level child0 nchild size offset
    add(sizeof(float4) * (size_t)node_cap);            // com
    add(256);                                          // counters + box
    return bytes;
}

// Spatial (Morton, reference quadrant order) permutation of the bodies only: bbox + path keys + radix sort.
// Used to make the traversal of a HOST-built tree wave-coherent. *perm_dev points into the workspace.
hipError_t device_spatial_order(const float4* posm, int n, void* workspace, size_t workspace_bytes, const unsigned** perm_dev,
                                hipStream_t stream)
{
    *perm_dev = nullptr;
    if (n <= 0) return hipSuccess;
    size_t sort_tmp = 0;
    if (device_tree_workspace_bytes(n, 1, &sort_tmp) > workspace_bytes) return hipErrorInvalidValue;
    char* w = static_cast<char*>(workspace);
    auto take = [&](size_t b) { char* p = w; w += (b + 255) & ~(size_t)255; return p; };
    unsigned long long* keys0 = reinterpret_cast<unsigned long long*>(take(sizeof(unsigned long long) * (size_t)n * 2));
    unsigned long long* keys1 = keys0 + n;
    unsigned* idx0 = reinterpret_cast<unsigned*>(take(sizeof(unsigned) * (size_t)n * 2));
    unsigned* idx1 = idx0 + n;
    void* tmp = take(sort_tmp);
    int* ints = reinterpret_cast<int*>(take(sizeof(int) * 7));
    (void)take(sizeof(float4));
    int* counters = reinterpret_cast<int*>(take(256));
    unsigned* box = reinterpret_cast<unsigned*>(counters + 4);
    TreeArrays t;
    t.lo = ints; t.hi = ints + 1; t.level = ints + 2; t.child0 = ints + 3; t.nchild = ints + 4; t.size = ints + 5; t.offset = ints + 6;
    t.com = nullptr;
    const int nb = (n + kTile - 1) / kTile;
    hipLaunchKernelGGL(k_init_root, dim3(1), dim3(1), 0, stream, t, n, counters, counters + 1, box);
    hipLaunchKernelGGL(k_bbox, dim3(nb < 256 ? nb : 256), dim3(kTile), 0, stream, posm, n, box);
    hipLaunchKernelGGL(k_keys, dim3(nb), dim3(kTile), 0, stream, posm, n, box, keys0, idx0);
    hipError_t e = rocprim::radix_sort_pairs(tmp, sort_tmp, keys0, keys1, idx0, idx1, (size_t)n, 0, 2 * kLevels, stream);
    if (e != hipSuccess) return e;
    *perm_dev = idx1;
    return hipGetLastError();
}

// Builds the flattened tree for posm[0..n) into `out` (capacity node_cap records). Returns the node count in
// *n_nodes_host (host, valid after the stream work the function waits for) and the sorted body order in
// *perm_dev (device pointer inside the workspace: body handled by thread t = perm[t], a Morton order).
hipError_t device_tree_build(const float4* posm, int n, void* workspace, size_t workspace_bytes, int node_cap, BhNode* out,
                             int* host_counters /* pinned, >= 4 ints */, int* n_nodes_host, const unsigned** perm_dev,
                             int* status, hipStream_t stream)
{
    *status = 0;
    *n_nodes_host = 0;
    if (n <= 0) return hipSuccess;
    size_t sort_tmp = 0;
    if (device_tree_workspace_bytes(n, node_cap, &sort_tmp) > workspace_bytes) return hipErrorInvalidValue;
    char* w = static_cast<char*>(workspace);
    auto take = [&](size_t b) { char* p = w; w += (b + 255) & ~(size_t)255; return p; };
    unsigned long long* keys0 = reinterpret_cast<unsigned long long*>(take(sizeof(unsigned long long) * (size_t)n * 2));
    unsigned long long* keys1 = keys0 + n;
    unsigned* idx0 = reinterpret_cast<unsigned*>(take(sizeof(unsigned) * (size_t)n * 2));
    unsigned* idx1 = idx0 + n;
    void* tmp = take(sort_tmp);
    int* ints = reinterpret_cast<int*>(take(sizeof(int) * (size_t)node_cap * 7));
    TreeArrays t;
    t.lo = ints; t.hi = ints + node_cap; t.level = ints + 2 * (size_t)node_cap; t.child0 = ints + 3 * (size_t)node_cap;
    t.nchild = ints + 4 * (size_t)node_cap; t.size = ints + 5 * (size_t)node_cap; t.offset = ints + 6 * (size_t)node_cap;
    t.com = reinterpret_cast<float4*>(take(sizeof(float4) * (size_t)node_cap));
    int* counters = reinterpret_cast<int*>(take(256));   // [0] node_count, [1] overflow, [4..7] box (as unsigned)
    unsigned* box = reinterpret_cast<unsigned*>(counters + 4);

    const int nb = (n + kTile - 1) / kTile;
    hipLaunchKernelGGL(k_init_root, dim3(1), dim3(1), 0, stream, t, n, counters, counters + 1, box);
    hipLaunchKernelGGL(k_bbox, dim3(nb < 256 ? nb : 256), dim3(kTile), 0, stream, posm, n, box);
    hipLaunchKernelGGL(k_keys, dim3(nb), dim3(kTile), 0, stream, posm, n, box, keys0, idx0);
    hipError_t e = rocprim::radix_sort_pairs(tmp, sort_tmp, keys0, keys1, idx0, idx1, (size_t)n, 0, 2 * kLevels, stream);
    if (e != hipSuccess) return e;
    *perm_dev = idx1;

    // breadth-first splitting, one launch per level; the level's node range comes back through pinned memory
    int level_first[kLevels + 4];
    int first = 0, last = 1, levels = 0;
    level_first[0] = 0;
    while (first < last && levels <= kLevels) {
        hipLaunchKernelGGL(k_split_level, dim3((last - first + kTile - 1) / kTile), dim3(kTile), 0, stream, keys1, t, first,
                           last, counters, node_cap, counters + 1);
        e = hipMemcpyAsync(host_counters, counters, 2 * sizeof(int), hipMemcpyDeviceToHost, stream);
        if (e != hipSuccess) return e;
        e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return e;
        if (host_counters[1]) { *status = 1; return hipSuccess; }   // node pool exhausted (pathological input)
        levels++;
        level_first[levels] = last;
        first = last;
        last = host_counters[0];
    }
    const int n_nodes = last;
    level_first[levels + 1] = n_nodes;
    for (int l = levels; l >= 0; l--) {
        const int a = level_first[l], b = l == levels ? n_nodes : level_first[l + 1];
        if (b > a)
            hipLaunchKernelGGL(k_up_level, dim3((b - a + kTile - 1) / kTile), dim3(kTile), 0, stream, posm, idx1, t, a, b);
    }
    for (int l = 0; l <= levels; l++) {
        const int a = level_first[l], b = l == levels ? n_nodes : level_first[l + 1];
        if (b > a) hipLaunchKernelGGL(k_down_level, dim3((b - a + kTile - 1) / kTile), dim3(kTile), 0, stream, t, a, b);
    }
    hipLaunchKernelGGL(k_emit, dim3((n_nodes + kTile - 1) / kTile), dim3(kTile), 0, stream, posm, idx1, box, t, n_nodes, out);
    *n_nodes_host = n_nodes;
    return hipGetLastError();
}

}  // namespace nbx
