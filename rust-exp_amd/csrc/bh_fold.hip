// bh_fold.hip -- the reference's f32 running fold (nbody.rs:303-320) of the nodes too big for k_emit, and of the root: the part of
// the device tree build (bh_build.hip) that makes interior records equal the host tree's bit for bit (reference fold).
#include "bh_build_internal.h"

namespace nbx {

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

void launch_fold_root(const float4* posm, int n, BhNode* out, hipStream_t side)
{
    hipLaunchKernelGGL(k_fold_root, dim3(1), dim3(128), 0, side, posm, n, out);
}

void launch_fold_big(const float4* posm, const float4* sb, const unsigned* idx, const int4* big, int big_cap, const int* counters, int n,
                     BhNode* out, hipStream_t stream)
{
    // one pair of waves per queued node; the count lives on the device: enough workgroups for every plausible queue
    // (a uniform system queues ~n/5 nodes), they loop when there are more
    const int fb = n / 4 + 64;
    hipLaunchKernelGGL(k_fold_big, dim3((unsigned)(fb < 8192 ? fb : 8192)), dim3(128), 0, stream, posm, sb, idx, big, big_cap, counters, n, out);
}

}  // namespace nbx
