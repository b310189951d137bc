// bh_sort.hip -- the sort of a warm device tree build: it starts from last step's order (round 5; replaces the serial insert
// loop nbody.rs:410-415 together with the rest of the build, bh_build.hip).
#include <cstdlib>

#include "bh_build_internal.h"

namespace nbx {

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
// (spos: the bodies in the order perm names, when the last kick-drift left them so -- BhKick::sorted -- else nullptr: gather)
__global__ __launch_bounds__(kTile) void k_sample_rank(const float4* __restrict__ posm, const float4* __restrict__ spos, const int n, const float4* __restrict__ part,
                                                       const int parts, unsigned* __restrict__ box,
                                                       const unsigned* __restrict__ perm, const int samples,
                                                       unsigned long long* __restrict__ skeys, int* __restrict__ srank,
                                                       const int* __restrict__ poison)
{
    if (*poison) return;   // see launch_inc_sort: the order in `perm` is not to be trusted
    __shared__ __attribute__((aligned(16))) unsigned long long other[kTile];
    const int sb = (samples + kTile - 1) / kTile;
    const int a = blockIdx.x / sb, c = blockIdx.x - a * sb;
    const int tid = threadIdx.x;
    const int oc = c * kTile + tid, mine_i = a * kTile + tid;
    // the bodies first (their two dependent loads fly while the box is folded out of k_bbox's partials; workgroup 0 files it)
    const int s0 = sample_pos(mine_i < samples ? mine_i : 0, n, samples), s1 = sample_pos(oc < samples ? oc : 0, n, samples);
    const float4 p0 = spos ? spos[s0] : posm[perm[s0]];
    const float4 p1 = spos ? spos[s1] : posm[perm[s1]];
    float rx1, ry1, rx2, ry2;
    fold_box_partials(part, parts, box, blockIdx.x == 0, rx1, ry1, rx2, ry2);
    unsigned long long ok = kPadKey, mine = kPadKey;
    if (a == c) {
        if (mine_i < samples) {
            float ax1 = rx1, ay1 = ry1, ax2 = rx2, ay2 = ry2;
            unsigned long long k0 = 0;
#pragma unroll 1
            for (int l = 0; l < kLevels; l++) k0 = (k0 << 2) | (unsigned long long)descend(ax1, ay1, ax2, ay2, p0.x, p0.y);
            mine = k0;
        }
        ok = mine;
    } else {
        // two descents, interleaved
        float ax1 = rx1, ay1 = ry1, ax2 = rx2, ay2 = ry2;
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
__global__ __launch_bounds__(kTile) void k_keys_scatter(const float4* __restrict__ posm, const float4* __restrict__ spos, const int n, const unsigned* __restrict__ box,
                                                        const unsigned* __restrict__ perm, const unsigned long long* __restrict__ skeys,
                                                        const int* __restrict__ srank, const int samples,
                                                        const int buckets, int* __restrict__ gcount,
                                                        ulonglong2* __restrict__ slots, float4* __restrict__ slot_recs,
                                                        unsigned long long* __restrict__ cell_table, const int cell_slots,
                                                        const int* __restrict__ poison)
{
    if (*poison) return;
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
    float4 rec[EA];                  // the bodies' records travel with their (key, index) pairs: k_bucket_sort delivers them sorted
    float px[EA], py[EA];
    const int t0 = blockIdx.x * (kTile * EA) + tid;
#pragma unroll
    for (int r = 0; r < EA; r++) {
        const int t = t0 + r * kTile;
        id[r] = perm[t < n ? t : n - 1];
    }
#pragma unroll
    for (int r = 0; r < EA; r++) {
        const int t = t0 + r * kTile;
        const float4 p = spos ? spos[t < n ? t : n - 1] : posm[id[r]];   // (coalesced when the last kick-drift left the bodies in this order)
        rec[r] = p;
        px[r] = p.x; py[r] = p.y;
    }
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
            const size_t at = (size_t)bk[r] * kBucketCap + (size_t)slot;
            slots[at] = make_ulonglong2(key[r], (unsigned long long)id[r]);   // one 16-byte store
            if (slot_recs) slot_recs[at] = rec[r];
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
                                               const int out_base, unsigned* __restrict__ lds, const float4* __restrict__ posm,
                                               float4* __restrict__ sb_out)
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
                // partners in other waves: through LDS, at most eight pairs per thread at a time (the network of a 4 096-pair
                // bucket trades in two halves: the staging area holds kStage pairs)
                constexpr int H = E > 8 ? E / 8 : 1, EH = E / H, PH = kTile * EH;
                const int pt = tid ^ tj;
#pragma unroll
                for (int h = 0; h < H; h++) {
                    __syncthreads();
#pragma unroll
                    for (int r = 0; r < EH; r++) {
                        lds[r * kTile + tid] = (unsigned)k[h * EH + r];
                        lds[PH + r * kTile + tid] = (unsigned)(k[h * EH + r] >> 32);
                        lds[2 * PH + r * kTile + tid] = id[h * EH + r];
                    }
                    __syncthreads();
#pragma unroll
                    for (int r = 0; r < EH; r++) {
                        const unsigned long long ok = (unsigned long long)lds[r * kTile + pt] | ((unsigned long long)lds[PH + r * kTile + pt] << 32);
                        const unsigned oi = lds[2 * PH + r * kTile + pt];
                        const bool mine_less = pair_less(k[h * EH + r], id[h * EH + r], ok, oi);
                        if (mine_less != keep_min) { k[h * EH + r] = ok; id[h * EH + r] = oi; }
                    }
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
        if (e < cnt) {
            keys_out[out_base + e] = k[r]; idx_out[out_base + e] = id[r];
            if (sb_out) sb_out[out_base + e] = posm[id[r]];      // (the rare path gathers the records: they did not travel through the network)
        }
    }
}

// the common case: sub-buckets by interpolation, place inside a sub-bucket by counting.  false: the keys clump, nothing was written
// (and start() has not been called).  start() -- collective, called once -- returns where the bucket's pairs go in the output.
constexpr int kSub = 2048;          // sub-buckets per bucket
constexpr int kStage = 2560;        // pairs the LDS staging area holds: 4 x the bucket target (P(an Erlang-4 bucket is bigger) = 9e-5;
                                    // those take the network) -- 38 KB of LDS per workgroup, four workgroups per CU instead of two
                                    // (tools/ubench_bucket_sort.hip: 23 -> 17 us for the kernel alone)
constexpr int kClumpPerPair = 48;   // sum of squared sub-bucket counts per pair beyond which the network is cheaper
template <int E, class StartFn>
__device__ __forceinline__ bool bucket_by_counting(const ulonglong2* __restrict__ ps, const int cnt,
                                                   unsigned long long* __restrict__ keys_out, unsigned* __restrict__ idx_out,
                                                   unsigned* __restrict__ lds, StartFn start, const float4* __restrict__ precs,
                                                   float4* __restrict__ sb_out)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int* hist = reinterpret_cast<int*>(lds);                                              // [kSub + 1]
    unsigned long long* skey = reinterpret_cast<unsigned long long*>(lds + kSub + 2);     // [kStage]
    unsigned* sidx = reinterpret_cast<unsigned*>(skey + kStage);                          // [kStage]
    if (cnt > kStage) return false;                                                       // (uniform; start() not called)
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
    if (sb_out) {   // the records go straight from their slots to their places (16-byte stores inside the bucket's stretch of sb)
#pragma unroll
        for (int r = 0; r < E; r++)
            if (pos[r] >= 0) sb_out[out_base + pos[r]] = precs[r * kTile + tid];
    }
    __syncthreads();                                                 // every place is known: the staging arrays become the sorted bucket
#pragma unroll
    for (int r = 0; r < E; r++)
        if (pos[r] >= 0) { skey[pos[r]] = k[r]; sidx[pos[r]] = id[r]; }
    __syncthreads();
    for (int t = tid; t < cnt; t += kTile) { keys_out[out_base + t] = skey[t]; idx_out[out_base + t] = sidx[t]; }   // coalesced
    return true;
}

constexpr size_t kBucketSortLds = sizeof(unsigned) * (kSub + 2) + 12 * (size_t)kStage;
static_assert(12 * (size_t)kStage >= 12 * (size_t)kTile * 8, "the network trades eight pairs per thread at a time through the staging area");
__global__ __launch_bounds__(kTile) void k_bucket_sort(const ulonglong2* __restrict__ slots, const float4* __restrict__ slot_recs,
                                                       const float4* __restrict__ posm, const int* __restrict__ gcount, const int buckets,
                                                       const int n, unsigned long long* __restrict__ keys_out, unsigned* __restrict__ idx_out,
                                                       float4* __restrict__ sb_out, int* __restrict__ counters)
{
    if (counters[kTreePoisonWord]) return;
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
            for (int t = total + tid; t < n; t += kTile) {
                keys_out[t] = (1ull << (2 * kLevels)) - 1ull; idx_out[t] = 0u;
                if (sb_out) sb_out[t] = posm[0];
            }
        }
        return before;
    };
    int cnt = gcount[b];
    cnt = cnt > kBucketCap ? kBucketCap : cnt;
    if (cnt == 0) { (void)start(); return; }
    const ulonglong2* ps = slots + (size_t)b * kBucketCap;
    const float4* precs = slot_recs + (size_t)b * kBucketCap;
    bool done;
    if (cnt <= kTile) done = bucket_by_counting<1>(ps, cnt, keys_out, idx_out, lds_sort, start, precs, sb_out);
    else if (cnt <= 2 * kTile) done = bucket_by_counting<2>(ps, cnt, keys_out, idx_out, lds_sort, start, precs, sb_out);
    else if (cnt <= 4 * kTile) done = bucket_by_counting<4>(ps, cnt, keys_out, idx_out, lds_sort, start, precs, sb_out);
    else if (cnt <= 8 * kTile) done = bucket_by_counting<8>(ps, cnt, keys_out, idx_out, lds_sort, start, precs, sb_out);
    else done = bucket_by_counting<16>(ps, cnt, keys_out, idx_out, lds_sort, start, precs, sb_out);
    if (done) return;
    __syncthreads();
    const int before = start();
    __syncthreads();
    if (cnt <= kTile) bucket_network<1>(ps, cnt, keys_out, idx_out, before, lds_sort, posm, sb_out);
    else if (cnt <= 2 * kTile) bucket_network<2>(ps, cnt, keys_out, idx_out, before, lds_sort, posm, sb_out);
    else if (cnt <= 4 * kTile) bucket_network<4>(ps, cnt, keys_out, idx_out, before, lds_sort, posm, sb_out);
    else if (cnt <= 8 * kTile) bucket_network<8>(ps, cnt, keys_out, idx_out, before, lds_sort, posm, sb_out);
    else bucket_network<16>(ps, cnt, keys_out, idx_out, before, lds_sort, posm, sb_out);
}

// the sort of a warm build: bodies in last step's order (perm) -> sorted (key, body) pairs in keys_out / idx_out (perm == idx_out is fine:
// it is read by the first three kernels and written by the last)
// The poison word (kTreePoisonWord, kernels.h): a step enqueued behind a REFUSED one starts from what that one left -- after
// an overflow an idx that is no permutation (valid indices, some twice, some missing), and an order-sorted copy of the positions that the
// refused, gated walk never rewrote.  Every consumer of such a build is gated on the poison word and does nothing, and the host enqueues the
// step again from a cold sort; the three kernels here check the word themselves and leave the arrays as the refused build left them (in
// bounds: its own later kernels ran on them), instead of sorting garbage into a tree nobody may read.
hipError_t launch_inc_sort(const float4* posm, const float4* sorted_pos, int n, unsigned* box, const float4* part, int parts, const unsigned* perm, int* gcount,
                           unsigned long long* skeys, int* srank, ulonglong2* slots, float4* slot_recs, unsigned long long* keys_out,
                           unsigned* idx_out, float4* sb_out, int* counters, unsigned long long* cell_table, int cell_slots, hipStream_t stream)
{
    static_assert(kBucketSortLds <= 64 * 1024, "k_bucket_sort's LDS must stay within the default limit (no per-device opt-in)");
    const int buckets = inc_buckets(n);
    const int samples = kOversample * buckets;
    const int sb = (samples + kTile - 1) / kTile;
    hipLaunchKernelGGL(k_sample_rank, dim3((unsigned)(sb * sb)), dim3(kTile), 0, stream, posm, sorted_pos, n, part, parts, box, perm, samples, skeys, srank,
                       counters + kTreePoisonWord);
    const size_t shm = sizeof(unsigned long long) * (size_t)(buckets > 1 ? buckets - 1 : 1) + sizeof(int) * (size_t)buckets;
    if (n >= 262144)
        hipLaunchKernelGGL(k_keys_scatter<4>, dim3((unsigned)((n + 4 * kTile - 1) / (4 * kTile))), dim3(kTile), shm, stream, posm, sorted_pos, n, box, perm,
                           skeys, srank, samples, buckets, gcount, slots, sb_out ? slot_recs : nullptr, cell_table, cell_slots, counters + kTreePoisonWord);
    else
        hipLaunchKernelGGL(k_keys_scatter<1>, dim3((unsigned)((n + kTile - 1) / kTile)), dim3(kTile), shm, stream, posm, sorted_pos, n, box, perm, skeys,
                           srank, samples, buckets, gcount, slots, sb_out ? slot_recs : nullptr, cell_table, cell_slots, counters + kTreePoisonWord);
    hipLaunchKernelGGL(k_bucket_sort, dim3((unsigned)buckets), dim3(kTile), kBucketSortLds, stream, slots, slot_recs, posm, gcount, buckets, n,
                       keys_out, idx_out, sb_out, counters);
    return hipGetLastError();
}

}  // namespace nbx
