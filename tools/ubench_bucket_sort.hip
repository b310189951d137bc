// ubench_bucket_sort.hip -- the second half of the warm sort (bh_build.hip, round 5): (62-bit key, index) pairs already partitioned
// into buckets of ~640-800 pairs (fixed slots of 4096 per bucket), each bucket sorted by its own workgroup / wave. Which shape?
//   v0  256 threads per bucket, pairs blocked over the threads, lane partners through ds_bpermute, 3 stages through LDS
//   v1  ONE WAVE per bucket (four buckets per 256-thread workgroup, no barrier anywhere), E = P/64 pairs per lane:
//       partners in the lane's own registers for j < E, in other lanes through DPP / ds_swizzle / ds_bpermute by mask
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_bucket_sort.hip -o tools/ubench_bucket_sort
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned long long u64;
constexpr int kTile = 256, kCap = 4096;
constexpr u64 kPadKey = ~0ull;

__device__ __forceinline__ bool pair_less(u64 ka, unsigned ia, u64 kb, unsigned ib) { return ka < kb || (ka == kb && ia < ib); }
__device__ __forceinline__ u64 shfl_xor_u64(u64 v, int m)
{
    return (u64)(unsigned)__shfl_xor((int)(unsigned)v, m) | ((u64)(unsigned)__shfl_xor((int)(unsigned)(v >> 32), m) << 32);
}

// ---- v0: as first built --------------------------------------------------------------------------------------------------
template <int E>
__device__ __forceinline__ void v0_body(const u64* pk, const unsigned* pi, int cnt, u64* ko, unsigned* io, int base, unsigned* lds)
{
    constexpr int P = kTile * E;
    const int tid = threadIdx.x;
    u64 k[E]; unsigned id[E];
#pragma unroll
    for (int r = 0; r < E; r++) { const int e = tid * E + r; k[r] = kPadKey; id[r] = ~0u; if (e < cnt) { k[r] = pk[e]; id[r] = pi[e]; } }
#pragma unroll 1
    for (int kk = 2; kk <= P; kk <<= 1) {
#pragma unroll 1
        for (int j = kk >> 1; j >= E; j >>= 1) {
            const int tj = j / E;
            const bool up = ((tid * E) & kk) == 0;
            const bool keep_min = up == ((tid & tj) == 0);
            if (tj < 64) {
#pragma unroll
                for (int r = 0; r < E; r++) {
                    const u64 ok = shfl_xor_u64(k[r], tj); const unsigned oi = (unsigned)__shfl_xor((int)id[r], tj);
                    if (pair_less(k[r], id[r], ok, oi) != keep_min) { k[r] = ok; id[r] = oi; }
                }
            } else {
                __syncthreads();
#pragma unroll
                for (int r = 0; r < E; r++) { lds[r * kTile + tid] = (unsigned)k[r]; lds[P + r * kTile + tid] = (unsigned)(k[r] >> 32); lds[2 * P + r * kTile + tid] = id[r]; }
                __syncthreads();
                const int pt = tid ^ tj;
#pragma unroll
                for (int r = 0; r < E; r++) {
                    const u64 ok = (u64)lds[r * kTile + pt] | ((u64)lds[P + r * kTile + pt] << 32); const unsigned oi = lds[2 * P + r * kTile + pt];
                    if (pair_less(k[r], id[r], ok, oi) != keep_min) { k[r] = ok; id[r] = oi; }
                }
            }
        }
#pragma unroll
        for (int j = E / 2; j > 0; j >>= 1) {
            if (j > (kk >> 1)) continue;
#pragma unroll
            for (int a = 0; a < E; a++) {
                if (a & j) continue;
                const int b = a | j;
                const bool up = ((tid * E + a) & kk) == 0;
                if (pair_less(k[b], id[b], k[a], id[a]) == up) { const u64 tk = k[a]; k[a] = k[b]; k[b] = tk; const unsigned ti = id[a]; id[a] = id[b]; id[b] = ti; }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < E; r++) { const int e = tid * E + r; if (e < cnt) { ko[base + e] = k[r]; io[base + e] = id[r]; } }
}
__global__ __launch_bounds__(kTile) void k_v0(const u64* pkeys, const unsigned* pidx, const int* gcount, const int* start, u64* ko, unsigned* io)
{
    extern __shared__ unsigned lds_sort[];
    const int b = blockIdx.x;
    const int cnt = gcount[b], base = start[b];
    const u64* pk = pkeys + (size_t)b * kCap; const unsigned* pi = pidx + (size_t)b * kCap;
    if (cnt <= kTile) v0_body<1>(pk, pi, cnt, ko, io, base, lds_sort);
    else if (cnt <= 2 * kTile) v0_body<2>(pk, pi, cnt, ko, io, base, lds_sort);
    else if (cnt <= 4 * kTile) v0_body<4>(pk, pi, cnt, ko, io, base, lds_sort);
    else if (cnt <= 8 * kTile) v0_body<8>(pk, pi, cnt, ko, io, base, lds_sort);
    else v0_body<16>(pk, pi, cnt, ko, io, base, lds_sort);
}

// ---- v1: one wave per bucket ---------------------------------------------------------------------------------------------
template <int MASK, int MODE>
__device__ __forceinline__ unsigned xchg32(const unsigned v, const int lane)
{
    if constexpr (MODE == 0) return (unsigned)__builtin_amdgcn_ds_bpermute((lane ^ MASK) << 2, (int)v);
    else {
        if constexpr (MASK == 1) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);       // quad_perm [1,0,3,2]
        else if constexpr (MASK == 2) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
        else if constexpr (MASK == 8) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, true); // row_ror:8
        else if constexpr (MASK == 4) return (unsigned)__builtin_amdgcn_ds_swizzle((int)v, (4 << 10) | 0x1F);         // bit mode: xor 4
        else if constexpr (MASK == 16) return (unsigned)__builtin_amdgcn_ds_swizzle((int)v, (16 << 10) | 0x1F);
        else return (unsigned)__builtin_amdgcn_ds_bpermute((lane ^ MASK) << 2, (int)v);
    }
}
template <int E, int MASK, int MODE>
__device__ __forceinline__ void lane_stage(u64 (&k)[E], unsigned (&id)[E], const int lane, const bool keep_min)
{
#pragma unroll
    for (int r = 0; r < E; r++) {
        const unsigned lo = xchg32<MASK, MODE>((unsigned)k[r], lane), hi = xchg32<MASK, MODE>((unsigned)(k[r] >> 32), lane);
        const unsigned oi = xchg32<MASK, MODE>(id[r], lane);
        const u64 ok = (u64)lo | ((u64)hi << 32);
        if (pair_less(k[r], id[r], ok, oi) != keep_min) { k[r] = ok; id[r] = oi; }
    }
}
template <int E, int MODE>
__device__ __forceinline__ void v1_body(const u64* pk, const unsigned* pi, int cnt, u64* ko, unsigned* io, int base)
{
    constexpr int P = 64 * E;
    const int lane = threadIdx.x & 63;
    u64 k[E]; unsigned id[E];
#pragma unroll
    for (int r = 0; r < E; r++) { const int e = lane * E + r; k[r] = kPadKey; id[r] = ~0u; if (e < cnt) { k[r] = pk[e]; id[r] = pi[e]; } }
#pragma unroll 1
    for (int kk = 2; kk <= P; kk <<= 1) {
        const bool up_lane = ((lane * E) & kk) == 0;     // for kk >= E
#pragma unroll 1
        for (int j = kk >> 1; j >= E; j >>= 1) {
            const int tj = j / E;
            const bool keep_min = up_lane == ((lane & tj) == 0);
            switch (tj) {
                case 32: lane_stage<E, 32, MODE>(k, id, lane, keep_min); break;
                case 16: lane_stage<E, 16, MODE>(k, id, lane, keep_min); break;
                case 8: lane_stage<E, 8, MODE>(k, id, lane, keep_min); break;
                case 4: lane_stage<E, 4, MODE>(k, id, lane, keep_min); break;
                case 2: lane_stage<E, 2, MODE>(k, id, lane, keep_min); break;
                default: lane_stage<E, 1, MODE>(k, id, lane, keep_min); break;
            }
        }
#pragma unroll
        for (int j = E / 2; j > 0; j >>= 1) {
            if (j > (kk >> 1)) continue;
#pragma unroll
            for (int a = 0; a < E; a++) {
                if (a & j) continue;
                const int b = a | j;
                const bool up = ((lane * E + a) & kk) == 0;
                if (pair_less(k[b], id[b], k[a], id[a]) == up) { const u64 tk = k[a]; k[a] = k[b]; k[b] = tk; const unsigned ti = id[a]; id[a] = id[b]; id[b] = ti; }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < E; r++) { const int e = lane * E + r; if (e < cnt) { ko[base + e] = k[r]; io[base + e] = id[r]; } }
}
template <int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_v1(const u64* pkeys, const unsigned* pidx, const int* gcount, const int* start, const int buckets,
                                                   u64* ko, unsigned* io)
{
    const int b = blockIdx.x * WAVES + (threadIdx.x >> 6);
    if (b >= buckets) return;
    const int cnt = gcount[b], base = start[b];
    const u64* pk = pkeys + (size_t)b * kCap; const unsigned* pi = pidx + (size_t)b * kCap;
    if (cnt <= 256) v1_body<4, MODE>(pk, pi, cnt, ko, io, base);
    else if (cnt <= 512) v1_body<8, MODE>(pk, pi, cnt, ko, io, base);
    else if (cnt <= 1024) v1_body<16, MODE>(pk, pi, cnt, ko, io, base);
    else if (cnt <= 2048) v1_body<32, MODE>(pk, pi, cnt, ko, io, base);
    else v1_body<64, MODE>(pk, pi, cnt, ko, io, base);
}

// ---- v2: spread the bucket over D sub-buckets by interpolation, rank inside the sub-bucket by counting ---------------------------
// The keys of one bucket are ~800 neighbours on the Z-curve: between the bucket's smallest and largest key they lie about evenly.
// digit = (key - min) >> shift spreads them over D = 2048 sub-buckets (0.4 pairs each on average); LDS counters give every pair its
// sub-bucket's start (scan) and an arrival number; its place inside the sub-bucket = the number of smaller (key, index) pairs among
// the sub-bucket's members (1-3 reads).  A bucket with a crowded sub-bucket (> kCrowd pairs: clustered keys) takes the bitonic network.
constexpr int kD = 2048, kCrowd = 24;
template <int E, int STAGE>
__device__ __forceinline__ bool v2_body(const u64* pk, const unsigned* pi, int cnt, u64* ko, unsigned* io, int base, unsigned* lds)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int* hist = reinterpret_cast<int*>(lds);                       // [kD + 1]
    u64* skey = reinterpret_cast<u64*>(lds + kD + 2);              // [kCap]  (8-byte aligned: kD + 2 words in front)
    unsigned* sidx = reinterpret_cast<unsigned*>(skey + STAGE);    // [STAGE]
    __shared__ u64 red[2][kTile / 64];
    __shared__ int wsum[kTile / 64];
    __shared__ int crowded;
    u64 k[E]; unsigned id[E];
    u64 mn = kPadKey, mx = 0;
#pragma unroll
    for (int r = 0; r < E; r++) {
        const int e = r * kTile + tid;
        k[r] = kPadKey; id[r] = ~0u;
        if (e < cnt) { k[r] = pk[e]; id[r] = pi[e]; mn = k[r] < mn ? k[r] : mn; mx = k[r] > mx ? k[r] : mx; }
    }
    for (int t = tid; t <= kD; t += kTile) hist[t] = 0;
    if (tid == 0) crowded = 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const u64 a = shfl_xor_u64(mn, o), b = shfl_xor_u64(mx, o); mn = a < mn ? a : mn; mx = b > mx ? b : mx; }
    if (lane == 0) { red[0][wave] = mn; red[1][wave] = mx; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < kTile / 64; w++) { mn = red[0][w] < mn ? red[0][w] : mn; mx = red[1][w] > mx ? red[1][w] : mx; }
    const u64 W = mx - mn;                                          // digits 0 .. W >> shift, < kD
    int shift = 0;
    if (W >= (u64)kD) shift = 64 - __clzll((long long)W) - 11;     // bit length of W minus log2(kD)
    int dg[E], off[E];
#pragma unroll
    for (int r = 0; r < E; r++) {
        const int e = r * kTile + tid;
        dg[r] = -1;
        if (e < cnt) { dg[r] = (int)((k[r] - mn) >> shift); off[r] = atomicAdd(&hist[dg[r]], 1); }
    }
    __syncthreads();
    // exclusive scan of the kD counters (8 per thread), the largest count on the way
    {
        int c[kD / kTile], sum = 0, big = 0;
#pragma unroll
        for (int u = 0; u < kD / kTile; u++) { c[u] = hist[tid * (kD / kTile) + u]; sum += c[u]; big = c[u] > big ? c[u] : big; }
        int incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
        if (lane == 63) wsum[wave] = incl;
        if (big > kCrowd) crowded = 1;
        __syncthreads();
        int before = incl - sum;
        for (int w = 0; w < wave; w++) before += wsum[w];
#pragma unroll
        for (int u = 0; u < kD / kTile; u++) { hist[tid * (kD / kTile) + u] = before; before += c[u]; }
        if (tid == kTile - 1) hist[kD] = before;
    }
    __syncthreads();
    if (crowded) return false;
#pragma unroll
    for (int r = 0; r < E; r++)
        if (dg[r] >= 0) { const int p = hist[dg[r]] + off[r]; skey[p] = k[r]; sidx[p] = id[r]; }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < E; r++) {
        if (dg[r] < 0) continue;
        const int s0 = hist[dg[r]], s1 = hist[dg[r] + 1];
        int pos = s0;
        for (int t = s0; t < s1; t++) pos += pair_less(skey[t], sidx[t], k[r], id[r]) ? 1 : 0;
        ko[base + pos] = k[r]; io[base + pos] = id[r];
    }
    return true;
}
template <int STAGE>
__global__ __launch_bounds__(kTile) void k_v2(const u64* pkeys, const unsigned* pidx, const int* gcount, const int* start, u64* ko, unsigned* io)
{
    extern __shared__ unsigned lds_sort[];
    const int b = blockIdx.x;
    const int cnt = gcount[b], base = start[b];
    const u64* pk = pkeys + (size_t)b * kCap; const unsigned* pi = pidx + (size_t)b * kCap;
    bool done;
    if (cnt <= kTile) done = v2_body<1, STAGE>(pk, pi, cnt, ko, io, base, lds_sort);
    else if (cnt <= 2 * kTile) done = v2_body<2, STAGE>(pk, pi, cnt, ko, io, base, lds_sort);
    else if (cnt <= 4 * kTile) done = v2_body<4, STAGE>(pk, pi, cnt, ko, io, base, lds_sort);
    else if (cnt <= 8 * kTile) done = v2_body<8, STAGE>(pk, pi, cnt, ko, io, base, lds_sort);
    else done = v2_body<16, STAGE>(pk, pi, cnt, ko, io, base, lds_sort);
    if (done) return;
    __syncthreads();
    if (cnt <= kTile) v0_body<1>(pk, pi, cnt, ko, io, base, lds_sort);
    else if (cnt <= 2 * kTile) v0_body<2>(pk, pi, cnt, ko, io, base, lds_sort);
    else if (cnt <= 4 * kTile) v0_body<4>(pk, pi, cnt, ko, io, base, lds_sort);
    else if (cnt <= 8 * kTile) v0_body<8>(pk, pi, cnt, ko, io, base, lds_sort);
    else v0_body<16>(pk, pi, cnt, ko, io, base, lds_sort);
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 1048576;
    const int target = argc > 2 ? atoi(argv[2]) : 800;
    const int erlang = argc > 3 ? atoi(argv[3]) : 0;     // 0: equal buckets; k: sizes ~ Erlang-k
    std::mt19937_64 rng(12345);
    std::vector<u64> keys(n);
    const int clustered = argc > 4 ? atoi(argv[4]) : 0;   // percent of keys that sit in tight clumps (differ in the low 16 bits only)
    u64 clump = 0; int left = 0;
    for (auto& x : keys) {
        x = rng() >> 2;
        if (clustered && (int)(rng() % 100) < clustered) { if (left == 0) { clump = x & ~0xFFFFull; left = 1 + (int)(rng() % 40); } x = clump | (rng() & 0xFFFF); left--; }
    }
    std::vector<unsigned> order(n);
    for (int i = 0; i < n; i++) order[i] = (unsigned)i;
    std::sort(order.begin(), order.end(), [&](unsigned a, unsigned b) { return keys[a] < keys[b] || (keys[a] == keys[b] && a < b); });
    // bucket boundaries
    std::vector<int> start{0};
    std::exponential_distribution<double> ex(1.0);
    while (start.back() < n) {
        double s = 1.0;
        if (erlang > 0) { s = 0; for (int t = 0; t < erlang; t++) s += ex(rng); s /= erlang; }
        int sz = std::max(1, std::min(kCap, (int)(target * s)));
        start.push_back(std::min(n, start.back() + sz));
    }
    const int B = (int)start.size() - 1;
    std::vector<int> cnt(B);
    std::vector<u64> pk((size_t)B * kCap, 0); std::vector<unsigned> pi((size_t)B * kCap, 0);
    int maxc = 0;
    for (int b = 0; b < B; b++) {
        cnt[b] = start[b + 1] - start[b]; maxc = std::max(maxc, cnt[b]);
        std::vector<unsigned> mem(order.begin() + start[b], order.begin() + start[b + 1]);
        std::shuffle(mem.begin(), mem.end(), rng);
        for (int t = 0; t < cnt[b]; t++) { pk[(size_t)b * kCap + t] = keys[mem[t]]; pi[(size_t)b * kCap + t] = mem[t]; }
    }
    printf("n %d buckets %d target %d erlang %d largest %d\n", n, B, target, erlang, maxc);
    u64 *d_pk, *d_ko; unsigned *d_pi, *d_io; int *d_cnt, *d_start;
    CHECK(hipMalloc(&d_pk, 8 * pk.size())); CHECK(hipMalloc(&d_pi, 4 * pi.size())); CHECK(hipMalloc(&d_ko, 8 * (size_t)n)); CHECK(hipMalloc(&d_io, 4 * (size_t)n));
    CHECK(hipMalloc(&d_cnt, 4 * B)); CHECK(hipMalloc(&d_start, 4 * B));
    CHECK(hipMemcpy(d_pk, pk.data(), 8 * pk.size(), hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_pi, pi.data(), 4 * pi.size(), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_cnt, cnt.data(), 4 * B, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_start, start.data(), 4 * B, hipMemcpyHostToDevice));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_v0), hipFuncAttributeMaxDynamicSharedMemorySize, 12 * kCap));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto check = [&](const char* name, float us) {
        std::vector<unsigned> out(n);
        CHECK(hipMemcpy(out.data(), d_io, 4 * (size_t)n, hipMemcpyDeviceToHost));
        printf("%-44s %8.1f us  %s\n", name, us, out == order ? "sorted" : "WRONG");
        CHECK(hipMemset(d_io, 0xFF, 4 * (size_t)n));
    };
    auto time_it = [&](const char* name, auto&& launch) {
        for (int w = 0; w < 3; w++) launch();
        CHECK(hipEventRecord(e0));
        for (int w = 0; w < 20; w++) launch();
        CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        check(name, ms * 1e3f / 20);
    };
    time_it("v0 256 threads / bucket, bpermute + LDS", [&] { hipLaunchKernelGGL(k_v0, dim3(B), dim3(kTile), 12 * kCap, 0, d_pk, d_pi, d_cnt, d_start, d_ko, d_io); });
    time_it("v0 same, 24 KB LDS claimed (buckets <= 2048)", [&] { hipLaunchKernelGGL(k_v0, dim3(B), dim3(kTile), maxc <= 2048 ? 12 * 2048 : 12 * kCap, 0, d_pk, d_pi, d_cnt, d_start, d_ko, d_io); });
    time_it("v2 interpolation + counting (bitonic if crowded)", [&] { hipLaunchKernelGGL(k_v2<kCap>, dim3(B), dim3(kTile), 4 * (kD + 2) + 12 * kCap, 0, d_pk, d_pi, d_cnt, d_start, d_ko, d_io); });
    if (maxc <= 2048 && !clustered)   // (the network of a crowded bucket needs the full staging area)
        time_it("v2 same, staging for 2 048 pairs (32 KB LDS)", [&] { hipLaunchKernelGGL(k_v2<2048>, dim3(B), dim3(kTile), 4 * (kD + 2) + 12 * 2048, 0, d_pk, d_pi, d_cnt, d_start, d_ko, d_io); });
    time_it("v1 one wave / bucket, bpermute, 4 waves/wg", [&] { hipLaunchKernelGGL((k_v1<0, 4>), dim3((B + 3) / 4), dim3(256), 0, 0, d_pk, d_pi, d_cnt, d_start, B, d_ko, d_io); });
    time_it("v1 one wave / bucket, dpp/swizzle, 4 waves/wg", [&] { hipLaunchKernelGGL((k_v1<1, 4>), dim3((B + 3) / 4), dim3(256), 0, 0, d_pk, d_pi, d_cnt, d_start, B, d_ko, d_io); });
    time_it("v1 one wave / bucket, dpp/swizzle, 1 wave/wg", [&] { hipLaunchKernelGGL((k_v1<1, 1>), dim3(B), dim3(64), 0, 0, d_pk, d_pi, d_cnt, d_start, B, d_ko, d_io); });
    time_it("v1 one wave / bucket, bpermute, 1 wave/wg", [&] { hipLaunchKernelGGL((k_v1<0, 1>), dim3(B), dim3(64), 0, 0, d_pk, d_pi, d_cnt, d_start, B, d_ko, d_io); });
    return 0;
}
