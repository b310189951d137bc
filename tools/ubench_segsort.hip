// ubench_segsort.hip -- how fast does rocPRIM sort pre-partitioned (62-bit key, index) pairs segment by segment?  (The second half
// of a sample sort for the 1 M-body tree build: after one partition pass by sampled splitters the buckets are independent.)
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_segsort.hip -o tools/ubench_segsort
#include <hip/hip_runtime.h>
#include <rocprim/device/device_segmented_radix_sort.hpp>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static void run(int n, int segments, int bits_lo)
{
    typedef unsigned long long K;
    std::vector<K> h(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (K)(s >> 2); }
    // partition by the leading bits (what the splitter pass would deliver), order inside a segment untouched
    int lead = 0; while ((1 << lead) < segments) lead++;
    std::stable_sort(h.begin(), h.end(), [&](K a, K b) { return (a >> (62 - lead)) < (b >> (62 - lead)); });
    std::vector<unsigned> off(segments + 1, 0);
    for (int i = 0; i < n; i++) off[(h[i] >> (62 - lead)) + 1]++;
    for (int g = 0; g < segments; g++) off[g + 1] += off[g];
    K *k0, *k1; unsigned *v0, *v1, *d_off;
    CHECK(hipMalloc(&k0, 8 * (size_t)n)); CHECK(hipMalloc(&k1, 8 * (size_t)n));
    CHECK(hipMalloc(&v0, 4 * (size_t)n)); CHECK(hipMalloc(&v1, 4 * (size_t)n)); CHECK(hipMalloc(&d_off, 4 * (segments + 1)));
    CHECK(hipMemcpy(k0, h.data(), 8 * (size_t)n, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_off, off.data(), 4 * (segments + 1), hipMemcpyHostToDevice));
    size_t tmp = 0;
    CHECK(rocprim::segmented_radix_sort_pairs(nullptr, tmp, k0, k1, v0, v1, (unsigned)n, (unsigned)segments, d_off, d_off + 1, bits_lo, 62 - lead, 0));
    void* t; CHECK(hipMalloc(&t, tmp));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 3; w++) CHECK(rocprim::segmented_radix_sort_pairs(t, tmp, k0, k1, v0, v1, (unsigned)n, (unsigned)segments, d_off, d_off + 1, bits_lo, 62 - lead, 0));
    CHECK(hipEventRecord(e0));
    for (int w = 0; w < 20; w++) CHECK(rocprim::segmented_radix_sort_pairs(t, tmp, k0, k1, v0, v1, (unsigned)n, (unsigned)segments, d_off, d_off + 1, bits_lo, 62 - lead, 0));
    CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<K> out(n);
    CHECK(hipMemcpy(out.data(), k1, 8 * (size_t)n, hipMemcpyDeviceToHost));
    bool ok = true;
    for (int g = 0; g < segments && ok; g++)
        for (unsigned i = off[g] + 1; i < off[g + 1]; i++)
            if ((out[i - 1] >> bits_lo) > (out[i] >> bits_lo)) { ok = false; break; }
    printf("n %8d segments %5d (avg %6d) bits [%d,%d): %7.1f us per sort  %s\n", n, segments, n / segments, bits_lo, 62 - lead, ms * 1e3 / 20, ok ? "sorted" : "NOT SORTED");
    hipFree(k0); hipFree(k1); hipFree(v0); hipFree(v1); hipFree(d_off); hipFree(t);
}

int main()
{
    for (int n : {262144, 1048576})
        for (int segments : {128, 256, 512, 1024, 4096}) {
            run(n, segments, 0);
            run(n, segments, 30);     // only the leading 16 levels' bits
        }
    return 0;
}
