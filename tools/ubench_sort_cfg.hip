// ubench_sort_cfg.hip -- rocPRIM radix_sort_pairs on the device tree build's (62-bit key, index) pairs with other merge-sort
// shapes than the default (block sort of 1 024 items, then log2(n / 1 024) merge passes of two kernels each): does a bigger
// first-level block (fewer passes) pay at 262 144 / 1 048 576 pairs?
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_sort_cfg.hip -o tools/ubench_sort_cfg
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <typename Config, typename K = unsigned long long, int BITS = 62>
void run(const char* name, int n)
{
    std::vector<K> h(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (K)(s >> (64 - BITS)); }
    K *k0, *k1; unsigned *v0, *v1;
    CHECK(hipMalloc(&k0, sizeof(K) * n)); CHECK(hipMalloc(&k1, sizeof(K) * n));
    CHECK(hipMalloc(&v0, 4 * n)); CHECK(hipMalloc(&v1, 4 * n));
    CHECK(hipMemcpy(k0, h.data(), sizeof(K) * n, hipMemcpyHostToDevice));
    size_t tmp = 0;
    CHECK(rocprim::radix_sort_pairs<Config>(nullptr, tmp, k0, k1, v0, v1, (size_t)n, 0, BITS, 0));
    void* t; CHECK(hipMalloc(&t, tmp));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 3; w++) CHECK(rocprim::radix_sort_pairs<Config>(t, tmp, k0, k1, v0, v1, (size_t)n, 0, BITS, 0));
    CHECK(hipEventRecord(e0));
    for (int w = 0; w < 20; w++) CHECK(rocprim::radix_sort_pairs<Config>(t, tmp, k0, k1, v0, v1, (size_t)n, 0, BITS, 0));
    CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s n %8d : %7.1f us per sort\n", name, n, ms * 1e3 / 20);
    hipFree(k0); hipFree(k1); hipFree(v0); hipFree(v1); hipFree(t);
}

template <unsigned SB, unsigned SI, unsigned MB, unsigned MI>
using cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::merge_sort_config<512, SB, SI, 128, MB, MI>, rocprim::default_config,
                                       (size_t)1 << 22>;

int main()
{
    for (int n : {262144, 1048576}) {
        run<rocprim::default_config>("default", n);
        run<cfg<256, 4, 128, 4>>("sort 256x4 merge 128x4", n);
        run<cfg<256, 8, 128, 4>>("sort 256x8 merge 128x4", n);
        run<cfg<256, 16, 128, 4>>("sort 256x16 merge 128x4", n);
        run<cfg<512, 8, 128, 4>>("sort 512x8 merge 128x4", n);
        run<cfg<256, 8, 256, 4>>("sort 256x8 merge 256x4", n);
        run<cfg<256, 8, 256, 8>>("sort 256x8 merge 256x8", n);
        run<cfg<256, 8, 128, 8>>("sort 256x8 merge 128x8", n);
        run<cfg<256, 16, 256, 8>>("sort 256x16 merge 256x8", n);
        run<cfg<1024, 4, 256, 8>>("sort 1024x4 merge 256x8", n);
        // would a 32-bit leading key (the top 16 levels) + a fix-up of the ties be cheaper?  (u32 key, u32 index) pairs:
        run<rocprim::default_config, unsigned, 32>("u32 keys, default", n);
        run<cfg<512, 8, 128, 4>, unsigned, 32>("u32 keys, sort 512x8 merge 128x4", n);
        run<cfg<1024, 8, 128, 4>, unsigned, 32>("u32 keys, sort 1024x8 merge 128x4", n);
        run<rocprim::default_config, unsigned, 24>("u32 keys 24 bits, default", n);
    }
    return 0;
}
