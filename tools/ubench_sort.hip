// ubench_sort.hip -- rocPRIM radix_sort_pairs at the device quadtree build's sizes: 62-bit path keys vs 32-bit prefixes
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_sort.hip -o tools/ubench_sort
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <typename K, typename Config = rocprim::default_config>
void run(const char* name, int n, int bits)
{
    std::vector<K> h(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (K)(s >> (64 - bits)); }
    K *k0, *k1; unsigned *v0, *v1;
    CHECK(hipMalloc(&k0, sizeof(K) * n)); CHECK(hipMalloc(&k1, sizeof(K) * n));
    CHECK(hipMalloc(&v0, 4 * n)); CHECK(hipMalloc(&v1, 4 * n));
    CHECK(hipMemcpy(k0, h.data(), sizeof(K) * n, hipMemcpyHostToDevice));
    size_t tmp = 0;
    CHECK(rocprim::radix_sort_pairs<Config>(nullptr, tmp, k0, k1, v0, v1, (size_t)n, 0, bits, 0));
    void* t; CHECK(hipMalloc(&t, tmp));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 3; w++) CHECK(rocprim::radix_sort_pairs<Config>(t, tmp, k0, k1, v0, v1, (size_t)n, 0, bits, 0));
    CHECK(hipEventRecord(e0));
    for (int w = 0; w < 20; w++) CHECK(rocprim::radix_sort_pairs<Config>(t, tmp, k0, k1, v0, v1, (size_t)n, 0, bits, 0));
    CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-10s n %8d bits %2d : %7.1f us per sort\n", name, n, bits, ms * 1e3 / 20);
    hipFree(k0); hipFree(k1); hipFree(v0); hipFree(v1); hipFree(t);
}

int main()
{
    for (int n : {10000, 100000, 262144, 1048576, 4194304}) {
        run<unsigned long long>("u64 keys", n, 62);
        run<unsigned long long>("u64 keys", n, 40);
        run<unsigned>("u32 keys", n, 32);
        run<unsigned>("u32 keys", n, 24);
        using onesweep = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 32768>;
        run<unsigned long long, onesweep>("u64 1sweep", n, 62);
        run<unsigned long long, onesweep>("u64 1sweep", n, 48);
        run<unsigned long long, onesweep>("u64 1sweep", n, 40);
        run<unsigned, onesweep>("u32 1sweep", n, 32);
    }
    return 0;
}
