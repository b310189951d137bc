// ubench_boundary.hip -- what does a kernel boundary cost a chain of small dependent phases (the 10 000-body tree build: nine kernels
// of 5-25 us), against a grid barrier inside one persistent kernel?  Every phase: out[i] = in[perm(i)] + 1 over n floats (a dependent
// first touch of what the phase before wrote), ping-pong.
//   (1) one launch per phase, 40 workgroups x 256 threads
//   (2) one persistent kernel, W workgroups spread over all XCDs, barrier = system-scope atomic counter (the L2s are not coherent)
//   (3) one persistent kernel whose working workgroups all sit on XCD 0 (blockIdx % 8 == 0), barrier = device-scope atomic (one L2)
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_boundary.hip -o tools/ubench_boundary
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ int pidx(int i, int n) { return (int)(((long long)i * 7919 + 13) % n); }

__global__ void k_empty() {}
__global__ void k_phase(const float* __restrict__ in, float* __restrict__ out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[pidx(i, n)] + 1.0f;
}

template <bool ONE_XCD>
__global__ void k_persistent(float* a, float* b, int n, int phases, unsigned* counter, int workers)
{
    int w = (int)blockIdx.x;
    if (ONE_XCD) {
        if (blockIdx.x & 7u) return;           // workgroups are dealt round-robin to the XCDs: these all land on XCD 0
        w = (int)(blockIdx.x >> 3);
    }
    if (w >= workers) return;
    float* in = a;
    float* out = b;
    for (int ph = 0; ph < phases; ph++) {
        for (int i = w * (int)blockDim.x + (int)threadIdx.x; i < n; i += workers * (int)blockDim.x) {
            float v;
            if (ONE_XCD) v = in[pidx(i, n)];
            else v = __builtin_nontemporal_load(&in[pidx(i, n)]);
            if (ONE_XCD) out[i] = v + 1.0f;
            else __builtin_nontemporal_store(v + 1.0f, &out[i]);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            const unsigned target = (unsigned)(ph + 1) * (unsigned)workers;
            if (ONE_XCD) {
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
            } else {
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < target) __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        float* t = in; in = out; out = t;
    }
}

int main()
{
    const int n = 10000, phases = 1000;
    float *a, *b; unsigned* c;
    CHECK(hipMalloc(&a, 4 * n)); CHECK(hipMalloc(&b, 4 * n)); CHECK(hipMalloc(&c, 64));
    CHECK(hipMemset(a, 0, 4 * n)); CHECK(hipMemset(b, 0, 4 * n));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms;
    for (int rep = 0; rep < 2; rep++) {
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < phases; i++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0);
        CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize()); CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("empty kernel, back to back                         : %6.2f us per launch\n", ms * 1e3 / phases);
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < phases; i++) hipLaunchKernelGGL(k_phase, dim3((n + 255) / 256), dim3(256), 0, 0, (i & 1) ? b : a, (i & 1) ? a : b, n);
        CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize()); CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("dependent phase, one launch each (40 x 256)        : %6.2f us per phase\n", ms * 1e3 / phases);
        for (int workers : {8, 40}) {
            CHECK(hipMemset(c, 0, 64));
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_persistent<false>, dim3(workers), dim3(256), 0, 0, a, b, n, phases, c, workers);
            CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize()); CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("persistent, %2d workgroups over all XCDs, sc1 barrier: %6.2f us per phase\n", workers, ms * 1e3 / phases);
        }
        for (int workers : {8, 32}) {
            CHECK(hipMemset(c, 0, 64));
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_persistent<true>, dim3(workers * 8), dim3(256), 0, 0, a, b, n, phases, c, workers);
            CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize()); CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("persistent, %2d workgroups on XCD 0, device barrier  : %6.2f us per phase\n", workers, ms * 1e3 / phases);
        }
    }
    float h[4]; CHECK(hipMemcpy(h, a, 16, hipMemcpyDeviceToHost));
    printf("(a[0] = %.0f)\n", h[0]);
    return 0;
}
