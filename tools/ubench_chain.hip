// ubench_chain.hip -- gfx950 DEPENDENT-issue latencies behind the bit-exact kernels' design (strict path):
// how many cycles between two back-to-back dependent VALU instructions of one wave, what an 8-wave s_barrier round costs,
// and the LDS b128/b64 write+read rate of one wave.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_chain.hip -o tools/ubench_chain
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float v2f __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int KIND, int CHAINS>
__global__ __launch_bounds__(512) void k_chain(float* out, int iters, float seed, long long* cyc)
{
    float a[CHAINS];
    v2f p[CHAINS];
    const float x = seed + threadIdx.x * 1e-7f, y = 1.0f - seed * 1e-3f;
    const v2f px = {x, x * 1.01f}, py = {y, y};
    for (int k = 0; k < CHAINS; k++) { a[k] = x + k; p[k] = v2f{x + k, y + k}; }
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
#pragma unroll
            for (int k = 0; k < CHAINS; k++) {
                if (KIND == 0) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[k]) : "v"(x));
                if (KIND == 1) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[k]) : "v"(px));
                if (KIND == 2) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(x), "v"(y));
                if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[k]) : "v"(px), "v"(py));
                if (KIND == 4) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[k]) : "v"(py));
                if (KIND == 5) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
                if (KIND == 6) asm volatile("v_add_f32 %0, %2, %0\n\tv_add_f32 %1, %3, %1" : "+v"(p[k].x), "+v"(p[k].y) : "v"(x), "v"(y));
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int k = 0; k < CHAINS; k++) s += a[k] + p[k].x + p[k].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

__global__ __launch_bounds__(512) void k_barrier(float* out, int iters, long long* cyc)
{
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) __syncthreads();
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
    out[threadIdx.x] = 0.f;
}

// one wave writes (KIND 0) or reads (KIND 1) 16 rows of 64 x 16 B per trip; KIND 2/3 = 8-byte accesses
template <int KIND>
__global__ __launch_bounds__(512) void k_lds(float* out, int iters, long long* cyc)
{
    __shared__ float4 buf[32][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float4 acc = make_float4(lane, 1.f, 2.f, 3.f);
    for (int r = w; r < 32; r += blockDim.x / 64) buf[r][lane] = acc;
    __syncthreads();
    float2* b2 = reinterpret_cast<float2*>(&buf[0][0]);
    float4 keep[4] = {acc, acc, acc, acc}; float2 keep2[4] = {};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            if (KIND == 0) buf[r][lane] = acc;
            if (KIND == 1) { float4 t; asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"((unsigned)((r * 64 + lane) * 16)) : "memory"); keep[r & 3] = t; }
            if (KIND == 2) b2[r * 64 + lane] = make_float2(acc.x, acc.y);
            if (KIND == 3) { float2 t; asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"((unsigned)((r * 64 + lane) * 8)) : "memory"); keep2[r & 3] = t; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
    out[threadIdx.x] = acc.x + buf[3][lane].y + keep[0].x + keep[1].y + keep[2].z + keep[3].w + keep2[0].x + keep2[1].y + keep2[2].x + keep2[3].y;
}

int main()
{
    float* out; long long* cyc; CHECK(hipMalloc(&out, 1 << 20)); CHECK(hipMalloc(&cyc, 8));
    const int iters = 20000;
    long long c;
    const char* names[] = {"v_add_f32", "v_pk_add_f32", "v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_rcp_f32", "v_add_f32 x2 (fx,fy)"};
#define RUN(KIND, CH, THREADS)                                                                          \
    hipLaunchKernelGGL((k_chain<KIND, CH>), dim3(1), dim3(THREADS), 0, 0, out, 10, 1.0f, cyc);        \
    hipLaunchKernelGGL((k_chain<KIND, CH>), dim3(1), dim3(THREADS), 0, 0, out, iters, 1.0f, cyc);     \
    CHECK(hipDeviceSynchronize()); CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));                \
    printf("%-22s chains %d waves/SIMD %d : %.2f cycles per dependent step (s_memtime ticks x clock ratio not applied)\n", names[KIND], CH, THREADS / 256, (double)c / (iters * 16.0));
    RUN(0, 1, 256) RUN(1, 1, 256) RUN(2, 1, 256) RUN(3, 1, 256) RUN(4, 1, 256) RUN(5, 1, 256) RUN(6, 1, 256)
    RUN(0, 2, 256) RUN(1, 2, 256) RUN(3, 2, 256) RUN(3, 4, 256) RUN(1, 4, 256)
    RUN(0, 1, 512) RUN(1, 1, 512) RUN(3, 1, 512)
    for (int threads : {256, 512}) {
        hipLaunchKernelGGL(k_barrier, dim3(1), dim3(threads), 0, 0, out, iters, cyc);
        CHECK(hipDeviceSynchronize()); CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
        printf("s_barrier, %d waves: %.1f ticks per round\n", threads / 64, (double)c / iters);
    }
    const char* ln[] = {"ds_write_b128", "ds_read_b128", "ds_write_b64", "ds_read_b64"};
#define RUNL(KIND, THREADS)                                                                              \
    hipLaunchKernelGGL((k_lds<KIND>), dim3(1), dim3(THREADS), 0, 0, out, iters, cyc);                    \
    CHECK(hipDeviceSynchronize()); CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));                  \
    printf("%-14s %d waves: %.1f ticks per wave access (64 lanes)\n", ln[KIND], THREADS / 64, (double)c / (iters * 16.0));
    RUNL(0, 64) RUNL(1, 64) RUNL(2, 64) RUNL(3, 64) RUNL(0, 256) RUNL(1, 256) RUNL(0, 512) RUNL(1, 512)
    // tick rate vs core clock: time a known v_add chain by wall clock
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_chain<0, 1>), dim3(1), dim3(256), 0, 0, out, iters * 20, 1.0f, cyc);
    CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    printf("tick rate: %.1f MHz (%lld ticks in %.3f ms); v_add_f32 chain: %.2f ns per step\n", c / (ms * 1e3), c, ms, ms * 1e6 / (iters * 20 * 16.0));
    return 0;
}
