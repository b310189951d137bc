// ubench_valu.hip -- gfx950 VALU issue-rate microbenchmark behind the K1 design choices
// (DESIGN.md section 5): how many cycles does one wave64 instruction of each kind cost a SIMD,
// alone and in the inner-loop mix, at 1/2/4/8 waves per SIMD?
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o gpurun_out/ubench_valu
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

constexpr int kChains = 16;

template <int KIND>
__global__ __launch_bounds__(256) void k_ubench(float* out, int iters, float seed, long long* cyc)
{
    float a[kChains];
    v2f p[kChains];
    const float x = seed + threadIdx.x * 1e-7f, y = 1.0f - seed * 1e-3f;
    const v2f px = {x, x * 1.01f}, py = {y, y};
#pragma unroll
    for (int k = 0; k < kChains; k++) {
        a[k] = x + k;
        p[k] = v2f{x + k, y + k};
    }
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < kChains; k++) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(x), "v"(y));
            if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[k]) : "v"(px), "v"(py));
            if (KIND == 2) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[k]) : "v"(py));
            if (KIND == 3) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[k]) : "v"(px));
            if (KIND == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
            if (KIND == 5) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[k]) : "v"(y));
            if (KIND == 6) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[k]) : "s"(seed));
            if (KIND == 7) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[k]));
            if (KIND == 8) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[k]) : "v"(x), "v"(y));
            if (KIND == 9) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p[k]) : "v"(px), "v"(py));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < kChains; k++) s += a[k] + p[k].x + p[k].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// the K1 inner loop as an instruction mix, 4 independent interactions in flight per trip
// KIND 20: scalar (11 VALU / interaction, 3-D) ; KIND 21: packed pairs (12 VALU / 2 interactions)
template <int KIND>
__global__ __launch_bounds__(256) void k_mix(float* out, int iters, float seed, long long* cyc)
{
    const float xi = seed + threadIdx.x * 1e-3f, yi = seed * 0.5f, zi = seed * 0.25f;
    float ax[4] = {0, 0, 0, 0}, ay[4] = {0, 0, 0, 0}, az[4] = {0, 0, 0, 0};
    v2f pax[4], pay[4], paz[4];
    for (int k = 0; k < 4; k++) pax[k] = pay[k] = paz[k] = v2f{0.f, 0.f};
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f macc[4][2];
    for (int k = 0; k < 4; k++) macc[k][0] = macc[k][1] = v4f{0.f, 0.f, 0.f, 0.f};
    const v2f pxi = {xi, xi + 1.f}, pyi = {yi, yi + 1.f}, pzi = {zi, zi + 1.f};
    const v2f eps2 = {1e-4f, 1e-4f};
    float sx = seed * 3.f, sy = seed * 5.f, sz = seed * 7.f, sm = 1.0f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (KIND == 20) {
                float dx, dy, dz, r2, inv;
                asm volatile(
                    "v_sub_f32 %0, %8, %5\n\t"
                    "v_sub_f32 %1, %9, %6\n\t"
                    "v_sub_f32 %2, %10, %7\n\t"
                    "v_fma_f32 %3, %0, %0, %12\n\t"
                    "v_fma_f32 %3, %1, %1, %3\n\t"
                    "v_fma_f32 %3, %2, %2, %3\n\t"
                    "v_rcp_f32 %4, %3\n\t"
                    "v_mul_f32 %4, %11, %4\n\t"
                    : "=&v"(dx), "=&v"(dy), "=&v"(dz), "=&v"(r2), "=&v"(inv)
                    : "v"(xi), "v"(yi), "v"(zi), "v"(sx), "v"(sy), "v"(sz), "v"(sm), "v"(1e-4f));
                asm volatile(
                    "v_fmac_f32 %0, %3, %4\n\t"
                    "v_fmac_f32 %1, %3, %5\n\t"
                    "v_fmac_f32 %2, %3, %6\n\t"
                    : "+v"(ax[k]), "+v"(ay[k]), "+v"(az[k])
                    : "v"(inv), "v"(dx), "v"(dy), "v"(dz));
            } else {
                v2f dx, dy, dz, r2, inv;
                const v2f sxy = {sx, sy}, szm = {sz, sm};
                asm volatile(
                    "v_pk_add_f32 %0, %7, %4 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                    "v_pk_add_f32 %1, %7, %5 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                    "v_pk_add_f32 %2, %8, %6 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                    "v_pk_fma_f32 %3, %0, %0, %9\n\t"
                    "v_pk_fma_f32 %3, %1, %1, %3\n\t"
                    "v_pk_fma_f32 %3, %2, %2, %3\n\t"
                    : "=&v"(dx), "=&v"(dy), "=&v"(dz), "=&v"(r2)
                    : "v"(pxi), "v"(pyi), "v"(pzi), "v"(sxy), "v"(szm), "v"(eps2));
                inv = v2f{__builtin_amdgcn_rcpf(r2.x), __builtin_amdgcn_rcpf(r2.y)};
                asm volatile("v_pk_mul_f32 %0, %1, %0 op_sel:[1,0] op_sel_hi:[1,1]" : "+v"(inv) : "v"(szm));
                if (KIND == 22) {
                    // accumulation on the matrix pipe: rank-1 updates acc[4 comps] += a[comp] * s[target], one MFMA per
                    // half of the packed pair; a = the source's (x-c, y-c, z-c, 1)[lane % 4] (one v_mov stands in for it)
                    float aval;
                    asm volatile("v_mov_b32 %0, %1" : "=v"(aval) : "v"(sx));
                    macc[k][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(aval, inv.x, macc[k][0], 0, 0, 0);
                    macc[k][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(aval, inv.y, macc[k][1], 0, 0, 0);
                } else
                asm volatile(
                    "v_pk_fma_f32 %0, %3, %4, %0\n\t"
                    "v_pk_fma_f32 %1, %3, %5, %1\n\t"
                    "v_pk_fma_f32 %2, %3, %6, %2\n\t"
                    : "+v"(pax[k]), "+v"(pay[k]), "+v"(paz[k])
                    : "v"(inv), "v"(dx), "v"(dy), "v"(dz));
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int k = 0; k < 4; k++) s += ax[k] + ay[k] + az[k] + pax[k].x + pax[k].y + pay[k].x + pay[k].y + paz[k].x + paz[k].y;
    for (int k = 0; k < 4; k++)
        for (int h = 0; h < 2; h++) s += macc[k][h].x + macc[k][h].y + macc[k][h].z + macc[k][h].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <typename F>
static void run(const char* name, F launch, int cus, int wps, long long instr_per_wave, double inter_per_instr_block)
{
    float* out;
    long long* cyc;
    const int blocks = cus * wps;  // 256 threads = 4 waves = 1 wave per SIMD per block
    CHECK(hipMalloc(&out, sizeof(float) * 256 * (size_t)blocks));
    CHECK(hipMalloc(&cyc, sizeof(long long)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    launch(blocks, out, cyc);  // warm-up
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        CHECK(hipEventRecord(e0, 0));
        launch(blocks, out, cyc);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    long long hc = 0;
    CHECK(hipMemcpy(&hc, cyc, sizeof hc, hipMemcpyDeviceToHost));
    const double ns_per_instr_simd = (double)best * 1e6 / ((double)instr_per_wave * wps);
    printf("%-34s waves/SIMD=%d  %8.3f ms  ns/instr/SIMD=%6.3f  cyc@2.4GHz=%5.2f  clk-ticks/instr(wave0)=%6.3f", name, wps, best,
           ns_per_instr_simd, ns_per_instr_simd * 2.4, (double)hc / (double)instr_per_wave);
    if (inter_per_instr_block > 0) {
        const double inter = (double)blocks * 256.0 * inter_per_instr_block;
        printf("  => %.3e interactions/s", inter / (best * 1e-3));
    }
    printf("\n");
    CHECK(hipFree(out));
    CHECK(hipFree(cyc));
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("device %s arch %s CUs %d clock %d kHz\n", p.name, p.gcnArchName, cus, p.clockRate);
    const int iters = 4096;
#define RUN_KIND(K, NAME)                                                                                    \
    for (int w : {1, 2, 4, 8})                                                                               \
        run(NAME, [&](int b, float* o, long long* c) { hipLaunchKernelGGL(k_ubench<K>, dim3(b), dim3(256), 0, 0, o, iters, 1.5f, c); }, \
            cus, w, (long long)iters * kChains, 0.0);
    RUN_KIND(0, "v_fma_f32");
    RUN_KIND(8, "v_fmac_f32 (VOP2)");
    RUN_KIND(5, "v_mul_f32");
    RUN_KIND(6, "v_sub_f32 sgpr operand");
    RUN_KIND(1, "v_pk_fma_f32");
    RUN_KIND(9, "v_pk_fma_f32 op_sel broadcast");
    RUN_KIND(2, "v_pk_mul_f32");
    RUN_KIND(3, "v_pk_add_f32");
    RUN_KIND(4, "v_rcp_f32");
    RUN_KIND(7, "v_rsq_f32");
    for (int w : {1, 2, 4, 8})
        run("K1 mix scalar (11 VALU/inter)", [&](int b, float* o, long long* c) { hipLaunchKernelGGL(k_mix<20>, dim3(b), dim3(256), 0, 0, o, iters, 1.5f, c); },
            cus, w, (long long)iters * 4 * 11, (double)iters * 4);
    for (int w : {1, 2, 4, 8})
        run("K1 mix packed (12 VALU/2 inter)", [&](int b, float* o, long long* c) { hipLaunchKernelGGL(k_mix<21>, dim3(b), dim3(256), 0, 0, o, iters, 1.5f, c); },
            cus, w, (long long)iters * 4 * 12, (double)iters * 4 * 2);
    for (int w : {1, 2, 4, 8})
        run("K1 mix packed + 2 MFMA 4x4x1 (10 VALU + 2 MFMA / 2 inter)", [&](int b, float* o, long long* c) { hipLaunchKernelGGL(k_mix<22>, dim3(b), dim3(256), 0, 0, o, iters, 1.5f, c); },
            cus, w, (long long)iters * 4 * 12, (double)iters * 4 * 2);
    return 0;
}
