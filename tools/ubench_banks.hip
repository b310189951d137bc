// ubench_banks.hip -- does the issue cost of v_pk_fma_f32 / v_fma_f32 on gfx950 depend on WHICH VGPRs
// feed it (register-file bank/port conflicts)?  Fixed physical registers, 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

#define REP4(X) X X X X
#define REP16(X) REP4(X) REP4(X) REP4(X) REP4(X)

// every kernel: 64 instructions per loop trip, fixed registers v[10..89] (declared clobbered)
#define CLOB "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29", \
  "v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49", \
  "v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69"

#define KERNEL(NAME, BODY)                                                             \
    __global__ __launch_bounds__(256) void NAME(float* out, int iters)                 \
    {                                                                                  \
        asm volatile("v_mov_b32 v10, 1.0\n v_mov_b32 v11, 1.0\n v_mov_b32 v12, 0.5\n v_mov_b32 v13, 0.5\n" \
                     "v_mov_b32 v14, 0.25\n v_mov_b32 v15, 0.25\n v_mov_b32 v16, 0.125\n v_mov_b32 v17, 0.125\n" \
                     "v_mov_b32 v18, 1.0\n v_mov_b32 v19, 1.0\n v_mov_b32 v20, 1.0\n v_mov_b32 v21, 1.0\n" ::: CLOB); \
        for (int it = 0; it < iters; it++) { asm volatile(REP16(BODY) ::: CLOB); }      \
        float r; asm volatile("v_mov_b32 %0, v40" : "=v"(r)::CLOB);                     \
        out[blockIdx.x * 256 + threadIdx.x] = r;                                        \
    }

// 4 instructions per BODY, 4 independent accumulators
// A: pk_fma, sources pairs v[10:11], v[12:13]  (banks: 10%4=2, 12%4=0), acc even-aligned spread
KERNEL(k_pk_spread, "v_pk_fma_f32 v[40:41], v[10:11], v[12:13], v[40:41]\n v_pk_fma_f32 v[42:43], v[10:11], v[12:13], v[42:43]\n v_pk_fma_f32 v[44:45], v[10:11], v[12:13], v[44:45]\n v_pk_fma_f32 v[46:47], v[10:11], v[12:13], v[46:47]\n")
// B: all three operands in the same bank class (regs = 0 mod 4): v[12:13], v[16:17], acc v[40:41],v[44:45]...
KERNEL(k_pk_samebank, "v_pk_fma_f32 v[40:41], v[12:13], v[16:17], v[40:41]\n v_pk_fma_f32 v[44:45], v[12:13], v[16:17], v[44:45]\n v_pk_fma_f32 v[48:49], v[12:13], v[16:17], v[48:49]\n v_pk_fma_f32 v[52:53], v[12:13], v[16:17], v[52:53]\n")
// C: src0 == src1 (d*d + r2)
KERNEL(k_pk_sq, "v_pk_fma_f32 v[40:41], v[10:11], v[10:11], v[40:41]\n v_pk_fma_f32 v[42:43], v[10:11], v[10:11], v[42:43]\n v_pk_fma_f32 v[44:45], v[10:11], v[10:11], v[44:45]\n v_pk_fma_f32 v[46:47], v[10:11], v[10:11], v[46:47]\n")
// D: pk_mul two sources
KERNEL(k_pk_mul, "v_pk_mul_f32 v[40:41], v[10:11], v[40:41]\n v_pk_mul_f32 v[42:43], v[10:11], v[42:43]\n v_pk_mul_f32 v[44:45], v[10:11], v[44:45]\n v_pk_mul_f32 v[46:47], v[10:11], v[46:47]\n")
// E: pk_add with neg + op_sel broadcast as in K1
KERNEL(k_pk_addsel, "v_pk_add_f32 v[40:41], v[10:11], v[12:13] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 v[42:43], v[10:11], v[14:15] op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 v[44:45], v[12:13], v[16:17] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 v[46:47], v[14:15], v[18:19] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n")
// F: scalar fma spread
KERNEL(k_fma_spread, "v_fma_f32 v40, v10, v13, v40\n v_fma_f32 v41, v10, v13, v41\n v_fma_f32 v42, v10, v13, v42\n v_fma_f32 v43, v10, v13, v43\n")
// G: scalar fma same bank (12,16, acc 40,44,48,52)
KERNEL(k_fma_samebank, "v_fma_f32 v40, v12, v16, v40\n v_fma_f32 v44, v12, v16, v44\n v_fma_f32 v48, v12, v16, v48\n v_fma_f32 v52, v12, v16, v52\n")
// H: v_fmac (VOP2) and v_mul
KERNEL(k_fmac, "v_fmac_f32 v40, v10, v13\n v_fmac_f32 v41, v10, v13\n v_fmac_f32 v42, v10, v13\n v_fmac_f32 v43, v10, v13\n")
KERNEL(k_mul, "v_mul_f32 v40, v10, v40\n v_mul_f32 v41, v10, v41\n v_mul_f32 v42, v10, v42\n v_mul_f32 v43, v10, v43\n")
// I: rcp interleaved with pk_fma 1:3 (does the transcendental overlap with plain VALU?)
KERNEL(k_rcp_mix, "v_rcp_f32 v40, v40\n v_pk_fma_f32 v[42:43], v[10:11], v[12:13], v[42:43]\n v_pk_fma_f32 v[44:45], v[10:11], v[12:13], v[44:45]\n v_pk_fma_f32 v[46:47], v[10:11], v[12:13], v[46:47]\n")
KERNEL(k_rcp_only, "v_rcp_f32 v40, v40\n v_rcp_f32 v41, v41\n v_rcp_f32 v42, v42\n v_rcp_f32 v43, v43\n")
// J: pk_fma whose accumulator/result differs from sources and dst != src2 (full 4-address)
KERNEL(k_pk_4addr, "v_pk_fma_f32 v[40:41], v[10:11], v[12:13], v[14:15]\n v_pk_fma_f32 v[42:43], v[10:11], v[12:13], v[16:17]\n v_pk_fma_f32 v[44:45], v[10:11], v[12:13], v[18:19]\n v_pk_fma_f32 v[46:47], v[10:11], v[12:13], v[20:21]\n")
// K: pk_fma with only op_sel broadcast on src0 (reads one dword of the pair?)
KERNEL(k_pk_bcast, "v_pk_fma_f32 v[40:41], v[10:11], v[12:13], v[40:41] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[42:43], v[10:11], v[12:13], v[42:43] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[44:45], v[10:11], v[12:13], v[44:45] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[46:47], v[10:11], v[12:13], v[46:47] op_sel_hi:[0,1,1]\n")

template <typename K>
void run(const char* name, K kern, int cus)
{
    float* out;
    const int iters = 2048;
    for (int wps : {4, 8}) {
        const int blocks = cus * wps;
        CHECK(hipMalloc(&out, sizeof(float) * 256 * (size_t)blocks));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters);
        CHECK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int r = 0; r < 5; r++) {
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters);
            CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        const double ninstr = (double)iters * 64.0 * wps;
        printf("%-16s waves/SIMD=%d  %7.3f ms  ns/instr/SIMD=%6.3f (cycles @2.2GHz %5.2f)\n", name, wps, best, best * 1e6 / ninstr, best * 1e6 / ninstr * 2.2);
        CHECK(hipFree(out));
    }
}

int main()
{
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    run("pk_fma spread", k_pk_spread, cus);
    run("pk_fma samebank", k_pk_samebank, cus);
    run("pk_fma d*d", k_pk_sq, cus);
    run("pk_fma 4addr", k_pk_4addr, cus);
    run("pk_fma bcast", k_pk_bcast, cus);
    run("pk_mul", k_pk_mul, cus);
    run("pk_add opsel", k_pk_addsel, cus);
    run("fma spread", k_fma_spread, cus);
    run("fma samebank", k_fma_samebank, cus);
    run("fmac vop2", k_fmac, cus);
    run("mul", k_mul, cus);
    run("rcp only", k_rcp_only, cus);
    run("rcp+3pk_fma", k_rcp_mix, cus);
    return 0;
}
