// microbenchmark: issue cost of v_med3_u32 chains as the in-patch kNN uses them, with the register numbers hipcc chose
// (all three sources in one VGPR bank) against bank-spread numbers; one or two waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int V>
__global__ __launch_bounds__(512) void k(unsigned *out, unsigned long long *cyc, int iters)
{
    unsigned key = threadIdx.x * 2654435761u;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (V == 0) {   // as compiled: v16,v24,v23..v17,v15..v8 with key in v28
            asm volatile(
                "v_mov_b32 v28, %0\n"
                "v_med3_u32 v16, v24, v28, v16\n v_med3_u32 v24, v23, v28, v24\n v_med3_u32 v23, v22, v28, v23\n v_med3_u32 v22, v21, v28, v22\n"
                "v_med3_u32 v21, v20, v28, v21\n v_med3_u32 v20, v19, v28, v20\n v_med3_u32 v19, v18, v28, v19\n v_med3_u32 v18, v17, v28, v18\n"
                "v_med3_u32 v17, v15, v28, v17\n v_med3_u32 v15, v14, v28, v15\n v_med3_u32 v14, v13, v28, v14\n v_med3_u32 v13, v12, v28, v13\n"
                "v_med3_u32 v12, v11, v28, v12\n v_med3_u32 v11, v10, v28, v11\n v_med3_u32 v10, v9, v28, v10\n v_med3_u32 v9, v8, v28, v9\n"
                "v_min_u32 v8, v8, v28\n"
                :: "v"(key) : "v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v28");
        } else if (V == 1) {   // list in v40..v56 ascending, key in v58 (bank 2): sources (s-1, key, s)
            asm volatile(
                "v_mov_b32 v58, %0\n"
                "v_med3_u32 v56, v55, v58, v56\n v_med3_u32 v55, v54, v58, v55\n v_med3_u32 v54, v53, v58, v54\n v_med3_u32 v53, v52, v58, v53\n"
                "v_med3_u32 v52, v51, v58, v52\n v_med3_u32 v51, v50, v58, v51\n v_med3_u32 v50, v49, v58, v50\n v_med3_u32 v49, v48, v58, v49\n"
                "v_med3_u32 v48, v47, v58, v48\n v_med3_u32 v47, v46, v58, v47\n v_med3_u32 v46, v45, v58, v46\n v_med3_u32 v45, v44, v58, v45\n"
                "v_med3_u32 v44, v43, v58, v44\n v_med3_u32 v43, v42, v58, v43\n v_med3_u32 v42, v41, v58, v42\n v_med3_u32 v41, v40, v58, v41\n"
                "v_min_u32 v40, v40, v58\n"
                :: "v"(key) : "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v58");
        } else if (V == 2) {   // min/max pair form: 2 ops per slot but 2-source VOP2
            asm volatile(
                "v_mov_b32 v58, %0\n"
                "v_max_u32 v57, v55, v58\n v_min_u32 v56, v57, v56\n v_max_u32 v57, v54, v58\n v_min_u32 v55, v57, v55\n"
                "v_max_u32 v57, v53, v58\n v_min_u32 v54, v57, v54\n v_max_u32 v57, v52, v58\n v_min_u32 v53, v57, v53\n"
                :: "v"(key) : "v52","v53","v54","v55","v56","v57","v58");
        } else if (V == 3) {   // 17 independent v_add_u32 (VOP2, 2 sources)
            asm volatile(
                "v_add_u32 v40, v40, %0\n v_add_u32 v41, v41, %0\n v_add_u32 v42, v42, %0\n v_add_u32 v43, v43, %0\n v_add_u32 v44, v44, %0\n v_add_u32 v45, v45, %0\n"
                "v_add_u32 v46, v46, %0\n v_add_u32 v47, v47, %0\n v_add_u32 v48, v48, %0\n v_add_u32 v49, v49, %0\n v_add_u32 v50, v50, %0\n v_add_u32 v51, v51, %0\n"
                "v_add_u32 v52, v52, %0\n v_add_u32 v53, v53, %0\n v_add_u32 v54, v54, %0\n v_add_u32 v55, v55, %0\n v_add_u32 v56, v56, %0\n"
                :: "v"(key) : "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56");
        } else if (V == 4) {   // 17 v_fma_f32 independent (VOP3, 3 sources)
            asm volatile(
                "v_fma_f32 v40, v40, %0, v41\n v_fma_f32 v41, v41, %0, v42\n v_fma_f32 v42, v42, %0, v43\n v_fma_f32 v43, v43, %0, v44\n v_fma_f32 v44, v44, %0, v45\n v_fma_f32 v45, v45, %0, v46\n"
                "v_fma_f32 v46, v46, %0, v47\n v_fma_f32 v47, v47, %0, v48\n v_fma_f32 v48, v48, %0, v49\n v_fma_f32 v49, v49, %0, v50\n v_fma_f32 v50, v50, %0, v51\n v_fma_f32 v51, v51, %0, v52\n"
                "v_fma_f32 v52, v52, %0, v53\n v_fma_f32 v53, v53, %0, v54\n v_fma_f32 v54, v54, %0, v55\n v_fma_f32 v55, v55, %0, v56\n v_fma_f32 v56, v56, %0, v40\n"
                :: "v"(key) : "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56");
        }
        key = key * 1664525u + 1013904223u;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = key;
}

template <int V> void run(const char *name, int threads, int ninstr)
{
    unsigned *out; unsigned long long *cyc;
    const int blocks = 256, iters = 20000;
    CHECK(hipMalloc(&out, blocks * 512 * 4)); CHECK(hipMalloc(&cyc, blocks * 8 * 8));
    CHECK(hipMemset(cyc, 0, blocks * 8 * 8));
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters); CHECK(hipDeviceSynchronize()); }
    unsigned long long h[256 * 8]; CHECK(hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost));
    double s = 0; int n = 0;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < threads / 64; ++w) { s += (double)h[b * 8 + w]; ++n; }
    printf("%-34s %d waves/SIMD: %.2f cycles per wave-instruction (%d instr + 3 per iteration)\n", name, threads / 256, s / n / iters / (ninstr + 3), ninstr);
    CHECK(hipFree(out)); CHECK(hipFree(cyc));
}
int main()
{
    for (int th : {256, 512}) {
        if (th == 256) { run<0>("med3, hipcc's registers", 256, 18); run<1>("med3, bank-spread registers", 256, 18); run<2>("max+min pairs (VOP2)", 256, 9); run<3>("v_add_u32 x17 (VOP2)", 256, 17); run<4>("v_fma_f32 x17 (VOP3)", 256, 17); }
        else { run<0>("med3, hipcc's registers", 512, 18); run<1>("med3, bank-spread registers", 512, 18); run<2>("max+min pairs (VOP2)", 512, 9); run<3>("v_add_u32 x17 (VOP2)", 512, 17); run<4>("v_fma_f32 x17 (VOP3)", 512, 17); }
    }
    return 0;
}
