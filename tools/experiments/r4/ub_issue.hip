// microbenchmark (round 4): vector-instruction issue throughput per SIMD at 1 / 2 / 4 / 8 waves per SIMD for the candidate forms of the
// sorted-insert ladder of patch_knn16_kernel: v_med3_u32, v_med3_f32, v_max_u32 + v_min_u32, v_pk_max_u16 + v_pk_min_u16, and plain
// v_add_u32 / v_fma_f32 / v_max3_f32 for reference.  Build: hipcc --offload-arch=gfx950 -O3 -o ub_issue ub_issue.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define L16(OP) OP(16,15) OP(15,14) OP(14,13) OP(13,12) OP(12,11) OP(11,10) OP(10,9) OP(9,8) OP(8,7) OP(7,6) OP(6,5) OP(5,4) OP(4,3) OP(3,2) OP(2,1) OP(1,0)

template <int V>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters)
{
    unsigned t[17];
#pragma unroll
    for (int s = 0; s < 17; ++s) t[s] = 0x7F000000u + s;
    unsigned key = threadIdx.x * 2654435761u + blockIdx.x;
    for (int it = 0; it < iters; ++it) {
        const unsigned x = (key >> 2) & 0x7EFFFFFFu;
        if (V == 0) {
#define OP(a, b) asm volatile("v_med3_u32 %0, %1, %2, %0" : "+v"(t[a]) : "v"(t[b]), "v"(x));
            L16(OP)
#undef OP
        } else if (V == 1) {
#define OP(a, b) asm volatile("v_med3_f32 %0, %1, %2, %0" : "+v"(t[a]) : "v"(t[b]), "v"(x));
            L16(OP)
#undef OP
        } else if (V == 2) {
            unsigned tmp;
#define OP(a, b) asm volatile("v_max_u32 %1, %2, %3\n v_min_u32 %0, %1, %0" : "+v"(t[a]), "=&v"(tmp) : "v"(t[b]), "v"(x));
            L16(OP)
#undef OP
        } else if (V == 3) {
            unsigned tmp;
#define OP(a, b) asm volatile("v_pk_max_u16 %1, %2, %3\n v_pk_min_u16 %0, %1, %0" : "+v"(t[a]), "=&v"(tmp) : "v"(t[b]), "v"(x));
            L16(OP)
#undef OP
        } else if (V == 4) {
#define OP(a, b) asm volatile("v_add_u32 %0, %1, %0" : "+v"(t[a]) : "v"(x));
            L16(OP)
#undef OP
        } else if (V == 5) {
#define OP(a, b) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(t[a]) : "v"(t[b]), "v"(x));
            L16(OP)
#undef OP
        } else if (V == 6) {
#define OP(a, b) asm volatile("v_max3_f32 %0, %1, %2, %0" : "+v"(t[a]) : "v"(t[b]), "v"(x));
            L16(OP)
#undef OP
        } else if (V == 7) {
#define OP(a, b) asm volatile("v_max_f32 %0, %1, %0" : "+v"(t[a]) : "v"(x));
            L16(OP)
#undef OP
        } else if (V == 8) {
            unsigned tmp;
#define OP(a, b) asm volatile("v_max_f32 %1, %2, %3\n v_min_f32 %0, %1, %0" : "+v"(t[a]), "=&v"(tmp) : "v"(t[b]), "v"(x));
            L16(OP)
#undef OP
        } else if (V == 9) {
#define OP(a, b) asm volatile("v_med3_i32 %0, %1, %2, %0" : "+v"(t[a]) : "v"(t[b]), "v"(x));
            L16(OP)
#undef OP
        }
        asm volatile("v_min_u32 %0, %0, %1" : "+v"(t[0]) : "v"(x));
        key = key * 1664525u + 1013904223u;
    }
    unsigned r = key;
#pragma unroll
    for (int s = 0; s < 17; ++s) r ^= t[s];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int V> void run(const char *name, int per_iter, int wg_per_cu)
{
    const int iters = 20000, grid = 256 * wg_per_cu;
    unsigned *out;
    CHECK(hipMalloc(&out, (size_t)grid * 256 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, out, 200);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    // wave-instructions per SIMD: every WG has 4 waves = one per SIMD -> wg_per_cu waves per SIMD
    const double instr = (double)iters * (per_iter + 5) * wg_per_cu;     // + the 5 loop-body extras (shift, and, min, mul-add ~2)
    printf("%-28s waves/SIMD %d: %.3f ms  -> %.2f ns per wave-instruction per SIMD (x2.4 GHz = %.2f cycles)\n", name, wg_per_cu, ms,
           ms * 1e6 / instr, ms * 1e6 / instr * 2.4);
    CHECK(hipFree(out));
}

int main()
{
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_med3_u32 x16", 16, w);
        run<1>("v_med3_f32 x16", 16, w);
        run<9>("v_med3_i32 x16", 16, w);
        run<2>("v_max_u32+v_min_u32 x16", 32, w);
        run<8>("v_max_f32+v_min_f32 x16", 32, w);
        run<3>("v_pk_max_u16+v_pk_min_u16", 32, w);
        run<4>("v_add_u32 x16", 16, w);
        run<5>("v_fma_f32 x16", 16, w);
        run<6>("v_max3_f32 x16", 16, w);
        run<7>("v_max_f32 x16", 16, w);
    }
    return 0;
}
