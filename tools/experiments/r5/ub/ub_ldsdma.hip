// Micro-benchmark: how fast does ONE CU take L2-resident bytes, through LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction) and
// through plain register loads (global_load_dwordx4), with 4 / 8 / 12 waves per CU issuing?  The planes GEMM of csrc/planes.hip streams a
// 16 KiB weight chunk per k-step per workgroup through LDS-DMA: is that path the bound of its 512 -> 1024 layer (4.3 GB in 0.65 ms)?
//   hipcc --offload-arch=gfx950 -O3 ub_ldsdma.hip -o /tmp/ub_ldsdma && /tmp/ub_ldsdma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((address_space(3))) unsigned lds_u32;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <bool DMA>
__global__ __launch_bounds__(256) void stream_kernel(const uint4 *__restrict__ src, size_t table_frags, int iters, unsigned *__restrict__ sink)
{
    __shared__ __attribute__((aligned(16))) uint4 ring[3 * 16 * 64];          // 48 KiB, as the GEMM's ring
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // every workgroup walks the whole table (L2-resident: a few MB), its four waves taking four fragments each per 16-fragment chunk
    const unsigned mask = (unsigned)table_frags - 1u;                          // table_frags is a power of two
    unsigned f = (blockIdx.x * 977u) & mask;
    // (first version of the register form: inline-asm loads whose results were "not used" -- the compiler reused their destination
    //  registers for the next iteration's POINTER while the loads were in flight, the data landing there became an address: a memory
    //  fault on the box.  The hazard class tools/asm_load_lint.py exists for; here the loads are plain C++, one iteration ahead.)
    uint4 acc = make_uint4(0, 0, 0, 0), cur[4] = {acc, acc, acc, acc};
    for (int it = 0; it < iters; ++it) {
        uint4 nxt[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned fr = (f + 4u * w + q) & mask;
            const uint4 *p = src + (size_t)fr * 64 + lane;
            if (DMA) {
                uint4 *dst = ring + ((it % 3) * 16 + 4 * w + q) * 64;
                __builtin_amdgcn_global_load_lds((const void *)p, (lds_u32 *)(uintptr_t)dst, 16, 0, 0);
            } else {
                nxt[q] = *p;
            }
        }
        if (DMA) {
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                    // two chunks' worth in flight
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) { acc.x ^= cur[q].x ^ cur[q].y ^ cur[q].z ^ cur[q].w; cur[q] = nxt[q]; }   // consumes the PREVIOUS iteration's loads
        }
        f = (f + 16u) & mask;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (DMA) acc = ring[threadIdx.x];
    else acc.x ^= cur[0].x ^ cur[1].y ^ cur[2].z ^ cur[3].w;
    if (acc.x == 0x12345678u) sink[0] = acc.x;
}

int main()
{
    const size_t table_bytes = 2u << 20;                                        // 2 MiB: stays in every XCD's L2
    const size_t frags = table_bytes / 1024;
    uint4 *src; unsigned *sink;
    CHECK(hipMalloc(&src, table_bytes)); CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(src, 1, table_bytes));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 4096;
    for (int wg_per_cu = 1; wg_per_cu <= 3; ++wg_per_cu)
        for (int dma = 1; dma >= 0; --dma) {
            const int grid = 256 * wg_per_cu;
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipEventRecord(e0));
                if (dma) hipLaunchKernelGGL(stream_kernel<true>, dim3(grid), dim3(256), 0, 0, src, frags, iters, sink);
                else hipLaunchKernelGGL(stream_kernel<false>, dim3(grid), dim3(256), 0, 0, src, frags, iters, sink);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            }
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double bytes = (double)grid * iters * 16 * 1024;
            printf("%s  %d workgroup(s) of 4 waves per CU: %.2f TB/s chip, %.1f GB/s per CU (%.3f ms)\n", dma ? "LDS-DMA      " : "register load", wg_per_cu,
                   bytes / ms / 1e9, bytes / ms / 1e6 / 256, ms);
        }
    return 0;
}
