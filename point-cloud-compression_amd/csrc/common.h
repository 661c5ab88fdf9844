// Shared host/device helpers for libpccx (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/pccx.h"

#define PCCX_WAVE 64
#define PCCX_SUM_REPLICAS 8       // replicas of a column-sum accumulator (train.hip: workgroup b adds into replica b % 8; the consumer adds them up)

void pccx_set_error(const char *fmt, ...);

#define PCCX_CHECK_ARG(cond, ...)                    \
    do {                                             \
        if (!(cond)) {                               \
            pccx_set_error(__VA_ARGS__);             \
            return PCCX_ERR_ARG;                     \
        }                                            \
    } while (0)

#define PCCX_CHECK_HIP(expr)                                                              \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            pccx_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return PCCX_ERR_HIP;                                                          \
        }                                                                                 \
    } while (0)

#define PCCX_CHECK_LAUNCH() PCCX_CHECK_HIP(hipGetLastError())

#ifdef __HIPCC__
// Squared distance exactly as the oracle / torch evaluate it: three products, two adds,
// no FMA contraction (the library is built with -ffp-contract=off; the intrinsics make the
// intent explicit and survive flag changes).
__device__ __forceinline__ float pccx_sqdist(float ax, float ay, float az, float bx, float by, float bz)
{
    float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
    float d = __fmul_rn(dx, dx);
    d = __fadd_rn(d, __fmul_rn(dy, dy));
    d = __fadd_rn(d, __fmul_rn(dz, dz));
    return d;
}

__device__ __forceinline__ int pccx_lane() { return threadIdx.x & 63; }

// Zero `bytes` (a multiple of 4, 4-byte aligned) on the stream with a KERNEL rather than hipMemsetAsync.  The buffers cleared this way
// are accumulated into by the very next kernel (column sums, scatter-adds, Chamfer gradients), and the training step is replayed as
// a hipGraph: with memset NODES between the kernel nodes a replay issued after a short idle gap left non-finite gradients about
// every other time, and never with kernels serialised (AMD_SERIALIZE_KERNEL=3; tools/experiments/r3/dbg_train4.py).  As a kernel
// node the clear is ordered like every other kernel of the captured stream.
static __global__ void pccx_zero_kernel(unsigned *__restrict__ p, size_t n4)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
static inline hipError_t pccx_zero_async(void *p, size_t bytes, hipStream_t st)
{
#ifdef PCCX_ZERO_WITH_MEMSET       // experiment builds only (tools/experiments/r4/graph_memset_probe.py): the round-2 form, memset NODES in the captured step
    return hipMemsetAsync(p, 0, bytes, st);
#endif
    const size_t n4 = bytes / 4;
    if (n4 == 0) return hipSuccess;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pccx_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (unsigned *)p, n4);
    return hipGetLastError();
}

// Inclusive-of-lower-lanes population count of a wave ballot.
__device__ __forceinline__ int pccx_ballot_rank(unsigned long long mask)
{
    // number of set bits strictly below this lane
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}
#endif
