// planes.hip -- the wide Conv1x1 / Linear stacks of the PointNet++ families (pointnet_sa_module.py:38-93, PPPF_AE.py:64-107)
// in the bf16x3 arithmetic, with the activations kept between layers as the three bf16 planes of the NEXT layer's MFMA B
// operand instead of fp32 rows:
//
//   planes[t][tile][plane][lane]   (16-byte vectors)   t = K/32 block, tile = 16 consecutive rows (points), lane (g, n):
//                                  the eight channels 32t + 16h + 4g + r (h = 0,1; r = 0..3) of row 16*tile + n
//
// which is both what b3_split8 makes of two adjacent C tiles of a layer's output and what v_mfma_f32_16x16x32_bf16 takes as its
// B operand, so a layer's epilogue writes the next layer's operand with coalesced 1 KiB stores and nothing is split twice.
//
//   group_planes_kernel : gather (ball-query / kNN indices, -1 -> row 0 as pointnet_sa_module.py:27) + concat [features, xyz]
//                         + split  ->  planes of the first layer; without indices: fp32 rows -> planes.
//   planes_gemm_kernel  : one layer.  Workgroup = 128 rows x (16*MB) output channels, 4 waves x (2 row tiles x MB m-tiles);
//                         the weight planes of the m-block stream through a 4-deep LDS-DMA ring shared by the waves, the B
//                         planes of the wave's two tiles are loaded two k-steps ahead into rotating register sets (the scheme
//                         of dec_main_kernel<true>, decoder.hip).  Epilogues: planes (bias + ReLU + split), fp32 rows, or the
//                         max over groups of `group` consecutive rows (torch.max over nsample, pointnet_sa_module.py:91).
// MFMA-bound for K, N >= 256; narrower layers are bound by the 6 bytes per activation they read and write.
#include <math.h>
#include <type_traits>

#include "common.h"
#include "mfma_chain.h"

#define PG_NB 4                          // ring depth (DMA three chunks ahead)

// The arithmetic of a planes kernel: P = 3 -> bf16x3 (three bf16 planes per operand, six products), P = 2 -> f16x2 (two fp16 planes of the
// operand times an exact power of two, three products; mfma_chain.h).  A ring chunk holds 4 m-tiles x P planes (1 KiB fragments), so a wave
// issues P DMA loads per chunk and 2 P plane loads per k-step: the counted waits below are written in terms of P.
template <int P> struct PgArith;
template <> struct PgArith<3> {
    typedef bf16x8 vec;
    static constexpr int NQ = 6;
    static __device__ __forceinline__ constexpr int pa(int q) { return q == 0 ? 2 : (q == 1 ? 0 : (q == 2 ? 1 : (q == 3 ? 1 : 0))); }   // (lo,hi) (hi,lo) (mid,mid)
    static __device__ __forceinline__ constexpr int pb(int q) { return q == 0 ? 0 : (q == 1 ? 2 : (q == 2 ? 1 : (q == 3 ? 0 : (q == 4 ? 1 : 0)))); }   // (mid,hi) (hi,mid) (hi,hi)
    static __device__ __forceinline__ void split(const f32x4 &a, const f32x4 &b, float, vec (&pl)[3]) { b3_split8(a, b, pl); }
    static __device__ __forceinline__ f32x4 mfma(const vec &a, const vec &b, const f32x4 &c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct PgArith<2> {
    typedef f16x8 vec;
    static constexpr int NQ = 3;
    static __device__ __forceinline__ constexpr int pa(int q) { return q == 0 ? 1 : 0; }      // smallest first: (lo,hi) (hi,lo) (hi,hi)
    static __device__ __forceinline__ constexpr int pb(int q) { return q == 1 ? 1 : 0; }
    static __device__ __forceinline__ void split(const f32x4 &a, const f32x4 &b, float rho, vec (&pl)[2]) { h2_split8(a, b, rho, pl); }
    static __device__ __forceinline__ f32x4 mfma(const vec &a, const vec &b, const f32x4 &c) { return H2_MFMA(a, b, c); }
};
// f16x2 only: the stack's dynamic input normalisation (pccx_dyn_scale): dyn[0] = s, a power of two with |input| s <= 1, dyn[1] = 1 / s.
// Conv / ReLU stacks are positively homogeneous in (input, biases): the kernels multiply the gathered input and every bias by s and the
// stack's final rows by 1 / s, so the static interval bounds of the layers (for inputs of magnitude <= 1) hold whatever the data.
__device__ __forceinline__ float pg_dyn(const float *dyn, int i) { return dyn ? dyn[i] : 1.f; }

// ---- dynamic input normalisation of an f16x2 stack ---------------------------------------------------------------------------------
// |x| maxima are collected as the bit patterns of non-negative floats (monotone as unsigned integers) with atomicMax into one of
// PG_AMAX_REP replicas (the same-address atomics of a large grid spread over eight words); pccx_dyn_scale turns one or two such maxima
// into the stack's power-of-two scale.  The slots are cleared by the caller once per forward pass (pccx_zero_bytes).
#define PG_AMAX_REP 8
__device__ __forceinline__ float pg_wave_max(float m)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    return m;
}
__device__ __forceinline__ void pg_amax_commit(float *amax, float m)          // m: the lane's largest |value|
{
    m = pg_wave_max(m);
    if ((threadIdx.x & 63) == 0) {
        // the maximum settles after the first few waves: a plain (L2) read first, the atomic only when this wave would raise the word --
        // 32 768 waves of a million-row stack otherwise queue 4096 deep on each of the eight words (measured +0.15 ms on a 0.3 ms stack)
        unsigned *w = (unsigned *)amax + (blockIdx.x & (PG_AMAX_REP - 1));
        const unsigned mine = __float_as_uint(m);
        if (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < mine) atomicMax(w, mine);
    }
}
__global__ __launch_bounds__(256) void absmax_kernel(const float *__restrict__ x, size_t n, float *__restrict__ amax)
{
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(x[i]));
    pg_amax_commit(amax, m);
}
// |x|max over n floats, folded into amax (PG_AMAX_REP floats, non-negative; cleared by the caller before the first contribution)
extern "C" int pccx_absmax(const float *x, int64_t n, float *amax8, void *stream)
{
    if (n == 0) return PCCX_OK;
    PCCX_CHECK_ARG(x && amax8 && n > 0, "pccx_absmax: bad arguments");
    long long blocks = (n + 16383) / 16384;                 // >= 64 values per thread: few waves, few atomics (they start together and all see 0)
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (size_t)n, amax8);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}
// bound = combine(a1 max(m1), a2 max(m2)) + add  (combine: 0 = sum, 1 = max; m2 may be NULL), inflated by 2^-20 for the roundings of this
// very expression; dyn[0] = s = the largest power of two <= 1 with bound * s <= 1 (at least 2^-60; 1 when the bound is 0), dyn[1] = 1 / s
__global__ void dyn_scale_kernel(const float *__restrict__ m1, float a1, const float *__restrict__ m2, float a2, float add, int combine,
                                 float *__restrict__ dyn)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float v1 = 0.f, v2 = 0.f;
    for (int i = 0; i < PG_AMAX_REP; ++i) {
        v1 = fmaxf(v1, m1[i]);
        if (m2) v2 = fmaxf(v2, m2[i]);
    }
    float bound = (combine ? fmaxf(a1 * v1, a2 * v2) : a1 * v1 + a2 * v2) + add;
    bound *= 1.f + 0x1p-20f;
    float s_ = 1.f;
    if (bound > 0.f && bound < INFINITY) {
        int e;
        const float f = frexpf(bound, &e);                   // bound = f 2^e, f in [0.5, 1)
        int k = f == 0.5f ? e - 1 : e;                       // ceil(log2 bound)
        k = k < 0 ? 0 : (k > 60 ? 60 : k);                   // s <= 1: small inputs are not blown up (the layer bounds count the biases at full size)
        s_ = ldexpf(1.f, -k);
    }
    dyn[0] = s_;
    dyn[1] = 1.f / s_;
}
extern "C" int pccx_dyn_scale(const float *m1_8, float a1, const float *m2_8, float a2, float add, int combine, float *dyn2, void *stream)
{
    PCCX_CHECK_ARG(m1_8 && dyn2 && a1 >= 0.f && a2 >= 0.f && add >= 0.f && (combine == 0 || combine == 1), "pccx_dyn_scale: bad arguments");
    hipLaunchKernelGGL(dyn_scale_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, m1_8, a1, m2_8, a2, add, combine, dyn2);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

static inline int pg_kt32(int K) { return ((K + 15) / 16 + 1) / 2; }
static inline int pg_mb(int N) { return (N + 15) / 16 <= 4 ? 4 : 8; }

static size_t pg_planes_floats(int64_t M, int K, int P)
{
    const size_t ntiles = (size_t)((M > 0 ? M : 0) + 15) / 16;
    return (size_t)pg_kt32(K > 0 ? K : 1) * ntiles * P * 256;
}
extern "C" size_t pccx_planes_floats(int64_t M, int K) { return pg_planes_floats(M, K, 3); }
extern "C" size_t pccx_planes_floats_h2(int64_t M, int K) { return pg_planes_floats(M, K, 2); }

// ---- gather + concat + split ------------------------------------------------------------------------------------
// One wave per row tile.  Row r takes source row s = idx ? (r / rows_per_batch) * n_src + max(idx[r], 0) : r; its channels are
// f0[s][0..C0) followed by f1[s][0..C1).
template <int P>
__global__ __launch_bounds__(256) void group_planes_kernel(const float *__restrict__ f0, int C0, int ld0, const float *__restrict__ f1,
                                                           int C1, int ld1, const int64_t *__restrict__ idx, long long M,
                                                           long long rows_per_batch, long long n_src, int KT32, long long ntiles,
                                                           uint4 *__restrict__ planes, long long mod0, long long div1, float rho,
                                                           const float *__restrict__ dyn)
{
    typedef PgArith<P> AR;
    rho *= pg_dyn(dyn, 0);
    const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
    const long long tile = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    long long r = tile * 16 + n;
    if (r >= M) r = M - 1;                                   // padded rows repeat the last one (never read back as results)
    long long s = r;
    if (idx) {
        const long long j = idx[r];
        s = (r / rows_per_batch) * n_src + (j < 0 ? 0 : j);
    }
    // without indices the two sources may be rows of different tables: f0 row r % mod0 (mod0 > 0), f1 row r / div1
    const float *p0 = f0 ? f0 + (size_t)(mod0 > 0 ? s % mod0 : s) * ld0 : nullptr;
    const float *p1 = f1 ? f1 + (size_t)(s / div1) * ld1 : nullptr;
    const bool vec0 = p0 && (ld0 % 4 == 0) && ((uintptr_t)f0 % 16 == 0);
    const int C = C0 + C1;
    for (int t = 0; t < KT32; ++t) {
        f32x4 v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = 32 * t + 16 * h + 4 * g;
            if (vec0 && c + 3 < C0) {
                v[h] = *(const f32x4 *)(p0 + c);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int cc = c + q;
                    v[h][q] = cc < C0 ? p0[cc] : (cc < C ? p1[cc - C0] : 0.f);
                }
            }
        }
        typename AR::vec pl[P];
        AR::split(v[0], v[1], rho, pl);
        uint4 *d = planes + (((size_t)t * ntiles + tile) * P) * 64 + lane;
#pragma unroll
        for (int p = 0; p < P; ++p) d[p * 64] = __builtin_bit_cast(uint4, pl[p]);
    }
}

template <int P>
static int group_planes_launch(const float *f0, int C0, int ld0, const float *f1, int C1, int ld1, const int64_t *idx, int64_t M,
                               int64_t rows_per_batch, int64_t n_src, float *planes, float rho, const float *dyn, void *stream, const char *who)
{
    if (M == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(planes && M > 0, "%s: null output or negative M", who);
    PCCX_CHECK_ARG(C0 >= 0 && C1 >= 0 && C0 + C1 >= 1 && (C0 == 0 || (f0 && ld0 >= C0)) && (C1 == 0 || (f1 && ld1 >= C1)),
                   "%s: bad sources C0=%d C1=%d", who, C0, C1);
    PCCX_CHECK_ARG(!idx || (rows_per_batch >= 1 && n_src >= 1), "%s: indices need rows_per_batch and n_src", who);
    const long long ntiles = (M + 15) / 16;
    PCCX_CHECK_ARG((ntiles + 3) / 4 <= 0x7fffffffLL, "%s: M too large", who);
    hipLaunchKernelGGL(group_planes_kernel<P>, dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, (hipStream_t)stream, C0 ? f0 : nullptr, C0,
                       ld0, C1 ? f1 : nullptr, C1, ld1, idx, (long long)M, (long long)(idx ? rows_per_batch : 1),
                       (long long)(idx ? n_src : 1), pg_kt32(C0 + C1), ntiles, (uint4 *)planes, 0LL, 1LL, rho, dyn);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}
extern "C" int pccx_group_planes(const float *f0, int C0, int ld0, const float *f1, int C1, int ld1, const int64_t *idx, int64_t M,
                                 int64_t rows_per_batch, int64_t n_src, float *planes, void *stream)
{
    return group_planes_launch<3>(f0, C0, ld0, f1, C1, ld1, idx, M, rows_per_batch, n_src, planes, 1.f, nullptr, stream, "pccx_group_planes");
}
// f16x2: planes of (rho * dyn[0]) * value, two fp16 pieces (pccx_planes_floats_h2(M, C0 + C1) floats); dyn = NULL: no dynamic factor
extern "C" int pccx_group_planes_h2(const float *f0, int C0, int ld0, const float *f1, int C1, int ld1, const int64_t *idx, int64_t M,
                                    int64_t rows_per_batch, int64_t n_src, float rho, const float *dyn, float *planes, void *stream)
{
    PCCX_CHECK_ARG(rho > 0.f, "pccx_group_planes_h2: rho must be a positive power of two");
    return group_planes_launch<2>(f0, C0, ld0, f1, C1, ld1, idx, M, rows_per_batch, n_src, planes, rho, dyn, stream, "pccx_group_planes_h2");
}

// torch.cat([a, b.unsqueeze(1).repeat(1, P, 1)], -1) as planes (the inputs of FoldingNet's two stacks, PPPF_AE.py:99-106): row r has
// the C0 channels of f0 row (mod0 > 0 ? r % mod0 : r) followed by the C1 channels of f1 row r / div1.  Nothing is concatenated or
// repeated in memory.
template <int P>
static int fold_planes_launch(const float *f0, int C0, int ld0, int64_t mod0, const float *f1, int C1, int ld1, int64_t div1, int64_t M,
                              float *planes, float rho, const float *dyn, void *stream, const char *who)
{
    if (M == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(planes && f0 && f1 && M > 0, "%s: null pointer or negative M", who);
    PCCX_CHECK_ARG(C0 >= 1 && C1 >= 1 && ld0 >= C0 && ld1 >= C1 && mod0 >= 0 && div1 >= 1, "%s: bad arguments", who);
    const long long ntiles = (M + 15) / 16;
    PCCX_CHECK_ARG((ntiles + 3) / 4 <= 0x7fffffffLL, "%s: M too large", who);
    hipLaunchKernelGGL(group_planes_kernel<P>, dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, (hipStream_t)stream, f0, C0, ld0, f1, C1, ld1,
                       (const int64_t *)nullptr, (long long)M, 1LL, 1LL, pg_kt32(C0 + C1), ntiles, (uint4 *)planes, (long long)mod0,
                       (long long)div1, rho, dyn);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}
extern "C" int pccx_fold_planes(const float *f0, int C0, int ld0, int64_t mod0, const float *f1, int C1, int ld1, int64_t div1, int64_t M,
                                float *planes, void *stream)
{
    return fold_planes_launch<3>(f0, C0, ld0, mod0, f1, C1, ld1, div1, M, planes, 1.f, nullptr, stream, "pccx_fold_planes");
}
extern "C" int pccx_fold_planes_h2(const float *f0, int C0, int ld0, int64_t mod0, const float *f1, int C1, int ld1, int64_t div1, int64_t M,
                                   float rho, const float *dyn, float *planes, void *stream)
{
    PCCX_CHECK_ARG(rho > 0.f, "pccx_fold_planes_h2: rho must be a positive power of two");
    return fold_planes_launch<2>(f0, C0, ld0, mod0, f1, C1, ld1, div1, M, planes, rho, dyn, stream, "pccx_fold_planes_h2");
}

// ---- the per-point part of FoldingNet's first layers straight into operand planes -------------------------------------------------
// act(base[r / div] + x[mod ? r % mod : r] @ w^T) for r < M (linear.hip: rows_affine_small_kernel -- the same fmaf chain per element, k
// ascending on top of base, so the values are bit-identical) written as the planes of the NEXT layer's operand instead of fp32 rows:
// round 3 wrote the rows (1.07 GB for the 512-wide MLP of 2048 patches), read them back in group_planes_kernel and wrote the planes.
// One wave per row tile, as group_planes_kernel; lane (g, n) forms channels 32 t + 16 h + 4 g .. + 3 of row 16 tile + n.
template <int P>
__global__ __launch_bounds__(256) void rows_affine_planes_kernel(const float *__restrict__ base, int C, unsigned div, const float *__restrict__ x,
                                                                 int ldx, int Ks, unsigned mod, const float *__restrict__ w, int relu,
                                                                 long long M, int KT32, long long ntiles, uint4 *__restrict__ planes,
                                                                 float rho, const float *__restrict__ dyn)
{
    typedef PgArith<P> AR;
    rho *= pg_dyn(dyn, 0);
    const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
    const long long tile = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    long long r = tile * 16 + n;
    if (r >= M) r = M - 1;                                   // padded rows repeat the last one (never read back as results)
    const float *xr = x + (size_t)(mod ? (unsigned long long)r % mod : (unsigned long long)r) * ldx;
    float xk[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) xk[k] = k < Ks ? xr[k] : 0.f;
    const float *br = base + (size_t)((unsigned long long)r / div) * C;
    for (int t = 0; t < KT32; ++t) {
        f32x4 v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = 32 * t + 16 * h + 4 * g;
            if (c + 3 < C) {
                const f32x4 b4 = *(const f32x4 *)(br + c);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float a = b4[u];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (k < Ks) a = fmaf(xk[k], w[(size_t)(c + u) * Ks + k], a);
                    v[h][u] = relu ? fmaxf(a, 0.f) : a;
                }
            } else {
                v[h] = f32x4{0.f, 0.f, 0.f, 0.f};             // C % 4 == 0: a block is inside the layer or beyond it
            }
        }
        typename AR::vec pl[P];
        AR::split(v[0], v[1], rho, pl);
        uint4 *d = planes + (((size_t)t * ntiles + tile) * P) * 64 + lane;
#pragma unroll
        for (int p = 0; p < P; ++p) d[p * 64] = __builtin_bit_cast(uint4, pl[p]);
    }
}

template <int P>
static int rows_affine_planes_launch(const float *base, int C, int64_t div, const float *x, int ldx, int Ks, int64_t mod, const float *w,
                                     int relu, int64_t M, float *planes, float rho, const float *dyn, void *stream, const char *who)
{
    if (M == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(base && x && w && planes, "%s: null pointer", who);
    PCCX_CHECK_ARG(M > 0 && C >= 4 && C % 4 == 0 && Ks >= 1 && Ks <= 4 && ldx >= Ks && div >= 1 && div < 0x7fffffffLL && mod >= 0 && mod < 0x7fffffffLL &&
                       ((uintptr_t)base & 15) == 0,
                   "%s: bad arguments (C=%d a multiple of 4, Ks=%d in 1..4, base 16-byte aligned)", who, C, Ks);
    const long long ntiles = (M + 15) / 16;
    PCCX_CHECK_ARG((ntiles + 3) / 4 <= 0x7fffffffLL, "%s: M too large", who);
    hipLaunchKernelGGL(rows_affine_planes_kernel<P>, dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, (hipStream_t)stream, base, C, (unsigned)div, x, ldx, Ks,
                       (unsigned)mod, w, relu, (long long)M, pg_kt32(C), ntiles, (uint4 *)planes, rho, dyn);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}
extern "C" int pccx_rows_affine_planes(const float *base, int C, int64_t div, const float *x, int ldx, int Ks, int64_t mod, const float *w,
                                       int relu, int64_t M, float *planes, void *stream)
{
    return rows_affine_planes_launch<3>(base, C, div, x, ldx, Ks, mod, w, relu, M, planes, 1.f, nullptr, stream, "pccx_rows_affine_planes");
}
extern "C" int pccx_rows_affine_planes_h2(const float *base, int C, int64_t div, const float *x, int ldx, int Ks, int64_t mod, const float *w,
                                          int relu, int64_t M, float rho, const float *dyn, float *planes, void *stream)
{
    PCCX_CHECK_ARG(rho > 0.f, "pccx_rows_affine_planes_h2: rho must be a positive power of two");
    return rows_affine_planes_launch<2>(base, C, div, x, ldx, Ks, mod, w, relu, M, planes, rho, dyn, stream, "pccx_rows_affine_planes_h2");
}

// ---- weight stream: [m-block][t][MB m-tiles][plane] fragments out of pccx_pack_linear_b3's (pccx_pack_linear_h2's) [t][MT][plane] ----
static size_t pg_weight_floats(int N, int K, int P)
{
    const int MT = ((N > 0 ? N : 1) + 15) / 16, MB = pg_mb(N), MBS = (MT + MB - 1) / MB;
    return (size_t)MBS * pg_kt32(K > 0 ? K : 1) * MB * P * 256;
}
extern "C" size_t pccx_planes_gemm_weight_floats(int N, int K) { return pg_weight_floats(N, K, 3); }
extern "C" size_t pccx_planes_gemm_weight_floats_h2(int N, int K) { return pg_weight_floats(N, K, 2); }

__global__ void planes_weight_kernel(const uint4 *__restrict__ wpl, int KT32, int MT, int MB, int MBS, uint4 *__restrict__ ws, int P)
{
    const size_t total = (size_t)MBS * KT32 * MB * P * 64;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(e & 63);
        size_t f = e >> 6;
        const int p = (int)(f % P); f /= P;
        const int m = (int)(f % MB); f /= MB;
        const int t = (int)(f % KT32);
        const int mb = (int)(f / KT32);
        const int mt = mb * MB + m;
        ws[e] = mt < MT ? wpl[(((size_t)t * MT + mt) * P + p) * 64 + lane] : make_uint4(0, 0, 0, 0);
    }
}

static int pack_planes_gemm(const float *wplanes_dev, int N, int K, float *wstream_dev, int P, void *stream, const char *who)
{
    PCCX_CHECK_ARG(wplanes_dev && wstream_dev && N >= 1 && K >= 1, "%s: bad argument", who);
    const int MT = (N + 15) / 16, MB = pg_mb(N), MBS = (MT + MB - 1) / MB, KT32 = pg_kt32(K);
    const size_t total = (size_t)MBS * KT32 * MB * P * 64;
    hipLaunchKernelGGL(planes_weight_kernel, dim3((unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, (const uint4 *)wplanes_dev, KT32, MT, MB, MBS, (uint4 *)wstream_dev, P);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}
extern "C" int pccx_pack_planes_gemm(const float *wplanes_dev, int N, int K, float *wstream_dev, void *stream)
{
    return pack_planes_gemm(wplanes_dev, N, K, wstream_dev, 3, stream, "pccx_pack_planes_gemm");
}
extern "C" int pccx_pack_planes_gemm_h2(const float *wplanes_dev, int N, int K, float *wstream_dev, void *stream)
{
    return pack_planes_gemm(wplanes_dev, N, K, wstream_dev, 2, stream, "pccx_pack_planes_gemm_h2");
}

// f16x2 weight planes [t][MT][2] of tau * W out of the packed f32 fragments (pccx_pack_linear / pccx_pack_linear_device): the fp16 pair
// (hi, lo) of every weight times tau, a power of two the caller picks with max|W| tau <= 2^14 (pack_h2.hip: the lo piece of all but
// negligible weights stays a normal fp16 number)
extern "C" size_t pccx_packed_linear_h2_floats(int N, int K)
{
    if (N < 1 || K < 1) return 0;
    return (size_t)pg_kt32(K) * (size_t)((N + 15) / 16) * 2 * 256;
}
__global__ __launch_bounds__(256) void linear_h2_pack_kernel(const f32x4 *__restrict__ wp, int KT16, int MT, float tau, uint4 *__restrict__ out)
{
    const int lane = threadIdx.x & 63, T = (KT16 + 1) / 2;
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= (long)T * MT) return;
    const int mt = (int)(item % MT), t = (int)(item / MT);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const f32x4 v0 = wp[((size_t)(2 * t) * MT + mt) * 64 + lane];
    const f32x4 v1 = 2 * t + 1 < KT16 ? wp[((size_t)(2 * t + 1) * MT + mt) * 64 + lane] : zero;   // an odd last k-tile pairs with zeros
    f16x8 pl[2];
    h2_split8(v0, v1, tau, pl);
#pragma unroll
    for (int p = 0; p < 2; ++p) out[(((size_t)t * MT + mt) * 2 + p) * 64 + lane] = __builtin_bit_cast(uint4, pl[p]);
}
extern "C" int pccx_pack_linear_h2(const float *wp_dev, int N, int K, float tau, float *out_dev, void *stream)
{
    PCCX_CHECK_ARG(wp_dev && out_dev && N >= 1 && K >= 1 && tau > 0.f, "pccx_pack_linear_h2: bad arguments");
    const int KT16 = (K + 15) / 16, MT = (N + 15) / 16;
    const long items = (long)((KT16 + 1) / 2) * MT;
    hipLaunchKernelGGL(linear_h2_pack_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const f32x4 *)wp_dev,
                       KT16, MT, tau, (uint4 *)out_dev);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// ---- one layer ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint4 pg_load_async(const uint4 *p)    // placed exactly here; completion rides on the ring's s_waitcnt
{
    uint4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// The end of a layer-0 k loop: wait for every load still in flight -- the B loads of the last two (clamped, unused) k-steps among them --
// with the three rotating register sets named as read-write operands of the wait.  The compiler does not know that an asm load's
// result arrives later: on paths where a set's value is dead (a layer with one or two k-steps never reads the third set) it would
// hand the registers to something else while the load is still in flight, and the data landing afterwards would overwrite that
// something (found by tools/asm_load_lint.py on the f16x2 chain of sa1, one k-step: the next layer's operand planes were built in
// those registers -- results changed from call to call).  Tied to the wait, the sets stay allocated until their loads have landed.
typedef unsigned int pg_u32x4 __attribute__((ext_vector_type(4)));
#define PG_R(x) __builtin_bit_cast(pg_u32x4, x)
#define PG_SETS2(bs)                                                                                                                          \
    "v"(PG_R(bs[0][0][0])), "v"(PG_R(bs[0][0][1])), "v"(PG_R(bs[0][1][0])), "v"(PG_R(bs[0][1][1])), "v"(PG_R(bs[1][0][0])),                     \
        "v"(PG_R(bs[1][0][1])), "v"(PG_R(bs[1][1][0])), "v"(PG_R(bs[1][1][1])), "v"(PG_R(bs[2][0][0])), "v"(PG_R(bs[2][0][1])),                 \
        "v"(PG_R(bs[2][1][0])), "v"(PG_R(bs[2][1][1]))
#define PG_SETS3(bs)                                                                                                                          \
    "v"(PG_R(bs[0][0][0])), "v"(PG_R(bs[0][0][1])), "v"(PG_R(bs[0][0][2])), "v"(PG_R(bs[0][1][0])), "v"(PG_R(bs[0][1][1])),                     \
        "v"(PG_R(bs[0][1][2])), "v"(PG_R(bs[1][0][0])), "v"(PG_R(bs[1][0][1])), "v"(PG_R(bs[1][0][2])), "v"(PG_R(bs[1][1][0])),                 \
        "v"(PG_R(bs[1][1][1])), "v"(PG_R(bs[1][1][2])), "v"(PG_R(bs[2][0][0])), "v"(PG_R(bs[2][0][1])), "v"(PG_R(bs[2][0][2])),                 \
        "v"(PG_R(bs[2][1][0])), "v"(PG_R(bs[2][1][1])), "v"(PG_R(bs[2][1][2]))
// inputs only (as native vectors: a HIP_vector_type is an aggregate the constraint cannot take): the statement READS the sets, so their
// values (as the compiler sees them) must still sit in their registers here
template <int NBV>
__device__ __forceinline__ void pg_drain_loads(const uint4 (&bs)[3][2][NBV])
{
    if constexpr (NBV == 2) asm volatile("s_waitcnt vmcnt(0)" ::PG_SETS2(bs) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::PG_SETS3(bs) : "memory");
}
// A counted wait that covers the register set about to be CONSUMED: the set goes through the statement as read-write operands, so no
// instruction that uses the loaded values can be scheduled above the wait (a "memory" clobber does not order register arithmetic: the
// compiler hoisted a conversion of gathered rows above a bare s_waitcnt once the register allocation shifted), and the registers stay
// allocated to the set until it.  Sets still in flight are untouched until their own wait (or the final drain) names them.
template <int N, int NBV>
__device__ __forceinline__ void pg_wait_set(uint4 (&b)[2][NBV])
{
    pg_u32x4 r00 = PG_R(b[0][0]), r01 = PG_R(b[0][1]), r10 = PG_R(b[1][0]), r11 = PG_R(b[1][1]);
    if constexpr (NBV == 2) {
        asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r00), "+v"(r01), "+v"(r10), "+v"(r11) : "n"(N) : "memory");
    } else {
        pg_u32x4 r02 = PG_R(b[0][2]), r12 = PG_R(b[1][2]);
        asm volatile("s_waitcnt vmcnt(%6)" : "+v"(r00), "+v"(r01), "+v"(r02), "+v"(r10), "+v"(r11), "+v"(r12) : "n"(N) : "memory");
        b[0][2] = __builtin_bit_cast(uint4, r02);
        b[1][2] = __builtin_bit_cast(uint4, r12);
    }
    b[0][0] = __builtin_bit_cast(uint4, r00);
    b[0][1] = __builtin_bit_cast(uint4, r01);
    b[1][0] = __builtin_bit_cast(uint4, r10);
    b[1][1] = __builtin_bit_cast(uint4, r11);
}
#undef PG_SETS2
#undef PG_SETS3
#undef PG_R

// The barrier of a ring boundary WITHOUT __syncthreads()' fences.  The release fence in __syncthreads() makes the compiler wait for
// EVERY outstanding vector-memory operation (s_waitcnt vmcnt(0)) in front of the s_barrier -- LDS-DMA fills are tracked by vmcnt and
// write LDS, so it cannot tell them from the plane loads -- which throws away the counted waits above it: the plane loads of k-step
// t + 2 and the ring fills of the next chunks, issued to stay in flight across the boundary, were all drained at every boundary
// (matrix pipe 0.43 busy in the 512 -> 1024 layer).  What the protocol needs is already explicit: each wave's counted wait covers its own
// pieces of chunk c (in-order completion), the barrier then says everyone's have landed and everyone has finished reading chunk c - 1
// (those reads feed MFMAs issued before the barrier; the compiler's own lgkmcnt wait for them precedes their use).
// BARE = false keeps __syncthreads(): the bf16x3 forms, the one-chunk forms and the chains gain nothing measurable from the bare barrier,
// and with the compiler's drain in place tools/asm_load_lint.py can check them.
template <bool BARE>
__device__ __forceinline__ void pg_ring_barrier()
{
    if constexpr (BARE) {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    } else {
        __syncthreads();
    }
}

enum { PG_EPI_PLANES = 0, PG_EPI_ROWS = 1, PG_EPI_MAX = 2 };

// P = 2 (f16x2): the accumulators are sigma_in tau (W y + b s); `bias` holds sigma_in tau b, rho_in = sigma_in scales gathered rows,
// scale_out = sigma_next / (sigma_in tau) for the planes epilogue, 1 / (sigma_in tau) for the row / max epilogues (times 1 / s there).
template <int P, int MB, int EPI, bool GATHER>
__global__ __launch_bounds__(256, 2) void planes_gemm_kernel(const uint4 *__restrict__ bin, const int64_t *__restrict__ idx,
                                                             long long rows_per_batch, long long n_src, int ldp, long long M,
                                                             long long ntiles, int KT32,
                                                             const float *__restrict__ wstream, int MBS, const float *__restrict__ bias,
                                                             int N, int relu, int group, float *__restrict__ out, int ldo, float rho_in,
                                                             float scale_out, const float *__restrict__ dyn, float *__restrict__ amax,
                                                             const unsigned char *__restrict__ member)
{
    typedef PgArith<P> AR;
    typedef typename AR::vec avec;
    // ring chunk: MQC m-tiles x P planes (1 KiB fragments).  The wide f16x2 form takes all eight m-tiles of a k-step in ONE chunk -- one
    // barrier per k-step instead of two: f16x2 issues half the MFMAs of bf16x3 between barriers -- through a three-deep ring of 16 KiB
    // chunks (48 KiB per workgroup: three workgroups per CU as before), reading the chunk's A fragments four m-tiles at a time (with all
    // sixteen in registers the kernel needs 188 VGPRs = two workgroups per CU, and was slower: tools/experiments/r5/README.md)
    constexpr int MQC = (P == 2 && MB == 8) ? 8 : 4;
    constexpr int PG_CHUNK = MQC * P;
    constexpr int NBUF = MQC == 8 ? 3 : PG_NB;                 // ring depth
    constexpr int DMA = PG_CHUNK / 4;                          // LDS-DMA loads per wave and chunk
    constexpr int HALVES = MB / MQC;                           // ring chunks per k-step
    const float dyn_s = P == 2 ? pg_dyn(dyn, 0) : 1.f;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, n = lane & 15;
    // Block order: XCD-aware.  Consecutive workgroup ids go round the 8 XCDs, so the MBS m-blocks of one 128-row block are
    // given to the SAME XCD one after the other: the row block's B planes are fetched into that XCD's L2 once, and every L2
    // holds the layer's whole weight stream (<= 3 MB).
    const long long nblk = (ntiles + 7) / 8;
    const long long s = blockIdx.x >> 3;
    const long long blk = (s / MBS) * 8 + (blockIdx.x & 7);
    const int mb = (int)(s % MBS);
    if (blk >= nblk) return;                                  // whole workgroup, before any barrier
    const long long tile0 = blk * 8 + 2 * w;
    __shared__ __attribute__((aligned(16))) f32x4 swt[NBUF * PG_CHUNK * 64];
    const int wu = __builtin_amdgcn_readfirstlane(w);
    const int nch = HALVES * KT32;
    const WStreamT<PG_CHUNK, NBUF> ws{wstream + (size_t)mb * nch * PG_CHUNK * 256, swt, nch, lane, wu, false};
    // DMA of chunk c (a chunk past the end re-reads chunk 0 into a free buffer, so every boundary issues the same loads and
    // the counted waits below hold to the last k-step)
    auto dma = [&](int c) { ws.issue(c < nch ? c : 0, c % NBUF); };
#pragma unroll
    for (int c = 0; c < NBUF - 1; ++c) dma(c);

    f32x4 acc[2][MB];
#pragma unroll
    for (int mt = 0; mt < MB; ++mt) {
        f32x4 b;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 16 * (mb * MB + mt) + 4 * g + r;
            b[r] = (bias && c < N) ? (P == 2 ? bias[c] * dyn_s : bias[c]) : 0.f;
        }
        acc[0][mt] = b; acc[1][mt] = b;
    }
    const long long t0 = tile0 < ntiles ? tile0 : ntiles - 1, t1 = tile0 + 1 < ntiles ? tile0 + 1 : ntiles - 1;
    {
        // VMEM issue order per wave and k-step t:  HALVES = 2:  boundary(2t): DMA(2t+3) [DMA], B(t+2) [2 P];  boundary(2t+1): DMA(2t+4) [DMA]
        //                                          HALVES = 1:  boundary(t):  DMA(t+NBUF-1) [DMA],  B(t+2) [2 P]
        // loads complete in order, so boundary(c) may leave in flight everything issued after the youngest load it needs.
        // GATHER: bin = fp32 source rows (n_src per batch, ldp = 32 * KT32 floats, zero padded); row r reads source row
        // (r / rows_per_batch) * n_src + max(idx[r], 0) and is split in registers at use (4 loads per k-step instead of 6).
        constexpr int NBL = GATHER ? 4 : 2 * P, NBV = GATHER ? 2 : P;
        const float rho_g = rho_in * dyn_s;
        uint4 bs[3][2][NBV];
        const uint4 *gsrc[2] = {nullptr, nullptr};
        if constexpr (GATHER) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                long long r = (nt ? t1 : t0) * 16 + n;
                if (r >= M) r = M - 1;
                const long long j = idx[r];
                gsrc[nt] = (const uint4 *)((const float *)bin + (size_t)((r / rows_per_batch) * n_src + (j < 0 ? 0 : j)) * ldp + 4 * g);
            }
        }
        auto load_b = [&](uint4 (&dst)[2][NBV], int t) {
            const int tc = t < KT32 ? t : KT32 - 1;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int pl = 0; pl < NBV; ++pl)
                    dst[nt][pl] = GATHER ? pg_load_async(gsrc[nt] + 8 * tc + 4 * pl)
                                         : pg_load_async(bin + (((size_t)tc * ntiles + (nt ? t1 : t0)) * P + pl) * 64 + lane);
        };
        auto kstep = [&](int t, uint4 (&braw)[2][NBV], uint4 (&bload)[2][NBV], bool first) {
            avec bc[2][P];
#pragma unroll
            for (int half = 0; half < HALVES; ++half) {
                const int c = HALVES * t + half;
                if (half == 0) {
                    if (first) pg_wait_set<0, NBV>(braw);
                    else if (HALVES == 2) pg_wait_set<2 * DMA + NBL, NBV>(braw);
                    else pg_wait_set<DMA + NBL, NBV>(braw);
                    pg_ring_barrier<P == 2 && MB == 8>();
                    dma(c + NBUF - 1);
                    load_b(bload, t + 2);
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        if constexpr (GATHER)
                            AR::split(__builtin_bit_cast(f32x4, braw[nt][0]), __builtin_bit_cast(f32x4, braw[nt][1]), rho_g, bc[nt]);
                        else
#pragma unroll
                            for (int pl = 0; pl < P; ++pl) bc[nt][pl] = __builtin_bit_cast(avec, braw[nt][pl < NBV ? pl : 0]);
                    }
                } else {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DMA + 2 * NBL) : "memory");
                    pg_ring_barrier<P == 2 && MB == 8>();
                    dma(c + NBUF - 1);
                }
                const f32x4 *buf = ws.chunk(c);
#pragma unroll
                for (int sub = 0; sub < MQC / 4; ++sub) {
                    avec a[4][P];
#pragma unroll
                    for (int mq = 0; mq < 4; ++mq)
#pragma unroll
                        for (int pl = 0; pl < P; ++pl) a[mq][pl] = __builtin_bit_cast(avec, buf[((4 * sub + mq) * P + pl) * 64]);
                    __builtin_amdgcn_sched_barrier(0);
                    // the products, smallest first (bf16x3: (lo,hi) (hi,lo) (mid,mid) (mid,hi) (hi,mid) (hi,hi); f16x2: (lo,hi) (hi,lo) (hi,hi))
#pragma unroll
                    for (int q = 0; q < AR::NQ; ++q)
#pragma unroll
                        for (int mq = 0; mq < 4; ++mq)
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt)
                                acc[nt][MQC * half + 4 * sub + mq] = AR::mfma(a[mq][AR::pa(q)], bc[nt][AR::pb(q)], acc[nt][MQC * half + 4 * sub + mq]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        load_b(bs[0], 0);
        load_b(bs[1], 1);
        kstep(0, bs[0], bs[2], true);                     // waits for everything issued so far
        if (KT32 > 1) kstep(1, bs[1], bs[0], false);
#pragma unroll 1
        for (int t = 2; t < KT32; t += 3) {               // three k-steps per trip: static register sets
            kstep(t, bs[2], bs[1], false);
            if (t + 1 < KT32) kstep(t + 1, bs[0], bs[2], false);
            if (t + 2 < KT32) kstep(t + 2, bs[1], bs[0], false);
        }
        pg_drain_loads<NBV>(bs);                                          // the last (clamped, unused) B loads and DMAs
    }

    if constexpr (EPI == PG_EPI_PLANES) {
        // next layer's operand: k-tile j of this m-block = C tiles 2j, 2j+1
        uint4 *o = (uint4 *)out;
        const int KTo = ((N + 15) / 16 + 1) / 2;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            if (tile0 + nt >= ntiles) continue;
#pragma unroll
            for (int j = 0; j < MB / 2; ++j) {
                const int to = mb * (MB / 2) + j;
                if (to >= KTo) continue;
                avec pl[P];
                if (relu) AR::split(relu4(acc[nt][2 * j]), relu4(acc[nt][2 * j + 1]), scale_out, pl);
                else AR::split(acc[nt][2 * j], acc[nt][2 * j + 1], scale_out, pl);
                uint4 *d = o + (((size_t)to * ntiles + tile0 + nt) * P) * 64 + lane;
#pragma unroll
                for (int p = 0; p < P; ++p) d[p * 64] = __builtin_bit_cast(uint4, pl[p]);
            }
        }
    } else if constexpr (EPI == PG_EPI_ROWS) {
        const float unscale = P == 2 ? scale_out * pg_dyn(dyn, 1) : 1.f;
        float vmax = 0.f;                                             // f16x2: the largest |value| written (the next stack's input bound)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const long long row = (tile0 + nt) * 16 + n;
            if (tile0 + nt >= ntiles || row >= M) continue;
#pragma unroll
            for (int mt = 0; mt < MB; ++mt) {
                const int c = 16 * (mb * MB + mt) + 4 * g;
                f32x4 v = relu ? relu4(acc[nt][mt]) : acc[nt][mt];
                if constexpr (P == 2) {
                    v = v * unscale;                                  // powers of two: exact
#pragma unroll
                    for (int r = 0; r < 4; ++r) vmax = fmaxf(vmax, fabsf(v[r]));      // channels past N are exact zeros
                }
                float *po = out + (size_t)row * ldo + c;
                if (c + 3 < N && ldo % 4 == 0) {
                    *(f32x4 *)po = v;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (c + r < N) po[r] = v[r];
                }
            }
        }
        if constexpr (P == 2) {
            if (amax) pg_amax_commit(amax, vmax);                     // whole waves reach this (no divergent exit above)
        }
    } else {
        // max over groups of `group` rows (32, 64 or 128; M is a multiple of it, so no group holds padded rows).  In the wave:
        // the two tiles elementwise, then the 16 rows of the tile by DPP; across the waves of a group through LDS.
        __syncthreads();                                   // every wave is done with the ring
        float *smax = (float *)swt;                        // [4 waves][16 * MB channels]
        if (member) {                                      // the maximum over the MEMBER rows of each group only (pccx_group_members)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const long long row = (tile0 + nt) * 16 + n;
                if (tile0 + nt < ntiles && row < M && member[row]) continue;
#pragma unroll
                for (int mt = 0; mt < MB; ++mt) acc[nt][mt] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            }
        }
#pragma unroll
        for (int mt = 0; mt < MB; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = fmaxf(acc[0][mt][r], acc[1][mt][r]);
                v = row16_max(v);
                if (n == 0) smax[w * (16 * MB) + 16 * mt + 4 * g + r] = v;
            }
        __syncthreads();
        const int gpb = 128 / group, wpg = group / 32;     // groups per block, waves per group
        const long long G = M / group;
        for (int e = tid; e < gpb * 16 * MB; e += 256) {
            const int gi = e / (16 * MB), c = e % (16 * MB);
            float v = smax[(gi * wpg) * (16 * MB) + c];
            for (int q = 1; q < wpg; ++q) v = fmaxf(v, smax[(gi * wpg + q) * (16 * MB) + c]);
            if (relu) v = fmaxf(v, 0.f);                   // max(relu(x)) = relu(max(x))
            if constexpr (P == 2) v *= scale_out * pg_dyn(dyn, 1);       // a positive power of two commutes with max and relu
            const long long grp = blk * gpb + gi;
            const int ch = mb * 16 * MB + c;
            if (grp < G && ch < N) out[(size_t)grp * ldo + ch] = v;
        }
    }
}

// out: epilogue 0 -> planes of the N output channels (pccx_planes_floats(M, N) floats); 1 -> fp32 rows (M, ldo);
// 2 -> fp32 (M / group, ldo), the max over each `group` consecutive rows (group in {32, 64, 128}, M % group == 0).
template <int P>
static int planes_gemm_launch(const float *x, const int64_t *idx, int64_t rows_per_batch, int64_t n_src, int ldp, int64_t M, int K,
                              const float *wstream, const float *bias, int N, int relu, int epilogue, int group, float *out, int ldo,
                              void *stream, const char *who, float rho_in = 1.f, float scale_out = 1.f, const float *dyn = nullptr,
                              float *amax = nullptr, const unsigned char *member = nullptr)
{
    if (M == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(x && wstream && out, "%s: null pointer", who);
    PCCX_CHECK_ARG(M > 0 && K >= 1 && N >= 1, "%s: bad shape M=%lld K=%d N=%d", who, (long long)M, K, N);
    PCCX_CHECK_ARG(epilogue >= 0 && epilogue <= 2, "%s: epilogue %d", who, epilogue);
    PCCX_CHECK_ARG(epilogue == PG_EPI_PLANES || ldo >= N, "%s: ldo=%d < N=%d", who, ldo, N);
    PCCX_CHECK_ARG(epilogue != PG_EPI_MAX || ((group == 32 || group == 64 || group == 128) && M % group == 0),
                   "%s: group max needs group in {32,64,128} dividing M (group=%d M=%lld)", who, group, (long long)M);
    const long long ntiles = (M + 15) / 16, nblk = (ntiles + 7) / 8;
    const int MT = (N + 15) / 16, MB = pg_mb(N), MBS = (MT + MB - 1) / MB, KT32 = pg_kt32(K);
    const long long blocks = (nblk + 7) / 8 * 8 * MBS;
    PCCX_CHECK_ARG(blocks <= 0x7fffffffLL, "%s: M=%lld too large", who, (long long)M);
    PCCX_CHECK_ARG(!idx || (rows_per_batch >= 1 && n_src >= 1 && ldp == 32 * KT32 && (uintptr_t)x % 16 == 0),
                   "%s: source rows must be 16-byte aligned with a stride of %d floats (got %d)", who, 32 * KT32, ldp);
    hipStream_t st = (hipStream_t)stream;
    relu &= 1;
#define PG_LAUNCH(MB_, E_, G_)                                                                                                  \
    hipLaunchKernelGGL((planes_gemm_kernel<P, MB_, E_, G_>), dim3((unsigned)blocks), dim3(256), 0, st, (const uint4 *)x, idx,    \
                       (long long)rows_per_batch, (long long)n_src, ldp, (long long)M, ntiles, KT32, wstream, MBS, bias, N, relu, \
                       group, out, ldo, rho_in, scale_out, dyn, amax, member)
#define PG_LAUNCH_E(MB_, G_)                                                                                                    \
    do {                                                                                                                        \
        if (epilogue == PG_EPI_PLANES) PG_LAUNCH(MB_, PG_EPI_PLANES, G_);                                                       \
        else if (epilogue == PG_EPI_ROWS) PG_LAUNCH(MB_, PG_EPI_ROWS, G_);                                                      \
        else PG_LAUNCH(MB_, PG_EPI_MAX, G_);                                                                                    \
    } while (0)
    if (MB == 8) { if (idx) PG_LAUNCH_E(8, true); else PG_LAUNCH_E(8, false); }
    else { if (idx) PG_LAUNCH_E(4, true); else PG_LAUNCH_E(4, false); }
#undef PG_LAUNCH_E
#undef PG_LAUNCH
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

extern "C" int pccx_planes_gemm(const float *planes_in, int64_t M, int K, const float *wstream, const float *bias, int N, int relu,
                                int epilogue, int group, float *out, int ldo, void *stream)
{
    return planes_gemm_launch<3>(planes_in, nullptr, 1, 1, 0, M, K, wstream, bias, N, relu, epilogue, group, out, ldo, stream,
                                 "pccx_planes_gemm");
}
// The f16x2 form (two fp16 planes per operand, three products per fp32 product: mfma_chain.h): planes_in = pccx_planes_floats_h2 planes of
// sigma_in * input, wstream = pccx_pack_planes_gemm_h2 of the planes of tau * W, bias = sigma_in tau b (the kernel multiplies it by
// dyn[0]).  scale_out: epilogue 0 -> sigma_next / (sigma_in tau), the planes written are those of sigma_next * output; epilogues 1, 2 ->
// 1 / (sigma_in tau), the rows are multiplied by it and by dyn[1].  Every scale is a power of two (exact).
extern "C" int pccx_planes_gemm_h2(const float *planes_in, int64_t M, int K, const float *wstream, const float *bias, int N, int relu,
                                   int epilogue, int group, float scale_out, const float *dyn, float *amax8, float *out, int ldo, void *stream)
{
    PCCX_CHECK_ARG(scale_out > 0.f, "pccx_planes_gemm_h2: scale_out must be a positive power of two");
    PCCX_CHECK_ARG(!amax8 || epilogue == PG_EPI_ROWS, "pccx_planes_gemm_h2: the |value| maximum is collected by the row epilogue only");
    return planes_gemm_launch<2>(planes_in, nullptr, 1, 1, 0, M, K, wstream, bias, N, relu, epilogue, group, out, ldo, stream,
                                 "pccx_planes_gemm_h2", 1.f, scale_out, dyn, amax8);
}

// The last layer of a set-abstraction stack whose groups are only ever reduced TOGETHER (PPPF_AE's third level: PPPF_AE.py:44 takes the
// maximum over the 32 centroids of the maxima over their 128 samples, pointnet_sa_module.py:91 -- and the stack acts on each un-centred
// source row by itself, so that double maximum is the maximum over the rows that are a sample of ANY centroid).  member = one byte per
// row (pccx_group_members), group = the source rows of one batch element (32, 64 or 128): out (M / group, ldo) = the maximum over the
// member rows of each group.  The layer's fp32 rows (1 GB per 2048 patches for 512 -> 1024) are never written, nor gathered per centroid.
// A group without a member gives 0 with relu, -inf without.
extern "C" int pccx_planes_gemm_h2_member_max(const float *planes_in, int64_t M, int K, const float *wstream, const float *bias, int N, int relu,
                                              int group, const unsigned char *member, float scale_out, const float *dyn, float *out, int ldo,
                                              void *stream)
{
    PCCX_CHECK_ARG(scale_out > 0.f, "pccx_planes_gemm_h2_member_max: scale_out must be a positive power of two");
    PCCX_CHECK_ARG(member || M == 0, "pccx_planes_gemm_h2_member_max: null member table");
    return planes_gemm_launch<2>(planes_in, nullptr, 1, 1, 0, M, K, wstream, bias, N, relu, PG_EPI_MAX, group, out, ldo, stream,
                                 "pccx_planes_gemm_h2_member_max", 1.f, scale_out, dyn, nullptr, member);
}

// member[(e / per_batch) * n_src + max(idx[e], 0)] = 1 for every entry e of idx (n_idx entries, per_batch = centroids * nsample per
// batch element; -1 = the ball query's padding, which the reference's gather clamps to row 0: pointnet_sa_module.py:27); all other
// bytes of member (n_batches * n_src) are cleared.
__global__ __launch_bounds__(256) void group_members_kernel(const int64_t *__restrict__ idx, long long n_idx, long long per_batch, long long n_src,
                                                            unsigned char *__restrict__ member)
{
    for (long long e = blockIdx.x * 256ll + threadIdx.x; e < n_idx; e += gridDim.x * 256ll) {
        long long j = idx[e];
        j = j < 0 ? 0 : (j < n_src ? j : n_src - 1);
        member[(e / per_batch) * n_src + j] = 1;
    }
}

extern "C" int pccx_group_members(const int64_t *idx, int64_t n_idx, int64_t per_batch, int64_t n_src, unsigned char *member, void *stream)
{
    if (n_idx == 0) return PCCX_OK;
    PCCX_CHECK_ARG(idx && member, "pccx_group_members: null pointer");
    PCCX_CHECK_ARG(n_idx > 0 && per_batch >= 1 && n_src >= 1 && n_idx % per_batch == 0, "pccx_group_members: bad shape n_idx=%lld per_batch=%lld n_src=%lld",
                   (long long)n_idx, (long long)per_batch, (long long)n_src);
    PCCX_CHECK_ARG((n_idx / per_batch * n_src) % 4 == 0 && (uintptr_t)member % 4 == 0,
                   "pccx_group_members: the member table (%lld bytes) is cleared in 4-byte words: size and address must be multiples of 4",
                   (long long)(n_idx / per_batch * n_src));
    hipStream_t st = (hipStream_t)stream;
    PCCX_CHECK_HIP(pccx_zero_async(member, (size_t)(n_idx / per_batch * n_src), st));       // a kernel, never a memset node (common.h)
    const long long blocks = (n_idx + 255) / 256;
    hipLaunchKernelGGL(group_members_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, idx, (long long)n_idx,
                       (long long)per_batch, (long long)n_src, member);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// The same layer with its input gathered in the kernel (see pccx_planes_chain4_gather): src = fp32 rows of ldp = 32 * ceil(K / 32)
// floats, row r of the layer's input = source row (r / rows_per_batch) * n_src + max(idx[r], 0).
extern "C" int pccx_planes_gemm_gather(const float *src, int ldp, const int64_t *idx, int64_t rows_per_batch, int64_t n_src, int64_t M,
                                       int K, const float *wstream, const float *bias, int N, int relu, int epilogue, int group,
                                       float *out, int ldo, void *stream)
{
    PCCX_CHECK_ARG(idx || M == 0, "pccx_planes_gemm_gather: null indices");
    return planes_gemm_launch<3>(src, idx, rows_per_batch, n_src, ldp, M, K, wstream, bias, N, relu, epilogue, group, out, ldo, stream,
                                 "pccx_planes_gemm_gather");
}
// f16x2: the gathered fp32 rows are multiplied by rho_in * dyn[0] (= sigma_in s) as they are split in registers
extern "C" int pccx_planes_gemm_gather_h2(const float *src, int ldp, const int64_t *idx, int64_t rows_per_batch, int64_t n_src, int64_t M,
                                          int K, const float *wstream, const float *bias, int N, int relu, int epilogue, int group,
                                          float rho_in, float scale_out, const float *dyn, float *amax8, float *out, int ldo, void *stream)
{
    PCCX_CHECK_ARG(idx || M == 0, "pccx_planes_gemm_gather_h2: null indices");
    PCCX_CHECK_ARG(rho_in > 0.f && scale_out > 0.f, "pccx_planes_gemm_gather_h2: scales must be positive powers of two");
    PCCX_CHECK_ARG(!amax8 || epilogue == PG_EPI_ROWS, "pccx_planes_gemm_gather_h2: the |value| maximum is collected by the row epilogue only");
    return planes_gemm_launch<2>(src, idx, rows_per_batch, n_src, ldp, M, K, wstream, bias, N, relu, epilogue, group, out, ldo, stream,
                                 "pccx_planes_gemm_gather_h2", rho_in, scale_out, dyn, amax8);
}

// ---- four-layer stack in one kernel ------------------------------------------------------------------------------
// Conv-BN-ReLU x 4 + max over nsample (pointnet_sa_module.py:90-91) for stacks whose first three widths are <= 128: layer 0 is the
// GEMM above on the planes of the gathered input; its accumulators (128 rows x <= 128 channels per workgroup, 2 row tiles x <= 8
// m-tiles per wave) are split in registers into the next layer's B planes (the chain of mfma_chain.h), and so on; the last layer
// runs in passes of 128 output channels, each reduced over the groups of `group` rows as in the epilogue above.  One weight stream
// for the whole stack (the four layers' pccx_pack_planes_gemm streams back to back) goes through the LDS-DMA ring; no activation
// of the stack exists in HBM.
//   MQ0 / MQ1 / MQ2 : output m-quads (4 m-tiles) of layers 0..2 (1 or 2);  KT1..KT3 : K/32 blocks of the inputs of layers 1..3;
//   NP : passes (of 8 m-tiles) of the last layer.
template <int P, int KT, int MQ, class WS>
__device__ __forceinline__ void pg_chain_layer(const WS &ws, int &c, int nch, const typename PgArith<P>::vec (&in)[2][KT][P], f32x4 (&acc)[2][4 * MQ])
{
    typedef PgArith<P> AR;
    typedef typename AR::vec avec;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int mqq = 0; mqq < MQ; ++mqq) {
            // only DMAs are in flight here: chunk c's was issued three boundaries ago, two chunks (2 P loads) may stay in flight
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * P) : "memory");
            pg_ring_barrier<false>();
            {
                const int nx = c + PG_NB - 1;
                ws.issue(nx < nch ? nx : 0, nx % PG_NB);
            }
            const f32x4 *buf = ws.chunk(c);
            avec a[4][P];
#pragma unroll
            for (int mq = 0; mq < 4; ++mq)
#pragma unroll
                for (int pl = 0; pl < P; ++pl) a[mq][pl] = __builtin_bit_cast(avec, buf[(mq * P + pl) * 64]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < AR::NQ; ++q)
#pragma unroll
                for (int mq = 0; mq < 4; ++mq)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[nt][4 * mqq + mq] = AR::mfma(a[mq][AR::pa(q)], in[nt][kt][AR::pb(q)], acc[nt][4 * mqq + mq]);
            __builtin_amdgcn_sched_barrier(0);
            ++c;
        }
}

template <int NTILES>
__device__ __forceinline__ void pg_bias_init(f32x4 (&acc)[2][NTILES], const float *__restrict__ bias, int N, int m0, int g, float mul = 1.f, bool scaled = false)
{
#pragma unroll
    for (int mt = 0; mt < NTILES; ++mt) {
        f32x4 b;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ch = 16 * (m0 + mt) + 4 * g + r;
            b[r] = (bias && ch < N) ? (scaled ? bias[ch] * mul : bias[ch]) : 0.f;
        }
        acc[0][mt] = b; acc[1][mt] = b;
    }
}

// relu + split of a layer's accumulators into the next layer's planes (k-tile j = C tiles 2j, 2j+1; a missing odd tile is zero)
template <int P, int NTILES, int KT>
__device__ __forceinline__ void pg_to_planes(const f32x4 (&acc)[2][NTILES], typename PgArith<P>::vec (&pl)[2][KT][P], float rho)
{
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int j = 0; j < KT; ++j)
            PgArith<P>::split(relu4(acc[nt][2 * j]), 2 * j + 1 < NTILES ? relu4(acc[nt][2 * j + 1]) : zero, rho, pl[nt][j]);
}

// P = 2 (f16x2): b0..b3 hold sigma_l tau_l b_l (multiplied by dyn[0] here), sc = {rho_in, rho_1, rho_2, rho_3, 1 / (sigma_3 tau_3)}:
// rho_in scales the gathered rows, rho_l = sigma_l / (sigma_{l-1} tau_{l-1}) the split in front of layer l, the last the output rows.
struct PgChainScales { float rho_in, rho1, rho2, rho3, inv3; };
template <int P, int MQ0, int KT1, int MQ1, int KT2, int MQ2, int KT3, int NP, bool GATHER>
__global__ __launch_bounds__(256, 2) void planes_chain4_kernel(const uint4 *__restrict__ bin, const int64_t *__restrict__ idx,
                                                               long long rows_per_batch, long long n_src, int ldp, long long M,
                                                               long long ntiles, int KT0,
                                                               const float *__restrict__ wstream, const float *__restrict__ b0, int N0,
                                                               const float *__restrict__ b1, int N1, const float *__restrict__ b2, int N2,
                                                               const float *__restrict__ b3, int N3, int group, float *__restrict__ out,
                                                               int ldo, PgChainScales sc, const float *__restrict__ dyn, float *__restrict__ amax)
{
    typedef PgArith<P> AR;
    typedef typename AR::vec avec;
    constexpr int PG_CHUNK = 4 * P;
    constexpr bool H2 = P == 2;
    const float dyn_s = H2 ? pg_dyn(dyn, 0) : 1.f;
    static_assert(2 * KT1 <= 4 * MQ0 + 1 && 2 * KT2 <= 4 * MQ1 + 1 && 2 * KT3 <= 4 * MQ2 + 1, "a layer's K blocks come from the previous layer's tiles");
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, n = lane & 15;
    const long long blk = blockIdx.x;
    const long long tile0 = blk * 8 + 2 * w;
    __shared__ __attribute__((aligned(16))) f32x4 swt[PG_NB * PG_CHUNK * 64];
    __shared__ float smax[4 * 128];
    const int wu = __builtin_amdgcn_readfirstlane(w);
    const int nch = MQ0 * KT0 + KT1 * MQ1 + KT2 * MQ2 + NP * KT3 * 2;
    const WStreamT<PG_CHUNK, PG_NB> ws{wstream, swt, nch, lane, wu, false};
    auto dma = [&](int c) { ws.issue(c < nch ? c : 0, c % PG_NB); };
#pragma unroll
    for (int c = 0; c < PG_NB - 1; ++c) dma(c);

    const long long t0 = tile0 < ntiles ? tile0 : ntiles - 1, t1 = tile0 + 1 < ntiles ? tile0 + 1 : ntiles - 1;
    // ---- layer 0: as planes_gemm_kernel (HALVES = MQ0).  GATHER: the B operand is not read as planes but gathered here --
    // bin = fp32 source rows (n_src per batch, ldp = 32 * KT0 floats each, zero padded), row r reads source row
    // (r / rows_per_batch) * n_src + max(idx[r], 0) (pointnet_sa_module.py:27,73-83) -- and split in registers at use.
    f32x4 acc0[2][4 * MQ0];
    pg_bias_init<4 * MQ0>(acc0, b0, N0, 0, g, dyn_s, H2);
    {
        constexpr int NBL = GATHER ? 4 : 2 * P;              // B loads per k-step
        constexpr int NBV = GATHER ? 2 : P;
        const float rho_g = sc.rho_in * dyn_s;
        uint4 bs[3][2][NBV];
        const uint4 *gsrc[2] = {nullptr, nullptr};
        if constexpr (GATHER) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                long long r = (nt ? t1 : t0) * 16 + n;
                if (r >= M) r = M - 1;
                const long long j = idx[r];
                gsrc[nt] = (const uint4 *)((const float *)bin + (size_t)((r / rows_per_batch) * n_src + (j < 0 ? 0 : j)) * ldp + 4 * g);
            }
        }
        auto load_b = [&](uint4 (&dst)[2][NBV], int t) {
            const int tc = t < KT0 ? t : KT0 - 1;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int pl = 0; pl < NBV; ++pl)
                    dst[nt][pl] = GATHER ? pg_load_async(gsrc[nt] + 8 * tc + 4 * pl)
                                         : pg_load_async(bin + (((size_t)tc * ntiles + (nt ? t1 : t0)) * P + pl) * 64 + lane);
        };
        auto kstep = [&](int t, uint4 (&braw)[2][NBV], uint4 (&bload)[2][NBV], bool first) {
            avec bc[2][P];
#pragma unroll
            for (int half = 0; half < MQ0; ++half) {
                const int c = MQ0 * t + half;
                if (half == 0) {
                    if (first) pg_wait_set<0, NBV>(braw);
                    else if (MQ0 == 2) pg_wait_set<2 * P + NBL, NBV>(braw);
                    else pg_wait_set<P + NBL, NBV>(braw);
                    pg_ring_barrier<false>();
                    dma(c + PG_NB - 1);
                    load_b(bload, t + 2);
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        if constexpr (GATHER)
                            AR::split(__builtin_bit_cast(f32x4, braw[nt][0]), __builtin_bit_cast(f32x4, braw[nt][1]), rho_g, bc[nt]);
                        else
#pragma unroll
                            for (int pl = 0; pl < P; ++pl) bc[nt][pl] = __builtin_bit_cast(avec, braw[nt][pl < NBV ? pl : 0]);
                    }
                } else {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * P + 2 * NBL) : "memory");
                    pg_ring_barrier<false>();
                    dma(c + PG_NB - 1);
                }
                const f32x4 *buf = ws.chunk(c);
                avec a[4][P];
#pragma unroll
                for (int mq = 0; mq < 4; ++mq)
#pragma unroll
                    for (int pl = 0; pl < P; ++pl) a[mq][pl] = __builtin_bit_cast(avec, buf[(mq * P + pl) * 64]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < AR::NQ; ++q)
#pragma unroll
                    for (int mq = 0; mq < 4; ++mq)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
                            acc0[nt][4 * half + mq] = AR::mfma(a[mq][AR::pa(q)], bc[nt][AR::pb(q)], acc0[nt][4 * half + mq]);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        load_b(bs[0], 0);
        load_b(bs[1], 1);
        kstep(0, bs[0], bs[2], true);
        if (KT0 > 1) kstep(1, bs[1], bs[0], false);
#pragma unroll 1
        for (int t = 2; t < KT0; t += 3) {
            kstep(t, bs[2], bs[1], false);
            if (t + 1 < KT0) kstep(t + 1, bs[0], bs[2], false);
            if (t + 2 < KT0) kstep(t + 2, bs[1], bs[0], false);
        }
        pg_drain_loads<NBV>(bs);                                          // the last (clamped, unused) B loads land before their
    }                                                                     // registers are reused; the ring's DMAs with them
    int c = MQ0 * KT0;
    // ---- layers 1, 2: registers to registers
    avec i1[2][KT1][P];
    pg_to_planes<P, 4 * MQ0, KT1>(acc0, i1, sc.rho1);
    f32x4 acc1[2][4 * MQ1];
    pg_bias_init<4 * MQ1>(acc1, b1, N1, 0, g, dyn_s, H2);
    pg_chain_layer<P, KT1, MQ1>(ws, c, nch, i1, acc1);
    avec i2[2][KT2][P];
    pg_to_planes<P, 4 * MQ1, KT2>(acc1, i2, sc.rho2);
    f32x4 acc2[2][4 * MQ2];
    pg_bias_init<4 * MQ2>(acc2, b2, N2, 0, g, dyn_s, H2);
    pg_chain_layer<P, KT2, MQ2>(ws, c, nch, i2, acc2);
    avec i3[2][KT3][P];
    pg_to_planes<P, 4 * MQ2, KT3>(acc2, i3, sc.rho3);
    const float unscale = H2 ? sc.inv3 * pg_dyn(dyn, 1) : 1.f;
    float vmax = 0.f;                                         // f16x2, group == 1: the largest value written (the next stack's input bound)
    // ---- last layer in passes of 8 m-tiles, each reduced over the row groups (group == 1: no reduction, the rows themselves)
    const int gpb = 128 / group, wpg = group / 32;
    const long long G = M / group;
#pragma unroll 1
    for (int ps = 0; ps < NP; ++ps) {
        f32x4 acc3[2][8];
        pg_bias_init<8>(acc3, b3, N3, 8 * ps, g, dyn_s, H2);
        pg_chain_layer<P, KT3, 2>(ws, c, nch, i3, acc3);
        if (group == 1) {
            // relu(L3(...)) as fp32 rows (M, ldo): lane (g, n) holds channels 128 ps + 16 mt + 4 g .. + 3 of row 16 (tile0 + nt) + n
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const long long row = (tile0 + nt) * 16 + n;
                if (tile0 + nt < ntiles && row < M) {
#pragma unroll
                    for (int mt = 0; mt < 8; ++mt) {
                        const int co = 128 * ps + 16 * mt + 4 * g;
                        f32x4 v = relu4(acc3[nt][mt]);
                        if constexpr (H2) {
                            v = v * unscale;
#pragma unroll
                            for (int r = 0; r < 4; ++r) vmax = fmaxf(vmax, v[r]);
                        }
                        if (co + 3 < N3 && (ldo & 3) == 0) {
                            *(f32x4 *)(out + (size_t)row * ldo + co) = v;
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (co + r < N3) out[(size_t)row * ldo + co + r] = v[r];
                        }
                    }
                }
            }
            continue;
        }
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = fmaxf(acc3[0][mt][r], acc3[1][mt][r]);
                v = row16_max(v);
                if (n == 0) smax[w * 128 + 16 * mt + 4 * g + r] = v;
            }
        __syncthreads();
        for (int e = tid; e < gpb * 128; e += 256) {
            const int gi = e >> 7, ch = e & 127;
            float v = smax[(gi * wpg) * 128 + ch];
            for (int q = 1; q < wpg; ++q) v = fmaxf(v, smax[(gi * wpg + q) * 128 + ch]);
            v = fmaxf(v, 0.f);                               // max(relu(x)) = relu(max(x))
            if constexpr (H2) v *= unscale;
            const long long grp = blk * gpb + gi;
            const int co = 128 * ps + ch;
            if (grp < G && co < N3) out[(size_t)grp * ldo + co] = v;
        }
        // the next pass writes smax only after its first ring boundary (a barrier every thread reaches after these reads)
    }
    ws.drain();
    if constexpr (H2) {
        if (amax && group == 1) pg_amax_commit(amax, vmax);
    }
}

// out (M / group, ldo) = max over each `group` consecutive rows of relu(L3(relu(L2(relu(L1(relu(L0(x)))))))).
// wstream: the four layers' pccx_pack_planes_gemm streams back to back.  Supported stacks (PCCX_ERR_ARG otherwise; callers fall
// back to pccx_planes_gemm layer by layer): widths (N0..N3) in the two shapes of PPPF_AE.py:29-34, (<=32, 33..64, 33..64, 65..128)
// and (97..128 x3, 129..256), each layer's input being the previous layer's output.
//   pccx_planes_chain4        : x given as planes (pccx_group_planes / a previous layer);
//   pccx_planes_chain4_gather : x gathered in the kernel from fp32 rows src (n_src rows per batch of ldp = 32 * ceil(K0 / 32) floats,
//                               the K0 channels zero padded) by idx (M entries, -1 -> row 0): the grouped tensor of
//                               pointnet_sa_module.py:73-83 never exists in memory in any form.
template <int P>
static int planes_chain4_launch(const float *x, const int64_t *idx, int64_t rows_per_batch, int64_t n_src, int ldp, int64_t M, int K0,
                                const float *wstream, const float *b0, int N0, const float *b1, int N1, const float *b2, int N2,
                                const float *b3, int N3, int group, float *out, int ldo, void *stream, const char *who,
                                PgChainScales sc = PgChainScales{1.f, 1.f, 1.f, 1.f, 1.f}, const float *dyn = nullptr, float *amax = nullptr)
{
    if (M == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(x && wstream && out, "%s: null pointer", who);
    PCCX_CHECK_ARG(M > 0 && K0 >= 1 && N0 >= 1 && N1 >= 1 && N2 >= 1 && N3 >= 1 && ldo >= N3, "%s: bad shape", who);
    PCCX_CHECK_ARG((group == 1 || group == 32 || group == 64 || group == 128) && M % group == 0 && (group != 1 || (uintptr_t)out % 16 == 0),
                   "%s: group in {1,32,64,128} dividing M (group=%d M=%lld; group 1 = rows, out 16-byte aligned)", who, group, (long long)M);
    const long long ntiles = (M + 15) / 16, nblk = (ntiles + 7) / 8;
    PCCX_CHECK_ARG(nblk <= 0x7fffffffLL, "%s: M too large", who);
    const int KT0 = pg_kt32(K0);
    PCCX_CHECK_ARG(!idx || (rows_per_batch >= 1 && n_src >= 1 && ldp == 32 * KT0 && (uintptr_t)x % 16 == 0),
                   "%s: source rows must be 16-byte aligned with a stride of %d floats (got %d)", who, 32 * KT0, ldp);
    hipStream_t st = (hipStream_t)stream;
    auto kt = [](int N) { return pg_kt32(N); };
#define PG_CHAIN(G_, MQ0, KT1, MQ1, KT2, MQ2, KT3, NP)                                                                           \
    hipLaunchKernelGGL((planes_chain4_kernel<P, MQ0, KT1, MQ1, KT2, MQ2, KT3, NP, G_>), dim3((unsigned)nblk), dim3(256), 0, st,    \
                       (const uint4 *)x, idx, (long long)rows_per_batch, (long long)n_src, ldp, (long long)M, ntiles, KT0, wstream, b0, \
                       N0, b1, N1, b2, N2, b3, N3, group, out, ldo, sc, dyn, amax)
    if (N0 <= 32 && N1 > 32 && N1 <= 64 && N2 > 32 && N2 <= 64 && N3 > 64 && N3 <= 128) {
        // (3, 64, 64, 128): layer 1 reads one K block, layers 2 and 3 two
        PCCX_CHECK_ARG(kt(N0) == 1 && kt(N1) == 2 && kt(N2) == 2, "%s: unsupported widths %d %d %d %d", who, N0, N1, N2, N3);
        if (idx) PG_CHAIN(true, 1, 1, 1, 2, 1, 2, 1); else PG_CHAIN(false, 1, 1, 1, 2, 1, 2, 1);
    } else if (N0 > 96 && N0 <= 128 && N1 > 96 && N1 <= 128 && N2 > 96 && N2 <= 128 && N3 > 128 && N3 <= 256) {
        // (128, 128, 128, 256): four K blocks into every chained layer, two passes of the last
        if (idx) PG_CHAIN(true, 2, 4, 2, 4, 2, 4, 2); else PG_CHAIN(false, 2, 4, 2, 4, 2, 4, 2);
    } else {
        pccx_set_error("%s: unsupported widths %d %d %d %d", who, N0, N1, N2, N3);
        return PCCX_ERR_ARG;
    }
#undef PG_CHAIN
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

extern "C" int pccx_planes_chain4(const float *planes_in, int64_t M, int K0, const float *wstream, const float *b0, int N0,
                                  const float *b1, int N1, const float *b2, int N2, const float *b3, int N3, int group, float *out,
                                  int ldo, void *stream)
{
    return planes_chain4_launch<3>(planes_in, nullptr, 1, 1, 0, M, K0, wstream, b0, N0, b1, N1, b2, N2, b3, N3, group, out, ldo, stream,
                                   "pccx_planes_chain4");
}
// f16x2: planes_in / wstream / b0..b3 in the f16x2 forms (pccx_planes_gemm_h2); scales = {rho_in (gather only), rho_1, rho_2, rho_3,
// 1 / (sigma_3 tau_3)} with rho_l = sigma_l / (sigma_{l-1} tau_{l-1}); the output rows are also multiplied by dyn[1], the biases by dyn[0]
extern "C" int pccx_planes_chain4_h2(const float *planes_in, int64_t M, int K0, const float *wstream, const float *b0, int N0,
                                     const float *b1, int N1, const float *b2, int N2, const float *b3, int N3, int group,
                                     const float *scales5_host, const float *dyn, float *amax8, float *out, int ldo, void *stream)
{
    PCCX_CHECK_ARG(scales5_host, "pccx_planes_chain4_h2: null scales");
    PCCX_CHECK_ARG(!amax8 || group == 1, "pccx_planes_chain4_h2: the value maximum is collected with group == 1 (rows) only");
    const PgChainScales sc{scales5_host[0], scales5_host[1], scales5_host[2], scales5_host[3], scales5_host[4]};
    PCCX_CHECK_ARG(sc.rho_in > 0.f && sc.rho1 > 0.f && sc.rho2 > 0.f && sc.rho3 > 0.f && sc.inv3 > 0.f, "pccx_planes_chain4_h2: scales must be positive");
    return planes_chain4_launch<2>(planes_in, nullptr, 1, 1, 0, M, K0, wstream, b0, N0, b1, N1, b2, N2, b3, N3, group, out, ldo, stream,
                                   "pccx_planes_chain4_h2", sc, dyn, amax8);
}

extern "C" int pccx_planes_chain4_gather(const float *src, int ldp, const int64_t *idx, int64_t rows_per_batch, int64_t n_src, int64_t M,
                                         int K0, const float *wstream, const float *b0, int N0, const float *b1, int N1,
                                         const float *b2, int N2, const float *b3, int N3, int group, float *out, int ldo, void *stream)
{
    PCCX_CHECK_ARG(idx || M == 0, "pccx_planes_chain4_gather: null indices");
    return planes_chain4_launch<3>(src, idx, rows_per_batch, n_src, ldp, M, K0, wstream, b0, N0, b1, N1, b2, N2, b3, N3, group, out, ldo,
                                   stream, "pccx_planes_chain4_gather");
}
extern "C" int pccx_planes_chain4_gather_h2(const float *src, int ldp, const int64_t *idx, int64_t rows_per_batch, int64_t n_src, int64_t M,
                                            int K0, const float *wstream, const float *b0, int N0, const float *b1, int N1,
                                            const float *b2, int N2, const float *b3, int N3, int group, const float *scales5_host,
                                            const float *dyn, float *amax8, float *out, int ldo, void *stream)
{
    PCCX_CHECK_ARG(idx || M == 0, "pccx_planes_chain4_gather_h2: null indices");
    PCCX_CHECK_ARG(!amax8 || group == 1, "pccx_planes_chain4_gather_h2: the value maximum is collected with group == 1 (rows) only");
    PCCX_CHECK_ARG(scales5_host, "pccx_planes_chain4_gather_h2: null scales");
    const PgChainScales sc{scales5_host[0], scales5_host[1], scales5_host[2], scales5_host[3], scales5_host[4]};
    PCCX_CHECK_ARG(sc.rho_in > 0.f && sc.rho1 > 0.f && sc.rho2 > 0.f && sc.rho3 > 0.f && sc.inv3 > 0.f, "pccx_planes_chain4_gather_h2: scales must be positive");
    return planes_chain4_launch<2>(src, idx, rows_per_batch, n_src, ldp, M, K0, wstream, b0, N0, b1, N1, b2, N2, b3, N3, group, out, ldo,
                                   stream, "pccx_planes_chain4_gather_h2", sc, dyn, amax8);
}

// ---- three wide layers in one kernel -----------------------------------------------------------------------------
// The first three layers of sa3 (PPPF_AE.py:32-34: 259 -> 256 -> 256 -> 512) are too wide for the two-tiles-per-wave chain above
// (256 channels x 32 rows of accumulators and operand planes do not fit 256 registers), so they run as pn_forward_b3_kernel does:
// ONE 16-row tile per wave, eight waves per workgroup sharing the weight stream of the three layers through a double-buffered
// LDS-DMA ring (dense_b3_stream), every activation between the layers in registers; the last layer runs in two passes of 256
// output channels whose epilogue writes the operand planes of the layer that follows (512 -> 1024 + max: planes_gemm_kernel).
// The input rows are gathered in the kernel (fp32 source rows of 32 * KT0 floats, zero padded; idx -1 -> row 0).
#define PW_CHUNK 24
// a layer = one dense_b3_stream call per K block (the unroller gives up on a whole wide layer in one call)
template <int KT, class WS>
__device__ __forceinline__ void pw_layer(const WS &ws, int &f, const bf16x8 (&in)[1][KT][3], f32x4 (&acc)[1][16])
{
#pragma clang loop unroll(full)
    for (int kt = 0; kt < KT; ++kt) {
        bf16x8 pl[1][1][3];
#pragma unroll
        for (int p = 0; p < 3; ++p) pl[0][0][p] = in[0][kt][p];
        dense_b3_stream<1, 16, 1>(ws, f, pl, acc);
    }
}
template <int KT0>
__global__ __launch_bounds__(512, 1) void planes_chain_wide_kernel(const uint4 *__restrict__ src, const int64_t *__restrict__ idx,
                                                                   long long rows_per_batch, long long n_src, int ldp, long long M,
                                                                   long long ntiles, const float *__restrict__ wstream,
                                                                   const float *__restrict__ b0, const float *__restrict__ b1,
                                                                   const float *__restrict__ b2, int N0, int N1, int N2,
                                                                   uint4 *__restrict__ out)
{
    constexpr int FRAGS = (KT0 * 16 + 8 * 16 + 8 * 32) * 3;
    constexpr int NCH = (FRAGS + PW_CHUNK - 1) / PW_CHUNK;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, n = lane & 15;
    const long long tile = (long long)blockIdx.x * 8 + w;
    const bool valid = tile < ntiles;                         // an idle wave recomputes the last tile and discards it (barriers inside)
    const long long tc = valid ? tile : ntiles - 1;
    __shared__ __attribute__((aligned(16))) f32x4 swt[2 * PW_CHUNK * 64];
    const int wu = __builtin_amdgcn_readfirstlane(w);
    const WStreamT<PW_CHUNK, 2, 8> ws{wstream, swt, NCH, lane, wu, false};
    ws.prologue();

    bf16x8 in0[1][KT0][3];
    {
        long long r = tc * 16 + n;
        if (r >= M) r = M - 1;
        const long long j = idx ? idx[r] : r;
        const long long s = idx ? (r / rows_per_batch) * n_src + (j < 0 ? 0 : j) : r;
        const f32x4 *gp = (const f32x4 *)((const float *)src + (size_t)s * ldp + 4 * g);
#pragma unroll
        for (int t = 0; t < KT0; ++t) b3_split8(gp[8 * t], gp[8 * t + 4], in0[0][t]);
    }
    auto bias16 = [&](f32x4 (&acc)[1][16], const float *b, int N, int m0) {
#pragma unroll
        for (int mt = 0; mt < 16; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ch = 16 * (m0 + mt) + 4 * g + r;
                acc[0][mt][r] = (b && ch < N) ? b[ch] : 0.f;
            }
    };
    int f = 0;                                                // fragment cursor (constant-folds: the chain is fully unrolled)
    f32x4 a0[1][16];
    bias16(a0, b0, N0, 0);
    pw_layer<KT0>(ws, f, in0, a0);
    bf16x8 in1[1][8][3];
#pragma unroll
    for (int t = 0; t < 8; ++t) b3_split8(relu4(a0[0][2 * t]), relu4(a0[0][2 * t + 1]), in1[0][t]);
    f32x4 a1[1][16];
    bias16(a1, b1, N1, 0);
    pw_layer<8>(ws, f, in1, a1);
    bf16x8 in2[1][8][3];
#pragma unroll
    for (int t = 0; t < 8; ++t) b3_split8(relu4(a1[0][2 * t]), relu4(a1[0][2 * t + 1]), in2[0][t]);
#pragma clang loop unroll(full)
    for (int ps = 0; ps < 2; ++ps) {
        f32x4 a2[1][16];
        bias16(a2, b2, N2, 16 * ps);
        pw_layer<8>(ws, f, in2, a2);
        if (valid) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                bf16x8 pl[3];
                b3_split8(relu4(a2[0][2 * t]), relu4(a2[0][2 * t + 1]), pl);
                uint4 *d = out + (((size_t)(8 * ps + t) * ntiles + tile) * 3) * 64 + lane;
#pragma unroll
                for (int p = 0; p < 3; ++p) d[p * 64] = __builtin_bit_cast(uint4, pl[p]);
            }
        }
    }
    ws.drain();
}

// The weight stream of pccx_planes_chain_wide: the three layers' pccx_pack_linear_b3 planes, the third reordered into its two
// passes of 16 m-tiles, zero padded to whole ring chunks.
extern "C" size_t pccx_planes_chain_wide_weight_floats(int K0)
{
    const size_t frags = (size_t)(pg_kt32(K0 > 0 ? K0 : 1) * 16 + 8 * 16 + 8 * 32) * 3;
    return (frags + PW_CHUNK - 1) / PW_CHUNK * PW_CHUNK * 256;
}

extern "C" int pccx_pack_planes_chain_wide(const float *wp3_l0, const float *wp3_l1, const float *wp3_l2, int K0, int N0, int N1, int N2,
                                           float *wstream_dev, void *stream)
{
    PCCX_CHECK_ARG(wp3_l0 && wp3_l1 && wp3_l2 && wstream_dev, "pccx_pack_planes_chain_wide: null pointer");
    PCCX_CHECK_ARG(K0 >= 1 && N0 > 240 && N0 <= 256 && N1 > 240 && N1 <= 256 && N2 > 496 && N2 <= 512,
                   "pccx_pack_planes_chain_wide: widths %d %d %d are not (241..256, 241..256, 497..512)", N0, N1, N2);
    hipStream_t st = (hipStream_t)stream;
    const int KT0 = pg_kt32(K0);
    PCCX_CHECK_HIP(hipMemsetAsync(wstream_dev, 0, sizeof(float) * pccx_planes_chain_wide_weight_floats(K0), st));
    const size_t n0 = (size_t)KT0 * 16 * 3 * 256, n1 = (size_t)8 * 16 * 3 * 256;
    PCCX_CHECK_HIP(hipMemcpyAsync(wstream_dev, wp3_l0, sizeof(float) * n0, hipMemcpyDeviceToDevice, st));           // [t][16][3] as packed
    PCCX_CHECK_HIP(hipMemcpyAsync(wstream_dev + n0, wp3_l1, sizeof(float) * n1, hipMemcpyDeviceToDevice, st));
    const size_t total = (size_t)2 * 8 * 16 * 3 * 64;
    hipLaunchKernelGGL(planes_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const uint4 *)wp3_l2, 8, 32, 16, 2,
                       (uint4 *)(wstream_dev + n0 + n1), 3);                                                           // [pass][t][16][3]
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// relu(L2(relu(L1(relu(L0(x)))))) for widths (241..256, 241..256, 497..512) written as the operand planes of the next layer
// (pccx_planes_floats(M, N2) floats).  x: fp32 source rows of ldp = 32 * ceil(K0 / 32) floats (zero padded, 16-byte aligned), row r
// of the input = source row (r / rows_per_batch) * n_src + max(idx[r], 0), or row r itself when idx is NULL.  K0 <= 288 (nine K
// blocks: sa3's 259 channels); other shapes return PCCX_ERR_ARG and the caller runs the layers one by one.
extern "C" int pccx_planes_chain_wide(const float *src, int ldp, const int64_t *idx, int64_t rows_per_batch, int64_t n_src, int64_t M,
                                      int K0, const float *wstream, const float *b0, int N0, const float *b1, int N1, const float *b2,
                                      int N2, float *out_planes, void *stream)
{
    if (M == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(src && wstream && out_planes && M > 0, "pccx_planes_chain_wide: null pointer");
    const int KT0 = pg_kt32(K0);
    PCCX_CHECK_ARG(N0 > 240 && N0 <= 256 && N1 > 240 && N1 <= 256 && N2 > 496 && N2 <= 512 && (KT0 == 8 || KT0 == 9),
                   "pccx_planes_chain_wide: unsupported shape K0=%d widths %d %d %d", K0, N0, N1, N2);
    PCCX_CHECK_ARG(ldp == 32 * KT0 && (uintptr_t)src % 16 == 0 && (!idx || (rows_per_batch >= 1 && n_src >= 1)),
                   "pccx_planes_chain_wide: source rows must be 16-byte aligned with a stride of %d floats (got %d)", 32 * KT0, ldp);
    const long long ntiles = (M + 15) / 16, nblk = (ntiles + 7) / 8;
    PCCX_CHECK_ARG(nblk <= 0x7fffffffLL, "pccx_planes_chain_wide: M too large");
    hipStream_t st = (hipStream_t)stream;
#define PW_LAUNCH(KT_)                                                                                                          \
    hipLaunchKernelGGL((planes_chain_wide_kernel<KT_>), dim3((unsigned)nblk), dim3(512), 0, st, (const uint4 *)src, idx,           \
                       (long long)(idx ? rows_per_batch : 1), (long long)(idx ? n_src : 1), ldp, (long long)M, ntiles, wstream, b0, b1, \
                       b2, N0, N1, N2, (uint4 *)out_planes)
    if (KT0 == 9) PW_LAUNCH(9); else PW_LAUNCH(8);
#undef PW_LAUNCH
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}
