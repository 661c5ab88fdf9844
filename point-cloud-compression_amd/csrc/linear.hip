// linear.hip -- generic building blocks for the reference's other model families (PPPF_AE.py,
// pointnet_sa_module.py, pppe_pcd_ae.py:556-917): a runtime-shaped fp32 MFMA linear layer
// (1x1 Conv / Linear with eval-mode BatchNorm folded into weight and bias on the host, optional ReLU)
// and a max over the neighbour axis.  Activations are row-major "channels last" ((rows, C) with rows =
// batch x points x neighbours), which is exactly the lane map of the 16x16x4 B operand (one 16-byte
// load per lane) and of its C/D tile (one 16-byte store per lane), so no staging is needed.
// These rows are correctness-first (unfused, activations round-trip through HBM); the fused chains
// in encoder.hip / decoder.hip are the tuned path for the IPDAE configuration.
#include <math.h>
#include <stdlib.h>

#include "common.h"
#include "mfma_chain.h"

extern "C" size_t pccx_packed_linear_floats(int N, int K)
{
    if (N < 1 || K < 1) return 0;
    return (size_t)((K + 15) / 16) * (size_t)((N + 15) / 16) * 256;
}

// HOST: W (N,K) row-major -> fragments [kt][mt][lane][4] (mfma_chain.h), zero padded.
extern "C" int pccx_pack_linear(const float *W_host, int N, int K, float *wp_host)
{
    PCCX_CHECK_ARG(W_host && wp_host && N >= 1 && K >= 1, "pccx_pack_linear: bad arguments");
    const int KT = (K + 15) / 16, MT = (N + 15) / 16;
    for (int kt = 0; kt < KT; ++kt)
        for (int mt = 0; mt < MT; ++mt)
            for (int lane = 0; lane < 64; ++lane)
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * mt + (lane & 15), col = 16 * kt + 4 * (lane >> 4) + r;
                    wp_host[(((size_t)kt * MT + mt) * 64 + lane) * 4 + r] = (row < N && col < K) ? W_host[(size_t)row * K + col] : 0.f;
                }
    return PCCX_OK;
}

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ s16x4 to_bf16x4(const f32x4 &v)      // round to nearest even (v_cvt_pk_bf16_f32)
{
    return __builtin_bit_cast(s16x4, __builtin_convertvector(v, bf16x4v));
}
__device__ __forceinline__ f32x4 round_bf16x4(const f32x4 &v)
{
    return __builtin_convertvector(__builtin_convertvector(v, bf16x4v), f32x4);
}

// Column tiles per wave: wide (fewer re-reads of the rows) only while the grid still fills the chip several times over --
// a layer with few rows (the decoder's 4 x 1024 -> 24576 Linear of the pppe model) needs its parallelism from the columns.
static int pccx_linear_col_tiles(int M, int MT)
{
    const long rowblocks = ((long)M + 127) / 128;
    if (MT >= 16 && rowblocks * ((MT + 15) / 16) >= 2048) return 16;
    if (MT >= 8 && rowblocks * ((MT + 7) / 8) >= 2048) return 8;
    return 4;
}

// out[M][N] = act(x[M][K] . W^T + b).  Block = 4 waves; wave = 32 rows (2 point tiles) x MTB*16 columns.
// BF16 = the autocast form (train_pppe_pcd_ae.py:193-217, torch.cuda.amp.autocast around the forward): both operands rounded to
// bf16, products on the bf16 matrix cores (one v_mfma_f32_16x16x16_bf16 per k-tile: its lane map -- four consecutive k per
// lane -- is exactly the f32 fragment's), fp32 accumulate, result rounded to bf16 (the layer's output dtype under autocast).
// sum over the 16 lanes of a DPP row (the n index of a C/D tile), a fixed order; every lane ends with the sum
__device__ __forceinline__ float row16_sum(float v)
{
    int t;
    t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true);   // row_mirror
    v = v + __int_as_float(t);
    t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true);   // row_half_mirror
    v = v + __int_as_float(t);
    t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x1B, 0xf, 0xf, true);    // quad_perm [3,2,1,0]
    v = v + __int_as_float(t);
    t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true);    // quad_perm [1,0,3,2]
    v = v + __int_as_float(t);
    return v;
}

// MOM (the training step's Conv -> BatchNorm pairs, pppe_pcd_ae.py:556-568): the epilogue also accumulates the column moments sum z
// and sum z^2 of what it stores -- fp32 over the workgroup's 128 rows, then one double atomic per column and sum into replica
// blockIdx.x % nrep of `sums` ([nrep][2][N] doubles, cleared by the caller) -- so the BatchNorm that follows needs no pass of its own over
// the activation (col_reduce4_kernel<0>: one launch per layer, 13 per step).
// MOM = 2 (the backward of the same pairs: this GEMM is the dX of the layer BEHIND a BatchNorm-ReLU, its output is that BatchNorm's dY):
// the epilogue accumulates sum d xhat and sum d with d = (Y > 0 ? dY : 0), xhat = (Z - mean) rstd -- the two sums of
// col_reduce4_kernel<1> -- reading Y and Z of the rows it has just produced.
struct LinBn {
    const float *Y, *Z, *mean, *rstd;                            // (M, N) rows of the BatchNorm's output and input, (N) statistics
};
template <int MTB, bool VEC, bool BF16, int MOM = 0>
__global__ __launch_bounds__(256, 2) void linear_kernel(const float *__restrict__ x, int M, int K, int ldx,
                                                     const f32x4 *__restrict__ wp, int KT, int MT,
                                                     const float *__restrict__ bias, int N, int relu,
                                                     float *__restrict__ out, int ldo, double *__restrict__ sums = nullptr, int nrep = 1,
                                                     LinBn bn = LinBn{nullptr, nullptr, nullptr, nullptr})
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int g = lane >> 4, n = lane & 15;
    const int row0 = (blockIdx.x * 4 + w) * 32;
    if (MOM == 0 && row0 >= M) return;                            // whole wave (MOM: every wave reaches the barrier below; rows are clamped)
    const int mt0 = blockIdx.y * MTB;
    f32x4 acc[2][MTB];
#pragma unroll
    for (int m = 0; m < MTB; ++m) {
        f32x4 b;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 16 * (mt0 + m) + 4 * g + r;
            b[r] = (bias && c < N) ? bias[c] : 0.f;
        }
        acc[0][m] = b; acc[1][m] = b;
    }
    for (int kt = 0; kt < KT; ++kt) {
        f32x4 bx[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int row = row0 + nt * 16 + n;
            const int k = 16 * kt + 4 * g;
            const float *px = x + (size_t)(row < M ? row : M - 1) * ldx + k;
            if (VEC && k + 3 < K) {
                bx[nt] = *(const f32x4 *)px;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) bx[nt][r] = k + r < K ? px[r] : 0.f;
            }
        }
#pragma unroll
        for (int m = 0; m < MTB; ++m) {
            if (mt0 + m < MT) {                                   // uniform
                const f32x4 a = wp[((size_t)kt * MT + mt0 + m) * 64 + lane];
                if (BF16) {
                    const s16x4 a16 = to_bf16x4(a);
                    acc[0][m] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a16, to_bf16x4(bx[0]), acc[0][m], 0, 0, 0);
                    acc[1][m] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a16, to_bf16x4(bx[1]), acc[1][m], 0, 0, 0);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        acc[0][m] = mfma16(a[r], bx[0][r], acc[0][m]);
                        acc[1][m] = mfma16(a[r], bx[1][r], acc[1][m]);
                    }
                }
            }
        }
    }
    f32x4 dsum[MOM == 2 ? 2 : 1][MOM == 2 ? MTB : 1];
    (void)dsum;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int row = row0 + nt * 16 + n;
#pragma unroll
        for (int m = 0; m < MTB; ++m) {
            const int c = 16 * (mt0 + m) + 4 * g;
            f32x4 v = acc[nt][m];
            if (BF16) v = round_bf16x4(v);
            if (relu) v = relu4(v);
            if (MOM == 1) acc[nt][m] = row < M ? v : f32x4{0.f, 0.f, 0.f, 0.f};  // what is stored, for the moments below
            if (MOM == 2) {
                // d = relu'(y) dY and d xhat in place of the accumulators (acc[nt][m] <- d xhat; the d's go to a second array)
                f32x4 dd = {0.f, 0.f, 0.f, 0.f}, dx_ = {0.f, 0.f, 0.f, 0.f};
                if (row < M) {
                    const size_t e = (size_t)row * ldo + c;
                    if (VEC && c + 3 < N) {
                        const f32x4 y4 = *(const f32x4 *)(bn.Y + e), z4 = *(const f32x4 *)(bn.Z + e);
                        const f32x4 mu = *(const f32x4 *)(bn.mean + c), rs = *(const f32x4 *)(bn.rstd + c);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float d = y4[r] > 0.f ? v[r] : 0.f;
                            dd[r] = d;
                            dx_[r] = d * ((z4[r] - mu[r]) * rs[r]);
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (c + r < N) {
                                const float d = bn.Y[e + r] > 0.f ? v[r] : 0.f;
                                dd[r] = d;
                                dx_[r] = d * ((bn.Z[e + r] - bn.mean[c + r]) * bn.rstd[c + r]);
                            }
                    }
                }
                acc[nt][m] = dx_;
                dsum[nt][m] = dd;
            }
            if (row >= M) continue;
            float *po = out + (size_t)row * ldo + c;
            if (VEC && c + 3 < N) {
                *(f32x4 *)po = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (c + r < N) po[r] = v[r];
            }
        }
    }
    if constexpr (MOM != 0) {
        __shared__ float smom[4][2][MTB * 16];
#pragma unroll
        for (int m = 0; m < MTB; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a = acc[0][m][r], b = acc[1][m][r];
                const float s1 = row16_sum(a + b), s2 = MOM == 1 ? row16_sum(a * a + b * b) : row16_sum(dsum[0][m][r] + dsum[1][m][r]);
                if (n == 0) {
                    smom[w][0][16 * m + 4 * g + r] = s1;
                    smom[w][1][16 * m + 4 * g + r] = s2;
                }
            }
        __syncthreads();
        double *dst = sums + (size_t)(blockIdx.x % nrep) * 2 * N;
        for (int e = threadIdx.x; e < 2 * MTB * 16; e += 256) {
            const int which = e / (MTB * 16), cl = e % (MTB * 16), col = 16 * mt0 + cl;
            const float t = (smom[0][which][cl] + smom[1][which][cl]) + (smom[2][which][cl] + smom[3][which][cl]);
            if (col < N) atomicAdd(&dst[(size_t)which * N + col], (double)t);
        }
    }
}

// pccx_linear (no bias, no ReLU; flags bit 1 = autocast) whose epilogue accumulates the output's column moments into `sums`
// (pccx_train_sums_doubles(N) doubles: PCCX_SUM_REPLICAS x [sum z (N) | sum z^2 (N)]); flags bit 2 (4): `sums` was cleared by the caller.
extern "C" int pccx_linear_moments(const float *x, int M, int K, int ldx, const float *wp, int N, int flags, float *out, int ldo,
                                   double *sums, void *stream)
{
    if (M == 0) return PCCX_OK;
    PCCX_CHECK_ARG(x && wp && out && sums, "pccx_linear_moments: null pointer");
    PCCX_CHECK_ARG(M >= 0 && K >= 1 && N >= 1 && ldx >= K && ldo >= N, "pccx_linear_moments: bad shape M=%d K=%d N=%d ldx=%d ldo=%d", M, K, N, ldx, ldo);
    const int KT = (K + 15) / 16, MT = (N + 15) / 16;
    const bool vec = ldx % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)out % 16 == 0);
    const bool bf16 = (flags & 2) != 0;
    dim3 grid((M + 127) / 128, (MT + 3) / 4);
    PCCX_CHECK_ARG(grid.y <= 65535, "pccx_linear_moments: N=%d too large", N);
    hipStream_t st = (hipStream_t)stream;
    if (!(flags & 4)) PCCX_CHECK_HIP(pccx_zero_async(sums, sizeof(double) * PCCX_SUM_REPLICAS * 2 * (size_t)N, st));
#define PCCX_LINM_LAUNCH(V, B)                                                                                                   \
    hipLaunchKernelGGL((linear_kernel<4, V, B, 1>), grid, dim3(256), 0, st, x, M, K, ldx, (const f32x4 *)wp, KT, MT, (const float *)nullptr, N, \
                       0, out, ldo, sums, PCCX_SUM_REPLICAS)
    if (vec && bf16) PCCX_LINM_LAUNCH(true, true);
    else if (vec) PCCX_LINM_LAUNCH(true, false);
    else if (bf16) PCCX_LINM_LAUNCH(false, true);
    else PCCX_LINM_LAUNCH(false, false);
#undef PCCX_LINM_LAUNCH
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// pccx_linear as the dX GEMM behind a train-mode BatchNorm-ReLU: out = x . W^T (no bias; flags as pccx_linear_moments) is that
// BatchNorm's dY, and the epilogue accumulates its two column sums -- sum d xhat into sums[r][0][N], sum d into sums[r][1][N], d = (Y > 0 ?
// dY : 0), xhat = (Z - mean) rstd -- so pccx_bn_relu_train_backward (flags bit 3) needs no reduction pass.  Y, Z: (M, ldo-strided) rows.
extern "C" int pccx_linear_bnback(const float *x, int M, int K, int ldx, const float *wp, int N, int flags, float *out, int ldo,
                                  const float *Y, const float *Z, const float *mean, const float *rstd, double *sums, void *stream)
{
    if (M == 0) return PCCX_OK;
    PCCX_CHECK_ARG(x && wp && out && sums && Y && Z && mean && rstd, "pccx_linear_bnback: null pointer");
    PCCX_CHECK_ARG(M >= 0 && K >= 1 && N >= 1 && ldx >= K && ldo >= N, "pccx_linear_bnback: bad shape M=%d K=%d N=%d ldx=%d ldo=%d", M, K, N, ldx, ldo);
    const int KT = (K + 15) / 16, MT = (N + 15) / 16;
    const bool vec = ldx % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)out % 16 == 0);
    const bool bf16 = (flags & 2) != 0;
    dim3 grid((M + 127) / 128, (MT + 3) / 4);
    PCCX_CHECK_ARG(grid.y <= 65535, "pccx_linear_bnback: N=%d too large", N);
    hipStream_t st = (hipStream_t)stream;
    if (!(flags & 4)) PCCX_CHECK_HIP(pccx_zero_async(sums, sizeof(double) * PCCX_SUM_REPLICAS * 2 * (size_t)N, st));
    const LinBn bn{Y, Z, mean, rstd};
#define PCCX_LINB_LAUNCH(V, B)                                                                                                   \
    hipLaunchKernelGGL((linear_kernel<4, V, B, 2>), grid, dim3(256), 0, st, x, M, K, ldx, (const f32x4 *)wp, KT, MT, (const float *)nullptr, N, \
                       0, out, ldo, sums, PCCX_SUM_REPLICAS, bn)
    if (vec && bf16) PCCX_LINB_LAUNCH(true, true);
    else if (vec) PCCX_LINB_LAUNCH(true, false);
    else if (bf16) PCCX_LINB_LAUNCH(false, true);
    else PCCX_LINB_LAUNCH(false, false);
#undef PCCX_LINB_LAUNCH
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

extern "C" int pccx_linear(const float *x, int M, int K, int ldx, const float *wp, const float *bias, int N, int relu,
                           float *out, int ldo, void *stream)
{
    if (M == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(x && wp && out, "pccx_linear: null pointer");
    PCCX_CHECK_ARG(M >= 0 && K >= 1 && N >= 1 && ldx >= K && ldo >= N, "pccx_linear: bad shape M=%d K=%d N=%d ldx=%d ldo=%d", M,
                   K, N, ldx, ldo);
    if (M == 0) return PCCX_OK;
    const int KT = (K + 15) / 16, MT = (N + 15) / 16;
    const bool vec = ldx % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)out % 16 == 0);
    // columns per wave: every column block re-reads the wave's x rows, so wide layers take 16 column tiles per wave (a 512 -> 1024
    // layer on 8.4 M rows re-read its 17 GB input 16 times with 4 tiles per wave: 114 ms, bandwidth-bound)
    const bool bf16 = (relu & 2) != 0;                     // flags: bit 0 = ReLU, bit 1 = autocast (bf16 operands and result)
    relu &= 1;
    const int mtb = pccx_linear_col_tiles(M, MT);
    dim3 grid((M + 127) / 128, (MT + mtb - 1) / mtb);
    PCCX_CHECK_ARG(grid.y <= 65535, "pccx_linear: N=%d too large", N);
#define PCCX_LIN_LAUNCH(MTB_, V, B)                                                                                             \
    hipLaunchKernelGGL((linear_kernel<MTB_, V, B>), grid, dim3(256), 0, (hipStream_t)stream, x, M, K, ldx, (const f32x4 *)wp, KT, MT, \
                       bias, N, relu, out, ldo)
#define PCCX_LIN_LAUNCH_M(MTB_)                                                                                                 \
    do {                                                                                                                        \
        if (vec && bf16) PCCX_LIN_LAUNCH(MTB_, true, true);                                                                     \
        else if (vec) PCCX_LIN_LAUNCH(MTB_, true, false);                                                                       \
        else if (bf16) PCCX_LIN_LAUNCH(MTB_, false, true);                                                                      \
        else PCCX_LIN_LAUNCH(MTB_, false, false);                                                                               \
    } while (0)
    if (mtb == 16) PCCX_LIN_LAUNCH_M(16);
    else if (mtb == 8) PCCX_LIN_LAUNCH_M(8);
    else PCCX_LIN_LAUNCH_M(4);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// The same layer in the bf16x3 arithmetic of the tuned path: fp32 products formed from three bf16 pieces
// per operand, six v_mfma_f32_16x16x32_bf16 per K = 32 block, fp32 accumulate (fp32-level error, 2.6x the f32 MFMA rate).
// Weights: the packed f32 fragments are split ONCE on the device into planes [kt32][mt][plane] (pccx_pack_linear_b3);
// activations are split in registers per k-block (two f32x4 fragments = one K = 32 operand, b3_split8) and reused by the MTB
// column tiles of the wave.
// ------------------------------------------------------------------------------------------------------------------
extern "C" size_t pccx_packed_linear_b3_floats(int N, int K)
{
    if (N < 1 || K < 1) return 0;
    return (size_t)(((K + 15) / 16 + 1) / 2) * (size_t)((N + 15) / 16) * 3 * 256;
}

__global__ __launch_bounds__(256) void linear_b3_pack_kernel(const f32x4 *__restrict__ wp, int KT16, int MT, uint4 *__restrict__ out)
{
    const int lane = threadIdx.x & 63, T = (KT16 + 1) / 2;
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= (long)T * MT) return;
    const int mt = (int)(item % MT), t = (int)(item / MT);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const f32x4 v0 = wp[((size_t)(2 * t) * MT + mt) * 64 + lane];
    const f32x4 v1 = 2 * t + 1 < KT16 ? wp[((size_t)(2 * t + 1) * MT + mt) * 64 + lane] : zero;   // an odd last k-tile pairs with zeros
    bf16x8 pl[3];
    b3_split8(v0, v1, pl);
#pragma unroll
    for (int p = 0; p < 3; ++p) out[(((size_t)t * MT + mt) * 3 + p) * 64 + lane] = __builtin_bit_cast(uint4, pl[p]);
}

// wp_dev: the fragments of pccx_pack_linear (uploaded) or pccx_pack_linear_device -> out_dev: pccx_packed_linear_b3_floats(N,K) floats
extern "C" int pccx_pack_linear_b3(const float *wp_dev, int N, int K, float *out_dev, void *stream)
{
    PCCX_CHECK_ARG(wp_dev && out_dev && N >= 1 && K >= 1, "pccx_pack_linear_b3: bad arguments");
    const int KT16 = (K + 15) / 16, MT = (N + 15) / 16;
    const long items = (long)((KT16 + 1) / 2) * MT;
    hipLaunchKernelGGL(linear_b3_pack_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const f32x4 *)wp_dev,
                       KT16, MT, (uint4 *)out_dev);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

template <int MTB, bool VEC>
__global__ __launch_bounds__(256, 2) void linear_b3_kernel(const float *__restrict__ x, int M, int K, int ldx, const uint4 *__restrict__ wpl,
                                                        int KT32, int MT, const float *__restrict__ bias, int N, int relu,
                                                        float *__restrict__ out, int ldo)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int g = lane >> 4, n = lane & 15;
    const int row0 = (blockIdx.x * 4 + w) * 32;
    if (row0 >= M) return;                                        // whole wave
    const int mt0 = blockIdx.y * MTB;
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};     // smallest products first
    f32x4 acc[2][MTB];
#pragma unroll
    for (int m = 0; m < MTB; ++m) {
        f32x4 b;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 16 * (mt0 + m) + 4 * g + r;
            b[r] = (bias && c < N) ? bias[c] : 0.f;
        }
        acc[0][m] = b; acc[1][m] = b;
    }
    auto kstep = [&](int t) {
        bf16x8 pl[2][3];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int row = row0 + nt * 16 + n;
            const float *px = x + (size_t)(row < M ? row : M - 1) * ldx;
            f32x4 v[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = 32 * t + 16 * h + 4 * g;
                if (VEC && k + 3 < K) {
                    v[h] = *(const f32x4 *)(px + k);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[h][r] = k + r < K ? px[k + r] : 0.f;
                }
            }
            b3_split8(v[0], v[1], pl[nt]);
        }
#pragma unroll
        for (int m = 0; m < MTB; ++m) {
            if (mt0 + m < MT) {                                   // uniform
                const uint4 *wq = wpl + (((size_t)t * MT + mt0 + m) * 3) * 64 + lane;
                bf16x8 a[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) a[p] = __builtin_bit_cast(bf16x8, wq[p * 64]);
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    acc[0][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[PA[q]], pl[0][PB[q]], acc[0][m], 0, 0, 0);
                    acc[1][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[PA[q]], pl[1][PB[q]], acc[1][m], 0, 0, 0);
                }
            }
        }
    };
    for (int t = 0; t < KT32; ++t) kstep(t);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int row = row0 + nt * 16 + n;
        if (row >= M) continue;
#pragma unroll
        for (int m = 0; m < MTB; ++m) {
            const int c = 16 * (mt0 + m) + 4 * g;
            f32x4 v = acc[nt][m];
            if (relu) v = relu4(v);
            float *po = out + (size_t)row * ldo + c;
            if (VEC && c + 3 < N) {
                *(f32x4 *)po = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (c + r < N) po[r] = v[r];
            }
        }
    }
}

// pccx_linear with the weights given as bf16x3 planes (pccx_pack_linear_b3); relu: bit 0 only
extern "C" int pccx_linear_b3(const float *x, int M, int K, int ldx, const float *wplanes, const float *bias, int N, int relu,
                              float *out, int ldo, void *stream)
{
    if (M == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(x && wplanes && out, "pccx_linear_b3: null pointer");
    PCCX_CHECK_ARG(M >= 0 && K >= 1 && N >= 1 && ldx >= K && ldo >= N, "pccx_linear_b3: bad shape M=%d K=%d N=%d ldx=%d ldo=%d", M, K,
                   N, ldx, ldo);
    const int KT32 = ((K + 15) / 16 + 1) / 2, MT = (N + 15) / 16;
    const bool vec = ldx % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)out % 16 == 0);
    int mtb = pccx_linear_col_tiles(M, MT);                // as pccx_linear: fewer re-reads of x for wide layers
    // a handful of row blocks (the per-patch Linears of PPPF_AE: 2048 rows) cannot fill 256 CUs four column tiles at a time: one tile per
    // wave then (the rows are re-read per tile, from L2)
    if (mtb == 4 && (long)((M + 127) / 128) * ((MT + 3) / 4) < 256 && MT > 1) mtb = 1;
    dim3 grid((M + 127) / 128, (MT + mtb - 1) / mtb);
    PCCX_CHECK_ARG(grid.y <= 65535, "pccx_linear_b3: N=%d too large", N);
    relu &= 1;
#define PCCX_LINB3_LAUNCH(MTB_, V)                                                                                              \
    hipLaunchKernelGGL((linear_b3_kernel<MTB_, V>), grid, dim3(256), 0, (hipStream_t)stream, x, M, K, ldx, (const uint4 *)wplanes, \
                       KT32, MT, bias, N, relu, out, ldo)
    if (mtb == 16) { if (vec) PCCX_LINB3_LAUNCH(16, true); else PCCX_LINB3_LAUNCH(16, false); }
    else if (mtb == 8) { if (vec) PCCX_LINB3_LAUNCH(8, true); else PCCX_LINB3_LAUNCH(8, false); }
    else if (mtb == 1) { if (vec) PCCX_LINB3_LAUNCH(1, true); else PCCX_LINB3_LAUNCH(1, false); }
    else { if (vec) PCCX_LINB3_LAUNCH(4, true); else PCCX_LINB3_LAUNCH(4, false); }
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// out[g][c] = max_k x[g][k][c]   (torch.max(new_features, 3)[0], pointnet_sa_module.py:91)
__global__ void group_max_kernel(const float *__restrict__ x, long long G, int Kn, int C, float *__restrict__ out)
{
    const long long total = G * C;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long gi = e / C;
        const int c = (int)(e % C);
        const float *p = x + (gi * Kn) * C + c;
        float m = -INFINITY;
        for (int k = 0; k < Kn; ++k) m = fmaxf(m, p[(size_t)k * C]);
        out[e] = m;
    }
}

// out[b][g][c] = max_s y[b][max(idx[b][g][s], 0)][c]: torch.max over nsample of knn_gather(features, idx.clamp(min=0))
// (pointnet_sa_module.py:27-28,91) WITHOUT the (B, M, nsample, C) tensor.  It is what makes the de-duplicated PointnetSAModule
// possible (families.PointnetSAModule.run): that module gathers features and xyz un-centred, so every grouped row is a copy of a
// source row and the Conv-BN(eval)-ReLU stack needs to run on the N source rows only; the groups then take their maxima from the
// (B, N, C) result.  Workgroup = (cloud b, chunk of channels): the chunk of all N rows sits in LDS (<= 64 KB), a thread owns four
// channels of one group and walks the group's nsample indices (16-byte LDS reads, rows contiguous: conflict-free within a group).
// LDS-bound by design: 4 B read per (group, sample, channel); N too large for LDS reads through L2 instead.
// out[r][c] = act(base[r / div][c] + sum_{k < Ks} x[(mod ? r % mod : r)][k] * w[c][k]),  Ks <= 4.
// The first layer of FoldingNet's two MLPs (PPPF_AE.py:99-107) acts on [grid | latent] and [coarse | latent] rows whose 1024-wide
// latent part is the SAME for the 256 points of a patch: W [a ; latent] = W_a a + (W_lat latent + bias), so the latent part is one
// row per patch (a Linear on B rows, `base`) and what is left per point is this 2- or 3-term update -- instead of a K = 1026 / 1027
// product on every one of the B x 256 rows (0.8 of the decoder's matrix work) and the planes of their concatenation (3 GB per 2048
// patches).  fp32 fmaf chain on top of base; not bit-identical to the single long dot product (summation order), well inside the
// 1e-5 parity bar of the family tests.
__global__ __launch_bounds__(256) void rows_affine_small_kernel(const float *__restrict__ base, int C, unsigned div, const float *__restrict__ x,
                                                               int ldx, int Ks, unsigned mod, const float *__restrict__ w, int relu,
                                                               unsigned M, unsigned rows_per_block, float *__restrict__ out)
{
    // a thread keeps FOUR fixed channels (its Ks x 4 weights in registers) and walks the rows of its block: no per-element division
    const int c4n = C >> 2;
    const int cq = threadIdx.x % c4n, rl = threadIdx.x / c4n, rstep = 256 / c4n;      // 256 % c4n == 0 (checked by the launcher)
    const int c = 4 * cq;
    float wk[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int u = 0; u < 4; ++u) wk[k][u] = k < Ks ? w[(size_t)(c + u) * Ks + k] : 0.f;
    const unsigned r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, M);
    for (unsigned r = r0 + rl; r < r1; r += rstep) {
        const float *xr = x + (size_t)(mod ? r % mod : r) * ldx;
        const float4 v = *(const float4 *)(base + (size_t)(r / div) * C + c);
        float a[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < Ks) {
                const float xk = xr[k];
#pragma unroll
                for (int u = 0; u < 4; ++u) a[u] = fmaf(xk, wk[k][u], a[u]);
            }
        }
        if (relu) {
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] = fmaxf(a[u], 0.f);
        }
        *(float4 *)(out + (size_t)r * C + c) = make_float4(a[0], a[1], a[2], a[3]);
    }
}

extern "C" int pccx_rows_affine_small(const float *base, int C, int64_t div, const float *x, int ldx, int Ks, int64_t mod, const float *w,
                                      int relu, int64_t M, float *out, void *stream)
{
    if (M == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(base && x && w && out, "pccx_rows_affine_small: null pointer");
    const int c4n = C >> 2;
    PCCX_CHECK_ARG(M > 0 && M < 0x7fffffffLL && C >= 4 && C % 4 == 0 && c4n <= 256 && 256 % c4n == 0 && Ks >= 1 && Ks <= 4 && ldx >= Ks && div >= 1 &&
                       div < 0x7fffffffLL && mod >= 0 && mod < 0x7fffffffLL,
                   "pccx_rows_affine_small: bad arguments (C=%d must be 4 x a power of two <= 1024, Ks=%d in 1..4)", C, Ks);
    const unsigned rstep = 256 / c4n;
    unsigned rpb = (unsigned)((M + 8191) / 8192);                              // about 32 workgroups per CU
    if (rpb < 8 * rstep) rpb = 8 * rstep;
    rpb = (rpb + rstep - 1) / rstep * rstep;
    hipLaunchKernelGGL(rows_affine_small_kernel, dim3((unsigned)((M + rpb - 1) / rpb)), dim3(256), 0, (hipStream_t)stream, base, C, (unsigned)div, x, ldx,
                       Ks, (unsigned)mod, w, relu, (unsigned)M, rpb, out);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// softmax over the L levels of every (centre, latent dim) row of logits (rows, L) + pmf_to_cdf + torchac's integer CDF: the
// epilogue of prob_forward_kernel (AE.py:121-123, pn_kit.py:452-461, torchac 0.9.3) for shapes its fused form does not cover
// (--d > 16 or --d * --L > 128: compress.py:30-34 accepts any).  One thread per row; the same operation order as the epilogue.
__global__ __launch_bounds__(256) void softmax_cdf_kernel(const float *__restrict__ logits, long rows, int L, float *__restrict__ pmf,
                                                         float *__restrict__ cdf, int32_t *__restrict__ cdf_int)
{
    const long row = (long)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    const float *lg = logits + row * L;
    const int Lp = L + 1;
    float mx = -INFINITY;
    for (int l = 0; l < L; ++l) mx = fmaxf(mx, lg[l]);
    float sum = 0.f;
    for (int l = 0; l < L; ++l) sum += expf(lg[l] - mx);
    float run_c = 0.f;
    if (cdf) cdf[row * Lp] = 0.f;
    if (cdf_int) cdf_int[row * Lp] = 0;
    for (int l = 0; l < L; ++l) {
        const float pv = expf(lg[l] - mx) / sum;
        if (pmf) pmf[row * L + l] = pv;
        run_c = run_c + pv;
        const float cv = fminf(run_c, 1.0f);
        if (cdf) cdf[row * Lp + l + 1] = cv;
        if (cdf_int) cdf_int[row * Lp + l + 1] = ((int)rintf(cv * (float)(65536 - (Lp - 1))) + (l + 1)) & 0xFFFF;
    }
}

extern "C" int pccx_softmax_cdf(const float *logits, int64_t rows, int L, float *pmf, float *cdf, int32_t *cdf_int, void *stream)
{
    if (rows == 0) return PCCX_OK;
    PCCX_CHECK_ARG(logits && (pmf || cdf || cdf_int), "pccx_softmax_cdf: null pointer / no output requested");
    PCCX_CHECK_ARG(rows > 0 && L >= 1 && L <= 65535, "pccx_softmax_cdf: bad shape rows=%lld L=%d", (long long)rows, L);
    hipLaunchKernelGGL(softmax_cdf_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, logits, (long)rows, L, pmf, cdf,
                       cdf_int);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// decompress.py:104-116 after the generic decoder: out[(patch * k + p)] = ((patches[patch][p] / scale) + centres[patch] - 0.5) * longest[b]
// / (1 - margin) + center[b], b = patch / S -- the fused decoders' epilogue (decoder.hip) as an op of its own, same operation order.
__global__ __launch_bounds__(256) void reassemble_kernel(const float *__restrict__ patches, long P, int k, float scale,
                                                        const float *__restrict__ centres, const float *__restrict__ nrm_center,
                                                        const float *__restrict__ nrm_longest, int S, float one_minus_margin,
                                                        float *__restrict__ out)
{
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= P * k) return;
    const long patch = e / k;
    const int b = (int)(patch / S);
    const float lg = nrm_longest[b];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float t = __fdiv_rn(patches[3 * e + a], scale);
        t = __fadd_rn(t, centres[patch * 3 + a]);
        t = __fsub_rn(t, 0.5f);
        t = __fdiv_rn(__fmul_rn(t, lg), one_minus_margin);
        out[3 * e + a] = __fadd_rn(t, nrm_center[3 * b + a]);
    }
}

extern "C" int pccx_reassemble(const float *patches, int64_t P, int k, float scale, const float *centres, const float *nrm_center,
                               const float *nrm_longest, int S, double margin, float *out, void *stream)
{
    if (P == 0) return PCCX_OK;
    PCCX_CHECK_ARG(patches && centres && nrm_center && nrm_longest && out, "pccx_reassemble: null pointer");
    PCCX_CHECK_ARG(P > 0 && k >= 1 && S >= 1 && scale != 0.f, "pccx_reassemble: bad arguments");
    hipLaunchKernelGGL(reassemble_kernel, dim3((unsigned)((P * k + 255) / 256)), dim3(256), 0, (hipStream_t)stream, patches, (long)P, k, scale, centres,
                       nrm_center, nrm_longest, S, (float)(1.0 - margin), out);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

#define GM_THREADS 1024                    // 2 workgroups of 64 KB per CU = all 32 wave slots: the walk is a chain of dependent reads
// the tail of cloud b's padded output rows (pccx_gather_max_rows): columns C .. C + 2 <- the centroids' coordinates, the rest <- 0
__device__ __forceinline__ void gm_write_tail(float *__restrict__ out, const float *__restrict__ xyz, int b, int M, int C, int ldo, int tid)
{
    const int tw = ldo - C;
    for (int e = tid; e < M * tw; e += GM_THREADS) {
        const int g = e / tw, c = e - g * tw;
        out[((size_t)b * M + g) * ldo + C + c] = c < 3 ? xyz[((size_t)b * M + g) * 3 + c] : 0.f;
    }
}
template <bool LDS_TILE>
__global__ __launch_bounds__(GM_THREADS) void gather_max_kernel(const float *__restrict__ y, int B, int N, int C, const int64_t *__restrict__ idx,
                                                                int M, int ns, int chunk, float *__restrict__ out, int ldo,
                                                                const float *__restrict__ xyz)
{
    extern __shared__ __attribute__((aligned(16))) float gm_tile[];
    const int c0 = blockIdx.x * chunk, tid = threadIdx.x;
    const int cw = min(chunk, C - c0), q4 = chunk >> 2;                      // chunk % 4 == 0, C % 4 == 0
    const int ldy4 = C >> 2;
  for (int b = blockIdx.y; b < B; b += gridDim.y) {                          // more clouds than the grid's y range: walk them
    if (xyz && blockIdx.x == 0) gm_write_tail(out, xyz, b, M, C, ldo, tid);
    const float4 *y4 = (const float4 *)(y + (size_t)b * N * C + c0);
    if (LDS_TILE) {
        float4 *t4 = (float4 *)gm_tile;
        for (int i = tid; i < N * q4; i += GM_THREADS) {
            const int row = i / q4, q = i - row * q4;
            if (4 * q < cw) t4[i] = y4[(size_t)row * ldy4 + q];
        }
        __syncthreads();
    }
    const float4 *t4 = (const float4 *)gm_tile;
    for (int it = tid; it < M * q4; it += GM_THREADS) {
        const int g = it / q4, q = it - g * q4;
        if (4 * q >= cw) continue;
        const int64_t *ig = idx + ((size_t)b * M + g) * ns;
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        int s_ = 0;
        for (; s_ + 8 <= ns; s_ += 8) {                                       // eight index loads in flight, then eight row reads
            long long j[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) j[u] = ig[s_ + u];
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const long long jj = j[u] < 0 ? 0 : j[u];                     // idx.clamp(min=0), pointnet_sa_module.py:27
                v[u] = LDS_TILE ? t4[(int)jj * q4 + q] : y4[(size_t)jj * ldy4 + q];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { m.x = fmaxf(m.x, v[u].x); m.y = fmaxf(m.y, v[u].y); m.z = fmaxf(m.z, v[u].z); m.w = fmaxf(m.w, v[u].w); }
        }
        for (; s_ < ns; ++s_) {
            long long jj = ig[s_];
            jj = jj < 0 ? 0 : jj;
            const float4 v = LDS_TILE ? t4[(int)jj * q4 + q] : y4[(size_t)jj * ldy4 + q];
            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
        ((float4 *)(out + ((size_t)b * M + g) * ldo + c0))[q] = m;
    }
    if (LDS_TILE) __syncthreads();                                            // the tile is rewritten for the next cloud
  }
}

// The same reduction with the cloud's WHOLE index table held in LDS as 16-bit entries (M * ns * 2 bytes: 32 / 16 / 8 KB for the three
// set-abstraction levels of PPPF_AE.py:29-34) and the channel chunks walked INSIDE the workgroup: the int64 table (8 bytes per entry)
// is read from memory once per cloud instead of once per channel chunk -- it was 1.07 GB of the first level's 1.6 GB per 2048 patches --
// and a group's walk chains LDS reads only.  Workgroup = cloud; tile = `chunk` channels of all N rows.
__global__ __launch_bounds__(GM_THREADS) void gather_max_lds_idx_kernel(const float *__restrict__ y, int B, int N, int C, const int64_t *__restrict__ idx,
                                                                        int M, int ns, int chunk, float *__restrict__ out, int ldo,
                                                                        const float *__restrict__ xyz)
{
    extern __shared__ __attribute__((aligned(16))) float gm_tile[];
    const int tid = threadIdx.x, q4 = chunk >> 2, ldy4 = C >> 2;
    unsigned short *sidx = (unsigned short *)(gm_tile + (size_t)N * chunk);          // [M][ns] after the tile
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    if (xyz) gm_write_tail(out, xyz, b, M, C, ldo, tid);
    const int64_t *ib = idx + (size_t)b * M * ns;
    for (int i = tid; i < M * ns; i += GM_THREADS) {
        const long long j = ib[i];
        sidx[i] = (unsigned short)(j < 0 ? 0 : j);                                    // idx.clamp(min=0), pointnet_sa_module.py:27
    }
    for (int c0 = 0; c0 < C; c0 += chunk) {
        const int cw = min(chunk, C - c0);
        const float4 *y4 = (const float4 *)(y + (size_t)b * N * C + c0);
        float4 *t4w = (float4 *)gm_tile;
        __syncthreads();                                                              // the previous chunk's walks are done (and sidx is written)
        for (int i = tid; i < N * q4; i += GM_THREADS) {
            const int row = i / q4, q = i - row * q4;
            if (4 * q < cw) t4w[i] = y4[(size_t)row * ldy4 + q];
        }
        __syncthreads();
        const float4 *t4 = (const float4 *)gm_tile;
        for (int it = tid; it < M * q4; it += GM_THREADS) {
            const int g = it / q4, q = it - g * q4;
            if (4 * q >= cw) continue;
            const unsigned short *ig = sidx + (size_t)g * ns;
            float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            int s_ = 0;
            for (; s_ + 8 <= ns; s_ += 8) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = t4[(int)ig[s_ + u] * q4 + q];
#pragma unroll
                for (int u = 0; u < 8; ++u) { m.x = fmaxf(m.x, v[u].x); m.y = fmaxf(m.y, v[u].y); m.z = fmaxf(m.z, v[u].z); m.w = fmaxf(m.w, v[u].w); }
            }
            for (; s_ < ns; ++s_) {
                const float4 v = t4[(int)ig[s_] * q4 + q];
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
            ((float4 *)(out + ((size_t)b * M + g) * ldo + c0))[q] = m;
        }
    }
    __syncthreads();                                                                  // sidx / the tile are rewritten for the next cloud
  }
}

static int gather_max_launch(const float *y, int B, int N, int C, const int64_t *idx, int M, int ns, float *out, int ldo, void *stream,
                             const float *xyz = nullptr)
{
    if (B == 0 || M == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(y && idx && out, "pccx_gather_max: null pointer");
    PCCX_CHECK_ARG(B > 0 && N >= 1 && M >= 1 && ns >= 1 && C >= 4 && C % 4 == 0, "pccx_gather_max: bad shape B=%d N=%d C=%d M=%d ns=%d (C %% 4 == 0)",
                   B, N, C, M, ns);
    PCCX_CHECK_ARG(ldo >= C && ldo % 4 == 0 && (uintptr_t)out % 16 == 0, "pccx_gather_max: output rows of %d floats for %d channels (a multiple of 4, 16-byte aligned)", ldo, C);
    const unsigned gy = (unsigned)(B < 65535 ? B : 65535);
    {
        // index table in LDS (16-bit) + a 32 KB tile = at most 64 KB per workgroup: two workgroups per CU as before
        const size_t ib = ((size_t)M * ns * 2 + 15) / 16 * 16;
        int ck = (int)((size_t)32768 / ((size_t)N * 4)) & ~3;
        if (ck > C) ck = C;
        if (N <= 65536 && ck >= 16 && ib <= 32768 && !getenv("PCCX_GATHER_MAX_PLAIN")) {
            PCCX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gather_max_lds_idx_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               64 * 1024));
            hipLaunchKernelGGL(gather_max_lds_idx_kernel, dim3(gy), dim3(GM_THREADS), (size_t)N * ck * 4 + ib, (hipStream_t)stream, y, B, N, C, idx, M,
                               ns, ck, out, ldo, xyz);
            PCCX_CHECK_LAUNCH();
            return PCCX_OK;
        }
    }
    int chunk = (int)((size_t)65536 / ((size_t)N * 4)) & ~3;                  // channels of all N rows in 64 KB
    if (chunk > C) chunk = C;
    if (chunk >= 16) {
        PCCX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gather_max_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           64 * 1024));
        hipLaunchKernelGGL(gather_max_kernel<true>, dim3((C + chunk - 1) / chunk, gy), dim3(GM_THREADS), (size_t)N * chunk * 4, (hipStream_t)stream, y,
                           B, N, C, idx, M, ns, chunk, out, ldo, xyz);
    } else {
        chunk = C < 256 ? C : 256;                                            // rows through L2: a workgroup per 256 channels
        hipLaunchKernelGGL(gather_max_kernel<false>, dim3((C + chunk - 1) / chunk, gy), dim3(GM_THREADS), 0, (hipStream_t)stream, y, B, N, C, idx,
                           M, ns, chunk, out, ldo, xyz);
    }
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

extern "C" int pccx_gather_max(const float *y, int B, int N, int C, const int64_t *idx, int M, int ns, float *out, void *stream)
{
    return gather_max_launch(y, B, N, C, idx, M, ns, out, C, stream);
}

// pccx_gather_max writing the NEXT set-abstraction level's input rows directly (pointnet_sa_module.py:83: features first, xyz last):
// out (B, M, ldo) with ldo = 32 * ceil((C + 3) / 32) floats per row = [the C maxima | xyz (B, M, 3) of the centroids | zeros] -- the
// padded fp32 rows the gathering forms of the planes kernels read (pccx_planes_chain4_gather_h2 / pccx_planes_gemm_gather_h2), so
// the level's operand-plane pass (pccx_group_planes_h2: 1.2 GB of traffic for the second level of PPPF_AE on 2048 patches) is not run.
extern "C" int pccx_gather_max_rows(const float *y, int B, int N, int C, const int64_t *idx, int M, int ns, const float *xyz, float *out, int ldo,
                                    void *stream)
{
    if (B == 0 || M == 0) return PCCX_OK;
    PCCX_CHECK_ARG(xyz, "pccx_gather_max_rows: null coordinates");
    PCCX_CHECK_ARG(ldo == (C + 3 + 31) / 32 * 32, "pccx_gather_max_rows: rows of %d floats for %d + 3 channels, need %d", ldo, C, (C + 3 + 31) / 32 * 32);
    return gather_max_launch(y, B, N, C, idx, M, ns, out, ldo, stream, xyz);     // the tails are written by the same kernel
}

extern "C" int pccx_group_max(const float *x, int64_t G, int Kn, int C, float *out, void *stream)
{
    if (G == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(x && out && G >= 0 && Kn >= 1 && C >= 1, "pccx_group_max: bad arguments");
    if (G == 0) return PCCX_OK;
    long long blocks = ((long long)G * C + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(group_max_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (long long)G, Kn, C, out);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// y = sigmoid(x) * spread - spread/2 (PPPF_AE.py:136-137), optionally rounded (AE.py:79-81).
__global__ void sigmoid_spread_kernel(const float *__restrict__ x, long n, float spread, float half, int do_round,
                                      float *__restrict__ y)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float s = 1.0f / (1.0f + expf(-x[i]));
        float v = __fsub_rn(__fmul_rn(s, spread), half);
        y[i] = do_round ? rintf(v) : v;
    }
}

extern "C" int pccx_sigmoid_spread(const float *x, int64_t n, int L, int do_round, float *y, void *stream)
{
    if (n == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(x && y && n >= 0 && L >= 1, "pccx_sigmoid_spread: bad arguments");
    if (n == 0) return PCCX_OK;
    long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(sigmoid_spread_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (long)n,
                       (float)((double)L - 0.2), (float)(((double)L - 0.2) / 2), do_round, y);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// y = rint(x) (AE.STEQuantize.forward, AE.py:79-81)
__global__ void round_kernel(const float *__restrict__ x, long n, float *__restrict__ y)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = rintf(x[i]);
}

extern "C" int pccx_round(const float *x, int64_t n, float *y, void *stream)
{
    if (n == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(x && y && n >= 0, "pccx_round: bad arguments");
    if (n == 0) return PCCX_OK;
    long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(round_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (long)n, y);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// quantize_st forward value + dequantisation (pppe_pcd_ae.py:719-735, :873):
//   scaled = (clamp(x,min,max) - min) / (max - min + 1e-9) * (levels-1); y_q = clamp(round(scaled), 0, levels-1)
//   y_deq  = y_q / (levels-1) * (max - min) + min
__global__ void quantize_st_kernel(const float *__restrict__ x, long n, float qmin, float qmax, float denom, float lm1,
                                   float *__restrict__ yq, float *__restrict__ ydeq)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float xc = fminf(fmaxf(x[i], qmin), qmax);
        const float scaled = __fmul_rn(__fdiv_rn(__fsub_rn(xc, qmin), denom), lm1);
        const float q = fminf(fmaxf(rintf(scaled), 0.f), lm1);
        yq[i] = q;
        if (ydeq) ydeq[i] = __fadd_rn(__fmul_rn(__fdiv_rn(q, lm1), __fsub_rn(qmax, qmin)), qmin);
    }
}

extern "C" int pccx_quantize_st(const float *x, int64_t n, float qmin, float qmax, int levels, float *y_q, float *y_deq,
                                void *stream)
{
    if (n == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(x && y_q && n >= 0 && levels >= 2 && qmax > qmin, "pccx_quantize_st: bad arguments");
    if (n == 0) return PCCX_OK;
    long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(quantize_st_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (long)n, qmin, qmax,
                       (float)((double)(qmax - qmin) + 1e-9), (float)(levels - 1), y_q, y_deq);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}
