// train.hip -- backward / training-mode primitives for the step of train_pppe_pcd_ae.py:184-226
// (BASELINE configs[4]): the pppe PointCloudAE forward in TRAIN mode (BatchNorm on batch
// statistics), Chamfer + smooth-L1 loss, backward, gradient clipping, Adam.
//
// Everything works on row-major "channels last" activations (rows = batch x points x neighbours), like
// linear.hip.  GEMM-shaped backward passes run on the fp32 matrix cores:
//   dX = dZ . W         -> pccx_linear with the transposed weight packed on the device
//   dW = dZ^T . X       -> linear_dw_kernel below (reduction over rows = the MFMA k dimension)
// Column reductions (BatchNorm moments, bias / gamma / beta gradients) accumulate in double through
// atomics; Adam / clipping are elementwise.  Correctness-first: parity against torch autograd on the
// oracle restatement (tests/test_train_step.py), not tuned.
#include <math.h>

#include "common.h"
#include "mfma_chain.h"

#define ROWS_PER_BLOCK 256

// ---- device-side packing of W (N,K) [or its transpose] into MFMA A fragments (weights change every step)
__global__ void pack_linear_dev_kernel(const float *__restrict__ W, int N, int K, int transpose, float *__restrict__ wp)
{
    // logical matrix A (rows R, cols Cc): A = W (R=N, Cc=K) or W^T (R=K, Cc=N)
    const int R = transpose ? K : N, Cc = transpose ? N : K;
    const int KT = (Cc + 15) / 16, MT = (R + 15) / 16;
    const long total = (long)KT * MT * 256;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int r4 = (int)(e & 3), lane = (int)((e >> 2) & 63);
        const long f = e >> 8;
        const int mt = (int)(f % MT), kt = (int)(f / MT);
        const int row = 16 * mt + (lane & 15), col = 16 * kt + 4 * (lane >> 4) + r4;
        float v = 0.f;
        if (row < R && col < Cc) v = transpose ? W[(size_t)col * K + row] : W[(size_t)row * K + col];
        wp[e] = v;
    }
}

extern "C" int pccx_pack_linear_device(const float *W, int N, int K, int transpose, float *wp, void *stream)
{
    PCCX_CHECK_ARG(W && wp && N >= 1 && K >= 1, "pccx_pack_linear_device: bad arguments");
    const int R = transpose ? K : N, Cc = transpose ? N : K;
    const long total = (long)((Cc + 15) / 16) * ((R + 15) / 16) * 256;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_linear_dev_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, W, N, K, transpose, wp);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// ---- dW[N][K] += sum_m dZ[m][n] * X[m][k]: one wave per (16 n x 16 k) tile and row slice
__device__ __forceinline__ float round_bf16(float v) { return (float)(__bf16)v; }

// ---- dW[N][K] += sum_m dZ[m][n] * X[m][k]: the reduction runs over the rows (the MFMA k dimension).  One wave owns a
// (16 TN) x (16 TK) tile of dW and a slice of the rows: per step of four rows it loads TN + TK operand registers and issues
// TN * TK MFMAs (the first version loaded two registers per MFMA and was bound by its loads: 2.6 ms of the 13.7 ms step).
// BF16 = the autocast backward: dZ and X rounded to bf16 (their products are then exact in fp32), fp32 accumulate.
template <bool BF16, int TN, int TK>
__global__ __launch_bounds__(256) void linear_dw_kernel(const float *__restrict__ dZ, const float *__restrict__ X, long M, int N,
                                                        int K, int ldz, int ldx, int rows_per_slice, float *__restrict__ dW)
{
    // One workgroup = one (16 TN) x (16 TK) tile of dW and one row slice; its four waves take a quarter of the slice's rows each and
    // add their partial tiles through LDS, so the slice ends with ONE fp32 atomic per element (round 3: every wave owned a tile and a
    // slice of its own and 1024 slices x N x K same-address atomics were most of the kernel -- 4 M for a 64 x 64 layer).
    extern __shared__ float dw_red[];                                  // [4 waves][TN * TK tiles][256]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int nt0 = blockIdx.x * TN, kt0 = blockIdx.y * TK;
    const long s0 = (long)blockIdx.z * rows_per_slice;
    const long s1 = s0 + rows_per_slice < M ? s0 + rows_per_slice : M;
    const long per = ((s1 - s0 + 3) / 4 + 3) / 4 * 4;                  // rows per wave, a multiple of the MFMA's four
    const long m0 = s0 + w * per;
    const long m1 = m0 + per < s1 ? m0 + per : s1;
    f32x4 acc[TN][TK];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TK; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // U steps of four rows per trip: the U (TN + TK) operand loads of a trip are issued before its first MFMA (one step per trip exposed
    // an HBM latency per 16 MFMAs).  The MFMAs of a tile still see the wave's rows in ascending order.
    constexpr int U = (TN + TK <= 4) ? 8 : 4;
    // every load is UNCONDITIONAL from a clamped address and masked afterwards: written as `ok ? load : 0` the compiler puts each load in
    // a branch of its own, and in the bf16 form sinks the rounding into that branch behind an s_waitcnt -- 32 serialised loads per trip
    // (the bf16 kernel measured 36 us per call against the fp32 one's 21.5)
    int ncol[TN], kcol[TK];
    bool nok[TN], kok[TK];
#pragma unroll
    for (int a = 0; a < TN; ++a) {
        const int n = (nt0 + a) * 16 + c;
        nok[a] = n < N;
        ncol[a] = nok[a] ? n : N - 1;
    }
#pragma unroll
    for (int b = 0; b < TK; ++b) {
        const int k = (kt0 + b) * 16 + c;
        kok[b] = k < K;
        kcol[b] = kok[b] ? k : K - 1;
    }
    for (long m = m0; m < m1; m += 4 * U) {
        float av[U][TN], bv[U][TK];
        bool rok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long mm = m + 4 * u + g;                           // MFMA k index = row within the 4-row step
            rok[u] = mm < m1;
            const long mr = rok[u] ? mm : m1 - 1;
            const float *zr = dZ + (size_t)mr * ldz, *xr = X + (size_t)mr * ldx;
#pragma unroll
            for (int a = 0; a < TN; ++a) av[u][a] = zr[ncol[a]];     // A[i = n][k = g]
#pragma unroll
            for (int b = 0; b < TK; ++b) bv[u][b] = xr[kcol[b]];     // B[k = g][j = k]
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int a = 0; a < TN; ++a) {
                av[u][a] = (rok[u] && nok[a]) ? av[u][a] : 0.f;
                if (BF16) av[u][a] = round_bf16(av[u][a]);
            }
#pragma unroll
            for (int b = 0; b < TK; ++b) {
                bv[u][b] = (rok[u] && kok[b]) ? bv[u][b] : 0.f;
                if (BF16) bv[u][b] = round_bf16(bv[u][b]);
            }
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TK; ++b) acc[a][b] = mfma16(av[u][a], bv[u][b], acc[a][b]);
        }
    }
    // partial tiles to LDS; wave w then owns tiles w, w + 4, ...: the four partials in wave order, one atomic per element
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TK; ++b) *(f32x4 *)(dw_red + ((size_t)(w * TN * TK + a * TK + b) * 64 + lane) * 4) = acc[a][b];
    __syncthreads();
    for (int t = w; t < TN * TK; t += 4) {
        const int a = t / TK, b = t % TK;
        f32x4 v = *(const f32x4 *)(dw_red + ((size_t)t * 64 + lane) * 4);
#pragma unroll
        for (int q = 1; q < 4; ++q) {
            const f32x4 o = *(const f32x4 *)(dw_red + ((size_t)(q * TN * TK + t) * 64 + lane) * 4);
            v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
        }
        // D[i = n-row 4g+r][j = k-col c]
        const int k = (kt0 + b) * 16 + c;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int nn = (nt0 + a) * 16 + 4 * g + r;
            if (nn < N && k < K) atomicAdd(&dW[(size_t)nn * K + k], v[r]);
        }
    }
}

template <int TN, int TK>
static int launch_dw(const float *dZ, const float *X, int64_t M, int N, int K, int ldz, int ldx, float *dW, int flags, hipStream_t st)
{
    const int nt = (N + 15) / 16, kt = (K + 15) / 16;
    // tiles x slices ~ 512 workgroups (two per CU), each slice at least 256 rows (64 per wave): the atomics are slices x N x K
    const int tiles = ((nt + TN - 1) / TN) * ((kt + TK - 1) / TK);
    int slices = (512 + tiles - 1) / tiles;
    const int max_slices = (int)((M + 255) / 256);
    if (slices > max_slices) slices = max_slices;
    if (slices < 1) slices = 1;
    if (slices > 1024) slices = 1024;
    int rps = (int)((M + slices - 1) / slices);
    rps = (rps + 15) / 16 * 16;
    slices = (int)((M + rps - 1) / rps);
    dim3 grid((nt + TN - 1) / TN, (kt + TK - 1) / TK, slices);
    PCCX_CHECK_ARG(grid.y <= 65535 && grid.z <= 65535, "pccx_linear_dw: shape too large");
    const size_t lds = (size_t)4 * TN * TK * 256 * sizeof(float);
    if (lds > 48 * 1024) {
        PCCX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&linear_dw_kernel<true, TN, TK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        PCCX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&linear_dw_kernel<false, TN, TK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (flags & 2)
        hipLaunchKernelGGL((linear_dw_kernel<true, TN, TK>), grid, dim3(256), lds, st, dZ, X, (long)M, N, K, ldz, ldx, rps, dW);
    else
        hipLaunchKernelGGL((linear_dw_kernel<false, TN, TK>), grid, dim3(256), lds, st, dZ, X, (long)M, N, K, ldz, ldx, rps, dW);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

extern "C" int pccx_linear_dw(const float *dZ, const float *X, int64_t M, int N, int K, int ldz, int ldx, float *dW, int flags,
                              void *stream)
{
    if (M == 0) return PCCX_OK;
    PCCX_CHECK_ARG(dZ && X && dW && N >= 1 && K >= 1 && ldz >= N && ldx >= K, "pccx_linear_dw: bad arguments");
    const int nt = (N + 15) / 16, kt = (K + 15) / 16;
    if (nt >= 4 && kt >= 4) return launch_dw<4, 4>(dZ, X, M, N, K, ldz, ldx, dW, flags, (hipStream_t)stream);
    if (nt >= 2 && kt >= 2) return launch_dw<2, 2>(dZ, X, M, N, K, ldz, ldx, dW, flags, (hipStream_t)stream);
    if (nt >= 2) return launch_dw<2, 1>(dZ, X, M, N, K, ldz, ldx, dW, flags, (hipStream_t)stream);
    return launch_dw<1, 1>(dZ, X, M, N, K, ldz, ldx, dW, flags, (hipStream_t)stream);
}

// ---- layers with a handful of rows (the pppe model's global / decoder / probability Linears: M = batch = 4) ---------------------
// out[m][n] = act(sum_k x[m][k] W[n][k] + b[n]) and dX[m][k] = sum_n dZ[m][n] W[n][k] for M <= 8 rows straight from the UNPACKED
// row-major W (N, K): these products are weight streams (the 24576 x 1024 expansion layer is 100 MB), not matrix work.  The generic
// MFMA layer walks K serially per workgroup (one exposed L2 / HBM latency per k-tile: 2.2 ms for that layer's dX at 256
// workgroups); here every W row is read once, coalesced, by a wave (forward) or by n-chunks of a split-K grid (dX, fp32 atomics
// into a zeroed dX like the other gradient kernels), and no packed copy of W is made.  bf16 = the autocast rounding of linear_kernel.
__device__ __forceinline__ float sk_bf16(float v) { return (float)(__bf16)v; }

template <bool BF16>
__global__ __launch_bounds__(256) void skinny_fwd_kernel(const float *__restrict__ x, int M, int K, int ldx, const float *__restrict__ W,
                                                        const float *__restrict__ bias, int N, int relu, float *__restrict__ out, int ldo)
{
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;                                                // whole wave
    const float4 *w4 = (const float4 *)(W + (size_t)n * K);
    float acc[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[m] = 0.f;
    for (int k4 = lane; k4 < (K >> 2); k4 += 64) {
        float4 wv = w4[k4];
        if (BF16) { wv.x = sk_bf16(wv.x); wv.y = sk_bf16(wv.y); wv.z = sk_bf16(wv.z); wv.w = sk_bf16(wv.w); }
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (m < M) {
                float4 xv = *(const float4 *)(x + (size_t)m * ldx + 4 * k4);
                if (BF16) { xv.x = sk_bf16(xv.x); xv.y = sk_bf16(xv.y); xv.z = sk_bf16(xv.z); xv.w = sk_bf16(xv.w); }
                acc[m] = fmaf(wv.x, xv.x, fmaf(wv.y, xv.y, fmaf(wv.z, xv.z, fmaf(wv.w, xv.w, acc[m]))));
            }
        }
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        if (m < M) {
            float v = acc[m];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) {
                v += bias ? bias[n] : 0.f;
                if (relu) v = fmaxf(v, 0.f);
                out[(size_t)m * ldo + n] = BF16 ? sk_bf16(v) : v;
            }
        }
    }
}

template <bool BF16>
__global__ __launch_bounds__(256) void skinny_dx_kernel(const float *__restrict__ dz, int M, int N, int ldz, const float *__restrict__ W, int K,
                                                       int rows_per_chunk, float *__restrict__ dx, int ldd)
{
    const int k4 = blockIdx.x * 256 + threadIdx.x;
    if (4 * k4 >= K) return;
    const int n0 = blockIdx.y * rows_per_chunk, n1 = min(n0 + rows_per_chunk, N);
    float4 acc[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[m] = make_float4(0.f, 0.f, 0.f, 0.f);
    // eight W rows in flight per thread (one row per trip left the 100 MB expansion layer at 0.8 TB/s: 128 us per call, a quarter of it
    // the serial load, the rest the atomics of 1024 row chunks on the same M x K addresses -- the launch now cuts ~256 chunks)
    constexpr int U = 8;
    for (int n = n0; n < n1; n += U) {
        float4 wv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int nn = n + u < n1 ? n + u : n1 - 1;                // clamped: the load is unconditional, the row is masked below
            wv[u] = *(const float4 *)(W + (size_t)nn * K + 4 * k4);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (BF16) { wv[u].x = sk_bf16(wv[u].x); wv[u].y = sk_bf16(wv[u].y); wv[u].z = sk_bf16(wv[u].z); wv[u].w = sk_bf16(wv[u].w); }
            const bool ok = n + u < n1;                                // wave-uniform
            const int nn = ok ? n + u : n1 - 1;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                if (m < M) {
                    float d = ok ? dz[(size_t)m * ldz + nn] : 0.f;     // wave-uniform
                    if (BF16) d = sk_bf16(d);
                    acc[m].x = fmaf(d, wv[u].x, acc[m].x); acc[m].y = fmaf(d, wv[u].y, acc[m].y);
                    acc[m].z = fmaf(d, wv[u].z, acc[m].z); acc[m].w = fmaf(d, wv[u].w, acc[m].w);
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        if (m < M) {
            float *o = dx + (size_t)m * ldd + 4 * k4;
            atomicAdd(o, acc[m].x); atomicAdd(o + 1, acc[m].y); atomicAdd(o + 2, acc[m].z); atomicAdd(o + 3, acc[m].w);
        }
    }
}

extern "C" int pccx_linear_skinny(const float *x, int M, int K, int ldx, const float *W, const float *bias, int N, int flags, float *out,
                                  int ldo, void *stream)
{
    if (M == 0) return PCCX_OK;
    PCCX_CHECK_ARG(x && W && out, "pccx_linear_skinny: null pointer");
    PCCX_CHECK_ARG(M >= 1 && M <= 8 && K >= 4 && K % 4 == 0 && ldx % 4 == 0 && ldx >= K && N >= 1 && ldo >= N && ((uintptr_t)x & 15) == 0 &&
                       ((uintptr_t)W & 15) == 0,
                   "pccx_linear_skinny: needs 1 <= M <= 8 rows, K %% 4 == 0 and 16-byte aligned rows (M=%d K=%d ldx=%d)", M, K, ldx);
    if (flags & 2)
        hipLaunchKernelGGL(skinny_fwd_kernel<true>, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, M, K, ldx, W, bias, N, flags & 1, out, ldo);
    else
        hipLaunchKernelGGL(skinny_fwd_kernel<false>, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, M, K, ldx, W, bias, N, flags & 1, out, ldo);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// dX (M, K) += dZ (M, N) . W (N, K); dX must be zeroed by the caller (partial sums of the n-chunks are added atomically).
extern "C" int pccx_linear_skinny_dx(const float *dZ, int M, int N, int ldz, const float *W, int K, int flags, float *dX, int ldd, void *stream)
{
    if (M == 0) return PCCX_OK;
    PCCX_CHECK_ARG(dZ && W && dX, "pccx_linear_skinny_dx: null pointer");
    PCCX_CHECK_ARG(M >= 1 && M <= 8 && K >= 4 && K % 4 == 0 && N >= 1 && ldz >= N && ldd >= K && ((uintptr_t)W & 15) == 0,
                   "pccx_linear_skinny_dx: needs 1 <= M <= 8 rows and K %% 4 == 0 (M=%d K=%d)", M, K);
    const int kblocks = (K / 4 + 255) / 256;
    int chunks = 256 / kblocks;                                        // about one workgroup per CU in all: every chunk ends in M x K atomics
                                                                       // (the 100 MB expansion layer: 58 us at 256, 66 at 128, 80 at 512, 128 at the old 1024)
    if (chunks < 1) chunks = 1;
    int rpc = (N + chunks - 1) / chunks;
    if (rpc < 32) rpc = 32;
    rpc = (rpc + 7) / 8 * 8;
    chunks = (N + rpc - 1) / rpc;
    if (flags & 2)
        hipLaunchKernelGGL(skinny_dx_kernel<true>, dim3(kblocks, chunks), dim3(256), 0, (hipStream_t)stream, dZ, M, N, ldz, W, K, rpc, dX, ldd);
    else
        hipLaunchKernelGGL(skinny_dx_kernel<false>, dim3(kblocks, chunks), dim3(256), 0, (hipStream_t)stream, dZ, M, N, ldz, W, K, rpc, dX, ldd);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// ---- column reductions -----------------------------------------------------------------------------
// mode 0: out0[c] += sum z, out1[c] += sum z^2                       (BatchNorm moments)
// mode 1: out0[c] += sum dy*(y>0)*xhat, out1[c] += sum dy*(y>0)      (BatchNorm-ReLU backward: dgamma, dbeta)
// mode 2: out0[c] += sum dy                                          (bias gradient)
__global__ void col_reduce_kernel(int mode, const float *__restrict__ A, const float *__restrict__ Y, const float *__restrict__ Z,
                                  const float *__restrict__ mean, const float *__restrict__ rstd, long M, int C,
                                  double *__restrict__ out0, double *__restrict__ out1)
{
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;                                   // 4 row lanes per block
    if (c >= C) return;
    const long m0 = (long)blockIdx.y * ROWS_PER_BLOCK;
    const long m1 = m0 + ROWS_PER_BLOCK < M ? m0 + ROWS_PER_BLOCK : M;
    double s0 = 0, s1 = 0;
    const float mu = mode == 1 ? mean[c] : 0.f, rs = mode == 1 ? rstd[c] : 0.f;
    for (long m = m0 + rl; m < m1; m += 4) {
        const size_t e = (size_t)m * C + c;
        if (mode == 0) {
            const double z = A[e];
            s0 += z; s1 += z * z;
        } else if (mode == 1) {
            const float d = Y[e] > 0.f ? A[e] : 0.f;
            s0 += (double)d * (double)((Z[e] - mu) * rs);
            s1 += d;
        } else {
            s0 += A[e];
        }
    }
    atomicAdd(&out0[c], s0);
    if (mode != 2) atomicAdd(&out1[c], s1);
}

// The same reductions for the widths the models use (C = 4 * a power of two, 4..1024): 16-byte loads, a thread keeps FOUR fixed
// columns (256 % (C/4) == 0, so its column group does not change from one row step to the next), the threads that share a column
// group are added up through LDS, and one workgroup issues ONE double atomic per column and sum.  The first form sent an atomic
// per THREAD (256 per 256 rows) at the C addresses of a layer: 262 144 contended double atomics for a 131 072 x 64 activation,
// 1.3 ms per training step over its 31 calls.
template <int MODE>
__global__ __launch_bounds__(256) void col_reduce4_kernel(const float *__restrict__ A, const float *__restrict__ Y, const float *__restrict__ Z,
                                                         const float *__restrict__ mean, const float *__restrict__ rstd, long M, int C,
                                                         long rows_per_block, double *__restrict__ out0, double *__restrict__ out1, int nrep)
{
    // nrep > 1: workgroup b adds into replica b % nrep of the sums (replicas 2 C doubles apart; the consumer adds them up): the kernel
    // ends with one double atomic per column per workgroup on the same 2 C addresses, which serialise at L2 -- spread over eight
    // replicas that tail is an eighth as deep
    const size_t ro = nrep > 1 ? (size_t)(blockIdx.x % nrep) * 2 * C : 0;
    __shared__ double red[2][4][256];
    const int cv = C >> 2, tid = threadIdx.x;
    const int cq = tid % cv, rl = tid / cv, rstep = 256 / cv;          // column group, row lane, rows per step
    const long m0 = (long)blockIdx.x * rows_per_block;
    const long m1 = m0 + rows_per_block < M ? m0 + rows_per_block : M;
    double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
    float4 mu = make_float4(0, 0, 0, 0), rs = mu;
    if (MODE == 1) { mu = ((const float4 *)mean)[cq]; rs = ((const float4 *)rstd)[cq]; }
    const float muv[4] = {mu.x, mu.y, mu.z, mu.w}, rsv[4] = {rs.x, rs.y, rs.z, rs.w};
    auto add_row = [&](const float4 &a, const float4 &y, const float4 &z) {
        const float av[4] = {a.x, a.y, a.z, a.w};
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { const double zz = av[u]; s0[u] += zz; s1[u] += zz * zz; }
        } else if (MODE == 1) {
            const float yv[4] = {y.x, y.y, y.z, y.w}, zv[4] = {z.x, z.y, z.z, z.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float d = yv[u] > 0.f ? av[u] : 0.f;
                s0[u] += (double)d * (double)((zv[u] - muv[u]) * rsv[u]);
                s1[u] += d;
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) s0[u] += av[u];
        }
    };
    // FOUR row steps' loads in flight per thread (a thread's rows are still added in ascending order: the sums are bit-identical to the
    // one-row-at-a-time loop, which exposed one HBM latency per 16 bytes: 31-34 us per call, 0.84 ms of the round-3 training step)
    long m = m0 + rl;
    const float4 zero4 = make_float4(0, 0, 0, 0);
    for (; m + 3 * rstep < m1; m += 4 * rstep) {
        float4 a[4], y[4], z[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const size_t e = (size_t)(m + q * rstep) * cv + cq;
            a[q] = ((const float4 *)A)[e];
            y[q] = MODE == 1 ? ((const float4 *)Y)[e] : zero4;
            z[q] = MODE == 1 ? ((const float4 *)Z)[e] : zero4;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) add_row(a[q], y[q], z[q]);
    }
    for (; m < m1; m += rstep) {
        const size_t e = (size_t)m * cv + cq;
        add_row(((const float4 *)A)[e], MODE == 1 ? ((const float4 *)Y)[e] : zero4, MODE == 1 ? ((const float4 *)Z)[e] : zero4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { red[0][u][tid] = s0[u]; red[1][u][tid] = s1[u]; }
    __syncthreads();
    if (tid < cv) {                                                    // thread cq adds the row lanes of its column group
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            double t0 = 0, t1 = 0;
            for (int l = 0; l < rstep; ++l) { t0 += red[0][u][l * cv + tid]; t1 += red[1][u][l * cv + tid]; }
            atomicAdd(&out0[ro + 4 * tid + u], t0);
            if (MODE != 2) atomicAdd(&out1[ro + 4 * tid + u], t1);
        }
    }
}

// prezeroed: the caller cleared o0 / o1 already (train.py's step arena: ONE clear per training step instead of one per reduction)
static int launch_col_reduce(int mode, const float *A, const float *Y, const float *Z, const float *mean, const float *rstd,
                             int64_t M, int C, double *o0, double *o1, hipStream_t st, bool prezeroed = false, int nrep = 1)
{
    const int cv = C >> 2;
    if (C % 4 == 0 && cv >= 1 && cv <= 256 && (cv & (cv - 1)) == 0 && ((uintptr_t)A & 15) == 0 && (!Y || ((uintptr_t)Y & 15) == 0) &&
        (!Z || ((uintptr_t)Z & 15) == 0)) {
        // One workgroup per CU: every workgroup ends with one double atomic per column on the SAME 2 C addresses, and same-address atomics
        // serialise at L2 -- with ~1024 workgroups that tail was most of the kernel (31-35 us per call for 17-34 MB of input, round 4
        // profile); 256 workgroups with four 16-byte loads in flight per thread still cover the HBM latency.
        long rpb = (M + 255) / 256;
        const long rstep = 256 / cv;
        if (rpb < 8 * rstep) rpb = 8 * rstep;
        rpb = (rpb + rstep - 1) / rstep * rstep;
        const unsigned blocks = (unsigned)((M + rpb - 1) / rpb);
        if (prezeroed) {
        } else if (nrep > 1) {
            PCCX_CHECK_HIP(pccx_zero_async(o0, sizeof(double) * 2 * C * nrep, st));   // replicas [r][2][C], o1 == o0 + C
        } else if (o1 == o0 + C) {
            PCCX_CHECK_HIP(pccx_zero_async(o0, sizeof(double) * 2 * C, st));          // both sums in one launch
        } else {
            PCCX_CHECK_HIP(pccx_zero_async(o0, sizeof(double) * C, st));
            if (o1) PCCX_CHECK_HIP(pccx_zero_async(o1, sizeof(double) * C, st));
        }
        if (mode == 0) hipLaunchKernelGGL(col_reduce4_kernel<0>, dim3(blocks), dim3(256), 0, st, A, Y, Z, mean, rstd, (long)M, C, rpb, o0, o1, nrep);
        else if (mode == 1) hipLaunchKernelGGL(col_reduce4_kernel<1>, dim3(blocks), dim3(256), 0, st, A, Y, Z, mean, rstd, (long)M, C, rpb, o0, o1, nrep);
        else hipLaunchKernelGGL(col_reduce4_kernel<2>, dim3(blocks), dim3(256), 0, st, A, Y, Z, mean, rstd, (long)M, C, rpb, o0, o1 ? o1 : o0 + C, nrep);
        PCCX_CHECK_LAUNCH();
        return PCCX_OK;
    }
    dim3 grid((C + 63) / 64, (unsigned)((M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK));
    PCCX_CHECK_ARG(grid.y <= 65535u, "column reduction: M=%ld rows too many", (long)M);
    if (!prezeroed) {
        if (nrep > 1) {
            PCCX_CHECK_HIP(pccx_zero_async(o0, sizeof(double) * 2 * C * nrep, st));   // the generic kernel fills replica 0; the others stay zero
        } else {
            PCCX_CHECK_HIP(pccx_zero_async(o0, sizeof(double) * C, st));
            if (o1) PCCX_CHECK_HIP(pccx_zero_async(o1, sizeof(double) * C, st));
        }
    }
    hipLaunchKernelGGL(col_reduce_kernel, grid, dim3(256), 0, st, mode, A, Y, Z, mean, rstd, (long)M, C, o0, o1);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// BatchNorm (training) statistics: mean[c], rstd[c] = 1/sqrt(var_biased + eps); running stats updated as torch
// does (momentum, unbiased variance).  sums: workspace of 2*C doubles.
__global__ void bn_finalize_kernel(const double *__restrict__ s0, const double *__restrict__ s1, long M, int C, float eps,
                                   float momentum, float *__restrict__ mean, float *__restrict__ rstd,
                                   float *__restrict__ running_mean, float *__restrict__ running_var)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double mu = s0[c] / (double)M;
    double var = s1[c] / (double)M - mu * mu;
    if (var < 0) var = 0;
    mean[c] = (float)mu;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mu);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
    }
}

extern "C" int pccx_bn_train_stats(const float *Z, int64_t M, int C, float eps, float momentum, double *sums, float *mean,
                                   float *rstd, float *running_mean, float *running_var, void *stream)
{
    if (M == 0) return PCCX_OK;
    PCCX_CHECK_ARG(Z && sums && mean && rstd && C >= 1, "pccx_bn_train_stats: bad arguments");
    int rc = launch_col_reduce(0, Z, nullptr, nullptr, nullptr, nullptr, M, C, sums, sums + C, (hipStream_t)stream);
    if (rc) return rc;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, sums + C, (long)M, C, eps,
                       momentum, mean, rstd, running_mean, running_var);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// y = relu((z - mean) * rstd * gamma + beta)
__global__ void bn_relu_fwd_kernel(const float *__restrict__ Z, long n, int C, const float *__restrict__ mean,
                                   const float *__restrict__ rstd, const float *__restrict__ gamma, const float *__restrict__ beta,
                                   int relu, float *__restrict__ Y)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const float v = (Z[i] - mean[c]) * rstd[c] * gamma[c] + beta[c];
        Y[i] = relu ? fmaxf(v, 0.f) : v;
    }
}

extern "C" int pccx_bn_relu_forward(const float *Z, int64_t M, int C, const float *mean, const float *rstd, const float *gamma,
                                    const float *beta, int relu, float *Y, void *stream)
{
    if (M == 0) return PCCX_OK;
    PCCX_CHECK_ARG(Z && mean && rstd && gamma && beta && Y, "pccx_bn_relu_forward: null pointer");
    long blocks = ((long)M * C + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(bn_relu_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, Z, (long)M * C, C, mean, rstd,
                       gamma, beta, relu, Y);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// dz = gamma*rstd/M * (M*dyh - dbeta - xhat*dgamma), dyh = dy*(y>0)
__global__ void bn_relu_bwd_apply_kernel(const float *__restrict__ dY, const float *__restrict__ Y, const float *__restrict__ Z,
                                         long n, int C, long M, const float *__restrict__ mean, const float *__restrict__ rstd,
                                         const float *__restrict__ gamma, const double *__restrict__ dgamma,
                                         const double *__restrict__ dbeta, float *__restrict__ dZ)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const double d = Y[i] > 0.f ? dY[i] : 0.f;
        const double xh = (double)((Z[i] - mean[c]) * rstd[c]);
        dZ[i] = (float)((double)gamma[c] * rstd[c] / (double)M * ((double)M * d - dbeta[c] - xh * dgamma[c]));
    }
}

// dgamma/dbeta: 2*C doubles (also the parameter gradients, written as float to g_gamma / g_beta)
__global__ void cast_d2f_kernel(const double *__restrict__ a, int n, float *__restrict__ o, int accumulate)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = accumulate ? o[i] + (float)a[i] : (float)a[i];
}

// o[i] = sum over the nrep replicas (2 C doubles apart) of a[i]
__global__ void cast_rep_d2f_kernel(const double *__restrict__ a, int C, int nrep, float *__restrict__ o)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C) return;
    double t = 0;
    for (int r = 0; r < nrep; ++r) t += a[(size_t)r * 2 * C + i];
    o[i] = (float)t;
}

// the doubles of `sums` the training step's three reductions take (pccx_bn_relu_train_forward / _backward, pccx_col_sum_w)
extern "C" size_t pccx_train_sums_doubles(int C) { return (size_t)PCCX_SUM_REPLICAS * 2 * (size_t)(C > 0 ? C : 0); }

// g_gamma[i] += sums[i], g_beta[i] += sums[C + i] in one launch
__global__ void cast2_d2f_kernel(const double *__restrict__ a, int C, float *__restrict__ o0, float *__restrict__ o1)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < C) o0[i] = o0[i] + (float)a[i];
    else if (i < 2 * C) o1[i - C] = o1[i - C] + (float)a[i];
}

extern "C" int pccx_bn_relu_backward(const float *dY, const float *Y, const float *Z, int64_t M, int C, const float *mean,
                                     const float *rstd, const float *gamma, double *sums, float *dZ, float *g_gamma,
                                     float *g_beta, void *stream)
{
    if (M == 0) return PCCX_OK;
    PCCX_CHECK_ARG(dY && Y && Z && mean && rstd && gamma && sums && dZ && g_gamma && g_beta, "pccx_bn_relu_backward: null pointer");
    hipStream_t st = (hipStream_t)stream;
    int rc = launch_col_reduce(1, dY, Y, Z, mean, rstd, M, C, sums, sums + C, st);
    if (rc) return rc;
    long blocks = ((long)M * C + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(bn_relu_bwd_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, st, dY, Y, Z, (long)M * C, C, (long)M, mean,
                       rstd, gamma, sums, sums + C, dZ);
    hipLaunchKernelGGL(cast2_d2f_kernel, dim3((2 * C + 255) / 256), dim3(256), 0, st, sums, C, g_gamma, g_beta);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// ---- the training step's BatchNorm-ReLU in two launches each way (round 4) -----------------------------------------------------------
// forward: column moments (col_reduce4<0>) -> ONE kernel that turns the double sums into mean / rstd itself (every workgroup for the C
// columns, in LDS; workgroup 0 also stores them for the backward and updates the running statistics) and applies the layer: the same
// fp32 expressions as bn_finalize_kernel + bn_relu_fwd_kernel, so Y, mean, rstd and the running buffers are bit-identical to
// pccx_bn_train_stats + pccx_bn_relu_forward.  flags & 4: `sums` was cleared by the caller.
__global__ __launch_bounds__(256) void bn_relu_fwd_fused_kernel(const float *__restrict__ Z, long n, int C, long M, const double *__restrict__ s0,
                                                                const double *__restrict__ s1, float eps, float momentum,
                                                                const float *__restrict__ gamma, const float *__restrict__ beta, int relu,
                                                                float *__restrict__ Y, float *__restrict__ mean_out, float *__restrict__ rstd_out,
                                                                float *__restrict__ running_mean, float *__restrict__ running_var, int nrep)
{
    extern __shared__ float bnsm[];                                    // mean[C] | rstd[C]
    float *smean = bnsm, *srstd = bnsm + C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        double t0 = 0, t1 = 0;
        for (int r = 0; r < nrep; ++r) { t0 += s0[(size_t)r * 2 * C + c]; t1 += s1[(size_t)r * 2 * C + c]; }     // the replicas in order
        const double mu = t0 / (double)M;
        double var = t1 / (double)M - mu * mu;
        if (var < 0) var = 0;
        const float mf = (float)mu, rf = (float)(1.0 / sqrt(var + (double)eps));
        smean[c] = mf;
        srstd[c] = rf;
        if (blockIdx.x == 0) {
            mean_out[c] = mf;
            rstd_out[c] = rf;
            if (running_mean) {
                const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
                running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mu);
                running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
            }
        }
    }
    __syncthreads();
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const float v = (Z[i] - smean[c]) * srstd[c] * gamma[c] + beta[c];
        Y[i] = relu ? fmaxf(v, 0.f) : v;
    }
}

extern "C" int pccx_bn_relu_train_forward(const float *Z, int64_t M, int C, float eps, float momentum, double *sums, const float *gamma,
                                          const float *beta, int relu, float *mean, float *rstd, float *running_mean, float *running_var,
                                          float *Y, int flags, void *stream)
{
    if (M == 0) return PCCX_OK;
    PCCX_CHECK_ARG(Z && sums && gamma && beta && mean && rstd && Y && C >= 1 && C <= 4096, "pccx_bn_relu_train_forward: bad arguments (C=%d)", C);
    hipStream_t st = (hipStream_t)stream;
    if (!(flags & 8)) {                                    // flags & 8: `sums` already holds the moments (pccx_linear_moments produced Z)
        int rc = launch_col_reduce(0, Z, nullptr, nullptr, nullptr, nullptr, M, C, sums, sums + C, st, (flags & 4) != 0, PCCX_SUM_REPLICAS);
        if (rc) return rc;
    }
    long blocks = ((long)M * C + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(bn_relu_fwd_fused_kernel, dim3((unsigned)blocks), dim3(256), sizeof(float) * 2 * C, st, Z, (long)M * C, C, (long)M, sums,
                       sums + C, eps, momentum, gamma, beta, relu, Y, mean, rstd, running_mean, running_var, PCCX_SUM_REPLICAS);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// backward: dgamma / dbeta sums (col_reduce4<1>) -> the apply kernel, whose workgroup 0 also WRITES the two parameter gradients as float
// (pccx_bn_relu_backward adds them onto its zeroed outputs in a third launch: the same values).  flags & 4: `sums` cleared by the caller.
__global__ void bn_relu_bwd_apply_w_kernel(const float *__restrict__ dY, const float *__restrict__ Y, const float *__restrict__ Z,
                                           long n, int C, long M, const float *__restrict__ mean, const float *__restrict__ rstd,
                                           const float *__restrict__ gamma, const double *__restrict__ dgamma,
                                           const double *__restrict__ dbeta, float *__restrict__ dZ, float *__restrict__ g_gamma,
                                           float *__restrict__ g_beta, int nrep)
{
    extern __shared__ double bwsm[];                                   // dgamma[C] | dbeta[C]: the replicas added up, in order
    double *sg = bwsm, *sb = bwsm + C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        double t0 = 0, t1 = 0;
        for (int r = 0; r < nrep; ++r) { t0 += dgamma[(size_t)r * 2 * C + c]; t1 += dbeta[(size_t)r * 2 * C + c]; }
        sg[c] = t0;
        sb[c] = t1;
        if (blockIdx.x == 0) {
            g_gamma[c] = (float)t0;
            g_beta[c] = (float)t1;
        }
    }
    __syncthreads();
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const double d = Y[i] > 0.f ? dY[i] : 0.f;
        const double xh = (double)((Z[i] - mean[c]) * rstd[c]);
        dZ[i] = (float)((double)gamma[c] * rstd[c] / (double)M * ((double)M * d - sb[c] - xh * sg[c]));
    }
}

extern "C" int pccx_bn_relu_train_backward(const float *dY, const float *Y, const float *Z, int64_t M, int C, const float *mean,
                                           const float *rstd, const float *gamma, double *sums, float *dZ, float *g_gamma,
                                           float *g_beta, int flags, void *stream)
{
    if (M == 0) return PCCX_OK;
    PCCX_CHECK_ARG(dY && Y && Z && mean && rstd && gamma && sums && dZ && g_gamma && g_beta, "pccx_bn_relu_train_backward: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (!(flags & 8)) {                                    // flags & 8: `sums` already holds the two sums (pccx_linear_bnback produced dY)
        int rc = launch_col_reduce(1, dY, Y, Z, mean, rstd, M, C, sums, sums + C, st, (flags & 4) != 0, PCCX_SUM_REPLICAS);
        if (rc) return rc;
    }
    long blocks = ((long)M * C + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(bn_relu_bwd_apply_w_kernel, dim3((unsigned)blocks), dim3(256), sizeof(double) * 2 * C, st, dY, Y, Z, (long)M * C, C, (long)M, mean, rstd,
                       gamma, sums, sums + C, dZ, g_gamma, g_beta, PCCX_SUM_REPLICAS);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// bias gradient, written: g_bias[c] = sum_m dY[m][c]  (flags & 4: `sums` cleared by the caller)
extern "C" int pccx_col_sum_w(const float *dY, int64_t M, int C, double *sums, float *g_bias, int flags, void *stream)
{
    if (M == 0) return PCCX_OK;
    PCCX_CHECK_ARG(dY && sums && g_bias, "pccx_col_sum_w: null pointer");
    int rc = launch_col_reduce(2, dY, nullptr, nullptr, nullptr, nullptr, M, C, sums, nullptr, (hipStream_t)stream, (flags & 4) != 0, PCCX_SUM_REPLICAS);
    if (rc) return rc;
    hipLaunchKernelGGL(cast_rep_d2f_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, C, PCCX_SUM_REPLICAS, g_bias);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// clear `bytes` (a multiple of 4) on the stream with a kernel node (common.h: pccx_zero_async) -- train.py's step arena
extern "C" int pccx_zero_bytes(void *p, size_t bytes, void *stream)
{
    if (bytes == 0) return PCCX_OK;
    PCCX_CHECK_ARG(p && bytes % 4 == 0 && ((uintptr_t)p & 3) == 0, "pccx_zero_bytes: needs a 4-byte aligned buffer of a multiple of 4 bytes");
    PCCX_CHECK_HIP(pccx_zero_async(p, bytes, (hipStream_t)stream));
    return PCCX_OK;
}

// dst[0 .. bytes) = src[0 .. bytes) with a kernel (16-byte aligned buffers, a multiple of 16 bytes): the training loop hands the NEXT batch
// and its selection tables to the captured step's fixed buffers this way (train.py GraphedTrainStep(prefetch=True)), one launch in
// front of the graph replay instead of a copy node per tensor inside it
__global__ void copy16_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
extern "C" int pccx_copy_bytes(const void *src, void *dst, size_t bytes, void *stream)
{
    if (bytes == 0) return PCCX_OK;
    PCCX_CHECK_ARG(src && dst && bytes % 16 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0,
                   "pccx_copy_bytes: needs 16-byte aligned buffers of a multiple of 16 bytes");
    const size_t n16 = bytes / 16;
    size_t blocks = (n16 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(copy16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint4 *)src, (uint4 *)dst, n16);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// *table[i] += delta for n int64 counters whose addresses sit in a device table (BatchNorm's num_batches_tracked: one launch per step
// instead of one torch add per layer)
__global__ void add_i64_table_kernel(const int64_t *__restrict__ table, int n, long long delta)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) *(long long *)(uintptr_t)table[i] += delta;
}
extern "C" int pccx_add_i64_table(const int64_t *table_dev, int n, int64_t delta, void *stream)
{
    if (n == 0) return PCCX_OK;
    PCCX_CHECK_ARG(table_dev && n > 0, "pccx_add_i64_table: bad arguments");
    hipLaunchKernelGGL(add_i64_table_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, table_dev, n, (long long)delta);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// bias gradient: g_bias[c] += sum_m dY[m][c]
extern "C" int pccx_col_sum(const float *dY, int64_t M, int C, double *sums, float *g_bias, void *stream)
{
    if (M == 0) return PCCX_OK;
    PCCX_CHECK_ARG(dY && sums && g_bias, "pccx_col_sum: null pointer");
    int rc = launch_col_reduce(2, dY, nullptr, nullptr, nullptr, nullptr, M, C, sums, nullptr, (hipStream_t)stream);
    if (rc) return rc;
    hipLaunchKernelGGL(cast_d2f_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, C, g_bias, 1);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// dz = dy * (y > 0)
__global__ void relu_bwd_kernel(const float *__restrict__ dY, const float *__restrict__ Y, long n, float *__restrict__ dZ)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dZ[i] = Y[i] > 0.f ? dY[i] : 0.f;
}

extern "C" int pccx_relu_backward(const float *dY, const float *Y, int64_t n, float *dZ, void *stream)
{
    if (n == 0) return PCCX_OK;
    PCCX_CHECK_ARG(dY && Y && dZ, "pccx_relu_backward: null pointer");
    long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dY, Y, (long)n, dZ);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// ---- max over neighbours with argmax, and its backward ----------------------------------------------
__global__ void group_max_arg_kernel(const float *__restrict__ x, long G, int Kn, int C, float *__restrict__ out,
                                     int32_t *__restrict__ arg)
{
    const long total = G * C;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long gi = e / C;
        const int c = (int)(e % C);
        const float *p = x + (gi * Kn) * C + c;
        float m = -INFINITY;
        int a = 0;
        for (int k = 0; k < Kn; ++k) {
            const float v = p[(size_t)k * C];
            if (v > m) { m = v; a = k; }                               // first maximum, as torch.max
        }
        out[e] = m;
        arg[e] = a;
    }
}

__global__ void group_max_bwd_kernel(const float *__restrict__ dOut, const int32_t *__restrict__ arg, long G, int Kn, int C,
                                     float *__restrict__ dX)
{
    const long total = G * Kn * C;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        const long gk = e / C;
        const int k = (int)(gk % Kn);
        const long gi = gk / Kn;
        dX[e] = arg[gi * C + c] == k ? dOut[gi * C + c] : 0.f;
    }
}

extern "C" int pccx_group_max_arg(const float *x, int64_t G, int Kn, int C, float *out, int32_t *arg, void *stream)
{
    if (G == 0) return PCCX_OK;
    PCCX_CHECK_ARG(x && out && arg && Kn >= 1 && C >= 1, "pccx_group_max_arg: bad arguments");
    long blocks = ((long)G * C + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(group_max_arg_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (long)G, Kn, C, out, arg);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

extern "C" int pccx_group_max_backward(const float *dOut, const int32_t *arg, int64_t G, int Kn, int C, float *dX, void *stream)
{
    if (G == 0) return PCCX_OK;
    PCCX_CHECK_ARG(dOut && arg && dX && Kn >= 1 && C >= 1, "pccx_group_max_backward: bad arguments");
    long blocks = ((long)G * Kn * C + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(group_max_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dOut, arg, (long)G, Kn, C, dX);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// ---- backward of index_points / knn_gather: dF[b, idx[b,m], c0:c0+C] += dG[b, m, :]  (dG row stride ldg) ------
__global__ void scatter_add_rows_kernel(const float *__restrict__ dG, int ldg, const int64_t *__restrict__ idx, int Mrows, int N,
                                        int C, float *__restrict__ dF)
{
    const int b = blockIdx.y;
    const long total = (long)Mrows * C;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long m = e / C;
        const int c = (int)(e % C);
        long i = idx[(size_t)b * Mrows + m];
        if (i < 0) i = 0;
        atomicAdd(&dF[((size_t)b * N + (size_t)i) * C + c], dG[((size_t)b * Mrows + m) * ldg + c]);
    }
}

static int gather_backward_launch(const float *dG, int ldg, const int64_t *idx, int B, int Mrows, int N, int C, float *dF, bool prezeroed, void *stream);
extern "C" int pccx_gather_backward(const float *dG, int ldg, const int64_t *idx, int B, int Mrows, int N, int C, float *dF,
                                    void *stream)
{
    return gather_backward_launch(dG, ldg, idx, B, Mrows, N, C, dF, false, stream);
}
// the same into a dF the caller has cleared (flags & 4), i.e. without the clearing launch
extern "C" int pccx_gather_backward_acc(const float *dG, int ldg, const int64_t *idx, int B, int Mrows, int N, int C, float *dF, int flags,
                                        void *stream)
{
    return gather_backward_launch(dG, ldg, idx, B, Mrows, N, C, dF, (flags & 4) != 0, stream);
}
static int gather_backward_launch(const float *dG, int ldg, const int64_t *idx, int B, int Mrows, int N, int C, float *dF, bool prezeroed, void *stream)
{
    if (B == 0 || Mrows == 0) return PCCX_OK;
    PCCX_CHECK_ARG(dG && idx && dF && ldg >= C && B <= 65535, "pccx_gather_backward: bad arguments");
    if (!prezeroed) PCCX_CHECK_HIP(pccx_zero_async(dF, sizeof(float) * (size_t)B * N * C, (hipStream_t)stream));
    long blocks = ((long)Mrows * C + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3((unsigned)blocks, B), dim3(256), 0, (hipStream_t)stream, dG, ldg, idx, Mrows, N,
                       C, dF);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// ---- smooth L1 (beta = 1, mean reduction; pppe_pcd_ae.py:822,826): value into out[0] (double), gradient * scale ----
__global__ void smooth_l1_kernel(const float *__restrict__ a, const float *__restrict__ b, long n, float gscale,
                                 double *__restrict__ value, float *__restrict__ grad)
{
    double s = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float d = a[i] - b[i], ad = fabsf(d);
        s += ad < 1.f ? 0.5 * (double)d * d : (double)ad - 0.5;
        if (grad) grad[i] = gscale * (ad < 1.f ? d : (d > 0.f ? 1.f : -1.f));
    }
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(value, s);
}

extern "C" int pccx_smooth_l1(const float *a, const float *b, int64_t n, float grad_scale, double *value, float *grad, void *stream)
{
    PCCX_CHECK_ARG(a && b && value && n >= 1, "pccx_smooth_l1: bad arguments");
    PCCX_CHECK_HIP(pccx_zero_async(value, sizeof(double), (hipStream_t)stream));
    long blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(smooth_l1_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, b, (long)n, grad_scale, value,
                       grad);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// ---- backward of quantize_st + dequantise (pppe_pcd_ae.py:719-735,873): straight-through inside the clamps ----
__global__ void ste_bwd_kernel(const float *__restrict__ x, const float *__restrict__ dydeq, long n, float qmin, float qmax,
                               float factor, float lm1, float *__restrict__ dx)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = x[i];
        const float scaled = (fminf(fmaxf(v, qmin), qmax) - qmin) * factor;
        const bool pass = v >= qmin && v <= qmax && scaled >= 0.f && scaled <= lm1;
        dx[i] = pass ? dydeq[i] * ((qmax - qmin) / lm1) * factor : 0.f;
    }
}

extern "C" int pccx_quantize_st_backward(const float *x, const float *d_ydeq, int64_t n, float qmin, float qmax, int levels, float *dx,
                                         void *stream)
{
    if (n == 0) return PCCX_OK;
    PCCX_CHECK_ARG(x && d_ydeq && dx && levels >= 2 && qmax > qmin, "pccx_quantize_st_backward: bad arguments");
    long blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    const float factor = (float)((double)(levels - 1) / ((double)(qmax - qmin) + 1e-9));
    hipLaunchKernelGGL(ste_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, d_ydeq, (long)n, qmin, qmax, factor,
                       (float)(levels - 1), dx);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// ---- gradient clipping (clip_grad_norm_, train_pppe_pcd_ae.py:215,219) and Adam -------------------------------
__global__ void sumsq_kernel(const float *__restrict__ g, long n, double *__restrict__ acc)
{
    double s = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += (double)g[i] * g[i];
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(acc, s);
}

extern "C" int pccx_sumsq_accumulate(const float *g, int64_t n, double *acc, void *stream)
{
    if (n == 0) return PCCX_OK;
    PCCX_CHECK_ARG(g && acc, "pccx_sumsq_accumulate: null pointer");
    long blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, (long)n, acc);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// torch.optim.Adam (no weight decay, no amsgrad); the clip factor min(1, max_norm/(norm+1e-6)) is read from the
// device-side squared norm so no host sync sits inside the step.
__global__ void adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v, long n,
                            const double *__restrict__ gnorm_sq, float max_norm, float lr, float b1, float b2, float eps, float bc1,
                            float bc2)
{
    float clip = 1.f;
    if (gnorm_sq) {
        const float norm = (float)sqrt(*gnorm_sq);
        clip = fminf(1.f, max_norm / (norm + 1e-6f));
    }
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gi = g[i] * clip;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
        p[i] = p[i] - (lr / bc1) * (mi / denom);
    }
}

// The same update with lr and the two bias corrections read from DEVICE memory (hyper = {lr, 1 - beta1^t, 1 - beta2^t}): nothing
// that changes from step to step is a launch argument, so the whole training step can be captured once as a hipGraph and replayed.
__global__ void adam_dev_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v, long n,
                                const double *__restrict__ gnorm_sq, float max_norm, const float *__restrict__ hyper, float b1, float b2,
                                float eps)
{
    const float lr = hyper[0], bc1 = hyper[1], bc2 = hyper[2];
    float clip = 1.f;
    if (gnorm_sq) {
        const float norm = (float)sqrt(*gnorm_sq);
        clip = fminf(1.f, max_norm / (norm + 1e-6f));
    }
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gi = g[i] * clip;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
        p[i] = p[i] - (lr / bc1) * (mi / denom);
    }
}

// ---- clip_grad_norm_ + Adam over ALL parameter tensors in two launches ------------------------------------------------------------
// The pppe model has 49 parameter tensors; one sum-of-squares and one update launch per tensor were 98 of the step's 494 launches
// (0.75 ms of its 6.6 ms of kernel time, most tensors a few hundred bytes).  The tensors are described by a table in device memory,
// one row of six int64 per tensor: {param, grad, exp_avg, exp_avg_sq (pointers), n (elements), first block}; a workgroup of 256
// threads owns 1024 consecutive elements of one tensor and finds its row by a scan of the first-block column (T <= 1024 rows).
#define MT_ELEMS 1024
struct MtRow { float *p; const float *g; float *m; float *v; long n; long first; };

__device__ __forceinline__ int mt_find_row(const MtRow *__restrict__ tab, int T, long blk)
{
    int lo = 0, hi = T - 1;                                            // last row with first <= blk
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].first <= blk) lo = mid; else hi = mid - 1;
    }
    return lo;
}

#define MT_SUMSQ_SPAN 16                  // 1024-element blocks per workgroup: one double atomic per 16 K elements (a single address:
                                         // one per 1 K elements was 113 000 serialised atomics for the pppe model, 0.34 ms)
__global__ __launch_bounds__(256) void sumsq_multi_kernel(const MtRow *__restrict__ tab, int T, long total_blocks, double *__restrict__ acc)
{
    double s = 0;
    for (long blk = (long)blockIdx.x * MT_SUMSQ_SPAN; blk < min((long)(blockIdx.x + 1) * MT_SUMSQ_SPAN, total_blocks); ++blk) {
        const MtRow row = tab[mt_find_row(tab, T, blk)];
        const long base = (blk - row.first) * MT_ELEMS;
        for (int u = 0; u < MT_ELEMS / 256; ++u) {
            const long i = base + u * 256 + threadIdx.x;
            if (i < row.n) { const double gi = row.g[i]; s += gi * gi; }
        }
    }
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(acc, part[0] + part[1] + part[2] + part[3]);
}

__global__ __launch_bounds__(256) void adam_multi_kernel(const MtRow *__restrict__ tab, int T, const double *__restrict__ gnorm_sq, float max_norm,
                                                        const float *__restrict__ hyper, float lr, float bc1, float bc2, float b1, float b2,
                                                        float eps)
{
    if (hyper) { lr = hyper[0]; bc1 = hyper[1]; bc2 = hyper[2]; }
    float clip = 1.f;
    if (gnorm_sq) {
        const float norm = (float)sqrt(*gnorm_sq);
        clip = fminf(1.f, max_norm / (norm + 1e-6f));
    }
    const int r = mt_find_row(tab, T, blockIdx.x);
    const MtRow row = tab[r];
    const long base = ((long)blockIdx.x - row.first) * MT_ELEMS;
    for (int u = 0; u < MT_ELEMS / 256; ++u) {
        const long i = base + u * 256 + threadIdx.x;
        if (i >= row.n) break;
        const float gi = row.g[i] * clip;                              // the arithmetic of adam_kernel, element for element
        const float mi = b1 * row.m[i] + (1.f - b1) * gi;
        const float vi = b2 * row.v[i] + (1.f - b2) * gi * gi;
        row.m[i] = mi; row.v[i] = vi;
        const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
        row.p[i] = row.p[i] - (lr / bc1) * (mi / denom);
    }
}

extern "C" int pccx_sumsq_multi(const int64_t *table_dev, int ntensors, int64_t total_blocks, double *acc, void *stream)
{
    if (ntensors == 0 || total_blocks == 0) return PCCX_OK;
    PCCX_CHECK_ARG(table_dev && acc && ntensors >= 1 && total_blocks >= 1 && total_blocks <= 0x7fffffffLL, "pccx_sumsq_multi: bad arguments");
    hipLaunchKernelGGL(sumsq_multi_kernel, dim3((unsigned)((total_blocks + MT_SUMSQ_SPAN - 1) / MT_SUMSQ_SPAN)), dim3(256), 0, (hipStream_t)stream,
                       (const MtRow *)table_dev, ntensors, (long)total_blocks, acc);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

extern "C" int pccx_adam_multi(const int64_t *table_dev, int ntensors, int64_t total_blocks, const double *gnorm_sq, float max_norm,
                               const float *hyper_dev, float lr, int step, float beta1, float beta2, float eps, void *stream)
{
    if (ntensors == 0 || total_blocks == 0) return PCCX_OK;
    PCCX_CHECK_ARG(table_dev && ntensors >= 1 && total_blocks >= 1 && total_blocks <= 0x7fffffffLL && (hyper_dev || step >= 1),
                   "pccx_adam_multi: bad arguments");
    const float bc1 = hyper_dev ? 1.f : 1.f - powf(beta1, (float)step), bc2 = hyper_dev ? 1.f : 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, (const MtRow *)table_dev, ntensors,
                       gnorm_sq, max_norm, hyper_dev, lr, bc1, bc2, beta1, beta2, eps);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// The step counter and the bias corrections live on the DEVICE: one thread advances t and recomputes 1 - beta^t.  The launch sits
// inside the captured training step, so a replay needs no per-step host write at all (a pinned host buffer rewritten by a CPU that
// runs several replays ahead would be read late by the queued copies).  Layout of the 32-byte state (pccx.h):
//   float lr | float 1-b1^t | float 1-b2^t | int32 t | double b1^t | double b2^t
__global__ void adam_advance_kernel(float *__restrict__ hyper, double b1, double b2)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    int *t = (int *)(hyper + 3);
    double *pw = (double *)(hyper + 4);
    pw[0] *= b1; pw[1] *= b2;
    *t += 1;
    hyper[1] = (float)(1.0 - pw[0]);
    hyper[2] = (float)(1.0 - pw[1]);
}

extern "C" int pccx_adam_advance_dev(float *hyper, double beta1, double beta2, void *stream)
{
    PCCX_CHECK_ARG(hyper && ((uintptr_t)hyper & 7) == 0, "pccx_adam_advance_dev: hyper must be an 8-byte aligned 32-byte state");
    hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, hyper, beta1, beta2);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

extern "C" int pccx_adam_step_dev(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, const double *gnorm_sq,
                                  float max_norm, const float *hyper, float beta1, float beta2, float eps, void *stream)
{
    if (n == 0) return PCCX_OK;
    PCCX_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && hyper, "pccx_adam_step_dev: bad arguments");
    long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adam_dev_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, (long)n,
                       gnorm_sq, max_norm, hyper, beta1, beta2, eps);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

extern "C" int pccx_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, const double *gnorm_sq,
                              float max_norm, float lr, float beta1, float beta2, float eps, int step, void *stream)
{
    if (n == 0) return PCCX_OK;
    PCCX_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && step >= 1, "pccx_adam_step: bad arguments");
    long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, (long)n,
                       gnorm_sq, max_norm, lr, beta1, beta2, eps, bc1, bc2);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// bits per point of the conditional model as train_pppe_pcd_ae.py uses it (pppe_pcd_ae.py:882-917): softmax over
// the K bins of logits (B,K), probability of bin idx0[b] (first latent channel), -log2(clamp(p, 1e-9)), batch mean.
__global__ void rate_kernel(const float *__restrict__ logits, const float *__restrict__ yq0, int B, int Kb, int ldy,
                            float *__restrict__ out)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    double tot = 0;
    for (int b = 0; b < B; ++b) {
        const float *l = logits + (size_t)b * Kb;
        float mx = -INFINITY;
        for (int k = 0; k < Kb; ++k) mx = fmaxf(mx, l[k]);
        float sum = 0.f;
        for (int k = 0; k < Kb; ++k) sum += expf(l[k] - mx);
        int idx = (int)yq0[(size_t)b * ldy];
        idx = idx < 0 ? 0 : (idx > Kb - 1 ? Kb - 1 : idx);
        const float pr = fmaxf(fmaxf(expf(l[idx] - mx) / sum, 1e-9f), 1e-9f);   // softmax clamp(min=1e-9), then clamp again (:798,:912)
        tot += -log2f(pr);
    }
    out[0] = (float)(tot / B);
}

extern "C" int pccx_rate_from_logits(const float *logits, const float *y_q, int B, int bins, int ld_yq, float *out, void *stream)
{
    PCCX_CHECK_ARG(logits && y_q && out && B >= 1 && bins >= 1, "pccx_rate_from_logits: bad arguments");
    hipLaunchKernelGGL(rate_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, logits, y_q, B, bins, ld_yq, out);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}
