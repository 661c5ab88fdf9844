// EXPERIMENT RECORD (not built): planes_gemm_kernel generalised to 8-wave workgroups (256 rows per workgroup, the weight stream fetched
// from L2 once per 256 rows instead of once per 128; PCCX_PG_WAVES=4|8) and to a ring depth of 6 (-DPG_NB=6, DMA five chunks ahead).
// Measured with tools/experiments/pg_bench.py on the PPPF layer shapes (2048 patches), ms per layer, 4 waves / 8 waves:
//   512->1024 + max over 128: 36.3 / 37.8      256->256: 6.48 / 6.79      128->128: 4.78 / 4.93      128->256 + max: 6.35 / 7.04
// ring depth 6 against 4: the same within 0.5 % in every shape.  Neither the latency of the weight stream nor its L2 traffic is what
// holds the layer at 0.58 of the nominal bf16x3 peak: by the PMC pass of the decoder kernel built the same way the matrix pipe is busy
// 75 % of the cycles and the clock under this load is 2.0 GHz against the 2.4 GHz the peak is quoted at.  Not kept.
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
#include <stdlib.h>

#include "common.h"
#include "mfma_chain.h"

#define PG_CHUNK 12                      // ring chunk: 4 m-tiles x 3 planes (1 KiB fragments)
#ifndef PG_NB
#define PG_NB 4                          // ring depth (DMA PG_NB - 1 chunks ahead)
#endif

static inline int pg_kt32(int K) { return ((K + 15) / 16 + 1) / 2; }
static inline int pg_mb(int N) { return (N + 15) / 16 <= 4 ? 4 : 8; }

extern "C" size_t pccx_planes_floats(int64_t M, int K)
{
    const size_t ntiles = (size_t)((M > 0 ? M : 0) + 15) / 16;
    return (size_t)pg_kt32(K > 0 ? K : 1) * ntiles * 3 * 256;
}

// ---- gather + concat + split ------------------------------------------------------------------------------------
// One wave per row tile.  Row r takes source row s = idx ? (r / rows_per_batch) * n_src + max(idx[r], 0) : r; its channels are
// f0[s][0..C0) followed by f1[s][0..C1).
__global__ __launch_bounds__(256) void group_planes_kernel(const float *__restrict__ f0, int C0, int ld0, const float *__restrict__ f1,
                                                           int C1, int ld1, const int64_t *__restrict__ idx, long long M,
                                                           long long rows_per_batch, long long n_src, int KT32, long long ntiles,
                                                           uint4 *__restrict__ planes)
{
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
    const float *p0 = f0 ? f0 + (size_t)s * ld0 : nullptr;
    const float *p1 = f1 ? f1 + (size_t)s * ld1 : nullptr;
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
        bf16x8 pl[3];
        b3_split8(v[0], v[1], pl);
        uint4 *d = planes + (((size_t)t * ntiles + tile) * 3) * 64 + lane;
#pragma unroll
        for (int p = 0; p < 3; ++p) d[p * 64] = __builtin_bit_cast(uint4, pl[p]);
    }
}

extern "C" int pccx_group_planes(const float *f0, int C0, int ld0, const float *f1, int C1, int ld1, const int64_t *idx, int64_t M,
                                 int64_t rows_per_batch, int64_t n_src, float *planes, void *stream)
{
    if (M == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(planes && M > 0, "pccx_group_planes: null output or negative M");
    PCCX_CHECK_ARG(C0 >= 0 && C1 >= 0 && C0 + C1 >= 1 && (C0 == 0 || (f0 && ld0 >= C0)) && (C1 == 0 || (f1 && ld1 >= C1)),
                   "pccx_group_planes: bad sources C0=%d C1=%d", C0, C1);
    PCCX_CHECK_ARG(!idx || (rows_per_batch >= 1 && n_src >= 1), "pccx_group_planes: indices need rows_per_batch and n_src");
    const long long ntiles = (M + 15) / 16;
    PCCX_CHECK_ARG((ntiles + 3) / 4 <= 0x7fffffffLL, "pccx_group_planes: M too large");
    hipLaunchKernelGGL(group_planes_kernel, dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, (hipStream_t)stream, C0 ? f0 : nullptr, C0,
                       ld0, C1 ? f1 : nullptr, C1, ld1, idx, (long long)M, (long long)(idx ? rows_per_batch : 1),
                       (long long)(idx ? n_src : 1), pg_kt32(C0 + C1), ntiles, (uint4 *)planes);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// ---- weight stream: [m-block][t][MB m-tiles][plane] fragments out of pccx_pack_linear_b3's [t][MT][plane] ------------
extern "C" size_t pccx_planes_gemm_weight_floats(int N, int K)
{
    const int MT = ((N > 0 ? N : 1) + 15) / 16, MB = pg_mb(N), MBS = (MT + MB - 1) / MB;
    return (size_t)MBS * pg_kt32(K > 0 ? K : 1) * MB * 3 * 256;
}

__global__ void planes_weight_kernel(const uint4 *__restrict__ wpl, int KT32, int MT, int MB, int MBS, uint4 *__restrict__ ws)
{
    const size_t total = (size_t)MBS * KT32 * MB * 3 * 64;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(e & 63);
        size_t f = e >> 6;
        const int p = (int)(f % 3); f /= 3;
        const int m = (int)(f % MB); f /= MB;
        const int t = (int)(f % KT32);
        const int mb = (int)(f / KT32);
        const int mt = mb * MB + m;
        ws[e] = mt < MT ? wpl[(((size_t)t * MT + mt) * 3 + p) * 64 + lane] : make_uint4(0, 0, 0, 0);
    }
}

extern "C" int pccx_pack_planes_gemm(const float *wplanes_dev, int N, int K, float *wstream_dev, void *stream)
{
    PCCX_CHECK_ARG(wplanes_dev && wstream_dev && N >= 1 && K >= 1, "pccx_pack_planes_gemm: bad argument");
    const int MT = (N + 15) / 16, MB = pg_mb(N), MBS = (MT + MB - 1) / MB, KT32 = pg_kt32(K);
    const size_t total = (size_t)MBS * KT32 * MB * 3 * 64;
    hipLaunchKernelGGL(planes_weight_kernel, dim3((unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, (const uint4 *)wplanes_dev, KT32, MT, MB, MBS, (uint4 *)wstream_dev);
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

enum { PG_EPI_PLANES = 0, PG_EPI_ROWS = 1, PG_EPI_MAX = 2 };

template <int MB, int EPI, int NWV>
__global__ __launch_bounds__(64 * NWV, NWV == 8 ? 1 : 2) void planes_gemm_kernel(const uint4 *__restrict__ bin, long long M, long long ntiles, int KT32,
                                                             const float *__restrict__ wstream, int MBS, const float *__restrict__ bias,
                                                             int N, int relu, int group, float *__restrict__ out, int ldo)
{
    // NWV = 4: 128 rows per workgroup, ring chunk = 4 m-tiles x 3 planes, two workgroups per CU.
    // NWV = 8: 256 rows per workgroup, ring chunk = 8 m-tiles x 3 planes (one k-step), one workgroup per CU: the weight
    //          stream is fetched from L2 once per 256 rows instead of once per 128.
    static_assert(NWV == 4 || (NWV == 8 && MB == 8), "8-wave form: MB = 8 only");
    constexpr int CH = NWV == 8 ? 2 * PG_CHUNK : PG_CHUNK;    // fragments per ring chunk
    constexpr int MQ = CH / PG_CHUNK;                          // groups of 4 m-tiles per chunk
    constexpr int HALVES = MB / (4 * MQ);                      // ring chunks per k-step
    constexpr int TPB = 2 * NWV;                               // row tiles per workgroup
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, n = lane & 15;
    // Block order: XCD-aware.  Consecutive workgroup ids go round the 8 XCDs, so the MBS m-blocks of one 128-row block are
    // given to the SAME XCD one after the other: the row block's B planes are fetched into that XCD's L2 once, and every L2
    // holds the layer's whole weight stream (<= 3 MB).
    const long long nblk = (ntiles + TPB - 1) / TPB;
    const long long s = blockIdx.x >> 3;
    const long long blk = (s / MBS) * 8 + (blockIdx.x & 7);
    const int mb = (int)(s % MBS);
    if (blk >= nblk) return;                                  // whole workgroup, before any barrier
    const long long tile0 = blk * TPB + 2 * w;
    __shared__ __attribute__((aligned(16))) f32x4 swt[PG_NB * CH * 64];
    const int wu = __builtin_amdgcn_readfirstlane(w);
    const int nch = HALVES * KT32;
    const WStreamT<CH, PG_NB, NWV> ws{wstream + (size_t)mb * nch * CH * 256, swt, nch, lane, wu, false};
    // DMA of chunk c (a chunk past the end re-reads chunk 0 into a free buffer, so every boundary issues the same loads and
    // the counted waits below hold to the last k-step)
    auto dma = [&](int c) { ws.issue(c < nch ? c : 0, c % PG_NB); };
#pragma unroll
    for (int c = 0; c < PG_NB - 1; ++c) dma(c);

    f32x4 acc[2][MB];
#pragma unroll
    for (int mt = 0; mt < MB; ++mt) {
        f32x4 b;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 16 * (mb * MB + mt) + 4 * g + r;
            b[r] = (bias && c < N) ? bias[c] : 0.f;
        }
        acc[0][mt] = b; acc[1][mt] = b;
    }
    const long long t0 = tile0 < ntiles ? tile0 : ntiles - 1, t1 = tile0 + 1 < ntiles ? tile0 + 1 : ntiles - 1;
    {
        // VMEM issue order per wave and k-step t:  HALVES = 2:  boundary(2t): DMA(2t+3) [3], B(t+2) [6];  boundary(2t+1): DMA(2t+4) [3]
        //                                          HALVES = 1:  boundary(t):  DMA(t+3) [3],  B(t+2) [6]
        // loads complete in order, so boundary(c) may leave in flight everything issued after the youngest load it needs.
        uint4 bs[3][2][3];
        auto load_b = [&](uint4 (&dst)[2][3], int t) {
            const int tc = t < KT32 ? t : KT32 - 1;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    dst[nt][pl] = pg_load_async(bin + (((size_t)tc * ntiles + (nt ? t1 : t0)) * 3 + pl) * 64 + lane);
        };
        auto kstep = [&](int t, const uint4 (&bc)[2][3], uint4 (&bload)[2][3], bool first) {
#pragma unroll
            for (int half = 0; half < HALVES; ++half) {
                const int c = HALVES * t + half;
                if (half == 0) {
                    if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else if (HALVES == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PG_NB == 4 ? 12 : 15) : "memory");
                    else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
                    __syncthreads();
                    dma(c + PG_NB - 1);
                    load_b(bload, t + 2);
                } else {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PG_NB == 4 ? 18 : 30) : "memory");
                    __syncthreads();
                    dma(c + PG_NB - 1);
                }
                const f32x4 *buf = ws.chunk(c);
                // six products, smallest first: (lo,hi) (hi,lo) (mid,mid) (mid,hi) (hi,mid) (hi,hi)
                constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int mqq = 0; mqq < MQ; ++mqq) {
                    bf16x8 a[4][3];
#pragma unroll
                    for (int mq = 0; mq < 4; ++mq)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) a[mq][pl] = __builtin_bit_cast(bf16x8, buf[((mqq * 4 + mq) * 3 + pl) * 64]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < 6; ++q)
#pragma unroll
                        for (int mq = 0; mq < 4; ++mq)
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt)
                                acc[nt][4 * (half * MQ + mqq) + mq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                    a[mq][PA[q]], __builtin_bit_cast(bf16x8, bc[nt][PB[q]]), acc[nt][4 * (half * MQ + mqq) + mq], 0, 0, 0);
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
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the last (clamped, unused) B loads and DMAs
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
                bf16x8 pl[3];
                if (relu) b3_split8(relu4(acc[nt][2 * j]), relu4(acc[nt][2 * j + 1]), pl);
                else b3_split8(acc[nt][2 * j], acc[nt][2 * j + 1], pl);
                uint4 *d = o + (((size_t)to * ntiles + tile0 + nt) * 3) * 64 + lane;
#pragma unroll
                for (int p = 0; p < 3; ++p) d[p * 64] = __builtin_bit_cast(uint4, pl[p]);
            }
        }
    } else if constexpr (EPI == PG_EPI_ROWS) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const long long row = (tile0 + nt) * 16 + n;
            if (tile0 + nt >= ntiles || row >= M) continue;
#pragma unroll
            for (int mt = 0; mt < MB; ++mt) {
                const int c = 16 * (mb * MB + mt) + 4 * g;
                f32x4 v = relu ? relu4(acc[nt][mt]) : acc[nt][mt];
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
    } else {
        // max over groups of `group` rows (32, 64 or 128; M is a multiple of it, so no group holds padded rows).  In the wave:
        // the two tiles elementwise, then the 16 rows of the tile by DPP; across the waves of a group through LDS.
        __syncthreads();                                   // every wave is done with the ring
        float *smax = (float *)swt;                        // [NWV waves][16 * MB channels]
#pragma unroll
        for (int mt = 0; mt < MB; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = fmaxf(acc[0][mt][r], acc[1][mt][r]);
                v = row16_max(v);
                if (n == 0) smax[w * (16 * MB) + 16 * mt + 4 * g + r] = v;
            }
        __syncthreads();
        const int gpb = 32 * NWV / group, wpg = group / 32; // groups per block, waves per group
        const long long G = M / group;
        for (int e = tid; e < gpb * 16 * MB; e += 64 * NWV) {
            const int gi = e / (16 * MB), c = e % (16 * MB);
            float v = smax[(gi * wpg) * (16 * MB) + c];
            for (int q = 1; q < wpg; ++q) v = fmaxf(v, smax[(gi * wpg + q) * (16 * MB) + c]);
            if (relu) v = fmaxf(v, 0.f);                   // max(relu(x)) = relu(max(x))
            const long long grp = blk * gpb + gi;
            const int ch = mb * 16 * MB + c;
            if (grp < G && ch < N) out[(size_t)grp * ldo + ch] = v;
        }
    }
}

// out: epilogue 0 -> planes of the N output channels (pccx_planes_floats(M, N) floats); 1 -> fp32 rows (M, ldo);
// 2 -> fp32 (M / group, ldo), the max over each `group` consecutive rows (group in {32, 64, 128}, M % group == 0).
extern "C" int pccx_planes_gemm(const float *planes_in, int64_t M, int K, const float *wstream, const float *bias, int N, int relu,
                                int epilogue, int group, float *out, int ldo, void *stream)
{
    if (M == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(planes_in && wstream && out, "pccx_planes_gemm: null pointer");
    PCCX_CHECK_ARG(M > 0 && K >= 1 && N >= 1, "pccx_planes_gemm: bad shape M=%lld K=%d N=%d", (long long)M, K, N);
    PCCX_CHECK_ARG(epilogue >= 0 && epilogue <= 2, "pccx_planes_gemm: epilogue %d", epilogue);
    PCCX_CHECK_ARG(epilogue == PG_EPI_PLANES || ldo >= N, "pccx_planes_gemm: ldo=%d < N=%d", ldo, N);
    PCCX_CHECK_ARG(epilogue != PG_EPI_MAX || ((group == 32 || group == 64 || group == 128) && M % group == 0),
                   "pccx_planes_gemm: group max needs group in {32,64,128} dividing M (group=%d M=%lld)", group, (long long)M);
    const long long ntiles = (M + 15) / 16;
    const int MT = (N + 15) / 16, MB = pg_mb(N), MBS = (MT + MB - 1) / MB, KT32 = pg_kt32(K);
    // 8-wave workgroups (256 rows) when the layer is wide and long enough to be bound by the operand traffic from L2
    static int force = -1;
    if (force < 0) { const char *e = getenv("PCCX_PG_WAVES"); force = e ? atoi(e) : 0; }
    const int nwv = force ? force : ((MB == 8 && K >= 64 && M >= 65536) ? 8 : 4);
    const long long nblk = (ntiles + 2 * nwv - 1) / (2 * nwv);
    const long long blocks = (nblk + 7) / 8 * 8 * MBS;
    PCCX_CHECK_ARG(blocks <= 0x7fffffffLL, "pccx_planes_gemm: M=%lld too large", (long long)M);
    PCCX_CHECK_ARG(nwv == 4 || (nwv == 8 && MB == 8), "pccx_planes_gemm: PCCX_PG_WAVES=%d does not fit this layer", nwv);
    hipStream_t st = (hipStream_t)stream;
    relu &= 1;
#define PG_LAUNCH(MB_, E_, W_)                                                                                                  \
    hipLaunchKernelGGL((planes_gemm_kernel<MB_, E_, W_>), dim3((unsigned)blocks), dim3(64 * W_), 0, st, (const uint4 *)planes_in,     \
                       (long long)M, ntiles, KT32, wstream, MBS, bias, N, relu, group, out, ldo)
    if (MB == 8 && nwv == 8) {
        if (epilogue == PG_EPI_PLANES) PG_LAUNCH(8, PG_EPI_PLANES, 8);
        else if (epilogue == PG_EPI_ROWS) PG_LAUNCH(8, PG_EPI_ROWS, 8);
        else PG_LAUNCH(8, PG_EPI_MAX, 8);
    } else if (MB == 8) {
        if (epilogue == PG_EPI_PLANES) PG_LAUNCH(8, PG_EPI_PLANES, 4);
        else if (epilogue == PG_EPI_ROWS) PG_LAUNCH(8, PG_EPI_ROWS, 4);
        else PG_LAUNCH(8, PG_EPI_MAX, 4);
    } else {
        if (epilogue == PG_EPI_PLANES) PG_LAUNCH(4, PG_EPI_PLANES, 4);
        else if (epilogue == PG_EPI_ROWS) PG_LAUNCH(4, PG_EPI_ROWS, 4);
        else PG_LAUNCH(4, PG_EPI_MAX, 4);
    }
#undef PG_LAUNCH
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}
