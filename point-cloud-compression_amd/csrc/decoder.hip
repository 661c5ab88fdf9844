// decoder.hip -- the synthesis transform of AE.AE (AE.py:48-53; decompress.py:97-116) for a batch
// of patches: inv_pool = Linear d->256->1024->k*128 (ReLU each), view (BS,128,k), concat the tiled
// latent, inv_mlp = Conv 144->128->64->32->3; then un-scale, add the patch centre, denormalize.
//
//   dec_head_kernel : the two small Linears, 16 patches per wave, output already laid out as the
//                     B-operand fragments of the big GEMM.
//   dec_main_kernel : the 1024 -> k*128 Linear (2.15 of the decoder's 2.65 GFLOP per cloud) as an
//                     fp32 MFMA GEMM whose output rows are permuted to o' = p*128 + c (pack.hip), so
//                     a workgroup's 8 m-tiles are the 128 channels of ONE point p and its
//                     accumulators feed inv_mlp directly from registers (n index = patch).  The
//                     (BS,16384) activation of the reference never exists in memory.
// MFMA-bound; weights (64 MiB) and activations stream from L2 / Infinity Cache as 1 KiB fragments.
// dec_main_kernel<true> is the bf16x3 variant (the host layer's default mode); <false> is the exact-fp32 product.
#include <math.h>
#include <stdlib.h>

#include "blobs.h"
#include "common.h"
#include "mfma_chain.h"

__global__ __launch_bounds__(256, 2) void dec_head_kernel(const float *__restrict__ latent_q, int P, int d, int ntiles,
                                                          const float *__restrict__ blob, f32x4 *__restrict__ h2p)
{
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, n = lane & 15;
    const int tile = blockIdx.x * 4 + w;
    if (tile >= ntiles) return;                                   // whole wave exits
    const int patch = tile * 16 + n;
    f32x4 in[1][1];
#pragma unroll
    for (int r = 0; r < 4; ++r) in[0][0][r] = (patch < P && 4 * g + r < d) ? latent_q[(size_t)patch * d + 4 * g + r] : 0.f;
    f32x4 a1[1][16];
#pragma unroll
    for (int mt = 0; mt < 16; ++mt) a1[0][mt] = *(const f32x4 *)(blob + DEC_H_B1 + 16 * mt + 4 * g);
    dense_acc<1, 16, 1, 16>((const f32x4 *)(blob + DEC_H_W1), lane, in, a1);
#pragma unroll
    for (int mt = 0; mt < 16; ++mt) a1[0][mt] = relu4(a1[0][mt]);
    const f32x4 *w2 = (const f32x4 *)(blob + DEC_H_W2);
#pragma unroll 1
    for (int mc = 0; mc < 16; ++mc) {
        const f32x4 *w2c = opaque_uniform(w2) + (size_t)mc * 4 * 64;       // m-tiles 4mc..4mc+3
        f32x4 a2[1][4];
#pragma unroll
        for (int m = 0; m < 4; ++m) a2[0][m] = *(const f32x4 *)(blob + DEC_H_B2 + 16 * (4 * mc + m) + 4 * g);
        dense_acc<16, 4, 1, 64>(w2c, lane, a1, a2);
#pragma unroll
        for (int m = 0; m < 4; ++m) h2p[((size_t)(4 * mc + m) * ntiles + tile) * 64 + lane] = relu4(a2[0][m]);
    }
}

// the head for decoder_h2.hip (same kernel, same fp32 result; that file adds its own operand preparation)
int pccx_dec_head_launch(const float *latent_q, int P, int d, int ntiles, const float *dec_blob, float *h2p, hipStream_t st)
{
    hipLaunchKernelGGL(dec_head_kernel, dim3((ntiles + 3) / 4), dim3(256), 0, st, latent_q, P, d, ntiles, dec_blob, (f32x4 *)h2p);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

#ifndef DEC_GROUP
#define DEC_GROUP 64                       // patch blocks per group of the block order (dec_main_kernel)
#endif


// ---- bf16x3 operands: x = hi + mid + lo exactly, each a bf16 (round to nearest even)
__device__ __forceinline__ unsigned b3_rne(float x)
{
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ void b3_split(float x, unsigned &h, unsigned &m, unsigned &l)
{
    h = b3_rne(x);
    const float r1 = __fsub_rn(x, __uint_as_float(h << 16));
    m = b3_rne(r1);
    l = b3_rne(__fsub_rn(r1, __uint_as_float(m << 16)));
}

// Pairs of fp32 fragments (k-tiles 2t and 2t+1 of the 16x16x4 chain layout) -> the three bf16 planes of one K=32 operand
// fragment of v_mfma_f32_16x16x32_bf16.  Lane-local: lane (x, kg) holds channels 16*kt + 4*kg + r of both k-tiles, which
// are k-slots 8*kg + j of the bf16 operand.  src index = outer*src_outer + ((2t + h)*W + j)*64 + lane,
// dst index = outer*dst_outer + ((t*W + j)*3 + plane)*64 + lane (units: 16-byte vectors).
__global__ __launch_bounds__(256) void b3_split_kernel(const f32x4 *__restrict__ src, uint4 *__restrict__ dst, int n_outer, int KT16, int W,
                                                       size_t src_outer, size_t dst_outer, int src_w)
{
    // src_w: fragments per k-tile row of the SOURCE array (>= W: a column range of a wider layer can be taken)
    const int lane = threadIdx.x & 63, T = (KT16 + 1) / 2;     // an odd last k-tile pairs with zeros
    const size_t item = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= (size_t)n_outer * T * W) return;
    const int j = (int)(item % W), t = (int)((item / W) % T);
    const size_t o = item / ((size_t)W * T);
    const f32x4 v0 = src[o * src_outer + ((size_t)(2 * t) * src_w + j) * 64 + lane];
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const f32x4 v1 = 2 * t + 1 < KT16 ? src[o * src_outer + ((size_t)(2 * t + 1) * src_w + j) * 64 + lane] : zero;
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) { b3_split(v0[r], h[r], m[r], l[r]); b3_split(v1[r], h[4 + r], m[4 + r], l[4 + r]); }
    uint4 *d = dst + o * dst_outer + ((size_t)t * W + j) * 3 * 64 + lane;
    d[0] = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
    d[64] = make_uint4(m[0] | (m[1] << 16), m[2] | (m[3] << 16), m[4] | (m[5] << 16), m[6] | (m[7] << 16));
    d[128] = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
}

__device__ __forceinline__ uint4 b3_load_async(const uint4 *p)    // placed exactly here; completion is covered by the ring's s_waitcnt
{
    uint4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// grid: x = patch block (8 n-tiles = 128 patches), y = point p.  4 waves, wave w owns n-tiles 2w, 2w+1.
// B3 = false: exact fp32 MFMA GEMM (the product path).  B3 = true (the bf16x3 mode): the K = 1024 GEMM runs as six
// v_mfma_f32_16x16x32_bf16 passes over pre-split operands (blob3 / h2p hold bf16 planes) and inv_mlp as a bf16x3 chain on registers.
// NT = patch tiles per wave.  2 (the default): two workgroups of four waves per CU, two waves per SIMD.  4 (bf16x3 only): ONE wave per
// SIMD on the 512-register budget, a workgroup covers 256 patches -- point p's weight stream enters LDS and is read from it half as
// often per MFMA (the untried lever of round 2's review; selected with PCCX_DEC_NT=4, measured in DESIGN.md section 4).
template <bool B3, int NT = 2>
__global__ __launch_bounds__(256, NT == 2 ? 2 : 1) void dec_main_kernel(const f32x4 *__restrict__ h2p, const float *__restrict__ latent_q,
                                                          int P, int d, int k, int ntiles, const float *__restrict__ blob,
                                                          const float *__restrict__ blob3,
                                                          float *__restrict__ patches_out, float inv_scale_div,
                                                          const float *__restrict__ centres, const float *__restrict__ nrm_center,
                                                          const float *__restrict__ nrm_longest, int S, float one_minus_margin,
                                                          float *__restrict__ pc_out)
{
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, n = lane & 15;
    // Block order: groups of DEC_GROUP patch blocks (128 patches each) outermost, then the point p, then the block inside
    // the group.  Consecutive workgroups share point p's weight stream (L2), and a group's activation fragments
    // (DEC_GROUP x 8 tiles, 32-48 MB) stay in the Infinity Cache while all k points sweep over them, instead of the whole
    // activation array being re-streamed from HBM once per point.
    constexpr int GRP = DEC_GROUP * 2 / NT;                   // patch blocks per group: the same number of PATCHES per group for any NT
    const int nblk = (ntiles + 4 * NT - 1) / (4 * NT);
    const int grp = blockIdx.x / (GRP * k), rem = blockIdx.x % (GRP * k);
    const int p = rem / GRP, blk = grp * GRP + rem % GRP;
    if (blk >= nblk) return;                                  // whole workgroup (before any barrier)
    const int tile0 = blk * 4 * NT + NT * w;
    constexpr int CH = B3 ? DEC_B3_CHUNK : DEC_WS_CHUNK;
    constexpr int NB = B3 ? 4 : 2;                                        // ring depth: B3 chunks are short, their DMA needs 3 chunks of lead
    __shared__ __attribute__((aligned(16))) f32x4 swt[NB * CH * 64];     // ring: one k-tile (8 m-tiles) per chunk; B3: 4 m-tiles x 3 planes
    const int wu = __builtin_amdgcn_readfirstlane(w);
    const WStreamT<CH, NB> ws{B3 ? blob3 + (size_t)p * DEC_B3_STREAM_CHUNKS * DEC_B3_CHUNK * 256
                             : blob + DEC_G_W(k) + (size_t)p * DEC_STREAM_CHUNKS * DEC_WS_CHUNK * 256,
                          swt, B3 ? DEC_B3_STREAM_CHUNKS : DEC_STREAM_CHUNKS, lane, wu, false};
    ws.prologue();

    f32x4 acc[NT][8];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        const f32x4 b = *(const f32x4 *)(blob + DEC_G_B + p * 128 + 16 * mt + 4 * g);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt][mt] = b;
    }
    int tq[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) tq[nt] = tile0 + nt < ntiles ? tile0 + nt : ntiles - 1;
    const int t0 = tq[0], t1 = tq[NT > 1 ? 1 : 0];
    if constexpr (!B3) {
        // ---- GEMM over K = 1024 (64 k-tiles).  A (weights of point p, shared by the 4 waves) comes through
        // the LDS ring one chunk (2 k-tiles) ahead; B (this wave's 2 patch tiles) is prefetched one k-tile
        // ahead from global.
        f32x4 b_cur[2], b_nxt[2];
        b_cur[0] = h2p[((size_t)0 * ntiles + t0) * 64 + lane];
        b_cur[1] = h2p[((size_t)0 * ntiles + t1) * 64 + lane];
#pragma unroll 1
        for (int c = 0; c < 64 / (DEC_WS_CHUNK / 8); ++c) {
            ws.boundary(c);
            const f32x4 *buf = swt + (c & 1) * DEC_WS_CHUNK * 64 + lane;
#pragma unroll
            for (int h = 0; h < DEC_WS_CHUNK / 8; ++h) {
                const int kt = (DEC_WS_CHUNK / 8) * c + h;
                const int kn = kt + 1 < 64 ? kt + 1 : 63;
                b_nxt[0] = h2p[((size_t)kn * ntiles + t0) * 64 + lane];
                b_nxt[1] = h2p[((size_t)kn * ntiles + t1) * 64 + lane];
                f32x4 a[8];
#pragma unroll
                for (int mt = 0; mt < 8; ++mt) a[mt] = buf[(h * 8 + mt) * 64];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int mt = 0; mt < 8; ++mt) {
                        acc[0][mt] = mfma16(a[mt][r], b_cur[0][r], acc[0][mt]);
                        acc[1][mt] = mfma16(a[mt][r], b_cur[1][r], acc[1][mt]);
                    }
                __builtin_amdgcn_sched_barrier(0);
                b_cur[0] = b_nxt[0]; b_cur[1] = b_nxt[1];
            }
        }
    } else {
        // ---- GEMM over K = 1024 as 32 k-steps of 32.  A planes come through the 4-deep LDS ring (chunk = 4 m-tiles x 3
        // planes, DMA three chunks ahead); the B planes of this wave's two patch tiles sit in three rotating register sets,
        // loaded two k-steps ahead by asm loads whose completion rides on the ring's waits.  VMEM issue order per wave:
        //   boundary(2t):   DMA(2t+3) [3 loads], B(t+2) [6 loads]        boundary(2t+1): DMA(2t+4) [3 loads]
        // so boundary(2t) needs all but its 12 youngest loads (DMA(2t) was issued at boundary(2t-3), B(t) at 2t-4) and
        // boundary(2t+1) all but its 18 youngest; loads complete in order.
        const uint4 *h3 = (const uint4 *)h2p;
        uint4 bs[3][NT][3];
        auto load_b = [&](uint4 (&dst)[NT][3], int t) {
            const int tc = t < 32 ? t : 31;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) dst[nt][pl] = b3_load_async(h3 + (((size_t)tc * ntiles + tq[nt]) * 3 + pl) * 64 + lane);
        };
        auto kstep = [&](int t, const uint4 (&bc)[NT][3], uint4 (&bload)[NT][3], bool first) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int c = 2 * t + half;
                if (half == 0) {
                    if (first) ws.boundary(c); else ws.template boundary_keep<6 + 3 * NT>(c);      // NT = 2: 12 youngest may fly
                    load_b(bload, t + 2);
                } else
                    ws.template boundary_keep<6 + 6 * NT>(c);                                       // NT = 2: 18
                const f32x4 *buf = ws.chunk(c);
                bf16x8 a[4][3];
#pragma unroll
                for (int mq = 0; mq < 4; ++mq)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) a[mq][pl] = __builtin_bit_cast(bf16x8, buf[(mq * 3 + pl) * 64]);
                __builtin_amdgcn_sched_barrier(0);
                // six products, smallest first: (lo,hi) (hi,lo) (mid,mid) (mid,hi) (hi,mid) (hi,hi)
                constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int q = 0; q < 6; ++q)
#pragma unroll
                    for (int mq = 0; mq < 4; ++mq)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[nt][4 * half + mq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                a[mq][PA[q]], __builtin_bit_cast(bf16x8, bc[nt][PB[q]]), acc[nt][4 * half + mq], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        load_b(bs[0], 0);
        load_b(bs[1], 1);
        kstep(0, bs[0], bs[2], true);                     // boundary(0) waits for everything issued so far
        kstep(1, bs[1], bs[0], false);
#pragma unroll 1
        for (int t = 2; t < 32; t += 3) {                 // t = 2, 5, ..., 29: three k-steps per trip, static register sets
            kstep(t, bs[2], bs[1], false);
            kstep(t + 1, bs[0], bs[2], false);
            kstep(t + 2, bs[1], bs[0], false);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the last (clamped, unused) B loads
    }
    f32x4 m3[NT][1];
    if constexpr (!B3) {
        // ---- inv_mlp on registers: channels 0..127 = relu(inv_pool.4) of point p, 128..143 = latent
        f32x4 in[2][9];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) in[nt][mt] = relu4(acc[nt][mt]);
            const int patch = (tile0 + nt) * 16 + n;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                in[nt][8][r] = (patch < P && 4 * g + r < d) ? latent_q[(size_t)patch * d + 4 * g + r] : 0.f;
        }
        f32x4 m0[2][8];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) m0[nt][mt] = *(const f32x4 *)(blob + DEC_M_B0 + 16 * mt + 4 * g);
        int f = B3 ? DEC_B3_GEMM_FRAGS : DEC_STREAM_GEMM_FRAGS;   // the inv_mlp fragments follow in the same LDS ring
        dense_acc_stream<9, 8, 2>(ws, f, in, m0);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) m0[nt][mt] = relu4(m0[nt][mt]);
        f32x4 m1[2][4];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) m1[nt][mt] = *(const f32x4 *)(blob + DEC_M_B1 + 16 * mt + 4 * g);
        dense_acc_stream<8, 4, 2>(ws, f, m0, m1);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) m1[nt][mt] = relu4(m1[nt][mt]);
        f32x4 m2[2][2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) m2[nt][mt] = *(const f32x4 *)(blob + DEC_M_B2 + 16 * mt + 4 * g);
        dense_acc_stream<4, 2, 2>(ws, f, m1, m2);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) m2[nt][mt] = relu4(m2[nt][mt]);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) m3[nt][0] = *(const f32x4 *)(blob + DEC_M_B3 + 4 * g);
        dense_acc_stream<2, 1, 2>(ws, f, m2, m3);          // last layer: no ReLU (AE.py:27)
    } else {
        // ---- inv_mlp as a bf16x3 chain: every layer's input is split in registers (relu, then three bf16 planes per pair
        // of 16-channel tiles); the ninth input tile (the latent) pairs with zeros.
        int f = DEC_B3_GEMM_FRAGS;
        bf16x8 i0[NT][5][3];
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int t = 0; t < 4; ++t) b3_split8(relu4(acc[nt][2 * t]), relu4(acc[nt][2 * t + 1]), i0[nt][t]);
            const int patch = (tile0 + nt) * 16 + n;
            f32x4 lat;
#pragma unroll
            for (int r = 0; r < 4; ++r) lat[r] = (patch < P && 4 * g + r < d) ? latent_q[(size_t)patch * d + 4 * g + r] : 0.f;
            b3_split8(lat, zero, i0[nt][4]);
        }
        f32x4 m0[NT][8];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) m0[nt][mt] = *(const f32x4 *)(blob + DEC_M_B0 + 16 * mt + 4 * g);
        dense_b3_stream<5, 8, NT>(ws, f, i0, m0);
        bf16x8 i1[NT][4][3];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int t = 0; t < 4; ++t) b3_split8(relu4(m0[nt][2 * t]), relu4(m0[nt][2 * t + 1]), i1[nt][t]);
        f32x4 m1[NT][4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) m1[nt][mt] = *(const f32x4 *)(blob + DEC_M_B1 + 16 * mt + 4 * g);
        dense_b3_stream<4, 4, NT>(ws, f, i1, m1);
        bf16x8 i2[NT][2][3];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int t = 0; t < 2; ++t) b3_split8(relu4(m1[nt][2 * t]), relu4(m1[nt][2 * t + 1]), i2[nt][t]);
        f32x4 m2[NT][2];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) m2[nt][mt] = *(const f32x4 *)(blob + DEC_M_B2 + 16 * mt + 4 * g);
        dense_b3_stream<2, 2, NT>(ws, f, i2, m2);
        bf16x8 i3[NT][1][3];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b3_split8(relu4(m2[nt][0]), relu4(m2[nt][1]), i3[nt][0]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) m3[nt][0] = *(const f32x4 *)(blob + DEC_M_B3 + 4 * g);
        dense_b3_stream<1, 1, NT>(ws, f, i3, m3);          // last layer: no ReLU (AE.py:27)
    }
    ws.drain();

    // ---- epilogue: rows 0..2 of the last tile (g == 0, r = 0..2) are x,y,z of (patch, point p)
    if (g == 0) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int patch = (tile0 + nt) * 16 + n;
            if (tile0 + nt < ntiles && patch < P) {
                float v[3] = {m3[nt][0][0], m3[nt][0][1], m3[nt][0][2]};
                if (patches_out) {
                    float *o = patches_out + ((size_t)patch * k + p) * 3;       // new_xyz.transpose(2,1)
                    o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
                }
                if (pc_out) {
                    const int b = patch / S;
                    const float lg = nrm_longest[b];
                    float *o = pc_out + ((size_t)patch * k + p) * 3;            // (B, S*k, 3): index s*k + p
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        float t = __fdiv_rn(v[a], inv_scale_div);                            // decompress.py:107
                        t = __fadd_rn(t, centres[(size_t)patch * 3 + a]);                     // decompress.py:110
                        t = __fsub_rn(t, 0.5f);                                               // pn_kit.py:63
                        t = __fdiv_rn(__fmul_rn(t, lg), one_minus_margin);                   // pn_kit.py:64
                        o[a] = __fadd_rn(t, nrm_center[3 * b + a]);                           // pn_kit.py:65
                    }
                }
            }
        }
    }
}

static unsigned dec_grid(int ntiles, int k, int NT = 2)
{
    const int grp = DEC_GROUP * 2 / NT;
    const int nblk = (ntiles + 4 * NT - 1) / (4 * NT), groups = (nblk + grp - 1) / grp;
    return (unsigned)groups * grp * (unsigned)k;
}

extern "C" size_t pccx_ae_decode_workspace_floats(int P)
{
    const size_t ntiles = ((size_t)(P > 0 ? P : 0) + 15) / 16;
    return (size_t)64 * ntiles * 64 * 4;
}

extern "C" int pccx_ae_decode(const float *latent_q, int P, int d, int k, const float *dec_blob, float *workspace,
                              float *patches_out, float scale, const float *centres, const float *nrm_center,
                              const float *nrm_longest, int S, double margin, float *pc_out, void *stream)
{
    if (P == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(latent_q && dec_blob && workspace, "pccx_ae_decode: null pointer");
    PCCX_CHECK_ARG(patches_out || pc_out, "pccx_ae_decode: need patches_out and/or pc_out");
    PCCX_CHECK_ARG(P >= 0 && d >= 1 && d <= 16 && k >= 1 && k <= 65535, "pccx_ae_decode: unsupported P=%d d=%d k=%d", P, d, k);
    PCCX_CHECK_ARG(!pc_out || (centres && nrm_center && nrm_longest && S >= 1 && scale != 0.f),
                   "pccx_ae_decode: pc_out needs centres, center, longest, S >= 1 and scale != 0");
    if (P == 0) return PCCX_OK;
    hipStream_t st = (hipStream_t)stream;
    const int ntiles = (P + 15) / 16;
    hipLaunchKernelGGL(dec_head_kernel, dim3((ntiles + 3) / 4), dim3(256), 0, st, latent_q, P, d, ntiles, dec_blob,
                       (f32x4 *)workspace);
    PCCX_CHECK_LAUNCH();
    hipLaunchKernelGGL(dec_main_kernel<false>, dim3(dec_grid(ntiles, k)), dim3(256), 0, st, (const f32x4 *)workspace, latent_q, P, d,
                       k, ntiles, dec_blob, (const float *)nullptr, patches_out, scale, centres, nrm_center, nrm_longest,
                       S > 0 ? S : 1, (float)(1.0 - margin), pc_out);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// ---- bf16x3 decoder GEMM ---------------------------------------
extern "C" size_t pccx_dec_b3_blob_floats(int k) { return DEC_B3_BLOB_FLOATS(k > 0 ? k : 0); }

extern "C" int pccx_pack_ae_decoder_b3(const float *dec_blob_dev, int k, float *b3_blob_dev, void *stream)
{
    PCCX_CHECK_ARG(dec_blob_dev && b3_blob_dev && k >= 1, "pccx_pack_ae_decoder_b3: bad argument");
    hipStream_t st = (hipStream_t)stream;
    PCCX_CHECK_HIP(hipMemsetAsync(b3_blob_dev, 0, sizeof(float) * DEC_B3_BLOB_FLOATS(k), st));
    const size_t so = (size_t)DEC_STREAM_CHUNKS * DEC_WS_CHUNK * 64, dst_o = (size_t)DEC_B3_STREAM_CHUNKS * DEC_B3_CHUNK * 64;
    const f32x4 *src = (const f32x4 *)(dec_blob_dev + DEC_G_W(k));
    // GEMM: [64 kt16][8 mt] fp32 fragments -> [32][8][3]; then the four inv_mlp layers ([kt16][mt] each, kt-major) likewise
    const int KT16[5] = {64, 9, 8, 4, 2}, MTL[5] = {8, 8, 4, 2, 1};
    size_t s_off = 0, d_off = 0;
    for (int l = 0; l < 5; ++l) {
        const int T = (KT16[l] + 1) / 2;
        const size_t items = (size_t)k * T * MTL[l];
        hipLaunchKernelGGL(b3_split_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, src + s_off * 64,
                           (uint4 *)b3_blob_dev + d_off * 64, k, KT16[l], MTL[l], so, dst_o, MTL[l]);
        PCCX_CHECK_LAUNCH();
        s_off += (size_t)KT16[l] * MTL[l];
        d_off += (size_t)T * MTL[l] * 3;
    }
    return PCCX_OK;
}

// SetAbstraction conv1 / conv2 weight planes (encoder.hip: sa_forward_kernel<true>): [1 x 4 x 3][2 x 8 x 3] fragments
extern "C" size_t pccx_sa_b3_blob_floats(void) { return (size_t)(1 * 4 * 3 + 2 * 8 * 3) * 256; }

extern "C" int pccx_pack_sa_b3(const float *enc_blob_dev, float *sa_b3_blob_dev, void *stream)
{
    PCCX_CHECK_ARG(enc_blob_dev && sa_b3_blob_dev, "pccx_pack_sa_b3: null pointer");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(b3_split_kernel, dim3((1 * 4 + 3) / 4), dim3(256), 0, st, (const f32x4 *)(enc_blob_dev + ENC_SA_W1),
                       (uint4 *)sa_b3_blob_dev, 1, 2, 4, (size_t)0, (size_t)0, 4);
    PCCX_CHECK_LAUNCH();
    hipLaunchKernelGGL(b3_split_kernel, dim3((2 * 8 + 3) / 4), dim3(256), 0, st, (const f32x4 *)(enc_blob_dev + ENC_SA_W2),
                       (uint4 *)sa_b3_blob_dev + 12 * 64, 1, 4, 8, (size_t)0, (size_t)0, 8);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// PointNet weight stream (encoder.hip: pn_forward_b3_kernel): the four layers' [kt16][mt] fp32 fragments -> [kt32][mt][plane]
extern "C" size_t pccx_pn_b3_blob_floats(void) { return PN_B3_BLOB_FLOATS; }

extern "C" int pccx_pack_pn_b3(const float *enc_blob_dev, float *pn_b3_blob_dev, void *stream)
{
    PCCX_CHECK_ARG(enc_blob_dev && pn_b3_blob_dev, "pccx_pack_pn_b3: null pointer");
    hipStream_t st = (hipStream_t)stream;
    PCCX_CHECK_HIP(hipMemsetAsync(pn_b3_blob_dev, 0, sizeof(float) * PN_B3_BLOB_FLOATS, st));
    // stream order: L0 [5][8], L1 [4][16], then for each half h of layer 2's output tiles: L2 [8][16 tiles 16h..] and the
    // eight k-steps 8h.. of layer 3 that consume them
    struct Seg { size_t src; int kt16, w, src_w; };
    const Seg segs[6] = {{ENC_PN_W0, 9, 8, 8},
                         {ENC_PN_W1, 8, 16, 16},
                         {ENC_PN_W2, 16, 16, 32},
                         {ENC_PN_W3, 16, 1, 1},
                         {ENC_PN_W2 + (size_t)16 * 256, 16, 16, 32},
                         {ENC_PN_W3 + (size_t)16 * 256, 16, 1, 1}};
    size_t d_off = 0;
    for (int l = 0; l < 6; ++l) {
        const int T = (segs[l].kt16 + 1) / 2;
        const size_t items = (size_t)T * segs[l].w;
        hipLaunchKernelGGL(b3_split_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, (const f32x4 *)(enc_blob_dev + segs[l].src),
                           (uint4 *)pn_b3_blob_dev + d_off * 64, 1, segs[l].kt16, segs[l].w, (size_t)0, (size_t)0, segs[l].src_w);
        PCCX_CHECK_LAUNCH();
        d_off += (size_t)T * segs[l].w * 3;
    }
    return PCCX_OK;
}

extern "C" size_t pccx_ae_decode_b3_workspace_floats(int P)
{
    const size_t ntiles = ((size_t)(P > 0 ? P : 0) + 15) / 16;
    return (size_t)(64 + 96) * ntiles * 64 * 4;              // fp32 fragments of dec_head + their three bf16 planes
}

extern "C" int pccx_ae_decode_b3(const float *latent_q, int P, int d, int k, const float *dec_blob, const float *b3_blob,
                                 float *workspace, float *patches_out, float scale, const float *centres, const float *nrm_center,
                                 const float *nrm_longest, int S, double margin, float *pc_out, void *stream)
{
    if (P == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(latent_q && dec_blob && b3_blob && workspace, "pccx_ae_decode_b3: null pointer");
    PCCX_CHECK_ARG(patches_out || pc_out, "pccx_ae_decode_b3: need patches_out and/or pc_out");
    PCCX_CHECK_ARG(P >= 0 && d >= 1 && d <= 16 && k >= 1 && k <= 65535, "pccx_ae_decode_b3: unsupported P=%d d=%d k=%d", P, d, k);
    PCCX_CHECK_ARG(!pc_out || (centres && nrm_center && nrm_longest && S >= 1 && scale != 0.f),
                   "pccx_ae_decode_b3: pc_out needs centres, center, longest, S >= 1 and scale != 0");
    hipStream_t st = (hipStream_t)stream;
    const int ntiles = (P + 15) / 16;
    f32x4 *h2p = (f32x4 *)workspace;
    uint4 *h3 = (uint4 *)(workspace + (size_t)64 * ntiles * 64 * 4);
    hipLaunchKernelGGL(dec_head_kernel, dim3((ntiles + 3) / 4), dim3(256), 0, st, latent_q, P, d, ntiles, dec_blob, h2p);
    PCCX_CHECK_LAUNCH();
    const size_t items = (size_t)32 * ntiles;
    hipLaunchKernelGGL(b3_split_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, (const f32x4 *)h2p, h3, 1, 64, ntiles,
                       (size_t)0, (size_t)0, ntiles);
    PCCX_CHECK_LAUNCH();
    static const int dec_nt = []() { const char *e = getenv("PCCX_DEC_NT"); return e && atoi(e) == 4 ? 4 : 2; }();
    if (dec_nt == 4)
        hipLaunchKernelGGL((dec_main_kernel<true, 4>), dim3(dec_grid(ntiles, k, 4)), dim3(256), 0, st, (const f32x4 *)h3, latent_q, P, d, k,
                           ntiles, dec_blob, b3_blob, patches_out, scale, centres, nrm_center, nrm_longest, S > 0 ? S : 1,
                           (float)(1.0 - margin), pc_out);
    else
        hipLaunchKernelGGL((dec_main_kernel<true, 2>), dim3(dec_grid(ntiles, k)), dim3(256), 0, st, (const f32x4 *)h3, latent_q, P, d, k,
                           ntiles, dec_blob, b3_blob, patches_out, scale, centres, nrm_center, nrm_longest, S > 0 ? S : 1,
                           (float)(1.0 - margin), pc_out);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}
