// encoder.hip -- the analysis transform of AE.AE (AE.py:34-45) for a batch of patches:
//   SetAbstraction (pn_kit.py:146-211; AE.py:16: kNN-16 inside the patch, centre, Conv 3->32->64->128
//   + ReLU, max over the 16 neighbours)  ->  PointNet (pn_kit.py:98-144; AE.py:17: Conv
//   131->128->256->512->d, max over the K points)  ->  sigmoid spread + round (AE.py:43-45).
//
// The reference calls ae.sa / ae.pn once per patch from Python (compress.py:113-121: 128 tiny
// launches per cloud); here one launch covers every patch of every cloud in the batch, one
// workgroup per patch.  The GEMM-shaped layers run on the fp32 matrix cores
// (v_mfma_f32_16x16x4_f32, exact fp32) as register-resident chains (mfma_chain.h).
// MFMA-bound: 5.42 GFLOP (SA) + 6.19 GFLOP (PointNet) per 8192-point cloud.
//
// sa_forward_kernel<true> and pn_forward_b3_kernel are the bf16x3 variants that form the
// same fp32 products from three bf16 pieces per operand on the bf16 matrix cores; in that mode the default encoder is the fused
// kernel of encoder_fused.hip (these two then serve ae.sa / ae.pn called separately, and K > 512).
#include <math.h>

#include "blobs.h"
#include "common.h"
#include "mfma_chain.h"

// ------------------------------------------------------------------------------------------
// SetAbstraction.  Workgroup = one patch (K points), 4 waves.
//   phase 1: thread t finds the 16 nearest points of point t inside the patch (sorted insertion
//            over an LDS broadcast of the patch; (dist, index) order as the oracle's orc_knn).
//   phase 2: wave w walks its K/4 points; the 16 lanes of a DPP row are the 16 neighbours
//            (n index of the MFMA tile), so max-pool over neighbours is a 4-step row reduction.
// Output layout feat[P][8][K][16]: 16-channel groups are contiguous per point, which is what the
// PointNet kernel loads as its B operand (one 16-byte load per lane, 1 KiB per wave).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned umed3(unsigned a, unsigned b, unsigned c)
{
    unsigned r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// B3 (the bf16x3 mode): conv1 and conv2 as fp32 products of three bf16 pieces per operand on the
// bf16 matrix cores.  blob3 = [conv1: 1 kt32 x 4 mt x 3 planes][conv2: 2 x 8 x 3] fragments (pccx_pack_sa_b3).
#define SA_W1_FRAGS(b3) ((b3) ? 1 * 4 * 3 : 2 * 4)
#define SA_W2_FRAGS(b3) ((b3) ? 2 * 8 * 3 : 4 * 8)

template <bool B3>
__global__ __launch_bounds__(256, B3 ? 2 : 3) void sa_forward_kernel(const float *__restrict__ x, int K, const float *__restrict__ blob,
                                                                  const float *__restrict__ blob3, float *__restrict__ feat)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f32x4 *sw1 = (f32x4 *)smem;                          // fp32: [2*4*64] 8 KiB;  B3: 12 KiB
    f32x4 *sw2 = sw1 + SA_W1_FRAGS(B3) * 64;             // fp32: [4*8*64] 32 KiB; B3: 48 KiB
    float *sb1 = (float *)(sw2 + SA_W2_FRAGS(B3) * 64);  // [64]
    float *sb2 = sb1 + 64;                               // [128]
    float *sx = sb2 + 128;                               // [3K]
    unsigned char *nbr = (unsigned char *)(sx + 3 * K);  // [K][16] (index < K <= 1024 needs 10 bits)
    unsigned short *nbr16 = (unsigned short *)nbr;       // stored as u16

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, n = lane & 15;
    const size_t P = blockIdx.x;
    const float *xp = x + P * (size_t)K * 3;

    {   // stage weights + patch
        const f32x4 *gw1 = B3 ? (const f32x4 *)blob3 : (const f32x4 *)(blob + ENC_SA_W1);
        const f32x4 *gw2 = B3 ? (const f32x4 *)blob3 + SA_W1_FRAGS(true) * 64 : (const f32x4 *)(blob + ENC_SA_W2);
        for (int i = tid; i < SA_W1_FRAGS(B3) * 64; i += 256) sw1[i] = gw1[i];
        for (int i = tid; i < SA_W2_FRAGS(B3) * 64; i += 256) sw2[i] = gw2[i];
        if (tid < 64) sb1[tid] = blob[ENC_SA_B1 + tid];
        if (tid < 128) sb2[tid] = blob[ENC_SA_B2 + tid];
        for (int i = tid; i < 3 * K; i += 256) sx[i] = xp[i];
    }
    __syncthreads();

    // ---- phase 1: the 16 nearest points of every point inside the patch (pn_kit.py:190, K=16; only the SET matters, a max-pool
    // follows).  Selected by patch_knn.hip (its own kernel at 8 waves per SIMD; round 2 selected here, at this kernel's 2-3 waves
    // per SIMD, where the vector-ALU-only phase cost 4 ms per 1024 clouds).  launch_sa parks the table of patch P at the head of
    // patch P's OWN slice of `feat` (K x 16 or 32 bytes of its K x 512): read it into LDS before the first feature row is written.
    {
        const unsigned char *tab = (const unsigned char *)(feat + P * (size_t)K * 128);
        if (K <= 256) {
            for (int i = tid; i < K; i += 256) {
                const uint4 v = ((const uint4 *)tab)[i];
                const unsigned b[4] = {v.x, v.y, v.z, v.w};
                unsigned wd[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    wd[2 * q] = (b[q] & 0xFFu) | ((b[q] & 0xFF00u) << 8);
                    wd[2 * q + 1] = ((b[q] >> 16) & 0xFFu) | ((b[q] >> 8) & 0xFF0000u);
                }
                ((uint4 *)nbr16)[2 * i] = make_uint4(wd[0], wd[1], wd[2], wd[3]);
                ((uint4 *)nbr16)[2 * i + 1] = make_uint4(wd[4], wd[5], wd[6], wd[7]);
            }
        } else {
            for (int i = tid; i < 2 * K; i += 256) ((uint4 *)nbr16)[i] = ((const uint4 *)tab)[i];
        }
    }
    __syncthreads();

    // conv0 (3 -> 32) also runs on the matrix core: one K=4 MFMA per 16 output channels with the bias
    // folded in as a fourth input of 1.0.  A = [w0 w1 w2 b] of channel 16*kt + (lane&15) at k = lane>>4;
    // B = coordinate (lane>>4) of neighbour (lane&15).
    const float w0a = blob[ENC_SA_W0B0 + 4 * n + g], w0b = blob[ENC_SA_W0B0 + 4 * (16 + n) + g];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // Two points per iteration (K % 16 == 0, so every wave owns an even count): the weight fragments read
    // from LDS feed both points' MFMA tiles and the two dependency chains interleave.
    const int per_wave = K / 4;
    for (int i0 = w * per_wave; i0 < (w + 1) * per_wave; i0 += 2) {
        f32x4 h0[2][2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int i = i0 + nt;
            const int j = nbr16[i * 16 + n];
            // grouped_xyz -= new_xyz (pn_kit.py:191): this lane's coordinate g of neighbour n, or the bias input
            const float rel = g < 3 ? __fsub_rn(sx[3 * j + g], sx[3 * i + g]) : 1.0f;
            h0[nt][0] = relu4(mfma16(w0a, rel, zero4));                              // relu(conv0)
            h0[nt][1] = relu4(mfma16(w0b, rel, zero4));
        }
        f32x4 a1[2][4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) a1[0][mt] = a1[1][mt] = *(const f32x4 *)(sb1 + 16 * mt + 4 * g);
        // conv2 transposed (points x channels): bias of channel 16*mt + n in every register
        // conv2 transposed (points x channels).  f32 mode: bias of channel 16*mt + n in every register (the exact fmaf chain starts
        // from it).  bf16x3 mode: accumulators start at zero and the bias is added after the neighbour max (monotone, so exact:
        // max_j (y_j + b) == (max_j y_j) + b), which saves the 16 register fills per point.
        f32x4 a2[2][8];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            const float bv = B3 ? 0.f : sb2[16 * mt + n];
            f32x4 b4 = {bv, bv, bv, bv};
            a2[0][mt] = b4; a2[1][mt] = b4;
        }
        if constexpr (!B3) {
            dense_acc<2, 4, 2, 4>(sw1, lane, h0, a1);                                // conv1
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) a1[nt][mt] = relu4(a1[nt][mt]);
            dense_acc<4, 8, 2, 8, true>(sw2, lane, a1, a2);                          // conv2
        } else {
            bf16x8 i1[2][1][3];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) b3_split8(h0[nt][0], h0[nt][1], i1[nt][0]);
            dense_b3<1, 4, 2>(sw1, lane, i1, a1);                                    // conv1
            bf16x8 i2[2][2][3];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int t = 0; t < 2; ++t) b3_split8(relu4(a1[nt][2 * t]), relu4(a1[nt][2 * t + 1]), i2[nt][t]);
            dense_b3<2, 8, 2, true>(sw2, lane, i2, a2);                              // conv2
        }
        // relu then max over the 16 neighbours (pn_kit.py:204-207) == max then relu
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            float mx[2];
            max16_of_8_transposed_tiles(a2[nt], mx);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) { // lane (row g, j = n) holds channel 16*(2g + s2) + n
                const float v = B3 ? __fadd_rn(mx[s2], sb2[16 * (2 * g + s2) + n]) : mx[s2];
                feat[((P * 8 + (2 * g + s2)) * (size_t)K + i0 + nt) * 16 + n] = fmaxf(v, 0.f);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// PointNet + quantiser.  Workgroup = one patch, 4 waves; a wave walks point tiles of 16, keeps a
// running per-channel max, and the 4 waves combine through LDS.  Layers 2 and 3 are interleaved
// two output tiles at a time so the 512-channel activation never materialises.  All four waves
// consume the same 744-fragment weight sequence per tile, so it is streamed L2 -> LDS once per
// workgroup by LDS-DMA one chunk ahead of the MFMAs (WStream, mfma_chain.h).
// ------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256, 2) void pn_forward_kernel(const float *__restrict__ x, const float *__restrict__ feat, int K,
                                                            const float *__restrict__ blob, int d, float spread,
                                                            float half_spread, float *__restrict__ latent_raw,
                                                            float *__restrict__ latent, float *__restrict__ latent_q)
{
    __shared__ __attribute__((aligned(16))) f32x4 swt[2 * WS_CHUNK * 64];      // weight ring: 2 chunks of WS_CHUNK KiB
    __shared__ float smax[4][16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, n = lane & 15;
    const size_t P = blockIdx.x;
    const float *xp = x + P * (size_t)K * 3;
    const int ntiles = K >> 4;
    const int wu = __builtin_amdgcn_readfirstlane(w);
    WStreamT<WS_CHUNK> ws{blob + ENC_PN_STREAM, swt, ENC_PN_STREAM_CHUNKS, lane, wu, true};
    ws.prologue();

    f32x4 run;                                            // running max, channel 4g+r
    run[0] = run[1] = run[2] = run[3] = -INFINITY;
    const int passes = (ntiles + 4 * NT - 1) / (4 * NT);  // identical for all waves: barriers inside
    for (int it = 0; it < passes; ++it) {
        const int t0 = (it * 4 + w) * NT;
        blob = opaque_uniform(blob);                      // keep weight / bias addressing inside the pass (no LICM)
        ws.g = blob + ENC_PN_STREAM;
        f32x4 in[NT][9];
        bool valid[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int tile = t0 + nt;
            valid[nt] = tile < ntiles;
            const int p = (valid[nt] ? tile : 0) * 16 + n;
#pragma unroll
            for (int kt = 0; kt < 8; ++kt) in[nt][kt] = *(const f32x4 *)(feat + ((P * 8 + kt) * (size_t)K + p) * 16 + 4 * g);
            f32x4 xyz;
            xyz[0] = g == 0 ? xp[3 * p] : 0.f;            // channels 128,129,130 = x,y,z (g == 0, r = 0..2)
            xyz[1] = g == 0 ? xp[3 * p + 1] : 0.f;
            xyz[2] = g == 0 ? xp[3 * p + 2] : 0.f;
            xyz[3] = 0.f;
            in[nt][8] = xyz;
        }
        int f = 0;                                        // fragment cursor of this pass (constant-folds)
        f32x4 a0[NT][8];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) a0[nt][mt] = *(const f32x4 *)(blob + ENC_PN_B0 + 16 * mt + 4 * g);
        dense_acc_stream<9, 8, NT>(ws, f, in, a0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) a0[nt][mt] = relu4(a0[nt][mt]);

        f32x4 a1[NT][16];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < 16; ++mt) a1[nt][mt] = *(const f32x4 *)(blob + ENC_PN_B1 + 16 * mt + 4 * g);
        dense_acc_stream<8, 16, NT>(ws, f, a0, a1);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < 16; ++mt) a1[nt][mt] = relu4(a1[nt][mt]);

        f32x4 a3[NT][1];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) a3[nt][0] = *(const f32x4 *)(blob + ENC_PN_B3 + 4 * g);
#pragma clang loop unroll(full)
        for (int mp = 0; mp < 16; ++mp) {                 // pairs of layer-2 output tiles
            f32x4 a2[NT][2];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int m = 0; m < 2; ++m) a2[nt][m] = *(const f32x4 *)(blob + ENC_PN_B2 + 16 * (2 * mp + m) + 4 * g);
            dense_acc_stream<16, 2, NT>(ws, f, a1, a2);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int m = 0; m < 2; ++m) a2[nt][m] = relu4(a2[nt][m]);
            dense_acc_stream<2, 1, NT>(ws, f, a2, a3);     // last layer has no ReLU (AE.py:17 relu=[T,T,T,F])
        }
        if ((ENC_PN_STREAM_FRAGS + WS_CHUNK - 1) / WS_CHUNK < ENC_PN_STREAM_CHUNKS)
            ws.boundary(ENC_PN_STREAM_CHUNKS - 1);         // padding chunk: keeps the ring parity; prefetches chunk 0
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            if (valid[nt])
#pragma unroll
                for (int r = 0; r < 4; ++r) run[r] = fmaxf(run[r], row16_max(a3[nt][0][r]));
    }
    ws.drain();                                            // no DMA may land after the workgroup retires
    if (n == 0)
#pragma unroll
        for (int r = 0; r < 4; ++r) smax[w][4 * g + r] = run[r];
    __syncthreads();
    if (tid < 16 && tid < d) {
        const float m = fmaxf(fmaxf(smax[0][tid], smax[1][tid]), fmaxf(smax[2][tid], smax[3][tid]));   // torch.max(points, 2)
        // latent = sigmoid(latent) * spread - spread / 2 ; round  (AE.py:43-45, compress.py:125-127)
        const float s = 1.0f / (1.0f + expf(-m));
        const float y = __fsub_rn(__fmul_rn(s, spread), half_spread);
        latent_raw[P * d + tid] = m;
        latent[P * d + tid] = y;
        latent_q[P * d + tid] = rintf(y);
    }
}

// ------------------------------------------------------------------------------------------
// PointNet + quantiser on bf16x3 operands.  Workgroup = one patch, EIGHT
// waves, ONE 16-point tile per wave per pass (the bf16 planes of two tiles of the 256-channel activation do not fit the
// register file), so the weight stream is shared by 128 points per pass as in the fp32 kernel.  Layer 2 runs k-outer in
// two halves of 16 output tiles (64 accumulator VGPRs): each pair of input tiles is split into its three planes once per
// half, and layer 3 consumes the half's accumulators pair by pair.  Epilogue as pn_forward_kernel.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 1) void pn_forward_b3_kernel(const float *__restrict__ x, const float *__restrict__ feat, int K,
                                                               const float *__restrict__ blob, const float *__restrict__ blob3, int d,
                                                               float spread, float half_spread, float *__restrict__ latent_raw,
                                                               float *__restrict__ latent, float *__restrict__ latent_q)
{
    __shared__ __attribute__((aligned(16))) f32x4 swt[2 * PN_B3_CHUNK * 64];   // 48 KiB weight ring
    __shared__ float smax[8][16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, n = lane & 15;
    const size_t P = blockIdx.x;
    const float *xp = x + P * (size_t)K * 3;
    const int ntiles = K >> 4;
    const int wu = __builtin_amdgcn_readfirstlane(w);
    WStreamT<PN_B3_CHUNK, 2, 8> ws{blob3, swt, PN_B3_STREAM_CHUNKS, lane, wu, true};
    ws.prologue();
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

    f32x4 run;                                            // running max, channel 4g+r
    run[0] = run[1] = run[2] = run[3] = -INFINITY;
    const int passes = (ntiles + 7) / 8;                  // identical for all waves: barriers inside
    for (int it = 0; it < passes; ++it) {
        const int tile = it * 8 + w;
        const bool valid = tile < ntiles;
        const int p = (valid ? tile : 0) * 16 + n;
        blob = opaque_uniform(blob);                      // keep bias and DMA addressing inside the pass (no LICM)
        ws.g = opaque_uniform(blob3);
        int f = 0;                                        // fragment cursor of this pass (constant-folds)
        f32x4 a0[1][8];
        {
            f32x4 in[9];
#pragma unroll
            for (int kt = 0; kt < 8; ++kt) in[kt] = *(const f32x4 *)(feat + ((P * 8 + kt) * (size_t)K + p) * 16 + 4 * g);
            in[8][0] = g == 0 ? xp[3 * p] : 0.f;          // channels 128,129,130 = x,y,z (g == 0, r = 0..2)
            in[8][1] = g == 0 ? xp[3 * p + 1] : 0.f;
            in[8][2] = g == 0 ? xp[3 * p + 2] : 0.f;
            in[8][3] = 0.f;
            bf16x8 i0[1][5][3];
#pragma unroll
            for (int t = 0; t < 4; ++t) b3_split8(in[2 * t], in[2 * t + 1], i0[0][t]);
            b3_split8(in[8], zero, i0[0][4]);
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) a0[0][mt] = *(const f32x4 *)(blob + ENC_PN_B0 + 16 * mt + 4 * g);
            dense_b3_stream<5, 8, 1>(ws, f, i0, a0);
        }
        f32x4 a1[1][16];
        {
            bf16x8 i1[1][4][3];
#pragma unroll
            for (int t = 0; t < 4; ++t) b3_split8(relu4(a0[0][2 * t]), relu4(a0[0][2 * t + 1]), i1[0][t]);
#pragma unroll
            for (int mt = 0; mt < 16; ++mt) a1[0][mt] = *(const f32x4 *)(blob + ENC_PN_B1 + 16 * mt + 4 * g);
            dense_b3_stream<4, 16, 1>(ws, f, i1, a1);
        }
        f32x4 a3[1][1];
        a3[0][0] = *(const f32x4 *)(blob + ENC_PN_B3 + 4 * g);
#pragma clang loop unroll(full)
        for (int h = 0; h < 2; ++h) {                     // layer 2 in two halves of 16 output tiles (64 accumulator VGPRs each)
            f32x4 a2[1][16];
#pragma unroll
            for (int mt = 0; mt < 16; ++mt) a2[0][mt] = *(const f32x4 *)(blob + ENC_PN_B2 + 16 * (16 * h + mt) + 4 * g);
#pragma clang loop unroll(full)
            for (int kt = 0; kt < 8; ++kt) {              // k-outer: every input pair is split once per half
                bf16x8 pl[1][1][3];
                b3_split8(relu4(a1[0][2 * kt]), relu4(a1[0][2 * kt + 1]), pl[0][0]);
                dense_b3_stream<1, 16, 1>(ws, f, pl, a2);
            }
#pragma clang loop unroll(full)
            for (int kt = 0; kt < 8; ++kt) {              // layer 3 over these 256 channels (no ReLU after it, AE.py:17)
                bf16x8 pl[1][1][3];
                b3_split8(relu4(a2[0][2 * kt]), relu4(a2[0][2 * kt + 1]), pl[0][0]);
                dense_b3_stream<1, 1, 1>(ws, f, pl, a3);
            }
        }
        if ((PN_B3_STREAM_FRAGS + PN_B3_CHUNK - 1) / PN_B3_CHUNK < PN_B3_STREAM_CHUNKS)
            ws.boundary(PN_B3_STREAM_CHUNKS - 1);          // padding chunk: keeps the ring parity; prefetches chunk 0
        if (valid)
#pragma unroll
            for (int r = 0; r < 4; ++r) run[r] = fmaxf(run[r], row16_max(a3[0][0][r]));
    }
    ws.drain();                                            // no DMA may land after the workgroup retires
    if (n == 0)
#pragma unroll
        for (int r = 0; r < 4; ++r) smax[w][4 * g + r] = run[r];
    __syncthreads();
    if (tid < 16 && tid < d) {
        float m = smax[0][tid];
#pragma unroll
        for (int k8 = 1; k8 < 8; ++k8) m = fmaxf(m, smax[k8][tid]);                                    // torch.max(points, 2)
        const float s = 1.0f / (1.0f + expf(-m));
        const float y = __fsub_rn(__fmul_rn(s, spread), half_spread);
        latent_raw[P * d + tid] = m;
        latent[P * d + tid] = y;
        latent_q[P * d + tid] = rintf(y);
    }
}

int pccx_patch_knn16_strided(const float *patches, int P, int K, void *nbr, size_t patch_stride, hipStream_t stream);   // patch_knn.hip

static int launch_sa(const float *patches, int P, int K, const float *enc_blob, const float *sa_b3_blob, float *feat, hipStream_t st)
{
    const bool b3 = sa_b3_blob != nullptr;
    // the neighbour tables, one per patch, at the head of each patch's slice of the feature map (see the kernel's phase 1)
    const int rc = pccx_patch_knn16_strided(patches, P, K, feat, (size_t)K * 128 * sizeof(float), st);
    if (rc != PCCX_OK) return rc;
    const size_t sa_lds = (size_t)(SA_W1_FRAGS(b3) + SA_W2_FRAGS(b3)) * 64 * 16 + (64 + 128) * 4 + (size_t)K * 12 + (size_t)K * 32;
    if (b3) {
        PCCX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&sa_forward_kernel<true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL(sa_forward_kernel<true>, dim3(P), dim3(256), sa_lds, st, patches, K, enc_blob, sa_b3_blob, feat);
    } else {
        PCCX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&sa_forward_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL(sa_forward_kernel<false>, dim3(P), dim3(256), sa_lds, st, patches, K, enc_blob, (const float *)nullptr, feat);
    }
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

static int launch_pn(const float *patches, const float *feat, int P, int K, const float *enc_blob, int d, int L,
                     float *latent_raw, float *latent, float *latent_q, hipStream_t st)
{
    const float spread = (float)((double)L - 0.2);
    const float half = (float)(((double)L - 0.2) / 2);
    // two point tiles per wave per pass (shared weight fragments, 140 vs 123 TFLOP/s) when the tile count
    // divides evenly over 4 waves x 2; otherwise one.
    if ((K >> 4) % 8 == 0)
        hipLaunchKernelGGL(pn_forward_kernel<2>, dim3(P), dim3(256), 0, st, patches, feat, K, enc_blob, d, spread, half,
                           latent_raw, latent, latent_q);
    else
        hipLaunchKernelGGL(pn_forward_kernel<1>, dim3(P), dim3(256), 0, st, patches, feat, K, enc_blob, d, spread, half,
                           latent_raw, latent, latent_q);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

#define CHECK_PK(fn)                                                                                              \
    PCCX_CHECK_ARG(P >= 0 && K >= 16 && K <= 1024 && K % 16 == 0, fn ": need K %% 16 == 0, 16 <= K <= 1024 (K=%d)", K)

extern "C" int pccx_sa_forward(const float *patches, int P, int K, const float *enc_blob, float *feat, void *stream)
{
    if (P == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(patches && enc_blob && feat, "pccx_sa_forward: null pointer");
    CHECK_PK("pccx_sa_forward");
    if (P == 0) return PCCX_OK;
    return launch_sa(patches, P, K, enc_blob, nullptr, feat, (hipStream_t)stream);
}

// bf16x3 mode: see sa_forward_kernel<true>
extern "C" int pccx_sa_forward_b3(const float *patches, int P, int K, const float *enc_blob, const float *sa_b3_blob, float *feat,
                                  void *stream)
{
    if (P == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(patches && enc_blob && sa_b3_blob && feat, "pccx_sa_forward_b3: null pointer");
    CHECK_PK("pccx_sa_forward_b3");
    return launch_sa(patches, P, K, enc_blob, sa_b3_blob, feat, (hipStream_t)stream);
}

extern "C" int pccx_pn_forward(const float *patches, const float *feat, int P, int K, const float *enc_blob, int d, int L,
                               float *latent_raw, float *latent, float *latent_q, void *stream)
{
    if (P == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(patches && feat && enc_blob && latent_raw && latent && latent_q, "pccx_pn_forward: null pointer");
    CHECK_PK("pccx_pn_forward");
    PCCX_CHECK_ARG(d >= 1 && d <= 16 && L >= 1, "pccx_pn_forward: unsupported d=%d L=%d", d, L);
    if (P == 0) return PCCX_OK;
    return launch_pn(patches, feat, P, K, enc_blob, d, L, latent_raw, latent, latent_q, (hipStream_t)stream);
}

// bf16x3 mode: see pn_forward_b3_kernel
extern "C" int pccx_pn_forward_b3(const float *patches, const float *feat, int P, int K, const float *enc_blob, const float *pn_b3_blob,
                                  int d, int L, float *latent_raw, float *latent, float *latent_q, void *stream)
{
    if (P == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(patches && feat && enc_blob && pn_b3_blob && latent_raw && latent && latent_q, "pccx_pn_forward_b3: null pointer");
    CHECK_PK("pccx_pn_forward_b3");
    PCCX_CHECK_ARG(d >= 1 && d <= 16 && L >= 1, "pccx_pn_forward_b3: unsupported d=%d L=%d", d, L);
    const float spread = (float)((double)L - 0.2);
    const float half = (float)(((double)L - 0.2) / 2);
    hipLaunchKernelGGL(pn_forward_b3_kernel, dim3(P), dim3(512), 0, (hipStream_t)stream, patches, feat, K, enc_blob, pn_b3_blob, d, spread,
                       half, latent_raw, latent, latent_q);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

extern "C" int pccx_ae_encode(const float *patches, int P, int K, const float *enc_blob, int d, int L, float *feat_ws,
                              float *latent_raw, float *latent, float *latent_q, void *stream)
{
    if (P == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(patches && enc_blob && feat_ws && latent_raw && latent && latent_q, "pccx_ae_encode: null pointer");
    CHECK_PK("pccx_ae_encode");
    PCCX_CHECK_ARG(d >= 1 && d <= 16 && L >= 1, "pccx_ae_encode: unsupported d=%d L=%d", d, L);
    if (P == 0) return PCCX_OK;
    int rc = launch_sa(patches, P, K, enc_blob, nullptr, feat_ws, (hipStream_t)stream);
    if (rc != PCCX_OK) return rc;
    return launch_pn(patches, feat_ws, P, K, enc_blob, d, L, latent_raw, latent, latent_q, (hipStream_t)stream);
}
