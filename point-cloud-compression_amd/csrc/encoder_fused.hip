// encoder_fused.hip -- the whole analysis transform of AE.AE (AE.py:34-45) in ONE kernel, bf16x3 arithmetic:
//   SetAbstraction (pn_kit.py:146-211) -> PointNet (pn_kit.py:98-144) -> sigmoid spread + round (AE.py:43-45)
// without the (P,128,K) feature map ever leaving the CU.  encoder.hip runs the two modules as two kernels with the map in
// HBM between them (8.4 MB per cloud written and read back: 18.6 GB per 1024 clouds against 0.2 GB of patches and weights);
// here a wave's SetAbstraction output for 16 points IS its PointNet input tile and is handed over through 8 KiB of LDS:
//
//   workgroup = one patch, eight waves.  Per pass over 128 points (wave w owns points 16(8 it + w) ..):
//     SA   : the wave's 16 points, two at a time (the 16 lanes of a DPP row are the 16 neighbours), exactly the arithmetic of
//            sa_forward_kernel<true>; the 128 channel maxima of each point go to the wave's own staging rows in LDS
//            ([point][132] floats: the pad makes the transposed read-back conflict-free);
//     hand : the wave reads the rows back as PointNet's B operand (channel 16 kt + 4 g + r of point n in lane (g, n)) and
//            splits them into bf16 planes -- no barrier, the rows are private to the wave;
//     PN   : after one barrier (the weight ring shares LDS with the staging rows) the pass of pn_forward_b3_kernel: the
//            1128-fragment weight stream through the LDS-DMA ring, layers 2 and 3 interleaved, running channel maximum.
//   The ring starts cold in every pass (it may not prefetch into the staging rows): two exposed fills per 256-point patch.
//
// Arithmetic, weight blobs and results are those of pccx_sa_forward_b3 + pccx_pn_forward_b3 (same products, same order), so the
// parity tests of the two-kernel path apply unchanged; tests/test_gpu_model.py also compares the two paths bit for bit.
#include <math.h>

#include "blobs.h"
#include "common.h"
#include "mfma_chain.h"

#ifndef FU_CHUNK
#define FU_CHUNK 32                                      // fragments per ring chunk of the PointNet weight stream: 2 x 32 KiB fit under the
#endif                                                   // staging rows; same speed as 24 (measured), and the kernel builds without scratch
static_assert(((PN_B3_STREAM_FRAGS + FU_CHUNK - 1) / FU_CHUNK) * FU_CHUNK <= PN_B3_STREAM_CHUNKS * PN_B3_CHUNK, "the padded stream must cover the last chunk");
#ifndef FU_NB
#define FU_NB 2                                          // ring buffers (chunks in flight + the one being read)
#endif
#ifdef FU_COUNTED_WAITS
#define FU_DENSE dense_b3_stream_pw                      // ring reads as inline assembly with counted lgkmcnt waits (mfma_chain.h): measured
#else                                                    // equal to the compiler's full waits (48.29 / 48.59 against 48.44 / 48.62 ms): with
#define FU_DENSE dense_b3_stream                         // two waves per SIMD the partner covers the exposed LDS latency.  Default: the compiler's
#endif
#define FU_STAGE_STRIDE 132                               // floats per staged point: 128 channels + 4 (bank rotation)
#define FU_STAGE_WAVE (16 * FU_STAGE_STRIDE)              // floats per wave
#define FU_W1_FRAGS (1 * 4 * 3)
#define FU_W2_FRAGS (2 * 8 * 3)

__device__ __forceinline__ unsigned fu_umed3(unsigned a, unsigned b, unsigned c)
{
    unsigned r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// LDS map (bytes): [sw1 12 KiB][sw2 48 KiB][sb1 256][sb2 512][sx 12K][nbr 32K][region: max(ring 48 KiB, 8 staging blocks)]
__host__ __device__ inline size_t fu_region_bytes()
{
    const size_t ring = (size_t)FU_NB * FU_CHUNK * 1024, stage = (size_t)8 * FU_STAGE_WAVE * 4;
    return ring > stage ? ring : stage;
}
__host__ __device__ inline size_t fu_lds_bytes(int K)
{
    return (size_t)(FU_W1_FRAGS + FU_W2_FRAGS) * 1024 + (64 + 128) * 4 + (size_t)K * 12 + (size_t)K * 32 + fu_region_bytes() + 8 * 16 * 4 + 32;
}

// EXT_NBR: the neighbour table of every patch comes from patch_knn.hip (pccx_patch_knn16, one or two bytes per index) instead of
// being selected here at two waves per SIMD -- the default since round 3; the in-kernel selection stays for the workspace-free
// entry point pccx_ae_encode_b3.
template <bool EXT_NBR>
__global__ __launch_bounds__(512, 1) void sa_pn_forward_b3_kernel(const float *__restrict__ x, int npatches, int K, const float *__restrict__ blob,
                                                                  const float *__restrict__ sa3, const float *__restrict__ pn3, int d,
                                                                  float spread, float half_spread, float *__restrict__ latent_raw,
                                                                  float *__restrict__ latent, float *__restrict__ latent_q,
                                                                  const unsigned char *__restrict__ nbr_tab)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f32x4 *sw1 = (f32x4 *)smem;
    f32x4 *sw2 = sw1 + FU_W1_FRAGS * 64;
    float *sb1 = (float *)(sw2 + FU_W2_FRAGS * 64);
    float *sb2 = sb1 + 64;
    float *sx = sb2 + 128;
    unsigned short *nbr16 = (unsigned short *)(sx + 3 * K);
    unsigned char *region = (unsigned char *)(nbr16 + 16 * K);          // 16-byte aligned: K % 16 == 0
    f32x4 *swt = (f32x4 *)region;                                       // PointNet weight ring (2 x 24 KiB)
    float *stage_all = (float *)region;                                 // ... or the eight staging blocks
    float (*smax)[16] = (float (*)[16])(region + fu_region_bytes());
    int *sa_next = (int *)(region + fu_region_bytes() + 8 * 16 * 4);     // [8]: next SetAbstraction unit of pass it

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int ntiles = K >> 4;
    const int wu = __builtin_amdgcn_readfirstlane(w);

    {   // stage the SetAbstraction weights ONCE per workgroup: a workgroup walks patches blockIdx.x, + gridDim.x, ...
        const f32x4 *gw1 = (const f32x4 *)sa3, *gw2 = (const f32x4 *)sa3 + FU_W1_FRAGS * 64;
        for (int i = tid; i < FU_W1_FRAGS * 64; i += 512) sw1[i] = gw1[i];
        for (int i = tid; i < FU_W2_FRAGS * 64; i += 512) sw2[i] = gw2[i];
        if (tid < 64) sb1[tid] = blob[ENC_SA_B1 + tid];
        if (tid < 128) sb2[tid] = blob[ENC_SA_B2 + tid];
    }
  for (size_t P = blockIdx.x; P < (size_t)npatches; P += gridDim.x) {
    const float *xp = x + P * (size_t)K * 3;
    for (int i = tid; i < 3 * K; i += 512) sx[i] = xp[i];
    if (tid < 8) sa_next[tid] = 0;
    if (EXT_NBR) {                                         // the patch's neighbour table, widened to the u16 rows the units read
        if (K <= 256) {
            const uint4 *tab = (const uint4 *)nbr_tab + P * (size_t)K;
            for (int i = tid; i < K; i += 512) {
                const uint4 v = tab[i];
                const unsigned b[4] = {v.x, v.y, v.z, v.w};
                unsigned w[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    w[2 * q] = (b[q] & 0xFFu) | ((b[q] & 0xFF00u) << 8);
                    w[2 * q + 1] = ((b[q] >> 16) & 0xFFu) | ((b[q] >> 8) & 0xFF0000u);
                }
                ((uint4 *)nbr16)[2 * i] = make_uint4(w[0], w[1], w[2], w[3]);
                ((uint4 *)nbr16)[2 * i + 1] = make_uint4(w[4], w[5], w[6], w[7]);
            }
        } else {
            const uint4 *tab = (const uint4 *)nbr_tab + P * (size_t)K * 2;
            for (int i = tid; i < 2 * K; i += 512) ((uint4 *)nbr16)[i] = tab[i];
        }
    }
    __syncthreads();

    // ---- kNN-16 inside the patch (pn_kit.py:190), the selection of sa_forward_kernel with TWO threads per point: thread t and
    // t + 256 each keep the 17 smallest packed keys of one half of the candidates (key = distance bits with the candidate index in
    // the low log2(K) bits, one v_med3_u32 per slot per candidate); the upper thread hands its 17 keys over through LDS (the
    // region is idle until the first SetAbstraction phase) and the lower one inserts them: the 17 smallest of the union are
    // among the two lists.  A tie or near-tie at the 16th / 17th rank takes the exact (distance, index) selection over all
    // candidates, as before.
    unsigned jmask = 15u;
    while ((int)jmask < K - 1) jmask = 2u * jmask + 1u;
    unsigned *kmerge = (unsigned *)region;                 // [256][17]
    const int khalf = tid >> 8, kslot = tid & 255;
    for (int ib = 0; ib < (EXT_NBR ? 0 : K); ib += 256) {
        const int i = ib + kslot;
        const bool act = i < K;
        unsigned tk[17];
#pragma unroll
        for (int s = 0; s < 17; ++s) tk[s] = 0xFFFFFFFFu;
        float px = 0.f, py = 0.f, pz = 0.f;
        if (act) {
            px = sx[3 * i]; py = sx[3 * i + 1]; pz = sx[3 * i + 2];
            const int jb = khalf * (K >> 1), je = jb + (K >> 1);          // K % 16 == 0: halves are multiples of 8
            for (int j0 = jb; j0 < je; j0 += 4) {
                float dd[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    dd[u] = pccx_sqdist(px, py, pz, sx[3 * (j0 + u)], sx[3 * (j0 + u) + 1], sx[3 * (j0 + u) + 2]);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned key = (__float_as_uint(dd[u]) & ~jmask) | (unsigned)(j0 + u);
#pragma unroll
                    for (int s = 16; s >= 1; --s) tk[s] = fu_umed3(tk[s - 1], key, tk[s]);
                    tk[0] = min(tk[0], key);
                }
            }
            if (khalf == 1) {
#pragma unroll
                for (int s = 0; s < 17; ++s) kmerge[s * 256 + kslot] = tk[s];
            }
        }
        __syncthreads();
        if (act && khalf == 0) {
#pragma unroll
            for (int q = 0; q < 17; ++q) {
                const unsigned key = kmerge[q * 256 + kslot];
#pragma unroll
                for (int s = 16; s >= 1; --s) tk[s] = fu_umed3(tk[s - 1], key, tk[s]);
                tk[0] = min(tk[0], key);
            }
            if (((tk[15] ^ tk[16]) & ~jmask) != 0u) {
#pragma unroll
                for (int s = 0; s < 16; ++s) nbr16[i * 16 + s] = (unsigned short)(tk[s] & jmask);
            } else {
                float td[16];                            // tie or near-tie at the boundary: the exact (distance, index) rule
#pragma unroll
                for (int s = 0; s < 16; ++s) td[s] = INFINITY;
                for (int j = 0; j < K; ++j) {
                    const float dj = pccx_sqdist(px, py, pz, sx[3 * j], sx[3 * j + 1], sx[3 * j + 2]);
#pragma unroll
                    for (int s = 15; s >= 1; --s) td[s] = __builtin_amdgcn_fmed3f(td[s - 1], dj, td[s]);
                    td[0] = fminf(td[0], dj);
                }
                const float T = td[15];
                int need = 16;
#pragma unroll
                for (int s = 0; s < 16; ++s) need -= td[s] < T ? 1 : 0;
                int c = 0, ties = 0;
                for (int j = 0; j < K; ++j) {
                    const float dj = pccx_sqdist(px, py, pz, sx[3 * j], sx[3 * j + 1], sx[3 * j + 2]);
                    const bool tie = dj == T;
                    if (dj < T || (tie && ties < need)) {
                        if (c < 16) nbr16[i * 16 + c] = (unsigned short)j;
                        ++c;
                    }
                    ties += tie ? 1 : 0;
                }
            }
        }
        __syncthreads();
    }

    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    if (lane < 16) smax[wu][lane] = -INFINITY;            // running channel maximum of this wave, kept in LDS between passes
    const int lane0 = lane;

    const int passes = (ntiles + 7) / 8;                  // identical for all waves: barriers inside
    for (int it = 0; it < passes; ++it) {
        const int tile = it * 8 + wu;
        const bool valid = tile < ntiles;
        const int p0 = (valid ? tile : 0) * 16;           // an idle wave recomputes tile 0 and discards it
        // Each phase derives its lane indices from a freshly "laundered" lane id: otherwise the compiler computes every
        // lane-dependent address of BOTH phases once, ahead of the pass loop, and carries them (in scratch) through the other
        // phase: 57 spilled VGPRs, 94 KB of scratch writes per patch = 6 GB of HBM traffic per 1024 clouds.
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        int g = lane >> 4, n = lane & 15;
        const float w0a = blob[ENC_SA_W0B0 + 4 * n + g], w0b = blob[ENC_SA_W0B0 + 4 * (16 + n) + g];

        // ---- SetAbstraction for the pass's points, two per unit (sa_forward_kernel<true>'s body).  The units are NOT tied to the
        // wave that owns the points in the PointNet pass: every wave takes the next unit from a counter in LDS.  With a fixed
        // eight units per wave the older wave of each SIMD wins the MFMA arbitration, finishes 2.6 units ahead and waits 24 k
        // cycles at the barrier below while the younger one runs on alone at 42 % of the pipe (phase stamps);
        // taken from the counter, all waves finish within one unit of each other.  The rows of a unit go to the staging block of
        // the tile the points belong to, so the hand-over reads the same layout as before -- after a barrier now.
        const int pass_base = it * 128;
        const int units = ((K - pass_base < 128 ? K - pass_base : 128) + 1) >> 1;
        for (;;) {
            int unit = 0;
            if (lane0 == 0) unit = atomicAdd(&sa_next[it & 7], 1);
            unit = __builtin_amdgcn_readfirstlane(unit);
            if (unit >= units) break;
            const int i0 = pass_base + 2 * unit;
            f32x4 h0[2][2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int i = i0 + nt;
                const int j = nbr16[i * 16 + n];
                const float rel = g < 3 ? __fsub_rn(sx[3 * j + g], sx[3 * i + g]) : 1.0f;      // grouped_xyz -= new_xyz; bias input
                h0[nt][0] = relu4(mfma16(w0a, rel, zero4));
                h0[nt][1] = relu4(mfma16(w0b, rel, zero4));
            }
            f32x4 a1[2][4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) a1[0][mt] = a1[1][mt] = *(const f32x4 *)(sb1 + 16 * mt + 4 * g);
            // conv2's accumulators start at zero (an inline constant of the first MFMA, no register fill); its bias is added after
            // the neighbour max: max_j (y_j + b) == (max_j y_j) + b exactly, since adding b is monotone (16 v_mov saved per point)
            f32x4 a2[2][8];
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) a2[0][mt] = a2[1][mt] = zero4;
            bf16x8 i1[2][1][3];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) b3_split8(h0[nt][0], h0[nt][1], i1[nt][0]);
            dense_b3<1, 4, 2>(sw1, lane, i1, a1);                                    // conv1
            bf16x8 i2[2][2][3];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int t = 0; t < 2; ++t) b3_split8(relu4(a1[nt][2 * t]), relu4(a1[nt][2 * t + 1]), i2[nt][t]);
            dense_b3<2, 8, 2, true>(sw2, lane, i2, a2);                              // conv2, transposed
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                float mx[2];
                max16_of_8_transposed_tiles(a2[nt], mx);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {         // lane (row g, j = n) holds channel 16*(2g + s2) + n of point i0 + nt
                    const int ch = 16 * (2 * g + s2) + n;
                    stage_all[(i0 + nt - pass_base) * FU_STAGE_STRIDE + ch] = fmaxf(__fadd_rn(mx[s2], sb2[ch]), 0.f);
                }
            }
        }

        __syncthreads();                                  // every row of the pass is staged
        // ---- hand-over: the rows of this wave's tile, read back as PointNet's B operand and split into planes
        lane = lane0;
        asm volatile("" : "+v"(lane));
        g = lane >> 4; n = lane & 15;
        bf16x8 i0p[1][5][3];
        {
            f32x4 in[9];
            const float *stage_t = stage_all + (valid ? p0 - pass_base : 0) * FU_STAGE_STRIDE;   // an idle wave: block 0, discarded
#pragma unroll
            for (int kt = 0; kt < 8; ++kt) in[kt] = *(const f32x4 *)(stage_t + n * FU_STAGE_STRIDE + 16 * kt + 4 * g);
            const int p = p0 + n;
            in[8][0] = g == 0 ? sx[3 * p] : 0.f;          // channels 128,129,130 = x,y,z (g == 0, r = 0..2)
            in[8][1] = g == 0 ? sx[3 * p + 1] : 0.f;
            in[8][2] = g == 0 ? sx[3 * p + 2] : 0.f;
            in[8][3] = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) b3_split8(in[2 * t], in[2 * t + 1], i0p[0][t]);
            b3_split8(in[8], zero4, i0p[0][4]);
        }
        __syncthreads();                                  // every wave has its tile in registers: the region becomes the weight ring

        // ---- PointNet pass (pn_forward_b3_kernel's), ring started cold
        lane = lane0;
        asm volatile("" : "+v"(lane));
        g = lane >> 4; n = lane & 15;
        blob = opaque_uniform(blob);
        WStreamT<FU_CHUNK, FU_NB, 8> ws{opaque_uniform(pn3), swt, (PN_B3_STREAM_FRAGS + FU_CHUNK - 1) / FU_CHUNK, lane, wu, false};   // data chunks only
        ws.prologue();
        int f = 0;                                        // fragment cursor of this pass (constant-folds)
        f32x4 a0[1][8];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) a0[0][mt] = *(const f32x4 *)(blob + ENC_PN_B0 + 16 * mt + 4 * g);
        FU_DENSE<5, 8, 1>(ws, f, i0p, a0);
        f32x4 a1p[1][16];
        {
            bf16x8 i1p[1][4][3];
#pragma unroll
            for (int t = 0; t < 4; ++t) b3_split8(relu4(a0[0][2 * t]), relu4(a0[0][2 * t + 1]), i1p[0][t]);
#pragma unroll
            for (int mt = 0; mt < 16; ++mt) a1p[0][mt] = *(const f32x4 *)(blob + ENC_PN_B1 + 16 * mt + 4 * g);
            FU_DENSE<4, 16, 1>(ws, f, i1p, a1p);
        }
        f32x4 a3[1][1];
        a3[0][0] = *(const f32x4 *)(blob + ENC_PN_B3 + 4 * g);
#ifdef FU_CACHE_PLANES
        // layer 1's activation kept as its bf16 planes (96 registers) instead of as fp32 (64) and split again for each half of
        // layer 2: 8 splits per pass instead of 16, the same values
        bf16x8 p1[8][3];
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) b3_split8(relu4(a1p[0][2 * kt]), relu4(a1p[0][2 * kt + 1]), p1[kt]);
#endif
#pragma clang loop unroll(full)
        for (int h = 0; h < 2; ++h) {                     // layer 2 in two halves of 16 output tiles
            f32x4 a2p[1][16];
#pragma unroll
            for (int mt = 0; mt < 16; ++mt) a2p[0][mt] = *(const f32x4 *)(blob + ENC_PN_B2 + 16 * (16 * h + mt) + 4 * g);
#pragma clang loop unroll(full)
            for (int kt = 0; kt < 8; ++kt) {
                bf16x8 pl[1][1][3];
#ifdef FU_CACHE_PLANES
                pl[0][0][0] = p1[kt][0]; pl[0][0][1] = p1[kt][1]; pl[0][0][2] = p1[kt][2];
#else
                b3_split8(relu4(a1p[0][2 * kt]), relu4(a1p[0][2 * kt + 1]), pl[0][0]);
#endif
                FU_DENSE<1, 16, 1>(ws, f, pl, a2p);
            }
#pragma clang loop unroll(full)
            for (int kt = 0; kt < 8; ++kt) {              // layer 3 over these 256 channels (no ReLU after it, AE.py:17)
                bf16x8 pl[1][1][3];
                b3_split8(relu4(a2p[0][2 * kt]), relu4(a2p[0][2 * kt + 1]), pl[0][0]);
                FU_DENSE<1, 1, 1>(ws, f, pl, a3);
            }
        }
        ws.drain();
        if (valid) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = row16_max(a3[0][0][r]);
                if (n == 0) smax[wu][4 * g + r] = fmaxf(smax[wu][4 * g + r], v);
            }
        }
        __syncthreads();                                  // every wave is done reading the ring: the region is staging again
    }
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
    __syncthreads();                                      // smax / sx / nbr16 are rewritten for the next patch
  }
}

// 1 when the fused kernel can hold a K-point patch (its neighbour table grows with K), 0 when the caller must run
// pccx_sa_forward_b3 + pccx_pn_forward_b3 through a feature workspace instead.
extern "C" int pccx_ae_encode_b3_fused_ok(int K)
{
    return (K >= 16 && K <= 1024 && K % 16 == 0 && fu_lds_bytes(K) <= (size_t)160 * 1024) ? 1 : 0;
}

static int fu_launch(const float *patches, int P, int K, const float *enc_blob, const float *sa_b3_blob, const float *pn_b3_blob, int d,
                     int L, float *latent_raw, float *latent, float *latent_q, const unsigned char *nbr_tab, hipStream_t stream)
{
    const float spread = (float)((double)L - 0.2);
    const float half = (float)(((double)L - 0.2) / 2);
    // one workgroup per CU at a time (LDS): a grid of 8 workgroups per CU, each walking P / grid patches, keeps the SetAbstraction
    // weights staged and still balances the tail
    const int grid = P < 2048 ? P : 2048;
    if (nbr_tab) {
        PCCX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&sa_pn_forward_b3_kernel<true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL(sa_pn_forward_b3_kernel<true>, dim3(grid), dim3(512), fu_lds_bytes(K), stream, patches, P, K, enc_blob, sa_b3_blob,
                           pn_b3_blob, d, spread, half, latent_raw, latent, latent_q, nbr_tab);
    } else {
        PCCX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&sa_pn_forward_b3_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL(sa_pn_forward_b3_kernel<false>, dim3(grid), dim3(512), fu_lds_bytes(K), stream, patches, P, K, enc_blob, sa_b3_blob,
                           pn_b3_blob, d, spread, half, latent_raw, latent, latent_q, (const unsigned char *)nullptr);
    }
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

#define FU_CHECK(fn)                                                                                                                   \
    PCCX_CHECK_ARG(patches && enc_blob && sa_b3_blob && pn_b3_blob && latent_raw && latent && latent_q, fn ": null pointer");             \
    PCCX_CHECK_ARG(P >= 0 && pccx_ae_encode_b3_fused_ok(K), fn ": K=%d does not fit the fused kernel (pccx_ae_encode_b3_fused_ok)", K); \
    PCCX_CHECK_ARG(d >= 1 && d <= 16 && L >= 1, fn ": unsupported d=%d L=%d", d, L)

// workspace-free form: the in-patch neighbour selection runs inside the kernel (round 2's shape)
extern "C" int pccx_ae_encode_b3(const float *patches, int P, int K, const float *enc_blob, const float *sa_b3_blob,
                                 const float *pn_b3_blob, int d, int L, float *latent_raw, float *latent, float *latent_q, void *stream)
{
    if (P == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    FU_CHECK("pccx_ae_encode_b3");
    return fu_launch(patches, P, K, enc_blob, sa_b3_blob, pn_b3_blob, d, L, latent_raw, latent, latent_q, nullptr, (hipStream_t)stream);
}

// the default form: pccx_patch_knn16 (own kernel, 8 waves per SIMD) fills the neighbour table in `workspace`
// (pccx_ae_encode_b3_workspace_bytes(P, K) bytes, 16-byte aligned), then the fused kernel reads it.  Same results, bit for bit.
extern "C" size_t pccx_ae_encode_b3_workspace_bytes(int P, int K) { return pccx_patch_knn16_bytes(P, K); }

extern "C" int pccx_ae_encode_b3_ws(const float *patches, int P, int K, const float *enc_blob, const float *sa_b3_blob,
                                    const float *pn_b3_blob, int d, int L, float *latent_raw, float *latent, float *latent_q,
                                    void *workspace, void *stream)
{
    if (P == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    FU_CHECK("pccx_ae_encode_b3_ws");
    PCCX_CHECK_ARG(workspace, "pccx_ae_encode_b3_ws: null workspace");
    const int rc = pccx_patch_knn16(patches, P, K, workspace, stream);
    if (rc != PCCX_OK) return rc;
    return fu_launch(patches, P, K, enc_blob, sa_b3_blob, pn_b3_blob, d, L, latent_raw, latent, latent_q, (const unsigned char *)workspace,
                     (hipStream_t)stream);
}

// the fused kernel alone, on neighbour tables the caller has filled with pccx_patch_knn16 (see pccx_ae_encode_h2_tables)
extern "C" int pccx_ae_encode_b3_tables(const float *patches, int P, int K, const float *enc_blob, const float *sa_b3_blob,
                                        const float *pn_b3_blob, int d, int L, float *latent_raw, float *latent, float *latent_q,
                                        const void *workspace, void *stream)
{
    if (P == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    FU_CHECK("pccx_ae_encode_b3_tables");
    PCCX_CHECK_ARG(workspace && ((uintptr_t)workspace & 15) == 0, "pccx_ae_encode_b3_tables: null or misaligned tables");
    return fu_launch(patches, P, K, enc_blob, sa_b3_blob, pn_b3_blob, d, L, latent_raw, latent, latent_q, (const unsigned char *)workspace,
                     (hipStream_t)stream);
}
