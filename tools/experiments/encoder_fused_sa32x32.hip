// EXPERIMENT RECORD (round 2), NOT part of libpccx.so: csrc/encoder_fused.hip with its SetAbstraction phase on
// v_mfma_f32_32x32x16_bf16 (two points = 32 columns per tile, conv0 on v_mfma_f32_32x32x2_f32, accumulator tiles reused as the next
// layer's operand with the k order of cdna_hip_programming.md "An accumulator tile as the next MFMA's operand", neighbour max
// in-lane + one v_permlane32_swap, weight planes re-packed by sa32_pack_kernel).  Correct at the first run (all oracle / golden
// parity tests green in bf16x3 mode; results differ from the 16x16x32 form only by fp32 summation order), but NOT faster:
// 227.5 TFLOP/s against 234 for the 16x16x32 form, with or without the weight planes pipelined one step ahead.  Halving the
// number of MFMA instructions (and so the issue slots they hold) does not move this phase, i.e. it is not MFMA-issue bound:
// by the two-wave occupancy model each wave spends ~8000 cycles per pair of points outside its MFMAs
// against ~2000 cycles of VALU issue, so the time is in dependency / LDS-wait stalls that the compiler-scheduled stream
// does not overlap.  Kept for the next attempt (a hand-scheduled stream).
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
    const size_t ring = (size_t)2 * PN_B3_CHUNK * 1024, stage = (size_t)8 * FU_STAGE_WAVE * 4;
    return ring > stage ? ring : stage;
}
__host__ __device__ inline size_t fu_lds_bytes(int K)
{
    return (size_t)(FU_W1_FRAGS + FU_W2_FRAGS) * 1024 + (64 + 128) * 4 + (size_t)K * 12 + (size_t)K * 32 + fu_region_bytes() + 8 * 16 * 4;
}

// ---- SetAbstraction on 32x32x16 MFMAs ------------------------------------------------------------------------------
// One iteration handles TWO points = 32 columns (point c >> 4, neighbour c & 15).  v_mfma_f32_32x32x16_bf16 does the work of two
// 16x16x32 per instruction while holding the SIMD's issue port for the same 8 cycles, and this phase is bound by instruction
// issue (6 products per fp32 product plus the bf16 splits), not by the matrix pipe.  Lane l = (c = l & 31, h = l >> 5).
//   C/D (f32x16): register r is row (r & 3) + 8 (r >> 2) + 4 h, column c.
//   A/B fragment of k-step s taken from an accumulator tile X: registers 8s .. 8s+7 -> element j is row 16 s + 8 (j >> 2) + 4 h
//   + (j & 3) of X (cdna_hip_programming.md, "An accumulator tile as the next MFMA's operand"); the weight fragments are packed
//   with that k order (sa32_pack_kernel), so the chain conv0 -> conv1 -> conv2 runs out of registers:
//   conv0  H0 = W0b . [x y z 1]      two v_mfma_f32_32x32x2_f32 (exact fp32, the k-ordered chain x, y, z, bias of the 16x16x4 form)
//   conv1  A1 = W1 . relu(H0)        A = weight planes [mt 2][s 2], B = H0's planes
//   conv2  Z  = relu(A1)^T . W2^T    A = A1's planes, B = weight planes [ks 4][nt 4]: Z has the CHANNEL on the lane and the 32
//                                    columns in registers (0-7: point 0, 8-15: point 1), so the neighbour max is in-lane plus one
//                                    v_permlane32_swap; lane (c, h) ends with channel 32 nt + c of point h.
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define SA32_W1_FRAGS (2 * 2 * 3)
#define SA32_W2_FRAGS (4 * 4 * 3)

__device__ __forceinline__ void sa32_split(const f32x16 &x, int s, bool relu, bf16x8 (&pl)[3])
{
    f32x4 v0, v1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        v0[r] = relu ? fmaxf(x[8 * s + r], 0.f) : x[8 * s + r];
        v1[r] = relu ? fmaxf(x[8 * s + 4 + r], 0.f) : x[8 * s + 4 + r];
    }
    b3_split8(v0, v1, pl);
}

// W (out x in) read back from its 16x16x4 fragment table in the encoder blob ([kt][mt][lane][r], MT16 m-tiles per k-tile)
__device__ __forceinline__ float sa32_w(const float *frag, int MT16, int out, int in)
{
    const int kt = in >> 4, g = (in & 15) >> 2, r = in & 3, mt = out >> 4, lane = (out & 15) + 16 * g;
    return frag[(((size_t)kt * MT16 + mt) * 64 + lane) * 4 + r];
}

// sa32 blob: conv1 planes [mt 2][s 2][plane 3][64 lanes] then conv2 planes [ks 4][nt 4][plane 3][64 lanes] (uint4 each)
__global__ __launch_bounds__(64) void sa32_pack_kernel(const float *__restrict__ blob, uint4 *__restrict__ out)
{
    const int lane = threadIdx.x, c = lane & 31, h = lane >> 5, item = blockIdx.x;
    float v[8];
    if (item < 4) {                                        // conv1: out = 32 mt + c, in = 16 s + 8 (j >> 2) + 4 h + (j & 3)
        const int mt = item >> 1, sdx = item & 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = sa32_w(blob + ENC_SA_W1, 4, 32 * mt + c, 16 * sdx + 8 * (j >> 2) + 4 * h + (j & 3));
    } else {                                               // conv2 (B operand): out = 32 nt + c, in = 32 mtx + 16 s + ...
        const int ks = (item - 4) >> 2, nt = (item - 4) & 3;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = sa32_w(blob + ENC_SA_W2, 8, 32 * nt + c, 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3));
    }
    const f32x4 v0 = {v[0], v[1], v[2], v[3]}, v1 = {v[4], v[5], v[6], v[7]};
    bf16x8 pl[3];
    b3_split8(v0, v1, pl);
#pragma unroll
    for (int p = 0; p < 3; ++p) out[((size_t)item * 3 + p) * 64 + lane] = __builtin_bit_cast(uint4, pl[p]);
}

__global__ __launch_bounds__(512, 1) void sa_pn_forward_b3_kernel(const float *__restrict__ x, int npatches, int K, const float *__restrict__ blob,
                                                                  const float *__restrict__ sa3, const float *__restrict__ pn3, int d,
                                                                  float spread, float half_spread, float *__restrict__ latent_raw,
                                                                  float *__restrict__ latent, float *__restrict__ latent_q)
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

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int ntiles = K >> 4;
    const int wu = __builtin_amdgcn_readfirstlane(w);
    float *stage = stage_all + wu * FU_STAGE_WAVE;

    {   // stage the SetAbstraction weights ONCE per workgroup: a workgroup walks patches blockIdx.x, + gridDim.x, ...
        const f32x4 *gw1 = (const f32x4 *)sa3, *gw2 = (const f32x4 *)sa3 + FU_W1_FRAGS * 64;     // the sa32 blob: same 12 + 48 fragments
        for (int i = tid; i < FU_W1_FRAGS * 64; i += 512) sw1[i] = gw1[i];
        for (int i = tid; i < FU_W2_FRAGS * 64; i += 512) sw2[i] = gw2[i];
        if (tid < 64) sb1[tid] = blob[ENC_SA_B1 + tid];
        if (tid < 128) sb2[tid] = blob[ENC_SA_B2 + tid];
    }
  for (size_t P = blockIdx.x; P < (size_t)npatches; P += gridDim.x) {
    const float *xp = x + P * (size_t)K * 3;
    for (int i = tid; i < 3 * K; i += 512) sx[i] = xp[i];
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
    for (int ib = 0; ib < K; ib += 256) {
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

        // ---- SetAbstraction for points p0 .. p0+15, two per iteration, on 32x32 MFMAs (see above)
        {
            const int c = lane & 31, h = lane >> 5;
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};     // weight plane / activation plane, smallest products first
            const float w0k0 = blob[ENC_SA_W0B0 + 4 * c + h], w0k1 = blob[ENC_SA_W0B0 + 4 * c + 2 + h];   // (wx | wy), (wz | bias) of channel c
            const uint4 *w1p = (const uint4 *)sw1 + lane, *w2p = (const uint4 *)sw2 + lane;
            for (int i0 = p0; i0 < p0 + 16; i0 += 2) {
                const int i = i0 + (c >> 4);
                const int j = nbr16[i * 16 + (c & 15)];
                // grouped_xyz -= new_xyz (pn_kit.py:191): k = h of the first step (x | y), of the second (z | the bias input 1)
                const float b0 = __fsub_rn(sx[3 * j + h], sx[3 * i + h]);
                const float b1 = h == 0 ? __fsub_rn(sx[3 * j + 2], sx[3 * i + 2]) : 1.0f;
                f32x16 h0;
#pragma unroll
                for (int r = 0; r < 16; ++r) h0[r] = 0.f;
                h0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w0k0, b0, h0, 0, 0, 0);
                h0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w0k1, b1, h0, 0, 0, 0);
                bf16x8 i1[2][3];
#pragma unroll
                for (int sdx = 0; sdx < 2; ++sdx) sa32_split(h0, sdx, true, i1[sdx]);           // relu(conv0)
                f32x16 a1[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 bq = *(const f32x4 *)(sb1 + 32 * mt + 8 * q + 4 * h);       // bias of rows 8q + 4h + (0..3)
#pragma unroll
                        for (int r = 0; r < 4; ++r) a1[mt][4 * q + r] = bq[r];
                    }
                {   // conv1: four (mt, s) steps; the planes of step t+1 are fetched from LDS ahead of the MFMAs of step t and pinned
                    // there (left alone the compiler hoists every fragment load of the unrolled chain and spills)
                    bf16x8 cur[3], nxt[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) cur[p] = __builtin_bit_cast(bf16x8, w1p[p * 64]);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        if (t + 1 < 4) {
#pragma unroll
                            for (int p = 0; p < 3; ++p) nxt[p] = __builtin_bit_cast(bf16x8, w1p[((t + 1) * 3 + p) * 64]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int q = 0; q < 6; ++q)
                            a1[t >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[PA[q]], i1[t & 1][PB[q]], a1[t >> 1], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int p = 0; p < 3; ++p) cur[p] = nxt[p];
                    }
                }
                bf16x8 i2[4][3];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) sa32_split(a1[ks >> 1], ks & 1, true, i2[ks]);  // relu(conv1)
                f32x16 a2[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) a2[nt][r] = 0.f;                               // conv2's bias is added after the max
                {   // conv2, transposed: sixteen (ks, nt) steps, weight planes pipelined one step ahead as in conv1
                    bf16x8 cur[3], nxt[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) cur[p] = __builtin_bit_cast(bf16x8, w2p[p * 64]);
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        if (t + 1 < 16) {
#pragma unroll
                            for (int p = 0; p < 3; ++p) nxt[p] = __builtin_bit_cast(bf16x8, w2p[((t + 1) * 3 + p) * 64]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int q = 0; q < 6; ++q)
                            a2[t & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(i2[t >> 2][PB[q]], cur[PA[q]], a2[t & 3], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int p = 0; p < 3; ++p) cur[p] = nxt[p];
                    }
                }
                // max over the 16 neighbours: registers 0-7 are point 0's rows 4h + (0..3), 8 + 4h + (0..3); 8-15 point 1's
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    float m0 = a2[nt][0], m1 = a2[nt][8];
#pragma unroll
                    for (int r = 1; r < 8; ++r) { m0 = fmaxf(m0, a2[nt][r]); m1 = fmaxf(m1, a2[nt][8 + r]); }
                    auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m0), __float_as_uint(m1), false, false);
                    const float mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));   // lanes h = 0: point 0, h = 1: point 1
                    const int ch = 32 * nt + c;
                    stage[(i0 + h - p0) * FU_STAGE_STRIDE + ch] = fmaxf(__fadd_rn(mx, sb2[ch]), 0.f);
                }
            }
        }

        // ---- hand-over: the wave's own rows, read back as PointNet's B operand and split into planes
        bf16x8 i0p[1][5][3];
        {
            f32x4 in[9];
#pragma unroll
            for (int kt = 0; kt < 8; ++kt) in[kt] = *(const f32x4 *)(stage + n * FU_STAGE_STRIDE + 16 * kt + 4 * g);
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
        WStreamT<PN_B3_CHUNK, 2, 8> ws{opaque_uniform(pn3), swt, (PN_B3_STREAM_FRAGS + PN_B3_CHUNK - 1) / PN_B3_CHUNK, lane, wu, false};   // data chunks only
        ws.prologue();
        int f = 0;                                        // fragment cursor of this pass (constant-folds)
        f32x4 a0[1][8];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) a0[0][mt] = *(const f32x4 *)(blob + ENC_PN_B0 + 16 * mt + 4 * g);
        dense_b3_stream<5, 8, 1>(ws, f, i0p, a0);
        f32x4 a1p[1][16];
        {
            bf16x8 i1p[1][4][3];
#pragma unroll
            for (int t = 0; t < 4; ++t) b3_split8(relu4(a0[0][2 * t]), relu4(a0[0][2 * t + 1]), i1p[0][t]);
#pragma unroll
            for (int mt = 0; mt < 16; ++mt) a1p[0][mt] = *(const f32x4 *)(blob + ENC_PN_B1 + 16 * mt + 4 * g);
            dense_b3_stream<4, 16, 1>(ws, f, i1p, a1p);
        }
        f32x4 a3[1][1];
        a3[0][0] = *(const f32x4 *)(blob + ENC_PN_B3 + 4 * g);
#pragma clang loop unroll(full)
        for (int h = 0; h < 2; ++h) {                     // layer 2 in two halves of 16 output tiles
            f32x4 a2p[1][16];
#pragma unroll
            for (int mt = 0; mt < 16; ++mt) a2p[0][mt] = *(const f32x4 *)(blob + ENC_PN_B2 + 16 * (16 * h + mt) + 4 * g);
#pragma clang loop unroll(full)
            for (int kt = 0; kt < 8; ++kt) {
                bf16x8 pl[1][1][3];
                b3_split8(relu4(a1p[0][2 * kt]), relu4(a1p[0][2 * kt + 1]), pl[0][0]);
                dense_b3_stream<1, 16, 1>(ws, f, pl, a2p);
            }
#pragma clang loop unroll(full)
            for (int kt = 0; kt < 8; ++kt) {              // layer 3 over these 256 channels (no ReLU after it, AE.py:17)
                bf16x8 pl[1][1][3];
                b3_split8(relu4(a2p[0][2 * kt]), relu4(a2p[0][2 * kt + 1]), pl[0][0]);
                dense_b3_stream<1, 1, 1>(ws, f, pl, a3);
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

// The SetAbstraction weight planes of the fused kernel (32x32x16 fragment order), built on the device from the encoder blob
extern "C" size_t pccx_sa_b3x32_blob_floats(void) { return (size_t)(SA32_W1_FRAGS + SA32_W2_FRAGS) * 256; }

extern "C" int pccx_pack_sa_b3x32(const float *enc_blob_dev, float *sa32_blob_dev, void *stream)
{
    PCCX_CHECK_ARG(enc_blob_dev && sa32_blob_dev, "pccx_pack_sa_b3x32: null pointer");
    hipLaunchKernelGGL(sa32_pack_kernel, dim3(4 + 16), dim3(64), 0, (hipStream_t)stream, enc_blob_dev, (uint4 *)sa32_blob_dev);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// 1 when the fused kernel can hold a K-point patch (its neighbour table grows with K), 0 when the caller must run
// pccx_sa_forward_b3 + pccx_pn_forward_b3 through a feature workspace instead.
extern "C" int pccx_ae_encode_b3_fused_ok(int K)
{
    return (K >= 16 && K <= 1024 && K % 16 == 0 && fu_lds_bytes(K) <= (size_t)160 * 1024) ? 1 : 0;
}

extern "C" int pccx_ae_encode_b3(const float *patches, int P, int K, const float *enc_blob, const float *sa_b3_blob /* pccx_pack_sa_b3x32 */,
                                 const float *pn_b3_blob, int d, int L, float *latent_raw, float *latent, float *latent_q, void *stream)
{
    if (P == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(patches && enc_blob && sa_b3_blob && pn_b3_blob && latent_raw && latent && latent_q, "pccx_ae_encode_b3: null pointer");
    PCCX_CHECK_ARG(P >= 0 && pccx_ae_encode_b3_fused_ok(K), "pccx_ae_encode_b3: K=%d does not fit the fused kernel (pccx_ae_encode_b3_fused_ok)", K);
    PCCX_CHECK_ARG(d >= 1 && d <= 16 && L >= 1, "pccx_ae_encode_b3: unsupported d=%d L=%d", d, L);
    const float spread = (float)((double)L - 0.2);
    const float half = (float)(((double)L - 0.2) / 2);
    PCCX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&sa_pn_forward_b3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024));
    // one workgroup per CU at a time (LDS): a grid of 8 workgroups per CU, each walking P / grid patches, keeps the SetAbstraction
    // weights staged and still balances the tail
    const int grid = P < 2048 ? P : 2048;
    hipLaunchKernelGGL(sa_pn_forward_b3_kernel, dim3(grid), dim3(512), fu_lds_bytes(K), (hipStream_t)stream, patches, P, K, enc_blob,
                       sa_b3_blob, pn_b3_blob, d, spread, half, latent_raw, latent, latent_q);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}
