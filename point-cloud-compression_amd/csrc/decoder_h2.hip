// decoder_h2.hip -- the synthesis transform of AE.AE (AE.py:48-53; decompress.py:97-116) in f16x2 arithmetic.
// Same structure as decoder.hip's bf16x3 form: dec_head_kernel (fp32, unchanged) -> operand preparation -> dec_main: the
// 1024 -> k*128 Linear as a GEMM whose accumulators feed inv_mlp from registers, point p's weight stream through an LDS-DMA ring.
// What differs is the arithmetic of the products: two fp16 pieces per operand and three v_mfma_f32_16x16x32_f16 passes
// (mfma_chain.h, "f16x2 operands") with the static power-of-two scales of pack_h2.hip, plus ONE dynamic scale per patch:
//   s_n = 2^-e <= 1 with max(largest head activation, largest |latent|) * s_n <= 1   (dec_h2_prep_kernel)
// Head activations and latents are multiplied by s_n when their planes are formed, every bias by s_n when it initialises an
// accumulator (inv_pool.4 -> inv_mlp is positively homogeneous in (input, biases)), and the output is divided by s_n.
#include <math.h>
#include <stdlib.h>

#include "blobs.h"
#include "common.h"
#include "mfma_chain.h"

int pccx_dec_head_launch(const float *latent_q, int P, int d, int ntiles, const float *dec_blob, float *h2p, hipStream_t st);   // decoder.hip

#ifndef DEC_GROUP
#define DEC_GROUP 64                       // patch blocks per group of the block order (as decoder.hip)
#endif

// one wave per tile of 16 patches: the patch scales, then the two fp16 planes of the head activation
// h2p: [64 kt][ntiles][64 lanes] f32x4 (lane (g, n): channels 16 kt + 4 g + r of patch n); h3: [32 t][ntiles][2][64 lanes] uint4
__global__ __launch_bounds__(256) void dec_h2_prep_kernel(const f32x4 *__restrict__ h2p, const float *__restrict__ latent_q, int P, int d,
                                                          int ntiles, float sig_h, uint4 *__restrict__ h3, float *__restrict__ pscale)
{
    const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    float m = 0.f;
    for (int kt = 0; kt < 64; ++kt) {
        const f32x4 v = h2p[((size_t)kt * ntiles + tile) * 64 + lane];
        m = fmaxf(fmaxf(m, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
    }
    const int patch = tile * 16 + n;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (patch < P && 4 * g + r < d) m = fmaxf(m, fabsf(latent_q[(size_t)patch * d + 4 * g + r]));
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    const int rexp = (int)(__float_as_uint(m) >> 23) - 127;
    const float s = rexp >= 0 ? __uint_as_float((unsigned)(126 - rexp) << 23) : 1.0f;
    if (g == 0) pscale[patch] = s;
    const float sc = s * sig_h;
    for (int t = 0; t < 32; ++t) {
        const f32x4 v0 = h2p[((size_t)(2 * t) * ntiles + tile) * 64 + lane], v1 = h2p[((size_t)(2 * t + 1) * ntiles + tile) * 64 + lane];
        f16x8 pl[2];
        h2_split8(v0, v1, sc, pl);
        uint4 *o = h3 + (((size_t)t * ntiles + tile) * 2) * 64 + lane;
        o[0] = __builtin_bit_cast(uint4, pl[0]);
        o[64] = __builtin_bit_cast(uint4, pl[1]);
    }
}

__device__ __forceinline__ uint4 h2_load_async(const uint4 *p)    // placed exactly here; completion is covered by the ring's s_waitcnt
{
    uint4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// grid: groups of DEC_GROUP patch blocks outermost, then the point p, then the block inside the group (decoder.hip); 4 waves, wave w
// owns patch tiles NT w .. NT w + NT - 1 of its block.
// NT = 2: the shape of decoder.hip's bf16x3 kernel.  NT = 4 (the default): FOUR patch tiles per wave in the GEMM -- every weight block read
// from the ring feeds 12 MFMAs instead of 6 and point p's stream enters LDS once per 256 patches instead of once per 128 (the two fp16 planes of
// four tiles' operands fit the two-waves-per-SIMD register budget where three bf16 planes did not: decoder.hip's NT = 4 needs one wave per SIMD);
// inv_mlp then runs twice on two tiles each, from the tail's fragments stored twice in the stream (blobs.h).  Same products in the same order.
template <int NT>
__global__ __launch_bounds__(256, 2) void dec_main_h2_kernel(const uint4 *__restrict__ h3, const float *__restrict__ pscale,
                                                             const float *__restrict__ latent_q, int P, int d, int k, int ntiles,
                                                             const float *__restrict__ hb, float *__restrict__ patches_out,
                                                             float inv_scale_div, const float *__restrict__ centres,
                                                             const float *__restrict__ nrm_center, const float *__restrict__ nrm_longest, int S,
                                                             float one_minus_margin, float *__restrict__ pc_out)
{
    static_assert(NT == 2 || NT == 4, "two or four patch tiles per wave");
    constexpr int NP = NT / 2;                                // tile pairs: inv_mlp runs once per pair
    constexpr int SETS = NT == 2 ? 3 : 2;                     // rotating B register sets (NT = 4: one k-step ahead, 64 registers)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, n = lane & 15;
    constexpr int GRP = DEC_GROUP * 2 / NT;
    const int nblk = (ntiles + 4 * NT - 1) / (4 * NT);
    const int grp = blockIdx.x / (GRP * k), rem = blockIdx.x % (GRP * k);
    const int p = rem / GRP, blk = grp * GRP + rem % GRP;
    if (blk >= nblk) return;                                  // whole workgroup (before any barrier)
    const int tile0 = blk * 4 * NT + NT * w;
    constexpr int CH = DEC_H2_CHUNK, NB = 4;
    constexpr int DPW = CH / 4;                               // DMA loads per wave per chunk
    __shared__ __attribute__((aligned(16))) f32x4 swt[NB * CH * 64];
    const int wu = __builtin_amdgcn_readfirstlane(w);
    const WStreamT<CH, NB> ws{hb + DEC_H2_G_W(k) + (size_t)p * DEC_H2_STREAM_CHUNKS * DEC_H2_CHUNK * 256, swt, DEC_H2_STREAM_CHUNKS, lane, wu, false};
    ws.prologue();

    int tq[NT];
    float sn[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        tq[nt] = tile0 + nt < ntiles ? tile0 + nt : ntiles - 1;
        sn[nt] = pscale[tq[nt] * 16 + n];
    }
    f32x4 acc[NP][2][8];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        const f32x4 b = *(const f32x4 *)(hb + DEC_H2_G_B + (size_t)p * 128 + 16 * mt + 4 * g);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt >> 1][nt & 1][mt] = b * sn[nt];
    }
    {
        // ---- GEMM over K = 1024 as 32 k-steps of 32.  A planes through the 4-deep LDS ring (chunk = 4 m-tiles x 2 planes, DMA
        // three chunks ahead); the B planes of this wave's patch tiles in SETS rotating register sets, loaded SETS - 1 k-steps ahead by
        // asm loads whose completion rides on the ring's waits.  VMEM issue order per wave:
        //   boundary(2t):   DMA(2t+3) [DPW loads], B(t + SETS - 1) [2 NT loads]        boundary(2t+1): DMA(2t+4) [DPW loads]
        // three sets: boundary(2t) needs all but its 2 DPW + 2 NT youngest loads, boundary(2t+1) all but its 2 DPW + 4 NT youngest;
        // two sets:   boundary(2t) needs B(t), issued one k-step ago: all but the DPW loads of DMA(2t+2); boundary(2t+1) as above.
        uint4 bs[SETS][NT][2];
        auto load_b = [&](uint4 (&dst)[NT][2], int t) {
            const int tc = t < 32 ? t : 31;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) dst[nt][pl] = h2_load_async(h3 + (((size_t)tc * ntiles + tq[nt]) * 2 + pl) * 64 + lane);
        };
        auto kstep = [&](int t, const uint4 (&bc)[NT][2], uint4 (&bload)[NT][2], bool first) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int c = 2 * t + half;
                if (half == 0) {
                    if (first) ws.boundary(c); else ws.template boundary_keep<SETS == 3 ? 2 * DPW + 2 * NT : DPW>(c);
                    load_b(bload, t + SETS - 1);
                } else
                    ws.template boundary_keep<2 * DPW + 4 * NT>(c);
                const f32x4 *buf = ws.chunk(c);
                f16x8 a[4][2];
#pragma unroll
                for (int mq = 0; mq < 4; ++mq)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) a[mq][pl] = __builtin_bit_cast(f16x8, buf[(mq * 2 + pl) * 64]);
                __builtin_amdgcn_sched_barrier(0);
                constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};          // (lo,hi) (hi,lo) (hi,hi)
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int mq = 0; mq < 4; ++mq)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[nt >> 1][nt & 1][4 * half + mq] =
                                H2_MFMA(a[mq][PA[q]], __builtin_bit_cast(f16x8, bc[nt][PB[q]]), acc[nt >> 1][nt & 1][4 * half + mq]);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if constexpr (SETS == 3) {
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
        } else {
            load_b(bs[0], 0);
            kstep(0, bs[0], bs[1], true);
            kstep(1, bs[1], bs[0], false);
#pragma unroll 1
            for (int t = 2; t < 32; t += 2) {                 // two k-steps per trip, static register sets
                kstep(t, bs[0], bs[1], false);
                kstep(t + 1, bs[1], bs[0], false);
            }
        }
        // The last B loads (clamped, unused) are still in flight.  Their registers are dead to the compiler, which is
        // free to reuse them -- and to hoist register-only work of the tail above a bare wait: a build of this kernel with a larger
        // ring chunk did exactly that and had its tail's operands overwritten by the late loads (tools/asm_load_lint.py finds it).
        // Naming the registers after the wait keeps them allocated until the loads have landed.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < SETS; ++i)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) asm volatile("" ::"v"(__builtin_bit_cast(f32x4, bs[i][nt][pl])));
    }
    // ---- inv_mlp as an f16x2 chain on registers, two tiles at a time: channels 0..127 = relu(inv_pool.4) of point p, 128..143 = latent.
    // Pair pr reads the tail's fragments from their pr-th copy in the stream, which simply continues.
    // Everything the tail needs per lane (tile and patch indices, bias / latent / output addresses, the patch scales) is derived HERE from a
    // laundered thread id and re-read from L2, not carried through the GEMM loop: the loop holds 224 registers of accumulators and operand
    // planes, and values computed in the prologue for the tail were spilled around it (41 VGPRs, 152 B of scratch per lane, round 3).
#ifdef DEC_TAIL_PRIO                 // experiment knob (round 4): issue priority of the VALU-heavy tail over the other workgroup's GEMM waves
    __builtin_amdgcn_s_setprio(DEC_TAIL_PRIO);
#endif
    int lane_t = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(lane_t));
    const int g_t = lane_t >> 4, n_t = lane_t & 15;
    const int tile0_t = blk * 4 * NT + NT * wu;
    const float *meta = hb + DEC_H2_META;                      // the six layer multipliers: wave-uniform, read after the loop, kept in SGPRs
    asm volatile("" : "+s"(meta));                             // (the laundered pointer loses its address space: two FLAT loads per workgroup,
                                                               //  once; mfma_chain.h's opaque_uniform keeps it global but costs two spilled VGPRs here)
    auto uni = [](float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); };
    const float rho0 = uni(meta[H2D_RHO0]), sig_q = uni(meta[H2D_SIG_Q]), rho1 = uni(meta[H2D_RHO1]);
    const float rho2 = uni(meta[H2D_RHO2]), rho3 = uni(meta[H2D_RHO3]), inv_out = uni(meta[H2D_INV_OUT]);
    float sn_t[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int tt = tile0_t + nt < ntiles ? tile0_t + nt : ntiles - 1;
        sn_t[nt] = pscale[tt * 16 + n_t];
    }
    const WStreamT<CH, NB> ws_t{ws.g, swt, DEC_H2_STREAM_CHUNKS, lane_t, wu, false};       // the same ring, addressed from the tail's lane id
    // bias rows: wave-uniform base + an UNSIGNED 32-bit lane offset, so the loads take the scalar-base form and no 64-bit address is held per lane
    const unsigned goff = 16u * (unsigned)g_t;
    auto bias4 = [&](int at) { return *(const f32x4 *)((const char *)(hb + at) + goff); };
    int f = DEC_H2_GEMM_FRAGS;
#pragma unroll
    for (int pr = 0; pr < NP; ++pr) {
        f32x4 m3[2][1];
        {
            f16x8 i0[2][5][2];
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
                for (int t = 0; t < 4; ++t) h2_split8(relu4(acc[pr][nt][2 * t]), relu4(acc[pr][nt][2 * t + 1]), rho0, i0[nt][t]);
                const int patch = (tile0_t + 2 * pr + nt) * 16 + n_t;
                f32x4 lat;
#pragma unroll
                for (int r = 0; r < 4; ++r) lat[r] = (patch < P && 4 * g_t + r < d) ? latent_q[(size_t)patch * d + 4 * g_t + r] : 0.f;
                h2_split8(lat, zero, sn_t[2 * pr + nt] * sig_q, i0[nt][4]);
            }
            f32x4 m0[2][8];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 8; ++mt) m0[nt][mt] = bias4(DEC_H2_M_B0 + 16 * mt) * sn_t[2 * pr + nt];
            dense_h2_stream<5, 8, 2>(ws_t, f, i0, m0);
            f16x8 i1[2][4][2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int t = 0; t < 4; ++t) h2_split8(relu4(m0[nt][2 * t]), relu4(m0[nt][2 * t + 1]), rho1, i1[nt][t]);
            f32x4 m1[2][4];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) m1[nt][mt] = bias4(DEC_H2_M_B1 + 16 * mt) * sn_t[2 * pr + nt];
            dense_h2_stream<4, 4, 2>(ws_t, f, i1, m1);
            f16x8 i2[2][2][2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int t = 0; t < 2; ++t) h2_split8(relu4(m1[nt][2 * t]), relu4(m1[nt][2 * t + 1]), rho2, i2[nt][t]);
            f32x4 m2[2][2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) m2[nt][mt] = bias4(DEC_H2_M_B2 + 16 * mt) * sn_t[2 * pr + nt];
            dense_h2_stream<2, 2, 2>(ws_t, f, i2, m2);
            f16x8 i3[2][1][2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) h2_split8(relu4(m2[nt][0]), relu4(m2[nt][1]), rho3, i3[nt][0]);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) m3[nt][0] = bias4(DEC_H2_M_B3) * sn_t[2 * pr + nt];
            dense_h2_stream<1, 1, 2>(ws_t, f, i3, m3);          // last layer: no ReLU (AE.py:27)
        }
        // ---- epilogue of the pair: rows 0..2 of the last tile (g_t == 0, r = 0..2) are x,y,z of (patch, point p)
        if (g_t == 0) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int tile = tile0_t + 2 * pr + nt, patch = tile * 16 + n_t;
                if (tile < ntiles && patch < P) {
                    const float un = __fmul_rn(inv_out, __fdiv_rn(1.0f, sn_t[2 * pr + nt]));       // undo the operand scales (exact)
                    float v[3] = {__fmul_rn(m3[nt][0][0], un), __fmul_rn(m3[nt][0][1], un), __fmul_rn(m3[nt][0][2], un)};
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
    ws.drain();
}

static unsigned dec_h2_grid(int ntiles, int k, int NT)
{
    const int grp = DEC_GROUP * 2 / NT;
    const int nblk = (ntiles + 4 * NT - 1) / (4 * NT), groups = (nblk + grp - 1) / grp;
    return (unsigned)groups * grp * (unsigned)k;
}

extern "C" size_t pccx_ae_decode_h2_workspace_floats(int P)
{
    const size_t ntiles = ((size_t)(P > 0 ? P : 0) + 15) / 16;
    return (size_t)(64 + 64) * ntiles * 64 * 4 + ntiles * 16;     // fp32 fragments of dec_head, their two fp16 planes, the patch scales
}

extern "C" int pccx_ae_decode_h2(const float *latent_q, int P, int d, int k, const float *dec_blob, const float *h2_blob, float *workspace,
                                 float *patches_out, float scale, const float *centres, const float *nrm_center, const float *nrm_longest,
                                 int S, double margin, float *pc_out, void *stream)
{
    if (P == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(latent_q && dec_blob && h2_blob && workspace, "pccx_ae_decode_h2: null pointer");
    PCCX_CHECK_ARG(patches_out || pc_out, "pccx_ae_decode_h2: need patches_out and/or pc_out");
    PCCX_CHECK_ARG(P >= 0 && d >= 1 && d <= 16 && k >= 1 && k <= 65535, "pccx_ae_decode_h2: unsupported P=%d d=%d k=%d", P, d, k);
    PCCX_CHECK_ARG(!pc_out || (centres && nrm_center && nrm_longest && S >= 1 && scale != 0.f),
                   "pccx_ae_decode_h2: pc_out needs centres, center, longest, S >= 1 and scale != 0");
    hipStream_t st = (hipStream_t)stream;
    const int ntiles = (P + 15) / 16;
    float *h2p = workspace;
    uint4 *h3 = (uint4 *)(workspace + (size_t)64 * ntiles * 64 * 4);
    float *pscale = workspace + (size_t)128 * ntiles * 64 * 4;
    const int rc = pccx_dec_head_launch(latent_q, P, d, ntiles, dec_blob, h2p, st);
    if (rc != PCCX_OK) return rc;
    hipLaunchKernelGGL(dec_h2_prep_kernel, dim3((ntiles + 3) / 4), dim3(256), 0, st, (const f32x4 *)h2p, latent_q, P, d, ntiles, 32768.0f, h3, pscale);
    PCCX_CHECK_LAUNCH();
    // four patch tiles per wave unless PCCX_DEC_H2_NT=2 asks for the two-tile form (same results; read per call for A/B and tests)
    const char *e = getenv("PCCX_DEC_H2_NT");
    if (e && atoi(e) == 2)
        hipLaunchKernelGGL((dec_main_h2_kernel<2>), dim3(dec_h2_grid(ntiles, k, 2)), dim3(256), 0, st, (const uint4 *)h3, (const float *)pscale, latent_q, P,
                           d, k, ntiles, h2_blob, patches_out, scale, centres, nrm_center, nrm_longest, S > 0 ? S : 1, (float)(1.0 - margin), pc_out);
    else
        hipLaunchKernelGGL((dec_main_h2_kernel<4>), dim3(dec_h2_grid(ntiles, k, 4)), dim3(256), 0, st, (const uint4 *)h3, (const float *)pscale, latent_q, P,
                           d, k, ntiles, h2_blob, patches_out, scale, centres, nrm_center, nrm_longest, S > 0 ? S : 1, (float)(1.0 - margin), pc_out);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}
