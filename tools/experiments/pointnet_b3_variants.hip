// EXPERIMENT RECORD (round 2), NOT part of libpccx.so: variants of the bf16x3 PointNet kernel measured with
// tools/experiments/pn_bench.py (drop this file into csrc/, declare pccx_launch_pn_b3v2 and route pccx_pn_forward_b3 to it to
// re-run).  Results per launch of 65 536 patches, same box, first design = 26.7-28.9 ms depending on the box:
//   register pipe across layers +-0; staggered DMA issue +6 %; interleaved split +-0; two workgroups of four waves per CU +1 %;
//   three / four ring buffers with counted vmcnt +3 % / +13 % (spills); 48-fragment chunks +6..+20 % (LDS offsets past the
//   16-bit immediate cost VGPRs); DMA issued by four waves only, with or without s_setprio, +-1 %;
//   NO DMA (garbage weights) 21.9-22.2 ms, and then removing every chunk barrier as well changes nothing (22.0 ms).
// In-kernel stamps (FLAGS & 16): per chunk boundary ~180 cycles waiting for the DMA, ~145 issuing 3 pieces, ~500 at the barrier
// (the partner wave of the SIMD keeps the matrix pipe busy meanwhile).  PMC: the DMA costs 10.6 % more wave cycles AND 5 % of
// clock (2.22 -> 2.11 GHz).  DESIGN.md section 4 draws the conclusions.
// pointnet_b3.hip -- PointNet + quantiser (pn_kit.py:98-144 + AE.py:43-45) on bf16x3 operands, second design.
//
// Same arithmetic as encoder.hip's pn_forward_b3_kernel (fp32 products formed from three bf16 pieces per operand, six
// v_mfma_f32_16x16x32_bf16 per product block, fp32 accumulate): workgroup = one patch, eight waves, one
// 16-point tile per wave per pass, the 1128-fragment weight stream shared by the eight waves through a two-buffer LDS ring
// filled by LDS-DMA.  What changes is how the waves keep the matrix pipe fed:
//
//   * the weight blocks are taken from the ring through a register PIPE that runs ahead of the MFMAs across the boundaries
//     of the layers (four blocks in registers or in flight), so no layer starts on a cold ds_read;
//   * the LDS-DMA pieces of the next chunk are not all issued at the chunk boundary by every wave at once (a piece costs
//     its issuer 60-185 cycles, MI355X_MICROARCH.md, and the eight waves run in lock step): each wave issues ONE piece per
//     weight block, waves 0-3 behind the first blocks of a chunk and waves 4-7 (their SIMD partners) behind later ones, so
//     that while one wave of a SIMD is held by the DMA issue the other one feeds the MFMAs;
//   * the split of the NEXT input pair into its bf16 planes (VALU) is interleaved with the MFMAs of the current one.
#include <math.h>

#include "blobs.h"
#include "common.h"
#include "mfma_chain.h"

#define PNB_BLOCKS (PN_B3_STREAM_FRAGS / 3)

// FLAGS: 1 = staggered DMA issue, 2 = ablation: no DMA at all (garbage results), 4 = ablation: planes split once per pass
// (garbage results), 8 = interleave the next split with the MFMAs
// DIAGNOSTIC build only (FLAGS & 16): cycle stamps around the chunk boundary, summed over all waves
__device__ unsigned long long pn_dbg_acc[8];

template <int FLAGS, int NW, int NB, int CH, int NI>
struct PnRing {
    static constexpr int NCH = PN_B3_STREAM_CHUNKS * PN_B3_CHUNK / CH;          // chunks per pass (the blob is padded to 1152 fragments)
    static_assert(NCH % NB == 0 && NCH * CH == PN_B3_STREAM_CHUNKS * PN_B3_CHUNK, "the chunks of a pass must fill the ring a whole number of times");
    static constexpr int PW = CH / NI;   // DMA pieces per ISSUING wave per chunk (waves 0 .. NI-1 issue)
    const float *g;                       // global stream (wave-uniform)
    f32x4 *lds;                           // [NB][CH][64]
    int lane, wave;
    mutable unsigned long long t_wait = 0, t_bar = 0, t_issue = 0, n_bound = 0;     // diagnostic build only

    __device__ __forceinline__ void issue_piece(int c, int q) const   // piece q (0..2) of this wave, chunk c of the pass (wraps)
    {
        if (FLAGS & 2) return;
        const int cc = c >= NCH ? c - NCH : c;
        const int fr = wave * PW + q;
        const char *src = (const char *)g + ((size_t)cc * CH + fr) * 1024 + (unsigned)lane * 16u;
        f32x4 *dst = lds + ((c % NB) * CH + fr) * 64;
        __builtin_amdgcn_global_load_lds((const void *)src, (lds_u32 *)(uintptr_t)dst, 16, 0, 0);
    }
    __device__ __forceinline__ void issue_chunk(int c) const
    {
        if (NI < NW && wave >= NI) return;                 // only the first NI waves carry the DMA issue
#pragma unroll
        for (int q = 0; q < PW; ++q) issue_piece(c, q);
    }
    // Before the first read of chunk c.  The ring runs NB - 1 chunks ahead: the pieces of chunks c+1 .. c+NB-2 are this wave's
    // youngest vector-memory operations and may stay in flight (vmcnt completes in order; any other younger operation only
    // makes the wait stricter); an LDS-DMA piece takes about a microsecond from issue to landing under load, more than one
    // chunk of MFMAs.  After the barrier every wave has finished reading chunk c-1, whose buffer chunk c+NB-1 now takes.
    __device__ __forceinline__ void boundary(int c) const
    {
        if (FLAGS & 16) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // LDS prefetches first, so that t_wait is the DMA's alone
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((FLAGS & 2) ? 0 : (NB - 2) * PW) : "memory");
            const unsigned long long t1 = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned long long t2 = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (!(FLAGS & 1)) issue_chunk(c + NB - 1);
            const unsigned long long t3 = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            t_wait += t1 - t0; t_bar += t2 - t1; t_issue += t3 - t2; n_bound += 1;
            return;
        }
        if (FLAGS & 64) return;                     // ablation: no boundary at all (with FLAGS & 2)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((FLAGS & 2) ? 0 : (NB - 2) * PW) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");              // no LDS read of chunk c may move above the barrier
        if (!(FLAGS & 1)) issue_chunk(c + NB - 1);
    }
    __device__ __forceinline__ void prologue() const
    {
#pragma unroll
        for (int i = 0; i < NB - 1; ++i) issue_chunk(i);
    }
    // called once per loaded block (f = its first fragment): the staggered issue of chunk c+1
    __device__ __forceinline__ void tick(int f) const
    {
        if (!(FLAGS & 1) || NW != 8 || CH != 24 || NI != 8) return;
        const int c = f / CH, pos = (f % CH) / 3;            // block position 0..7 inside the chunk
        if (pos < 3) {
            if (wave < 4) issue_piece(c + NB - 1, pos);
        } else if (pos < 6) {
            if (wave >= 4) issue_piece(c + NB - 1, pos - 3);
        }
    }
    __device__ __forceinline__ f32x4 get(int f) const
    {
        if ((f % CH) == 0) boundary(f / CH);
        return lds[(((f / CH) % NB) * CH + (f % CH)) * 64 + lane];
    }
};

template <int FLAGS, int NW, int NB, int CH, int NI>
struct PnPipe {
    PnRing<FLAGS, NW, NB, CH, NI> ring;
    bf16x8 r[4][3];
    __device__ __forceinline__ void load(int b)                      // block b of the pass -> register set b & 3
    {
#pragma unroll
        for (int p = 0; p < 3; ++p) r[b & 3][p] = __builtin_bit_cast(bf16x8, ring.get(3 * b + p));
        ring.tick(3 * b);
    }
};

// One quarter (two of the eight values) of b3_split8: q = 0..3
__device__ __forceinline__ void b3_split_quarter(const f32x4 &v0, const f32x4 &v1, int q, unsigned (&w)[3][4])
{
    f32x2v x = q < 2 ? f32x2v{v0[2 * q], v0[2 * q + 1]} : f32x2v{v1[2 * q - 4], v1[2 * q - 3]};
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const bf16x2 h = __builtin_convertvector(x, bf16x2);
        w[p][q] = __builtin_bit_cast(unsigned, h);
        if (p < 2) x = x - __builtin_convertvector(h, f32x2v);
    }
}
__device__ __forceinline__ void b3_words_to_planes(const unsigned (&w)[3][4], bf16x8 (&pl)[3])
{
#pragma unroll
    for (int p = 0; p < 3; ++p) pl[p] = __builtin_bit_cast(bf16x8, make_uint4(w[p][0], w[p][1], w[p][2], w[p][3]));
}

// acc[mt] += W[kt][mt] * in[kt] for one 16-point tile, blocks taken from the pipe in stream order ([kt][mt]).
// FILL(gi) is VALU work of the caller (the split of the next input) placed among the MFMAs of group gi.
template <int KT, int MT, int FLAGS, int NW, int NB, int CH, int NI, class FILL>
__device__ __forceinline__ void dense_b3_pipe(PnPipe<FLAGS, NW, NB, CH, NI> &pp, int &b, const bf16x8 (&in)[KT][3], f32x4 (&acc)[MT], FILL &&fill)
{
    constexpr int MG = MT >= 2 ? 2 : 1;
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};     // smallest products first
    constexpr int NG = KT * (MT / MG);
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
        const int kt = gi / (MT / MG), m0 = (gi % (MT / MG)) * MG;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int m = 0; m < MG; ++m)
                acc[m0 + m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pp.r[(b + m) & 3][PA[q]], in[kt][PB[q]], acc[m0 + m], 0, 0, 0);
        fill(gi);
        if (FLAGS & 8) {                    // one MFMA, then up to two VALU, repeated: the filler rides in the MFMA gaps
#pragma unroll
            for (int i = 0; i < 6 * MG; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MG; ++m)
            if (b + 4 + m < PNB_BLOCKS) pp.load(b + 4 + m);
        b += MG;
    }
}

template <int FLAGS, int NW, int NB, int CH, int NI>
__global__ __launch_bounds__(64 * NW, 8 / NW) void pn_forward_b3v2_kernel(const float *__restrict__ x, const float *__restrict__ feat, int K,
                                                                 const float *__restrict__ blob, const float *__restrict__ blob3, int d,
                                                                 float spread, float half_spread, float *__restrict__ latent_raw,
                                                                 float *__restrict__ latent, float *__restrict__ latent_q)
{
    __shared__ __attribute__((aligned(16))) f32x4 swt[NB * CH * 64];       // weight ring, 24 KiB per buffer
    __shared__ __attribute__((aligned(16))) float sbias[128 + 256 + 512 + 16];  // the four layers' biases (ENC_PN_B0..B3 are contiguous)
    __shared__ float smax[NW][16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, n = lane & 15;
    const size_t P = blockIdx.x;
    const float *xp = x + P * (size_t)K * 3;
    const int ntiles = K >> 4;
    const int wu = __builtin_amdgcn_readfirstlane(w);
    PnPipe<FLAGS, NW, NB, CH, NI> pp;
    pp.ring = PnRing<FLAGS, NW, NB, CH, NI>{blob3, swt, lane, wu};
    if (FLAGS & 32) {                                      // the issuing waves run ahead of their SIMD partners
        if (wu < NI) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
    }
    pp.ring.prologue();
    for (int i = tid; i < 128 + 256 + 512 + 16; i += 64 * NW) sbias[i] = blob[ENC_PN_B0 + i];
    __syncthreads();
    const float *sb0 = sbias, *sb1 = sbias + 128, *sb2 = sbias + 384, *sb3 = sbias + 896;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    auto nofill = [](int) {};

    const unsigned long long t_start = (FLAGS & 16) ? __builtin_amdgcn_s_memtime() : 0ull;
    f32x4 run;                                            // running max, channel 4g+r
    run[0] = run[1] = run[2] = run[3] = -INFINITY;
    const int passes = (ntiles + NW - 1) / NW;            // identical for all waves: barriers inside
    for (int it = 0; it < passes; ++it) {
        const int tile = it * NW + w;
        const bool valid = tile < ntiles;
        const int p = (valid ? tile : 0) * 16 + n;
        blob = opaque_uniform(blob);                      // keep bias and DMA addressing inside the pass (no LICM)
        pp.ring.g = opaque_uniform(blob3);
        int b = 0;                                        // block cursor of this pass (constant-folds)
#pragma unroll
        for (int i = 0; i < 4; ++i) pp.load(i);
        f32x4 a0[8];
        {
            f32x4 in[9];
#pragma unroll
            for (int kt = 0; kt < 8; ++kt) in[kt] = *(const f32x4 *)(feat + ((P * 8 + kt) * (size_t)K + p) * 16 + 4 * g);
            in[8][0] = g == 0 ? xp[3 * p] : 0.f;          // channels 128,129,130 = x,y,z (g == 0, r = 0..2)
            in[8][1] = g == 0 ? xp[3 * p + 1] : 0.f;
            in[8][2] = g == 0 ? xp[3 * p + 2] : 0.f;
            in[8][3] = 0.f;
            bf16x8 i0[5][3];
#pragma unroll
            for (int t = 0; t < 4; ++t) b3_split8(in[2 * t], in[2 * t + 1], i0[t]);
            b3_split8(in[8], zero, i0[4]);
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) a0[mt] = *(const f32x4 *)(sb0 + 16 * mt + 4 * g);
            dense_b3_pipe<5, 8>(pp, b, i0, a0, nofill);
        }
        f32x4 a1[16];
        {
            bf16x8 i1[4][3];
#pragma unroll
            for (int t = 0; t < 4; ++t) b3_split8(relu4(a0[2 * t]), relu4(a0[2 * t + 1]), i1[t]);
#pragma unroll
            for (int mt = 0; mt < 16; ++mt) a1[mt] = *(const f32x4 *)(sb1 + 16 * mt + 4 * g);
            dense_b3_pipe<4, 16>(pp, b, i1, a1, nofill);
        }
#pragma unroll
        for (int mt = 0; mt < 16; ++mt) a1[mt] = relu4(a1[mt]);
        f32x4 a3[1];
        a3[0] = *(const f32x4 *)(sb3 + 4 * g);
        bf16x8 pl_once[1][3];
        if (FLAGS & 4) b3_split8(a1[0], a1[1], pl_once[0]);
#pragma clang loop unroll(full)
        for (int h = 0; h < 2; ++h) {                     // layer 2 in two halves of 16 output tiles (64 accumulator VGPRs each)
            f32x4 a2[16];
#pragma unroll
            for (int mt = 0; mt < 16; ++mt) a2[mt] = *(const f32x4 *)(sb2 + 16 * (16 * h + mt) + 4 * g);
            bf16x8 pl[1][3];
            unsigned wn[3][4];
            if (!(FLAGS & 4)) b3_split8(a1[0], a1[1], pl[0]);
#pragma clang loop unroll(full)
            for (int kt = 0; kt < 8; ++kt) {              // k-outer: every input pair is split once per half
                if (FLAGS & 4) {
                    dense_b3_pipe<1, 16>(pp, b, pl_once, a2, nofill);
                } else if (FLAGS & 8) {
                    // the planes of pair kt+1 are produced among the MFMAs of pair kt (a quarter per group, groups 0..3)
                    auto fill = [&](int gi) { if (kt < 7 && gi < 4) b3_split_quarter(a1[2 * kt + 2], a1[2 * kt + 3], gi, wn); };
                    dense_b3_pipe<1, 16>(pp, b, pl, a2, fill);
                    if (kt < 7) b3_words_to_planes(wn, pl[0]);
                } else {
                    dense_b3_pipe<1, 16>(pp, b, pl, a2, nofill);
                    if (kt < 7) b3_split8(a1[2 * kt + 2], a1[2 * kt + 3], pl[0]);
                }
            }
#pragma clang loop unroll(full)
            for (int kt = 0; kt < 8; ++kt) {              // layer 3 over these 256 channels (no ReLU after it, AE.py:17)
                if (FLAGS & 4) {
                    dense_b3_pipe<1, 1>(pp, b, pl_once, a3, nofill);
                } else {
                    b3_split8(relu4(a2[2 * kt]), relu4(a2[2 * kt + 1]), pl[0]);
                    dense_b3_pipe<1, 1>(pp, b, pl, a3, nofill);
                }
            }
        }
        // an all-padding last chunk (CH = 24) keeps the ring's phase over the pass; its boundary also starts a chunk of the next pass
        constexpr int NCHK = PN_B3_STREAM_CHUNKS * PN_B3_CHUNK / CH;
        if ((PN_B3_STREAM_FRAGS + CH - 1) / CH < NCHK) {
            pp.ring.boundary(NCHK - 1);
            if (FLAGS & 1) pp.ring.issue_chunk(NCHK - 1 + NB - 1);
        }
        if (valid)
#pragma unroll
            for (int r = 0; r < 4; ++r) run[r] = fmaxf(run[r], row16_max(a3[0][r]));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no DMA may land after the workgroup retires
    if ((FLAGS & 16) && lane == 0) {
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        atomicAdd(&pn_dbg_acc[0], pp.ring.t_wait);
        atomicAdd(&pn_dbg_acc[1], pp.ring.t_bar);
        atomicAdd(&pn_dbg_acc[2], pp.ring.t_issue);
        atomicAdd(&pn_dbg_acc[3], pp.ring.n_bound);
        atomicAdd(&pn_dbg_acc[4], t_end - t_start);
        atomicAdd(&pn_dbg_acc[5], 1ull);
    }
    if (n == 0)
#pragma unroll
        for (int r = 0; r < 4; ++r) smax[w][4 * g + r] = run[r];
    __syncthreads();
    if (tid < 16 && tid < d) {
        float m = smax[0][tid];
#pragma unroll
        for (int k8 = 1; k8 < NW; ++k8) m = fmaxf(m, smax[k8][tid]);                                    // torch.max(points, 2)
        const float s = 1.0f / (1.0f + expf(-m));
        const float y = __fsub_rn(__fmul_rn(s, spread), half_spread);
        latent_raw[P * d + tid] = m;
        latent[P * d + tid] = y;
        latent_q[P * d + tid] = rintf(y);
    }
}

#define PN_B3V2_LAUNCH(F, NW, NB, CH, NI)                                                                                                    \
    hipLaunchKernelGGL((pn_forward_b3v2_kernel<F, NW, NB, CH, NI>), dim3(P), dim3(64 * NW), 0, st, patches, feat, K, enc_blob, pn_b3_blob, d,  \
                       spread, half, latent_raw, latent, latent_q)

int pccx_launch_pn_b3v2(int variant, const float *patches, const float *feat, int P, int K, const float *enc_blob, const float *pn_b3_blob,
                        int d, float spread, float half, float *latent_raw, float *latent, float *latent_q, hipStream_t st)
{
    switch (variant) {
    case 0: PN_B3V2_LAUNCH(0, 8, 2, 24, 8); break;        // as the first design, with the register pipe
    case 1: PN_B3V2_LAUNCH(0, 8, 2, 24, 4); break;        // DMA issued by waves 0-3 only
    case 2: PN_B3V2_LAUNCH(32, 8, 2, 24, 4); break;       // ... and those waves at higher priority
    case 3: PN_B3V2_LAUNCH(0, 8, 2, 48, 8); break;        // 48-fragment chunks (24 boundaries per pass)
    case 4: PN_B3V2_LAUNCH(0, 8, 2, 48, 4); break;
    case 5: PN_B3V2_LAUNCH(32, 8, 2, 48, 4); break;
    case 6: PN_B3V2_LAUNCH(0, 8, 3, 48, 4); break;        // three buffers of 48
    case 7: PN_B3V2_LAUNCH(32, 8, 2, 24, 8); break;       // priority alone
    case 8: PN_B3V2_LAUNCH(66, 8, 2, 24, 8); break;       // no DMA, no chunk boundaries (garbage)
    case 10: PN_B3V2_LAUNCH(66, 4, 2, 24, 4); break;      // the same, two workgroups of four waves per CU
    case 9: PN_B3V2_LAUNCH(2, 8, 2, 24, 8); break;        // no DMA (garbage)
    default: pccx_set_error("pccx_pn_forward_b3: unknown variant %d", variant); return PCCX_ERR_ARG;
    }
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// DIAGNOSTIC: read (and clear) the cycle sums of the FLAGS & 16 build: wait, barrier, issue, boundaries, wave lifetime, waves
extern "C" PCCX_API int pccx_debug_pn_b3_stamps(unsigned long long *out8)
{
    PCCX_CHECK_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(pn_dbg_acc), sizeof(unsigned long long) * 8));
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    PCCX_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(pn_dbg_acc), z, sizeof(z)));
    return PCCX_OK;
}
