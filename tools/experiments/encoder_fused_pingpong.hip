// EXPERIMENT RECORD (round 2), NOT part of libpccx.so: the fused SetAbstraction + PointNet kernel with its two wave groups
// running ONE PHASE APART (a SetAbstraction wave and a PointNet wave on every SIMD), pasted from csrc/encoder_fused.hip at
// the time of the measurement (it uses that file's helpers).  Correct (bit-identical to the two-kernel path, 48 barriers per
// role per phase, no deadlock) but SLOWER: 66 ms per 1024 clouds against 53.6 ms for the lock-step schedule.  The matrix pipe
// stayed at ~59 % in both: pairing a VALU-heavy with an MFMA-heavy wave does not raise it, and the ramp (one idle role in five
// phases), the doubled LDS-DMA of a four-wave ring and 123 spilled VGPRs cost the rest.  DESIGN.md section 4 has the reading:
// the bf16x3 kernels with ONE 16-point tile per wave are bound by instruction issue, not by the pipe.
// ------------------------------------------------------------------------------------------------------------------
// PING-PONG schedule.  In the kernel above all eight waves are in the same phase at the same time: eight waves of
// SetAbstraction (VALU-heavy: kNN gathers, bf16 splits, the neighbour max) and then eight waves of PointNet (MFMA-heavy,
// paced by the weight ring), so the two waves of a SIMD always want the same resource.  Here the workgroup's two halves run
// ONE PHASE APART: while waves 0-3 run the PointNet pass of their tiles, waves 4-7 (their SIMD partners) run SetAbstraction
// for the next tiles, and vice versa; each SIMD then always holds one MFMA-bound and one VALU-bound wave.
//   phase ph:  group 0 (waves 0-3)   ph = 2s: SA(set s)    ph = 2s+1: PN(set s)
//              group 1 (waves 4-7)   ph = 2s+1: SA(set s)  ph = 2s+2: PN(set s)        set s, group q, wave i -> tile 8s + 4q + i
// gfx950 has one barrier per workgroup, and the ring needs one per chunk; every role therefore executes EXACTLY 48 s_barrier
// per phase: the PointNet pass its 47 chunk boundaries + 1 closing barrier, SetAbstraction 6 per pair of points x 8 pairs (bare
// s_barrier between its MFMA groups, no counter waits), an idle role (first / last phase) 48 bare ones.
// The ring (4 waves x 6 pieces per chunk) starts cold in every phase; the staging rows have their own LDS (not aliased).
// ------------------------------------------------------------------------------------------------------------------
#define PP_BARRIERS 48
static_assert((PN_B3_STREAM_FRAGS + PN_B3_CHUNK - 1) / PN_B3_CHUNK + 1 == PP_BARRIERS, "PointNet pass: 47 chunk boundaries + 1");

// dense_b3 (weights resident in LDS) with a bare s_barrier after every BAR_EVERY groups of MFMAs
template <int KT, int MT, int NT, bool SWAP, int BAR_EVERY>
__device__ __forceinline__ void dense_b3_bar(const f32x4 *w, int lane, const bf16x8 (&in)[NT][KT][3], f32x4 (&acc)[NT][MT])
{
    constexpr int MG = MT >= 2 ? 2 : 1;
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
    constexpr int NG = KT * (MT / MG);
    bf16x8 cur[MG][3], nxt[MG][3];
#pragma unroll
    for (int m = 0; m < MG; ++m)
#pragma unroll
        for (int p = 0; p < 3; ++p) cur[m][p] = __builtin_bit_cast(bf16x8, w[(size_t)(m * 3 + p) * 64 + lane]);
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
        if (gi + 1 < NG) {
#pragma unroll
            for (int m = 0; m < MG; ++m)
#pragma unroll
                for (int p = 0; p < 3; ++p) nxt[m][p] = __builtin_bit_cast(bf16x8, w[(size_t)(((gi + 1) * MG + m) * 3 + p) * 64 + lane]);
        }
        const int kt = gi / (MT / MG), m0 = (gi % (MT / MG)) * MG;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int m = 0; m < MG; ++m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt][m0 + m] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(in[nt][kt][PB[q]], cur[m][PA[q]], acc[nt][m0 + m], 0, 0, 0)
                                           : __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[m][PA[q]], in[nt][kt][PB[q]], acc[nt][m0 + m], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (BAR_EVERY > 0 && (gi + 1) % BAR_EVERY == 0) __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int m = 0; m < MG; ++m)
#pragma unroll
            for (int p = 0; p < 3; ++p) cur[m][p] = nxt[m][p];
    }
}

__host__ __device__ inline size_t pp_lds_bytes(int K)
{
    return (size_t)(FU_W1_FRAGS + FU_W2_FRAGS) * 1024 + (64 + 128) * 4 + (size_t)K * 12 + (size_t)K * 32 + (size_t)2 * PN_B3_CHUNK * 1024 +
           (size_t)4 * FU_STAGE_WAVE * 4 + 8 * 16 * 4;      // ring + FOUR staging blocks: only SA-role waves stage, block w & 3
}

__global__ __launch_bounds__(512, 1) void sa_pn_pingpong_b3_kernel(const float *__restrict__ x, int K, const float *__restrict__ blob,
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
    f32x4 *swt = (f32x4 *)(nbr16 + 16 * K);                              // PointNet weight ring (2 x 24 KiB)
    float *stage_all = (float *)(swt + 2 * PN_B3_CHUNK * 64);            // four staging blocks
    float (*smax)[16] = (float (*)[16])(stage_all + 4 * FU_STAGE_WAVE);

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, n = lane & 15;
    const size_t P = blockIdx.x;
    const float *xp = x + P * (size_t)K * 3;
    const int ntiles = K >> 4;
    const int wu = __builtin_amdgcn_readfirstlane(w);
    const int grp = wu >> 2, wi = wu & 3;
    float *stage = stage_all + wi * FU_STAGE_WAVE;

    {   // stage SetAbstraction weights + the patch
        const f32x4 *gw1 = (const f32x4 *)sa3, *gw2 = (const f32x4 *)sa3 + FU_W1_FRAGS * 64;
        for (int i = tid; i < FU_W1_FRAGS * 64; i += 512) sw1[i] = gw1[i];
        for (int i = tid; i < FU_W2_FRAGS * 64; i += 512) sw2[i] = gw2[i];
        if (tid < 64) sb1[tid] = blob[ENC_SA_B1 + tid];
        if (tid < 128) sb2[tid] = blob[ENC_SA_B2 + tid];
        for (int i = tid; i < 3 * K; i += 512) sx[i] = xp[i];
    }
    __syncthreads();

    // ---- kNN-16 inside the patch (pn_kit.py:190): one point per thread
    unsigned jmask = 15u;
    while ((int)jmask < K - 1) jmask = 2u * jmask + 1u;
    for (int i = tid; i < K; i += 512) {
        const float px = sx[3 * i], py = sx[3 * i + 1], pz = sx[3 * i + 2];
        unsigned tk[17];
#pragma unroll
        for (int s = 0; s < 17; ++s) tk[s] = 0xFFFFFFFFu;
        for (int j0 = 0; j0 < K; j0 += 4) {
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
        if (((tk[15] ^ tk[16]) & ~jmask) != 0u) {
#pragma unroll
            for (int s = 0; s < 16; ++s) nbr16[i * 16 + s] = (unsigned short)(tk[s] & jmask);
            continue;
        }
        float td[16];
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
    __syncthreads();

    const float w0a = blob[ENC_SA_W0B0 + 4 * n + g], w0b = blob[ENC_SA_W0B0 + 4 * (16 + n) + g];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 run;                                            // running max, channel 4g+r
    run[0] = run[1] = run[2] = run[3] = -INFINITY;
    const int sets = (ntiles + 7) / 8;                    // tile sets per group
    bf16x8 i0p[1][5][3];                                  // the tile handed from this wave's SA phase to its next PN phase
    bool tile_valid = false;

    for (int ph = 0; ph < 2 * sets + 1; ++ph) {           // identical for all waves: 48 barriers per phase whatever the role
        const int rel = ph - grp;                         // group q runs SA at rel = 2s, PN at rel = 2s + 1
        const int s = rel >> 1;
        const bool in_range = rel >= 0 && s < sets;
        if (in_range && (rel & 1) == 0) {
            // ================= SetAbstraction for this wave's tile of set s, 6 barriers per pair of points
            const int tile = 8 * s + 4 * grp + wi;
            tile_valid = tile < ntiles;
            const int p0 = (tile_valid ? tile : 0) * 16;  // a wave without a tile recomputes tile 0 and discards it
            for (int i0 = p0; i0 < p0 + 16; i0 += 2) {
                f32x4 h0[2][2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int i = i0 + nt;
                    const int j = nbr16[i * 16 + n];
                    const float relc = g < 3 ? __fsub_rn(sx[3 * j + g], sx[3 * i + g]) : 1.0f;
                    h0[nt][0] = relu4(mfma16(w0a, relc, zero4));
                    h0[nt][1] = relu4(mfma16(w0b, relc, zero4));
                }
                f32x4 a1[2][4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) a1[0][mt] = a1[1][mt] = *(const f32x4 *)(sb1 + 16 * mt + 4 * g);
                f32x4 a2[2][8];
#pragma unroll
                for (int mt = 0; mt < 8; ++mt) {
                    const float bv = sb2[16 * mt + n];
                    f32x4 b4 = {bv, bv, bv, bv};
                    a2[0][mt] = b4; a2[1][mt] = b4;
                }
                bf16x8 i1[2][1][3];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) b3_split8(h0[nt][0], h0[nt][1], i1[nt][0]);
                dense_b3_bar<1, 4, 2, false, 0>(sw1, lane, i1, a1);                   // conv1
                __builtin_amdgcn_s_barrier();                                         // 1
                bf16x8 i2[2][2][3];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int t = 0; t < 2; ++t) b3_split8(relu4(a1[nt][2 * t]), relu4(a1[nt][2 * t + 1]), i2[nt][t]);
                dense_b3_bar<2, 8, 2, true, 2>(sw2, lane, i2, a2);                    // conv2, transposed: 8 groups, barriers 2..5
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    float mx[2];
                    max16_of_8_transposed_tiles(a2[nt], mx);
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2)
                        stage[(i0 + nt - p0) * FU_STAGE_STRIDE + 16 * (2 * g + s2) + n] = fmaxf(mx[s2], 0.f);
                }
                __builtin_amdgcn_s_barrier();                                         // 6
            }
            // hand-over: the wave's own rows, read back as PointNet's B operand and split into planes
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
        } else if (in_range) {
            // ================= PointNet pass over the tile of the previous phase: 47 chunk boundaries + 1 closing barrier
            blob = opaque_uniform(blob);
            WStreamT<PN_B3_CHUNK, 2, 4> ws{opaque_uniform(pn3), swt, (PN_B3_STREAM_FRAGS + PN_B3_CHUNK - 1) / PN_B3_CHUNK, lane, wi, false};
            ws.prologue();
            int f = 0;                                    // fragment cursor of this pass (constant-folds)
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
            for (int h = 0; h < 2; ++h) {
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
                for (int kt = 0; kt < 8; ++kt) {
                    bf16x8 pl[1][1][3];
                    b3_split8(relu4(a2p[0][2 * kt]), relu4(a2p[0][2 * kt + 1]), pl[0][0]);
                    dense_b3_stream<1, 1, 1>(ws, f, pl, a3);
                }
            }
            ws.drain();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                 // 48: every reader of the ring is done before the next phase refills it
            if (tile_valid)
#pragma unroll
                for (int r = 0; r < 4; ++r) run[r] = fmaxf(run[r], row16_max(a3[0][0][r]));
        } else {
            // ================= no role in this phase (group 1 in the first phase, group 0 in the last)
            for (int i = 0; i < PP_BARRIERS; ++i) __builtin_amdgcn_s_barrier();
        }
    }
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

