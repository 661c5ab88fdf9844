// prob.hip -- AE.ConditionalProbabilityModel (AE.py:87-123) + pn_kit.pmf_to_cdf (pn_kit.py:452-461)
// + torchac's float->16-bit CDF conversion, for a batch of clouds.
//
// Workgroup = one cloud (S decoded patch centres, S % 16 == 0), 4 waves.
//   pass 1: PointNet 3->64->128->256 on every centre tile, max over all S centres -> LDS feature
//   pass 1b: the first Conv's contribution of the 256 feature channels, b0 + W0[:, :256] feat -- the same 512 numbers for every
//           centre of the cloud (AE.py:115-116 concatenates the repeated feature) -- ONCE per cloud, a quarter of the rows per wave,
//           by the same MFMA chain in the same order a centre's column would run (bit-identical to evaluating it per centre,
//           which rounds 1-3 did: 28 % of pass 2's matrix work)
//   pass 2: per tile, Conv 259->512->512->d*L (that vector + the xyz k-tile), logits -> LDS,
//           softmax over L per (centre, latent dim), cumsum, clamp, integer CDF.
// Runs on both sides of the codec from bit-identical centres with a fixed summation order, so the
// encoder's and the decoder's integer CDFs are identical (a range-coder requirement).
#include <math.h>

#include "blobs.h"
#include "common.h"
#include "mfma_chain.h"

__global__ __launch_bounds__(256, 2) void prob_forward_kernel(const float *__restrict__ centres, int S, int d, int L,
                                                              const float *__restrict__ blob, float *__restrict__ pmf,
                                                              float *__restrict__ cdf, int32_t *__restrict__ cdf_int)
{
    __shared__ __attribute__((aligned(16))) f32x4 swt[4 * PRB_WS_CHUNK * 64];   // 32 KiB weight ring, four chunks deep
    __shared__ __attribute__((aligned(16))) float sfeat[256];
    __shared__ __attribute__((aligned(16))) float su[512];
    __shared__ float smax[4][256];
    __shared__ __attribute__((aligned(16))) float slog[4][16][128];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, n = lane & 15;
    const size_t b = blockIdx.x;
    const float *cp = centres + b * (size_t)S * 3;
    const int ntiles = S >> 4;
    const int wu = __builtin_amdgcn_readfirstlane(w);
    WStreamT<PRB_WS_CHUNK, 4, 4> ws{blob + PRB_STREAM, swt, PRB_STREAM_CHUNKS, lane, wu, true};
    ws.prologue();                                        // the first chunks of model_mlp arrive while model_pn runs

    // ---- pass 1: model_pn (AE.py:96,112)
    f32x4 run[16];
#pragma unroll
    for (int mt = 0; mt < 16; ++mt) run[mt][0] = run[mt][1] = run[mt][2] = run[mt][3] = -INFINITY;
    for (int tile = w; tile < ntiles; tile += 4) {
        const float *bl = opaque_uniform(blob);
        const int c = tile * 16 + n;
        f32x4 in[1][1];
        in[0][0][0] = g == 0 ? cp[3 * c] : 0.f;
        in[0][0][1] = g == 0 ? cp[3 * c + 1] : 0.f;
        in[0][0][2] = g == 0 ? cp[3 * c + 2] : 0.f;
        in[0][0][3] = 0.f;
        f32x4 a0[1][4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) a0[0][mt] = *(const f32x4 *)(bl + PRB_P_B0 + 16 * mt + 4 * g);
        dense_acc<1, 4, 1, 4>((const f32x4 *)(bl + PRB_P_W0), lane, in, a0);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) a0[0][mt] = relu4(a0[0][mt]);
        f32x4 a1[1][8];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) a1[0][mt] = *(const f32x4 *)(bl + PRB_P_B1 + 16 * mt + 4 * g);
        dense_acc<4, 8, 1, 8>((const f32x4 *)(bl + PRB_P_W1), lane, a0, a1);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) a1[0][mt] = relu4(a1[0][mt]);
        f32x4 a2[1][16];
#pragma unroll
        for (int mt = 0; mt < 16; ++mt) a2[0][mt] = *(const f32x4 *)(bl + PRB_P_B2 + 16 * mt + 4 * g);
        dense_acc<8, 16, 1, 16>((const f32x4 *)(bl + PRB_P_W2), lane, a1, a2);
#pragma unroll
        for (int mt = 0; mt < 16; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) run[mt][r] = fmaxf(run[mt][r], fmaxf(row16_max(a2[0][mt][r]), 0.f));
    }
    if (n == 0)
#pragma unroll
        for (int mt = 0; mt < 16; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) smax[w][16 * mt + 4 * g + r] = run[mt][r];
    __syncthreads();
    sfeat[tid] = fmaxf(fmaxf(smax[0][tid], smax[1][tid]), fmaxf(smax[2][tid], smax[3][tid]));
    __syncthreads();

    // ---- pass 1b: u = b0 + W0[:, feature channels] feat, rows 128 w .. 128 w + 127 in wave w (fragments from L2, as pass 1's)
    {
        const float *bl = opaque_uniform(blob);
        f32x4 fin[1][16];
#pragma unroll
        for (int kt = 0; kt < 16; ++kt) fin[0][kt] = *(const f32x4 *)(sfeat + 16 * kt + 4 * g);
        f32x4 u[1][8];
#pragma unroll
        for (int m = 0; m < 8; ++m) u[0][m] = *(const f32x4 *)(bl + PRB_M_B0 + 16 * (8 * wu + m) + 4 * g);
        dense_acc<16, 8, 1, 32>((const f32x4 *)(bl + PRB_M_W0), lane, fin, u, 0, 8 * wu);
        if (n == 0)
#pragma unroll
            for (int m = 0; m < 8; ++m) *(f32x4 *)(su + 16 * (8 * wu + m) + 4 * g) = u[0][m];
    }
    __syncthreads();

    // ---- pass 2: model_mlp (AE.py:97-105,115-118) + softmax (AE.py:120) + cdf
    const int Lp = L + 1;
    for (int tile0 = 0; tile0 < ntiles; tile0 += 4) {
        const int tile = tile0 + w;
        {
            // All four waves walk the same weight sequence on their own tile (an idle wave of the last round walks it on a
            // clamped tile and discards the result: the ring's barriers need every wave), streamed L2 -> LDS three chunks ahead.
            const float *bl = opaque_uniform(blob);
            ws.g = bl + PRB_STREAM;
            // this pass's lane indices from a laundered lane id: nothing computed for pass 1 or for the softmax below is carried through
            // the 160 accumulator registers of the two wide layers (round 3: ten values spilled to scratch around them)
            int lane2 = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            asm volatile("" : "+v"(lane2));
            const int g = lane2 >> 4, n = lane2 & 15;
            ws.lane = lane2;
            const int c = (tile < ntiles ? tile : ntiles - 1) * 16 + n;
            int f = 0;
            f32x4 a0[1][32];
#pragma unroll
            for (int mt = 0; mt < 32; ++mt) a0[0][mt] = *(const f32x4 *)(su + 16 * mt + 4 * g);       // bias + the feature k-tiles (pass 1b)
            {
                f32x4 in[1][1];
                in[0][0][0] = g == 0 ? cp[3 * c] : 0.f;
                in[0][0][1] = g == 0 ? cp[3 * c + 1] : 0.f;
                in[0][0][2] = g == 0 ? cp[3 * c + 2] : 0.f;
                in[0][0][3] = 0.f;
                dense_acc_stream<1, 32, 1>(ws, f, in, a0);                                              // the xyz k-tile, last as before
            }
#pragma unroll
            for (int mt = 0; mt < 32; ++mt) a0[0][mt] = relu4(a0[0][mt]);
            f32x4 a2[1][8];
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) a2[0][mt] = *(const f32x4 *)(bl + PRB_M_B2 + 16 * mt + 4 * g);
#pragma clang loop unroll(full)
            for (int mp = 0; mp < 16; ++mp) {
                f32x4 a1[1][2];
#pragma unroll
                for (int m = 0; m < 2; ++m) a1[0][m] = *(const f32x4 *)(bl + PRB_M_B1 + 16 * (2 * mp + m) + 4 * g);
                dense_acc_stream<32, 2, 1>(ws, f, a0, a1);
#pragma unroll
                for (int m = 0; m < 2; ++m) a1[0][m] = relu4(a1[0][m]);
                dense_acc_stream<2, 8, 1>(ws, f, a1, a2);
            }
            if (tile < ntiles)
#pragma unroll
                for (int mt = 0; mt < 8; ++mt) *(f32x4 *)(&slog[w][n][16 * mt + 4 * g]) = a2[0][mt];
        }
        __syncthreads();
        // one thread per (centre, latent dim): the wave's own tile, 16 centres x 16 dims
        if (tile < ntiles) {
            for (int e = lane; e < 16 * d; e += 64) {
                const int cn = e / d, i = e % d;
                const float *lg = &slog[w][cn][i * L];          // output.view(B,S,d,L): channel i*L + l
                float mx = -INFINITY;
                for (int l = 0; l < L; ++l) mx = fmaxf(mx, lg[l]);
                float sum = 0.f;
                for (int l = 0; l < L; ++l) sum += expf(lg[l] - mx);
                const size_t row = (b * S + (size_t)tile * 16 + cn) * d + i;
                float run_c = 0.f;
                if (cdf) cdf[row * Lp] = 0.f;
                if (cdf_int) cdf_int[row * Lp] = 0;
                for (int l = 0; l < L; ++l) {
                    const float pv = expf(lg[l] - mx) / sum;
                    if (pmf) pmf[row * L + l] = pv;
                    run_c = run_c + pv;                                        // cumsum (pn_kit.py:453)
                    const float cv = fminf(run_c, 1.0f);                       // clamp(max=1) (pn_kit.py:460)
                    if (cdf) cdf[row * Lp + l + 1] = cv;
                    // torchac: round(cdf * (2^16 - (Lp-1))) + arange(Lp), kept to 16 bits
                    if (cdf_int) cdf_int[row * Lp + l + 1] = ((int)rintf(cv * (float)(65536 - (Lp - 1))) + (l + 1)) & 0xFFFF;
                }
            }
        }
        __syncthreads();
    }
    ws.drain();                                           // no DMA may land after the workgroup retires
}

extern "C" int pccx_prob_forward(const float *centres, int B, int S, int d, int L, const float *prob_blob, float *pmf,
                                 float *cdf, int32_t *cdf_int, void *stream)
{
    if (B == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(centres && prob_blob, "pccx_prob_forward: null pointer");
    PCCX_CHECK_ARG(pmf || cdf || cdf_int, "pccx_prob_forward: no output requested");
    PCCX_CHECK_ARG(B >= 0 && S >= 16 && S % 16 == 0, "pccx_prob_forward: need S %% 16 == 0 (S=%d)", S);
    PCCX_CHECK_ARG(d >= 1 && d <= 16 && L >= 1 && L <= 15 && d * L <= 128, "pccx_prob_forward: unsupported d=%d L=%d", d, L);
    hipLaunchKernelGGL(prob_forward_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, centres, S, d, L, prob_blob, pmf, cdf,
                       cdf_int);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}
