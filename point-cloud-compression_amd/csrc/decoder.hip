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
#include <math.h>

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

// grid: x = patch block (8 n-tiles = 128 patches), y = point p.  4 waves, wave w owns n-tiles 2w, 2w+1.
__global__ __launch_bounds__(256, 2) void dec_main_kernel(const f32x4 *__restrict__ h2p, const float *__restrict__ latent_q,
                                                          int P, int d, int k, int ntiles, const float *__restrict__ blob,
                                                          float *__restrict__ patches_out, float inv_scale_div,
                                                          const float *__restrict__ centres, const float *__restrict__ nrm_center,
                                                          const float *__restrict__ nrm_longest, int S, float one_minus_margin,
                                                          float *__restrict__ pc_out)
{
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, n = lane & 15;
    const int p = blockIdx.y;
    const int tile0 = blockIdx.x * 8 + 2 * w;
    __shared__ __attribute__((aligned(16))) f32x4 swt[2 * DEC_WS_CHUNK * 64];      // ring: one k-tile (8 m-tiles) per chunk
    const int wu = __builtin_amdgcn_readfirstlane(w);
    const WStreamT<DEC_WS_CHUNK> ws{blob + DEC_G_W(k) + (size_t)p * DEC_STREAM_CHUNKS * DEC_WS_CHUNK * 256, swt, DEC_STREAM_CHUNKS, lane, wu, false};
    ws.prologue();

    f32x4 acc[2][8];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        const f32x4 b = *(const f32x4 *)(blob + DEC_G_B + p * 128 + 16 * mt + 4 * g);
        acc[0][mt] = b; acc[1][mt] = b;
    }
    const int t0 = tile0 < ntiles ? tile0 : ntiles - 1, t1 = tile0 + 1 < ntiles ? tile0 + 1 : ntiles - 1;
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
    int f = DEC_STREAM_GEMM_FRAGS;                    // the inv_mlp fragments follow in the same LDS ring
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
    f32x4 m3[2][1];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) m3[nt][0] = *(const f32x4 *)(blob + DEC_M_B3 + 4 * g);
    dense_acc_stream<2, 1, 2>(ws, f, m2, m3);          // last layer: no ReLU (AE.py:27)
    ws.drain();

    // ---- epilogue: rows 0..2 of the last tile (g == 0, r = 0..2) are x,y,z of (patch, point p)
    if (g == 0) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
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
    hipLaunchKernelGGL(dec_main_kernel, dim3((ntiles + 7) / 8, k), dim3(256), 0, st, (const f32x4 *)workspace, latent_q, P, d, k,
                       ntiles, dec_blob, patches_out, scale, centres, nrm_center, nrm_longest, S > 0 ? S : 1,
                       (float)(1.0 - margin), pc_out);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}
