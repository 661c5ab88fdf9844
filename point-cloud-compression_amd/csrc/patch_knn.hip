// patch_knn.hip -- the 16 nearest neighbours of every point INSIDE its own patch, the selection SetAbstraction starts with
// (pn_kit.py:186-190: knn_points(xyz, xyz, K=16) on the (B, 256, 3) patch), as a kernel of its own.
//
// Why not inside the encoder kernel (where round 2 had it): the selection is pure vector-ALU work, 27 instructions per
// (point, candidate) pair, and the fused encoder runs at TWO waves per SIMD (its matrix phases need 256 registers).  At that
// occupancy a SIMD issues one vector instruction per ~3.5 cycles (measured, tools/experiments/r3/ub/ub_med3.hip: 5.5 cycles for a
// lone wave, 6.9 per wave with two), and nothing else can run beside it -- the phase was 6.5 ms of the 52.7 ms kernel.  Here the
// same instruction stream runs at 8 waves per SIMD (44 registers, 3 KB of LDS per workgroup), and the neighbour table travels to
// the encoder as one byte per index (K <= 256; two above): 4 KB per patch, written once and read once.
//
// Selection rule (bit-for-bit the round-2 in-kernel one, which the oracle tests pin): key = distance bits with the candidate
// index in the low log2(K) bits, 17 sorted keys by v_med3_u32; exact whenever ranks 16 and 17 differ in the kept distance bits,
// otherwise the thread falls back to the exact (distance, index) two-pass selection.  Only the SET matters (max-pool follows).
#include <math.h>

#include "common.h"

__device__ __forceinline__ unsigned pk_umed3(unsigned a, unsigned b, unsigned c)
{
    unsigned r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

template <typename IDX>
__global__ __launch_bounds__(256) void patch_knn16_kernel(const float *__restrict__ x, int npatches, int K, unsigned char *__restrict__ nbr_bytes,
                                                          size_t patch_stride)
{
    extern __shared__ __attribute__((aligned(16))) float sx[];      // [3K]
    const int tid = threadIdx.x;
    unsigned jmask = 15u;
    while ((int)jmask < K - 1) jmask = 2u * jmask + 1u;
    for (size_t P = blockIdx.x; P < (size_t)npatches; P += gridDim.x) {
        const float *xp = x + P * (size_t)K * 3;
        for (int i = tid; i < 3 * K; i += 256) sx[i] = xp[i];
        __syncthreads();
        for (int i = tid; i < K; i += 256) {
            const float px = sx[3 * i], py = sx[3 * i + 1], pz = sx[3 * i + 2];
            unsigned tk[17];
#pragma unroll
            for (int s = 0; s < 17; ++s) tk[s] = 0xFFFFFFFFu;
#ifndef PK_UNROLL
#define PK_UNROLL 4
#endif
#ifdef PK_SGPR
            const float *__restrict__ cand = xp;         // wave-uniform address: scalar loads, candidates arrive in SGPRs
#else
            const float *cand = sx;                      // LDS broadcast reads
#endif
            // Candidate ORDER and SKIPS (round 4).  The 17 smallest keys do not depend on the order the candidates arrive in (the
            // keys are distinct: the index is part of the key), and a key >= tk[16] leaves the ladder as it is -- so a group of
            // PK_UNROLL candidates whose keys are >= tk[16] in EVERY lane of the wave is skipped outright (one ballot), bit-identical
            // by construction.  To make that happen the wave starts with the 16-candidate chunks around its own points and walks
            // outwards in both directions: a patch is sorted by distance from its centre (compress.py:105-108, knn_points returns
            // sorted neighbours), a wave's 64 queries are a ring of it, after the nearby rings tk[16] is close to its final
            // value, and the far rings' points are farther than that from every query of the ring (|r_c - r_q| <= d(c, q)).
            const int nch = K / 16;
            const int c0 = __builtin_amdgcn_readfirstlane(i >> 4) & ~3;   // first chunk of the wave's own 64 points
            int up = c0, down = c0 - 1;
            for (int t = 0; t < nch; ++t) {
                // own four chunks first, then alternately one chunk above / one below while both sides last
                int ch;
                if (t < 4 || down < 0 || (up < nch && ((t & 1) == 0))) ch = up < nch ? up++ : down--;
                else ch = down--;
                for (int j0 = 16 * ch; j0 < 16 * ch + 16; j0 += PK_UNROLL) {  // PK_UNROLL candidates in flight
                    float d[PK_UNROLL];
                    unsigned key[PK_UNROLL];
#pragma unroll
                    for (int u = 0; u < PK_UNROLL; ++u)
                        d[u] = pccx_sqdist(px, py, pz, cand[3 * (j0 + u)], cand[3 * (j0 + u) + 1], cand[3 * (j0 + u) + 2]);
                    unsigned kmin = 0xFFFFFFFFu;
#pragma unroll
                    for (int u = 0; u < PK_UNROLL; ++u) {
                        key[u] = (__float_as_uint(d[u]) & ~jmask) | (unsigned)(j0 + u);
                        kmin = min(kmin, key[u]);
                    }
                    if (__ballot(kmin < tk[16]) == 0ull) continue;
#pragma unroll
                    for (int u = 0; u < PK_UNROLL; ++u) {
#pragma unroll
                        for (int s = 16; s >= 1; --s) tk[s] = pk_umed3(tk[s - 1], key[u], tk[s]);
                        tk[0] = min(tk[0], key[u]);
                    }
                }
            }
            IDX out[16];
            if (((tk[15] ^ tk[16]) & ~jmask) != 0u) {
#pragma unroll
                for (int s = 0; s < 16; ++s) out[s] = (IDX)(tk[s] & jmask);
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
#pragma unroll
                for (int s = 0; s < 16; ++s) out[s] = 0;
                for (int j = 0; j < K; ++j) {
                    const float dj = pccx_sqdist(px, py, pz, sx[3 * j], sx[3 * j + 1], sx[3 * j + 2]);
                    const bool tie = dj == T;
                    if (dj < T || (tie && ties < need)) {
#pragma unroll
                        for (int s = 0; s < 16; ++s)
                            if (s == c) out[s] = (IDX)j;
                        ++c;
                    }
                    ties += tie ? 1 : 0;
                }
            }
            // 16 (or 32) contiguous bytes per point: one (two) 16-byte store(s)
            uint4 *dst = (uint4 *)(nbr_bytes + P * patch_stride + (size_t)i * 16 * sizeof(IDX));
            if (sizeof(IDX) == 1) {
                unsigned w[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    w[q] = (unsigned)out[4 * q] | ((unsigned)out[4 * q + 1] << 8) | ((unsigned)out[4 * q + 2] << 16) | ((unsigned)out[4 * q + 3] << 24);
                dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
            } else {
                unsigned w[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) w[q] = (unsigned)out[2 * q] | ((unsigned)out[2 * q + 1] << 16);
                dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
                dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
            }
        }
        __syncthreads();                                  // sx is rewritten for the next patch
    }
}

// bytes per neighbour index in the table: 1 while an index fits a byte
extern "C" int pccx_patch_knn16_index_bytes(int K) { return K <= 256 ? 1 : 2; }

extern "C" size_t pccx_patch_knn16_bytes(int P, int K)
{
    return (size_t)(P > 0 ? P : 0) * (size_t)(K > 0 ? K : 0) * 16 * (size_t)pccx_patch_knn16_index_bytes(K);
}

// patch_stride: bytes between the tables of consecutive patches (>= K * 16 * index bytes, a multiple of 16).  The unfused
// SetAbstraction kernels park each patch's table at the head of that patch's own slice of the feature map, which they overwrite
// only after reading it (encoder.hip).
int pccx_patch_knn16_strided(const float *patches, int P, int K, void *nbr, size_t patch_stride, hipStream_t stream)
{
    // 8 workgroups of 4 waves per CU fill the 32 wave slots; no state is kept between patches, so the grid is one workgroup per
    // patch up to 64 per CU
    const int grid = P < 256 * 64 ? P : 256 * 64;
    if (K <= 256)
        hipLaunchKernelGGL(patch_knn16_kernel<uint8_t>, dim3(grid), dim3(256), (size_t)K * 12, stream, patches, P, K, (unsigned char *)nbr, patch_stride);
    else
        hipLaunchKernelGGL(patch_knn16_kernel<uint16_t>, dim3(grid), dim3(256), (size_t)K * 12, stream, patches, P, K, (unsigned char *)nbr, patch_stride);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

extern "C" int pccx_patch_knn16(const float *patches, int P, int K, void *nbr, void *stream)
{
    if (P == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(patches && nbr, "pccx_patch_knn16: null pointer");
    PCCX_CHECK_ARG(P >= 0 && K >= 16 && K <= 1024 && K % 16 == 0, "pccx_patch_knn16: need K %% 16 == 0, 16 <= K <= 1024 (K=%d)", K);
    PCCX_CHECK_ARG(((uintptr_t)nbr & 15) == 0, "pccx_patch_knn16: the table must be 16-byte aligned");
    return pccx_patch_knn16_strided(patches, P, K, nbr, (size_t)K * 16 * (size_t)pccx_patch_knn16_index_bytes(K), (hipStream_t)stream);
}
