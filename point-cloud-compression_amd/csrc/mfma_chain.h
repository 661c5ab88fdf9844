// mfma_chain.h -- fp32 MFMA building blocks for the per-point MLP stacks (gfx950).
//
// The reference's 1x1 Conv / Linear stacks (pn_kit.py:98-144,146-211,263-305; AE.py:19-27,96-105)
// are evaluated TRANSPOSED: activations are the B operand (k = channel, n = point), weights the A
// operand (m = output channel).  With v_mfma_f32_16x16x4_f32 the C/D tile then has
//     lane (g = lane>>4, n = lane&15), register r  <->  output channel 16*mt + 4*g + r of point n
// which is exactly the B-operand lane map of the NEXT layer's k-tile kt = mt, step r
// (B[k = lane>>4][n = lane&15], with the k order inside a 16-channel tile permuted to 4*g + r on
// both operands).  A whole Conv-ReLU-Conv-... chain therefore runs out of registers: no LDS round
// trip and no lane shuffles between layers; bias is the accumulator's initial value and ReLU is
// one v_max per register.  f32-in MFMA is bit-for-bit a k-ordered fmaf chain, so the arithmetic
// is plain fp32 (no TF32/bf16 anywhere on the path to the quantiser).
//
// Weight fragment layout ("packed", built once at model load by pccx/weights.py):
//     Wp[kt][mt][lane][r] = W[16*mt + (lane&15)][16*kt + 4*(lane>>4) + r]      (zero padded)
// so one 16-byte load per lane feeds 4 MFMAs and a wave's load is 1 KiB contiguous.
#pragma once
#include <hip/hip_runtime.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x4 relu4(f32x4 v)
{
    f32x4 o;
    o[0] = fmaxf(v[0], 0.f); o[1] = fmaxf(v[1], 0.f); o[2] = fmaxf(v[2], 0.f); o[3] = fmaxf(v[3], 0.f);
    return o;
}

// acc[nt][mt] += W[kt0..kt0+KT)[mt0..mt0+MT) * in[nt][kt]   for NT point tiles sharing the A fragments.
// `w` (wave-uniform, so loads use the scalar-base + lane-offset form) points at fragment (kt=0, mt=0); WMT = number of m-tiles in the
// packed layer (row stride of the fragment table).
template <int KT, int MT, int NT, int WMT>
__device__ __forceinline__ void dense_acc(const f32x4 *__restrict__ w, int lane, const f32x4 (&in)[NT][KT],
                                          f32x4 (&acc)[NT][MT], int kt0 = 0, int mt0 = 0)
{
    // Fragments are consumed in groups of MG m-tiles of one k-tile: consecutive MFMAs then hit
    // different accumulators (dependent-accumulator latency of 16x16x4_f32 is 40 cycles against a
    // 32-cycle issue).  The loads of group i+1 are issued ahead of group i's MFMAs and pinned there
    // with sched_barrier: left alone, hipcc hoists every fragment load of the unrolled chain to the
    // top and spills thousands of registers.
    constexpr int MG = MT >= 4 ? 4 : MT;
    static_assert(MT % MG == 0, "MT must be a multiple of the m-group");
    constexpr int GPK = MT / MG;          // groups per k-tile
    constexpr int NG = KT * GPK;
    f32x4 cur[MG], nxt[MG];
#pragma unroll
    for (int m = 0; m < MG; ++m) cur[m] = w[(kt0 * WMT + mt0 + m) * 64 + lane];
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
        const int kt = gi / GPK, m0 = (gi % GPK) * MG;
        if (gi + 1 < NG) {
            const int kt_n = (gi + 1) / GPK, m0_n = ((gi + 1) % GPK) * MG;
#pragma unroll
            for (int m = 0; m < MG; ++m) nxt[m] = w[((kt0 + kt_n) * WMT + mt0 + m0_n + m) * 64 + lane];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int m = 0; m < MG; ++m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt][m0 + m] = mfma16(cur[m][r], in[nt][kt][r], acc[nt][m0 + m]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MG; ++m) cur[m] = nxt[m];
    }
}

// Launders a wave-uniform pointer through an empty asm so loads through it are not treated as
// loop-invariant: weights do not depend on the point-tile loop, and without this hipcc's LICM hoists
// every fragment load of the chain out of that loop and parks the lot in scratch.
template <class T>
__device__ __forceinline__ const T *opaque_uniform(const T *p)
{
    unsigned zero = 0;                      // an offset the optimiser cannot see through keeps the
    asm volatile("" : "+s"(zero));          // pointer's global address space (no flat loads)
    return p + zero;
}

// max over the 16 lanes of a DPP row (the n index of a C/D tile); every lane ends with the max.
__device__ __forceinline__ float row16_max(float v)
{
    int t;
    t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true);   // row_mirror
    v = fmaxf(v, __int_as_float(t));
    t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true);   // row_half_mirror
    v = fmaxf(v, __int_as_float(t));
    t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x1B, 0xf, 0xf, true);    // quad_perm [3,2,1,0]
    v = fmaxf(v, __int_as_float(t));
    t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true);    // quad_perm [1,0,3,2]
    v = fmaxf(v, __int_as_float(t));
    return v;
}
