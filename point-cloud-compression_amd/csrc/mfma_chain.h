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
#include <stdint.h>

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
template <int KT, int MT, int NT, int WMT, bool SWAP = false>
__device__ __forceinline__ void dense_acc(const f32x4 *__restrict__ w, int lane, const f32x4 (&in)[NT][KT],
                                          f32x4 (&acc)[NT][MT], int kt0 = 0, int mt0 = 0)
{
    // Fragments are consumed in groups of MG m-tiles of one k-tile: consecutive MFMAs then hit
    // different accumulators (dependent-accumulator latency of 16x16x4_f32 is 40 cycles against a
    // 32-cycle issue).  The loads of group i+1 are issued ahead of group i's MFMAs and pinned there
    // with sched_barrier: left alone, hipcc hoists every fragment load of the unrolled chain to the
    // top and spills thousands of registers.
    //
    // SWAP exchanges the MFMA operands (the A and B lane maps of 16x16x4 are mirror images, so the
    // same registers serve either way): the tile comes out transposed, D[point][channel], i.e.
    // lane (g, j) register r holds channel 16*mt + j of point 4*g + r.  Used for a chain's LAST layer
    // when a max over the 16 points follows: it becomes an in-lane max over 4 registers plus two
    // cross-row steps instead of a 16-lane reduction per register.
    constexpr int MG = MT >= 4 ? 4 : MT;
    static_assert(MT % MG == 0, "MT must be a multiple of the m-group");
    constexpr int GPK = MT / MG;          // groups per k-tile
    constexpr int NG = KT * GPK;
    const char *wb = (const char *)w;     // wave-uniform base: loads take the scalar-base + lane-offset form
    const unsigned voff = (unsigned)lane * 16u;
    f32x4 cur[MG], nxt[MG];
#pragma unroll
    for (int m = 0; m < MG; ++m) cur[m] = *(const f32x4 *)(wb + (size_t)(kt0 * WMT + mt0 + m) * 1024 + voff);
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
        const int kt = gi / GPK, m0 = (gi % GPK) * MG;
        if (gi + 1 < NG) {
            const int kt_n = (gi + 1) / GPK, m0_n = ((gi + 1) % GPK) * MG;
#pragma unroll
            for (int m = 0; m < MG; ++m)
                nxt[m] = *(const f32x4 *)(wb + (size_t)((kt0 + kt_n) * WMT + mt0 + m0_n + m) * 1024 + voff);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int m = 0; m < MG; ++m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt][m0 + m] = SWAP ? mfma16(in[nt][kt][r], cur[m][r], acc[nt][m0 + m])
                                           : mfma16(cur[m][r], in[nt][kt][r], acc[nt][m0 + m]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MG; ++m) cur[m] = nxt[m];
    }
}

// Max over the 16 points of EIGHT transposed tiles (dense_acc<..., SWAP=true> outputs, lane (g, j)
// register r = channel 16*t + j of point 4*g + r), as a transpose-reduce: in-lane max over the 4
// registers, then v_permlane32_swap / v_permlane16_swap pair the lane groups so that each swap+max
// finishes two tiles at once.  Result: out[s] (s = 0,1) in lane (row, j) is the max of channel
// 16*(2*row + s) + j.  ~30 VALU instead of ~400 for the 16-lane DPP reduction per register.
__device__ __forceinline__ void max16_of_8_transposed_tiles(const f32x4 (&t)[8], float (&out)[2])
{
    float p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = fmaxf(fmaxf(t[i][0], t[i][1]), fmaxf(t[i][2], t[i][3]));
    float u[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {       // lanes 0-31 finish tile i over rows {0,2},{1,3}; lanes 32-63 tile i+4
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(p[i]), __float_as_uint(p[i + 4]), false, false);
        u[i] = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {       // row0: tile s, row1: tile s+2, row2: tile s+4, row3: tile s+6
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(u[s]), __float_as_uint(u[s + 2]), false, false);
        out[s] = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
}

// ------------------------------------------------------------------------------------------
// bf16x3 operands: fp32 products formed on the bf16 matrix cores from three bf16
// pieces per operand (x = hi + mid + lo exactly), the six products of weight i + j <= 4 accumulated in fp32.
// ------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// In-register split of two fp32 C tiles (8 values per lane = one K=32 B operand) into the three bf16 planes:
// v_cvt_pk_bf16_f32 (round to nearest even), widen back, exact residual, twice.
__device__ __forceinline__ void b3_split8(const f32x4 &v0, const f32x4 &v1, bf16x8 (&pl)[3])
{
    unsigned w[3][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x2v x = q < 2 ? f32x2v{v0[2 * q], v0[2 * q + 1]} : f32x2v{v1[2 * q - 4], v1[2 * q - 3]};
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const bf16x2 h = __builtin_convertvector(x, bf16x2);
            w[p][q] = __builtin_bit_cast(unsigned, h);
            if (p < 2) x = x - __builtin_convertvector(h, f32x2v);
        }
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) pl[p] = __builtin_bit_cast(bf16x8, make_uint4(w[p][0], w[p][1], w[p][2], w[p][3]));
}

// dense layer on bf16x3 operands with the weight blocks RESIDENT in LDS as [kt][mt][plane] fragments (K = 32 per kt).
// SWAP exchanges the MFMA operands (the lane maps of 16x16x32 mirror each other like those of 16x16x4): the tile comes out
// transposed, D[point][channel].
template <int KT, int MT, int NT, bool SWAP = false>
__device__ __forceinline__ void dense_b3(const f32x4 *w, int lane, const bf16x8 (&in)[NT][KT][3], f32x4 (&acc)[NT][MT])
{
    constexpr int MG = MT >= 2 ? 2 : 1;
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};     // smallest products first
    static_assert(MT % MG == 0, "MT must be a multiple of the block group");
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
#pragma unroll
        for (int m = 0; m < MG; ++m)
#pragma unroll
            for (int p = 0; p < 3; ++p) cur[m][p] = nxt[m][p];
    }
}

// ------------------------------------------------------------------------------------------
// Weight stream through LDS.  When every wave of a workgroup consumes the SAME fragment sequence
// (PointNet: each wave runs the whole layer stack on its own 16 points), the sequence is packed on
// the host in consumption order and streamed global -> LDS by LDS-DMA (global_load_lds_dwordx4, one
// 1 KiB fragment per wave-instruction, no VGPRs), one CH-fragment chunk ahead of the MFMAs (CH = 8 measured
// best on MI355X: 16 costs 1 %, 32 costs 3 %, 4 equals 8):
//   * L2 -> CU traffic drops by the number of waves sharing the stream,
//   * MFMAs read fragments with short, uniform LDS latency instead of exposed L2 latency.
// Protocol per chunk c (all waves in lock step, ONE barrier per chunk):
//   s_waitcnt vmcnt(0)   own DMA of chunk c has landed
//   __syncthreads()      everyone's has, and everyone finished reading chunk c-1's buffer
//   issue DMA of chunk c+1 into that buffer; run the MFMAs of chunk c.
// The fragment counter is a plain int that constant-folds after full unrolling, so the chunk hook
// costs nothing between boundaries.
// ------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) unsigned int lds_u32;

// NB buffers of CH fragments: the DMA of chunk c + NB - 1 is issued at the boundary of chunk c into the buffer chunk c - 1
// has just vacated (NB = 2: one chunk ahead).
template <int CH, int NB = 2, int NW = 4>
struct WStreamT {
    static constexpr int chunk_frags = CH, buffers = NB;
    const float *g;                       // global stream, NCH * CH fragments, wave-uniform
    f32x4 *lds;                           // [NB][CH][64]
    int nch;                              // chunks per pass (a multiple of NB when wrap)
    int lane, wave;
    bool wrap;                            // several passes over the same stream

    __device__ __forceinline__ void issue(int c, int buf) const
    {
        // NW waves x CH/NW fragments: wave w moves fragments (CH/NW)w .. of the chunk.  Source address =
        // wave-uniform fragment base (scalar registers) + lane*16 (one VGPR shared by every DMA).
        static_assert(CH % NW == 0, "chunk must divide over the waves");
        const unsigned voff = (unsigned)lane * 16u;
#pragma unroll
        for (int q = 0; q < CH / NW; ++q) {
            const int fr = wave * (CH / NW) + q;               // wave is scalar (readfirstlane)
            const char *src = (const char *)g + ((size_t)c * CH + fr) * 1024;
            f32x4 *dst = lds + (buf * CH + fr) * 64;          // wave-uniform; hardware adds lane*16
            __builtin_amdgcn_global_load_lds((const void *)(src + voff), (lds_u32 *)(uintptr_t)dst, 16, 0, 0);
        }
    }
    __device__ __forceinline__ void prologue() const
    {
#pragma unroll
        for (int i = 0; i < NB - 1; ++i) issue(i, i);
    }
    __device__ __forceinline__ void issue_ahead(int c) const
    {
        const int n = c + NB - 1;
        if (n < nch) issue(n, n % NB);
        else if (wrap) issue(n - nch, n % NB);      // next pass starts over
    }
    __device__ __forceinline__ void boundary(int c) const     // before the first read of chunk c
    {
        // vmcnt(0): the wave's own DMA pieces of chunk c have landed.  lgkmcnt(0): its LDS reads of chunk c - 1's buffer have
        // RETURNED before the barrier lets anyone refill that buffer -- the compiler adds this itself for reads it tracks, but the
        // counted-wait readers (dense_b3_stream_pw) issue theirs as inline assembly, which it does not see
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        issue_ahead(c);
    }
    // boundary that may leave the wave's N most recent VMEM loads in flight: the caller guarantees that the DMA of
    // chunk c was issued before them (in-order completion)
    template <int N>
    __device__ __forceinline__ void boundary_keep(int c) const
    {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
        // a bare s_barrier, not __syncthreads(): the release fence in __syncthreads() makes the compiler put s_waitcnt vmcnt(0) in front of
        // the barrier (LDS-DMA fills are vmcnt-tracked LDS writes, it cannot tell them from the loads meant to stay in flight), which
        // discarded the counted wait above it in 64 of the f16x2 decoder's 65 barriers.  The wait above is what the protocol needs
        // (in-order completion covers this wave's pieces of chunk c; reads of chunk c - 1 fed MFMAs issued before this point).
        // Same-box A/B: decoder stage 6.01 / 6.07 -> 5.92 / 5.99 ms per 1024 clouds (tools/experiments/r5/README.md).
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue_ahead(c);
    }
    __device__ __forceinline__ const f32x4 *chunk(int c) const { return lds + (c % NB) * CH * 64 + lane; }
    __device__ __forceinline__ f32x4 get(int f) const         // fragment f of the current pass
    {
        if ((f % CH) == 0) boundary(f / CH);
        return lds[(((f / CH) % NB) * CH + (f % CH)) * 64 + lane];
    }
    __device__ __forceinline__ void drain() const { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
};

// dense_acc fed from a WStream: fragments are taken in stream order (kt-major, m inner), `f` is the
// running fragment index of the pass.
template <int KT, int MT, int NT, bool SWAP = false, class WS>
__device__ __forceinline__ void dense_acc_stream(const WS &ws, int &f, const f32x4 (&in)[NT][KT], f32x4 (&acc)[NT][MT])
{
    constexpr int MG = MT >= 4 ? 4 : MT;
    static_assert(MT % MG == 0, "MT must be a multiple of the m-group");
    constexpr int GPK = MT / MG;
    constexpr int NG = KT * GPK;
    f32x4 cur[MG], nxt[MG];
#pragma unroll
    for (int m = 0; m < MG; ++m) cur[m] = ws.get(f + m);
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
        const int kt = gi / GPK, m0 = (gi % GPK) * MG;
        if (gi + 1 < NG) {
#pragma unroll
            for (int m = 0; m < MG; ++m) nxt[m] = ws.get(f + (gi + 1) * MG + m);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int m = 0; m < MG; ++m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt][m0 + m] = SWAP ? mfma16(in[nt][kt][r], cur[m][r], acc[nt][m0 + m])
                                           : mfma16(cur[m][r], in[nt][kt][r], acc[nt][m0 + m]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MG; ++m) cur[m] = nxt[m];
    }
    f += NG * MG;
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

// dense layer of the chain on bf16x3 operands: in[nt][kt][plane] are K=32 B operands, the weight blocks ([kt][mt][plane],
// three 1 KiB A fragments each) come from the LDS ring in stream order; six products per block, two blocks in flight.
template <int KT, int MT, int NT, class WS>
__device__ __forceinline__ void dense_b3_stream(const WS &ws, int &f, const bf16x8 (&in)[NT][KT][3], f32x4 (&acc)[NT][MT])
{
    constexpr int MG = MT >= 2 ? 2 : 1;
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};     // smallest products first
    static_assert(MT % MG == 0, "MT must be a multiple of the block group");
    constexpr int NG = KT * (MT / MG);
    bf16x8 cur[MG][3], nxt[MG][3];
#pragma unroll
    for (int m = 0; m < MG; ++m)
#pragma unroll
        for (int p = 0; p < 3; ++p) cur[m][p] = __builtin_bit_cast(bf16x8, ws.get(f + 3 * m + p));
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
        const int kt = gi / (MT / MG), m0 = (gi % (MT / MG)) * MG;
        if (gi + 1 < NG) {
#pragma unroll
            for (int m = 0; m < MG; ++m)
#pragma unroll
                for (int p = 0; p < 3; ++p) nxt[m][p] = __builtin_bit_cast(bf16x8, ws.get(f + 3 * ((gi + 1) * MG + m) + p));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int m = 0; m < MG; ++m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt][m0 + m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[m][PA[q]], in[nt][kt][PB[q]], acc[nt][m0 + m], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MG; ++m)
#pragma unroll
            for (int p = 0; p < 3; ++p) cur[m][p] = nxt[m][p];
    }
    f += 3 * KT * MT;
}


// ------------------------------------------------------------------------------------------
// dense_b3_stream with the LDS reads of the weight blocks issued as inline assembly and waited for by COUNT.
// Why: while an LDS-DMA (global_load_lds) is in flight -- i.e. during every chunk of the weight ring -- hipcc (ROCm 7.2) answers each
// use of a ds_read result with s_waitcnt lgkmcnt(0).  With the loads of block group g+1 issued ahead of group g's MFMAs, as here,
// that drains the six reads it has JUST issued before the first MFMA of every other group (the alternate groups then need no wait):
// the full LDS latency exposed once per 24 MFMAs.  Without a DMA in the kernel the same source compiles to the counted waits
// lgkmcnt(9) / lgkmcnt(6) (tools/experiments/r3: -DK_NODMA).  The compiler does not track these reads at all, so the wait is placed by
// hand: LDS returns in order, lgkmcnt(n) = "all but the youngest n LDS/SMEM operations are done", and any further operation the
// compiler puts in flight only makes the wait longer, never shorter.  The wait asm names the registers it guards as in/out operands,
// so no MFMA that reads them can be scheduled above it.
// ------------------------------------------------------------------------------------------
// base = LDS byte address of the ring's fragment 0 for this lane (one VGPR for every read); byte_off: a compile-time constant after
// unrolling (the ring is at most 64 KiB, the instruction's offset field 16 bits)
__device__ __forceinline__ bf16x8 lds_read_frag_async(unsigned base, int byte_off)
{
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(byte_off) : "memory");
    return v;
}

template <int N, int MG>
__device__ __forceinline__ void lds_wait_keep(bf16x8 (&c)[MG][3])
{
    if constexpr (MG == 2)
        asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(c[0][0]), "+v"(c[0][1]), "+v"(c[0][2]), "+v"(c[1][0]), "+v"(c[1][1]), "+v"(c[1][2]) : "n"(N) : "memory");
    else
        asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(c[0][0]), "+v"(c[0][1]), "+v"(c[0][2]) : "n"(N) : "memory");
}

template <int KT, int MT, int NT, class WS>
__device__ __forceinline__ void dense_b3_stream_pw(const WS &ws, int &f, const bf16x8 (&in)[NT][KT][3], f32x4 (&acc)[NT][MT])
{
    constexpr int MG = MT >= 2 ? 2 : 1;
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};     // smallest products first
    static_assert(MT % MG == 0, "MT must be a multiple of the block group");
    constexpr int NG = KT * (MT / MG);
    constexpr int CH = WS::chunk_frags, NB = WS::buffers;
    static_assert(NB * CH * 1024 <= 65536, "the ring must fit the 16-bit offset field of ds_read");
    unsigned base = (unsigned)(uintptr_t)(ws.lds + ws.lane);         // LDS offset = low half of the generic address
    asm volatile("" : "+v"(base));                                   // ONE address register; offsets are immediates
    auto frag = [&](int fi) {                                        // fragment fi of the pass: ring boundary, then the read
        if ((fi % CH) == 0) ws.boundary(fi / CH);
        return lds_read_frag_async(base, (((fi / CH) % NB) * CH + (fi % CH)) * 1024);
    };
    bf16x8 cur[MG][3], nxt[MG][3];
#pragma unroll
    for (int m = 0; m < MG; ++m)
#pragma unroll
        for (int p = 0; p < 3; ++p) cur[m][p] = frag(f + 3 * m + p);
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
        const int kt = gi / (MT / MG), m0 = (gi % (MT / MG)) * MG;
        if (gi + 1 < NG) {
#pragma unroll
            for (int m = 0; m < MG; ++m)
#pragma unroll
                for (int p = 0; p < 3; ++p) nxt[m][p] = frag(f + 3 * ((gi + 1) * MG + m) + p);
            lds_wait_keep<3 * MG, MG>(cur);                          // cur has landed; the 3 * MG reads of nxt may still fly
        } else
            lds_wait_keep<0, MG>(cur);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int m = 0; m < MG; ++m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt][m0 + m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[m][PA[q]], in[nt][kt][PB[q]], acc[nt][m0 + m], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MG; ++m)
#pragma unroll
            for (int p = 0; p < 3; ++p) cur[m][p] = nxt[m][p];
    }
    f += 3 * KT * MT;
}

// ------------------------------------------------------------------------------------------
// f16x2 operands ("f16x2"): fp32 products formed on the fp16 matrix cores from TWO fp16 pieces per operand,
// x ~ hi + lo with hi = rn16(x), lo = rn16(x - hi) (22-23 significant bits: the mantissa budget of "3xTF32"), and the three
// products hi*hi + hi*lo + lo*hi accumulated in fp32 -- half the MFMAs of bf16x3.  fp16 has 5 exponent bits, so every operand
// is brought into the format's range by an exact power-of-two scale chosen from RIGOROUS bounds (pack_h2.hip): activations to
// at most 2^15 (no overflow; the lo piece stays a normal number down to 2^-18 of the bound and degrades gracefully, as an
// absolute error of 2^-40 of the bound, below that), weights to 2^14.  The scales travel with the accumulators
// (acc = sigma * tau * (W y + b)) and are undone by the multiplier `rho` of the next split or of the epilogue: powers of two
// commute with fp32 rounding, so the scaled chain computes exactly what the unscaled one would.
// ------------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));

// split of two fp32 C tiles (8 values per lane = one K=32 B operand), multiplied by rho first, into the two fp16 planes
__device__ __forceinline__ void h2_split8(const f32x4 &v0, const f32x4 &v1, float rho, f16x8 (&pl)[2])
{
    unsigned w[2][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x2v x = q < 2 ? f32x2v{v0[2 * q], v0[2 * q + 1]} : f32x2v{v1[2 * q - 4], v1[2 * q - 3]};
        x = x * rho;                                                    // exact (power of two)
        const f16x2v h = __builtin_convertvector(x, f16x2v);           // round to nearest even
        w[0][q] = __builtin_bit_cast(unsigned, h);
        // exact residual x - hi, one v_fma_mix_f32 per value: the fp16 operand is widened inside the instruction (hipcc turns the
        // plain expression into two v_cvt_f32_f16 and a v_pk_add_f32)
        f32x2v r;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r[0]) : "v"(w[0][q]), "v"(x[0]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r[1]) : "v"(w[0][q]), "v"(x[1]));
        w[1][q] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2v));
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) pl[p] = __builtin_bit_cast(f16x8, make_uint4(w[p][0], w[p][1], w[p][2], w[p][3]));
}

#define H2_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#ifdef H2_PRIO       // experiment (tools/experiments/r5): raise the wave's issue priority over its MFMA groups
#define H2_PRIO_UP() __builtin_amdgcn_s_setprio(H2_PRIO)
#define H2_PRIO_DOWN() __builtin_amdgcn_s_setprio(0)
#else
#define H2_PRIO_UP()
#define H2_PRIO_DOWN()
#endif

// dense layer on f16x2 operands with the weight blocks RESIDENT in LDS as [kt][mt][plane] fragments (K = 32 per kt); the three
// products smallest first: (lo,hi) (hi,lo) (hi,hi).  SWAP as in dense_b3.
template <int KT, int MT, int NT, bool SWAP = false, int MGW = 2>
__device__ __forceinline__ void dense_h2(const f32x4 *w, int lane, const f16x8 (&in)[NT][KT][2], f32x4 (&acc)[NT][MT])
{
    constexpr int MG = MT >= MGW ? MGW : MT;
    constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
    static_assert(MT % MG == 0, "MT must be a multiple of the block group");
    constexpr int NG = KT * (MT / MG);
    f16x8 cur[MG][2], nxt[MG][2];
#pragma unroll
    for (int m = 0; m < MG; ++m)
#pragma unroll
        for (int p = 0; p < 2; ++p) cur[m][p] = __builtin_bit_cast(f16x8, w[(size_t)(m * 2 + p) * 64 + lane]);
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
        if (gi + 1 < NG) {
#pragma unroll
            for (int m = 0; m < MG; ++m)
#pragma unroll
                for (int p = 0; p < 2; ++p) nxt[m][p] = __builtin_bit_cast(f16x8, w[(size_t)(((gi + 1) * MG + m) * 2 + p) * 64 + lane]);
        }
        const int kt = gi / (MT / MG), m0 = (gi % (MT / MG)) * MG;
        __builtin_amdgcn_sched_barrier(0);
        H2_PRIO_UP();
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int m = 0; m < MG; ++m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt][m0 + m] = SWAP ? H2_MFMA(in[nt][kt][PB[q]], cur[m][PA[q]], acc[nt][m0 + m])
                                           : H2_MFMA(cur[m][PA[q]], in[nt][kt][PB[q]], acc[nt][m0 + m]);
        H2_PRIO_DOWN();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MG; ++m)
#pragma unroll
            for (int p = 0; p < 2; ++p) cur[m][p] = nxt[m][p];
    }
}

// the same layer with its weight blocks ([kt][mt][plane], two 1 KiB A fragments each) taken from the LDS ring in stream order
template <int KT, int MT, int NT, class WS, int MGW = 2>
__device__ __forceinline__ void dense_h2_stream(const WS &ws, int &f, const f16x8 (&in)[NT][KT][2], f32x4 (&acc)[NT][MT])
{
    constexpr int MG = MT >= MGW ? MGW : MT;
    constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
    static_assert(MT % MG == 0, "MT must be a multiple of the block group");
    constexpr int NG = KT * (MT / MG);
    f16x8 cur[MG][2], nxt[MG][2];
#pragma unroll
    for (int m = 0; m < MG; ++m)
#pragma unroll
        for (int p = 0; p < 2; ++p) cur[m][p] = __builtin_bit_cast(f16x8, ws.get(f + 2 * m + p));
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
        const int kt = gi / (MT / MG), m0 = (gi % (MT / MG)) * MG;
        if (gi + 1 < NG) {
#pragma unroll
            for (int m = 0; m < MG; ++m)
#pragma unroll
                for (int p = 0; p < 2; ++p) nxt[m][p] = __builtin_bit_cast(f16x8, ws.get(f + 2 * ((gi + 1) * MG + m) + p));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int m = 0; m < MG; ++m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt][m0 + m] = H2_MFMA(cur[m][PA[q]], in[nt][kt][PB[q]], acc[nt][m0 + m]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MG; ++m)
#pragma unroll
            for (int p = 0; p < 2; ++p) cur[m][p] = nxt[m][p];
    }
    f += 2 * KT * MT;
}

// ------------------------------------------------------------------------------------------
// Stream reader for the f16x2 layers: a register FIFO of D fragments over the LDS ring, filled by inline-assembly ds_read_b128 and
// waited for by COUNT.  Why (beyond dense_b3_stream_pw's reasons): with three MFMAs per weight block instead of six, a group of two
// blocks covers only 96 cycles of matrix work, less than an LDS round trip under load; the compiler's form (next group's reads
// issued ahead of this group's MFMAs, every use answered with s_waitcnt lgkmcnt(0) while an LDS-DMA is in flight) then exposes the
// full LDS latency once per group, and the first group of every layer call -- 48 calls per PointNet pass -- starts with its reads
// not even issued.  The FIFO runs ACROSS layer calls, always D fragments (D / 4 groups) ahead of the MFMAs, in stream order:
//   at(f)      the register holding fragment f (f is a compile-time constant after unrolling: no moves, no indexing)
//   need(f, G) before the first MFMA of a group consuming fragments [f, f + G): waits until at most the reads YOUNGER than that
//              group's are outstanding, i.e. fragments f + G .. min(f + D, total) - 1; LDS returns in order, and any other LDS
//              operation the compiler has in flight only makes the wait longer.  (No scalar loads are in flight in these loops:
//              SMEM returns out of order and would make a counted lgkmcnt unsafe.)
//   done(f)    after the MFMAs that read fragment f have been issued: its register is refilled with fragment f + D.  The ring's
//              chunk boundary (barrier + next DMA) is crossed by the refill, D fragments before the MFMAs get there; its
//              lgkmcnt(0) drains the FIFO's own reads of the chunk being vacated.
// ------------------------------------------------------------------------------------------
template <int D, int TOTAL, class WS>
struct H2Reader {
    static constexpr int CH = WS::chunk_frags, NB = WS::buffers;
    static_assert(NB * CH * 1024 <= 65536, "the ring must fit the 16-bit offset field of ds_read");
    const WS &ws;
    unsigned base;                                                   // LDS byte address of the ring's fragment 0 for this lane
    f16x8 q[D];
    __device__ __forceinline__ explicit H2Reader(const WS &w) : ws(w)
    {
        base = (unsigned)(uintptr_t)(w.lds + w.lane);
        asm volatile("" : "+v"(base));
    }
    __device__ __forceinline__ void load(int fi)
    {
        if ((fi % CH) == 0) ws.boundary(fi / CH);
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[fi % D]) : "v"(base), "n"((((fi / CH) % NB) * CH + (fi % CH)) * 1024) : "memory");
    }
    __device__ __forceinline__ void start()
    {
#pragma unroll
        for (int i = 0; i < D; ++i) load(i);
    }
    __device__ __forceinline__ const f16x8 &at(int f) const { return q[f % D]; }
    template <int G>
    __device__ __forceinline__ void need(int f)
    {
        static_assert(G == 2 || G == 4, "groups of one or two blocks");
        const int hi = f + D < TOTAL ? f + D : TOTAL;               // fragments issued so far: [0, hi)
        const int younger = hi - (f + G) > 0 ? hi - (f + G) : 0;
        // the asm names the group's registers as in/out operands, so no MFMA that reads them can be scheduled above the wait
        if constexpr (G == 4)
            asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(q[f % D]), "+v"(q[(f + 1) % D]), "+v"(q[(f + 2) % D]), "+v"(q[(f + 3) % D]) : "n"(younger) : "memory");
        else
            asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(q[f % D]), "+v"(q[(f + 1) % D]) : "n"(younger) : "memory");
    }
    __device__ __forceinline__ void done(int f)
    {
        if (f + D < TOTAL) load(f + D);
    }
};

// dense_h2_stream on a reader: fragments [f, f + 2 KT MT) of the stream, consumed in order
template <int KT, int MT, int NT, class RD>
__device__ __forceinline__ void dense_h2_rd(RD &rd, int &f, const f16x8 (&in)[NT][KT][2], f32x4 (&acc)[NT][MT])
{
    constexpr int MG = MT >= 2 ? 2 : 1;
    constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
    static_assert(MT % MG == 0, "MT must be a multiple of the block group");
    constexpr int NG = KT * (MT / MG);
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
        const int kt = gi / (MT / MG), m0 = (gi % (MT / MG)) * MG, f0 = f + 2 * gi * MG;
        rd.template need<2 * MG>(f0);
        __builtin_amdgcn_sched_barrier(0);
        H2_PRIO_UP();
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int m = 0; m < MG; ++m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt][m0 + m] = H2_MFMA(rd.at(f0 + 2 * m + PA[q]), in[nt][kt][PB[q]], acc[nt][m0 + m]);
        H2_PRIO_DOWN();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2 * MG; ++i) rd.done(f0 + i);
    }
    f += 2 * KT * MT;
}
