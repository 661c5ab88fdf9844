// octree.hip -- the integer path: occupancy-octree serialisation of the S sampled patch centres.
//
// Replaces pn_kit.encode_sampled_np (pn_kit.py:380-401: a Python depth search that re-runs
// octree_np.encode, an O(nodes*S) stack DFS, up to 16 times) with ONE pass in closed form
// (re-derived and checked bit-for-bit against octree_np.py):
//   * q = floor(p * 2^16) per axis is exact in fp32 (power-of-two scale), and the level-D cell is
//     q >> (16-D); a 63-bit Morton key of (q + 2^20) sorts all S centres once;
//   * level-D occupancy / uniqueness are boundary counts on the sorted keys, so the depth search
//     (bits(D)/N > min_bpp and all S cells distinct) is evaluated for D = 1..16 in the same loop
//     that emits level D;
//   * the reference's LIFO DFS visits occupied cells in DESCENDING Morton order and appends one
//     bit per child 7..0, so the bit of an occupied level-D cell c with parent rank r (ascending)
//     lands at  1 + 8*sum_{l<D-1} occ(l) + 8*(occ(D-1)-1-r) + (7 - (c & 7)).
// One workgroup per cloud (a wave when S <= 64).  Byte/integer work, HBM-bound: 12*S bytes in,
// ~(1 + 8*sum occ)/8 bytes out per cloud.
#include "common.h"

#define OCT_MAX_DEPTH 16
#define OCT_BIAS (1 << 20)

__device__ __forceinline__ unsigned long long spread3(unsigned v)   // 21 bits -> every third bit
{
    unsigned long long x = v & 0x1fffffull;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}
__device__ __forceinline__ unsigned compact3(unsigned long long x)  // inverse of spread3
{
    x &= 0x1249249249249249ull;
    x = (x ^ (x >> 2)) & 0x10c30c30c30c30c3ull;
    x = (x ^ (x >> 4)) & 0x100f00f00f00f00full;
    x = (x ^ (x >> 8)) & 0x1f0000ff0000ffull;
    x = (x ^ (x >> 16)) & 0x1f00000000ffffull;
    x = (x ^ (x >> 32)) & 0x1fffffull;
    return (unsigned)x;
}

__device__ __forceinline__ unsigned quant16(float p)
{
    // floor(p / 2^-16) as numpy's float32 floor_divide gives it (octree_np.py:130), biased by 2^20
    // and clamped so out-of-cube coordinates keep their identity for the uniqueness test.
    float f = floorf(p * 65536.0f);
    f = fminf(fmaxf(f, -(float)OCT_BIAS), (float)(OCT_BIAS - 1));
    return (unsigned)((int)f + OCT_BIAS);
}

// Inclusive block scan of a 0/1 flag; returns the inclusive prefix, *total gets the block sum.
// Two barriers.  s_w: int[17] scratch.
__device__ __forceinline__ int block_scan_flag(bool flag, int *s_w, int nwaves, int *total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned long long mask = __ballot(flag);
    int incl = pccx_ballot_rank(mask) + (flag ? 1 : 0);
    if (lane == 0) s_w[w] = __popcll(mask);
    __syncthreads();
    int base = 0, tot = 0;
    for (int k = 0; k < nwaves; ++k) {
        int v = s_w[k];
        if (k < w) base += v;
        tot += v;
    }
    __syncthreads();
    *total = tot;
    return incl + base;
}

__global__ __launch_bounds__(1024) void octree_encode_kernel(const float *__restrict__ centres, int S, int N, double min_bpp,
                                                             uint8_t *__restrict__ bits_all, int cap,
                                                             int32_t *__restrict__ nbits_out, int32_t *__restrict__ depth_out,
                                                             uint8_t *__restrict__ bytes_all, int bytes_stride,
                                                             int32_t *__restrict__ nbytes_out)
{
    extern __shared__ unsigned long long keys[];     // [T]
    __shared__ int s_w[17];
    const int T = blockDim.x, tid = threadIdx.x, b = blockIdx.x, nwaves = T >> 6;
    uint8_t *bits = bits_all + (size_t)b * cap;

    unsigned long long key = ~0ull;
    if (tid < S) {
        const float *p = centres + ((size_t)b * S + tid) * 3;
        key = (spread3(quant16(p[0])) << 2) | (spread3(quant16(p[1])) << 1) | spread3(quant16(p[2]));
    }
    keys[tid] = key;
    __syncthreads();
    for (int size = 2; size <= T; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (tid < (T >> 1)) {
                int lo = 2 * tid - (tid & (stride - 1)), hi = lo + stride;
                bool up = ((lo & size) == 0);
                unsigned long long a = keys[lo], c = keys[hi];
                if ((a > c) == up) { keys[lo] = c; keys[hi] = a; }
            }
            __syncthreads();
        }
    }
    key = keys[tid];
    const unsigned long long prev = tid > 0 ? keys[tid - 1] : 0ull;
    const bool valid = tid < S;
    const bool in = valid && (key >> 48) == 0x7000ull;               // all three q in [0, 2^16)
    const bool prev_in = tid > 0 && (prev >> 48) == 0x7000ull;

    // level 0
    int total;
    int cell_prev = block_scan_flag(in && !prev_in, s_w, nwaves, &total) - 1;   // index of my level-0 cell
    int occ_prev = total;                 // occ_in(0): 1 if any centre is inside the unit cube
    int sum_occ = 0;                      // sum_{l < D-1} occ_in(l)
    int accepted = 0, final_bits = 1, final_depth = 17;
    if (tid == 0) bits[0] = occ_prev ? 1 : 0;                                    // root bit
    for (int D = 1; D <= OCT_MAX_DEPTH; ++D) {
        const int sh = 3 * (16 - D);
        const bool f_all = valid && (tid == 0 || (key >> sh) != (prev >> sh));
        const bool f_in = in && (!prev_in || (key >> sh) != (prev >> sh));
        int occ_all;
        block_scan_flag(f_all, s_w, nwaves, &occ_all);
        int occ_in;
        const int cell = block_scan_flag(f_in, s_w, nwaves, &occ_in) - 1;
        // stream of depth D: 1 + 8*sum_{l<D} occ_in(l)
        const int off = 1 + 8 * sum_occ;                 // first bit of level D
        const int nb = occ_prev ? off + 8 * occ_prev : 1;
        if (!accepted) {
            // emit level D (every depth up to and including the accepted one contains it)
            if (occ_prev) {
                for (int j = tid; j < 8 * occ_prev; j += T) bits[off + j] = 0;
                __syncthreads();
                if (f_in) bits[off + 8 * (occ_prev - 1 - cell_prev) + (7 - (int)((key >> sh) & 7ull))] = 1;
            }
            final_bits = nb;
            // pn_kit.py:391-394: bpp = len(code)/N (Python float division); accept when
            // bpp > min_bpp and the snapped centres are all distinct (getDecodeFromPc shape test).
            if ((double)nb / (double)N > min_bpp && occ_all == S) { accepted = 1; final_depth = D; }
        }
        sum_occ += occ_prev;
        occ_prev = occ_in;
        cell_prev = cell;
    }
    __syncthreads();
    // pack (pn_kit.py:463-467): MSB-first; a final partial group is right-aligned in its byte.
    const int nby = (final_bits + 7) >> 3;
    uint8_t *bytes = bytes_all + (size_t)b * bytes_stride;
    for (int j = tid; j < nby; j += T) {
        int lo = 8 * j, hi = lo + 8 < final_bits ? lo + 8 : final_bits;
        unsigned v = 0;
        for (int t = lo; t < hi; ++t) v = (v << 1) | bits[t];
        bytes[j] = (uint8_t)v;
    }
    if (tid == 0) {
        nbits_out[b] = final_bits;
        depth_out[b] = final_depth;
        nbytes_out[b] = nby;
    }
}

extern "C" int pccx_octree_bits_capacity(int S) { return 1 + 8 * S * OCT_MAX_DEPTH; }

extern "C" int pccx_octree_encode(const float *centres, int B, int S, int N, double min_bpp, uint8_t *bits, int32_t *nbits,
                                  int32_t *depth, uint8_t *bytes, int32_t *nbytes, void *stream)
{
    if (B == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(centres && bits && nbits && depth && bytes && nbytes, "pccx_octree_encode: null pointer");
    PCCX_CHECK_ARG(B >= 0 && S >= 1 && S <= 1024 && N >= 1, "pccx_octree_encode: need 1 <= S <= 1024, N >= 1 (S=%d N=%d)", S, N);
    int T = 64;
    while (T < S) T <<= 1;
    const int cap = pccx_octree_bits_capacity(S);
    hipLaunchKernelGGL(octree_encode_kernel, dim3(B), dim3(T), (size_t)T * 8, (hipStream_t)stream, centres, S, N, min_bpp, bits,
                       cap, nbits, depth, bytes, (cap + 7) / 8, nbytes);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// ------------------------------------------------------------------------------------------
// decode
// ------------------------------------------------------------------------------------------
// mode 0: octree_np.decode AS WRITTEN (octree_np.py:47-112).  Line :61 overwrites the stream with
// its first group, so only 8 bits are consumed and depth == 1; those bits are read for the eight
// level-1 children popped 111..000 (the root bit is not skipped), giving <= 8 points in
// {0.25,0.75}^3, padded to S = 64 with the last one (:100-107) or zeros when empty.
// One wave per cloud, lane = output point (round 4: the first form ran one THREAD per cloud, 192 scattered stores each -- 99 us per
// 1024 clouds for a few kilobytes of work, twice per step).
__global__ __launch_bounds__(256) void octree_decode_reference_kernel(const uint8_t *__restrict__ bytes, int stride,
                                                                      const int32_t *__restrict__ nbytes, int B, float *__restrict__ out,
                                                                      int32_t *__restrict__ count)
{
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;                                // whole wave
    float *o = out + ((size_t)b * 64 + lane) * 3;
    const int nb = nbytes[b];
    if (nb <= 0) {
        // empty stream: bits_ls == [[1]], depth 0, the root is the single leaf
        o[0] = o[1] = o[2] = 0.5f;
        if (count && lane == 0) count[b] = 1;
        return;
    }
    const unsigned g = bytes[(size_t)b * stride];      // byte_array_to_binary_array: f'{b:08b}'
    const int np = __popc(g & 0xFFu);
    if (count && lane == 0) count[b] = np;
    if (np == 0) { o[0] = o[1] = o[2] = 0.f; return; }
    const int s = lane < np ? lane : np - 1;           // padded to 64 with the last point (octree_np.py:100-107)
    int c = 0, seen = 0;
#pragma unroll
    for (int t = 0; t < 8; ++t) {                      // children are popped 111, 110, ..., 000: the s-th set bit from the top
        if ((g >> (7 - t)) & 1u) {
            if (seen == s) c = 7 - t;
            ++seen;
        }
    }
    o[0] = (c & 4) ? 0.75f : 0.25f;
    o[1] = (c & 2) ? 0.75f : 0.25f;
    o[2] = (c & 1) ? 0.75f : 0.25f;
}

// mode 1 ("full", the build's extension): level-by-level decode.  The true stream is
// 1 + 8*sum(occ) bits long, i.e. always 8*(nbytes-1) + 1 bits: whole bytes MSB-first and one final
// bit right-aligned in the last byte (the packing quirk of pn_kit.py:465-466).
#define OCT_DEC_CAP 2048
__device__ __forceinline__ int stream_bit(const uint8_t *bytes, int nb, int pos)
{
    if (pos >= 8 * (nb - 1)) return bytes[nb - 1] & 1;
    return (bytes[pos >> 3] >> (7 - (pos & 7))) & 1;
}

__global__ __launch_bounds__(256) void octree_decode_full_kernel(const uint8_t *__restrict__ bytes_all, int stride,
                                                                 const int32_t *__restrict__ nbytes, int S_out,
                                                                 float *__restrict__ out, int32_t *__restrict__ count)
{
    __shared__ unsigned long long codes[2][OCT_DEC_CAP];
    __shared__ int s_w[17];
    const int b = blockIdx.x, tid = threadIdx.x;
    const uint8_t *bytes = bytes_all + (size_t)b * stride;
    const int nb = min(max(nbytes[b], 0), stride);     // never read past the row, whatever the caller's count says
    float *o = out + (size_t)b * S_out * 3;
    const int nbits = nb >= 1 ? 8 * (nb - 1) + 1 : 0;
    int parents = (nbits >= 1 && stream_bit(bytes, nb, 0)) ? 1 : 0;
    if (tid == 0) codes[0][0] = 0ull;
    __syncthreads();
    int cur = 0, depth = 0, pos = 1, bad = 0;
    while (parents > 0 && pos + 8 * parents <= nbits && depth < OCT_MAX_DEPTH) {
        const int n = 8 * parents;
        int base = 0;
        for (int j0 = 0; j0 < n; j0 += 256) {
            const int j = j0 + tid;
            const bool f = j < n && stream_bit(bytes, nb, pos + j);
            int tot;
            const int r = base + block_scan_flag(f, s_w, 4, &tot) - 1;
            if (f && r < OCT_DEC_CAP) codes[cur ^ 1][r] = codes[cur][j >> 3] * 8ull + (unsigned long long)(7 - (j & 7));
            base += tot;
        }
        if (base > OCT_DEC_CAP) { bad = 1; break; }
        __syncthreads();
        parents = base; pos += n; cur ^= 1; ++depth;
    }
    if (bad) {
        if (tid == 0 && count) count[b] = -1;
        return;
    }
    if (tid == 0 && count) count[b] = parents;
    const float cell = __uint_as_float((unsigned)(127 - depth) << 23);          // 2^-depth
    for (int i = tid; i < S_out; i += 256) {
        float x = 0.f, y = 0.f, z = 0.f;
        if (parents > 0) {
            const unsigned long long c = codes[cur][i < parents ? i : parents - 1];
            x = ((float)compact3(c >> 2) + 0.5f) * cell;
            y = ((float)compact3(c >> 1) + 0.5f) * cell;
            z = ((float)compact3(c) + 0.5f) * cell;
        }
        o[3 * i] = x; o[3 * i + 1] = y; o[3 * i + 2] = z;
    }
}

extern "C" int pccx_octree_decode(const uint8_t *bytes, int stride, const int32_t *nbytes, int B, int mode, int S_out,
                                  float *out, int32_t *count, void *stream)
{
    if (B == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(bytes && nbytes && out, "pccx_octree_decode: null pointer");
    PCCX_CHECK_ARG(B >= 0 && stride >= 1 && S_out >= 1, "pccx_octree_decode: bad shape");
    PCCX_CHECK_ARG(mode == 0 || mode == 1, "pccx_octree_decode: mode must be 0 (reference) or 1 (full)");
    if (mode == 0) {
        PCCX_CHECK_ARG(S_out == 64, "pccx_octree_decode: reference mode always yields 64 points (octree_np.py:100), S_out=%d",
                       S_out);
        hipLaunchKernelGGL(octree_decode_reference_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, bytes, stride,
                           nbytes, B, out, count);
    } else {
        hipLaunchKernelGGL(octree_decode_full_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, bytes, stride, nbytes, S_out,
                           out, count);
    }
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}
