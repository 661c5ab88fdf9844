// rangecoder.hip -- torchac-style range coder (compress.py:136 encode_float_cdf, decompress.py:93
// decode_float_cdf) on the GPU: 32-bit low/high, 16-bit CDFs, E1/E2/E3 renormalisation with pending
// bits, MSB-first bit packing, zero-padded last byte.  Byte layout PARITY UNPINNED (torchac is not in
// the image); the oracle (orc_range_encode / orc_range_decode) restates the same published algorithm
// and must agree byte-for-byte.
//
// The coder is inherently serial per stream (S*d = 1024 symbols per cloud).  One wave per cloud:
// the 64 lanes stage the cloud's tables and bytes through LDS with coalesced transfers and run the
// serial recurrence wave-uniformly on the scalar unit (see "wave-uniform execution" below).  Streams
// larger than the LDS budget fall back to one lane per cloud working from global memory.
#include "common.h"

struct BitWriter {
    uint8_t *buf;
    int cap, n;
    unsigned cache;
    int count;
    __device__ void bit(int b)
    {
        cache = (cache << 1) | (unsigned)(b & 1);
        if (++count == 8) {
            if (n < cap) buf[n] = (uint8_t)cache;
            ++n;
            count = 0; cache = 0;
        }
    }
    __device__ void bit_pending(int b, unsigned long long &pending)
    {
        bit(b);
        while (pending > 0) { bit(!b); --pending; }
    }
    __device__ void flush()
    {
        if (count > 0) {
            cache <<= (8 - count);
            if (n < cap) buf[n] = (uint8_t)cache;
            ++n;
            count = 0; cache = 0;
        }
    }
};

__global__ void range_encode_kernel(const int32_t *__restrict__ cdf_int, const float *__restrict__ latent_q, int B, int nsym,
                                    int Lp, int sym_offset, uint8_t *__restrict__ out, int cap, int32_t *__restrict__ nbytes)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    BitWriter w{out + (size_t)b * cap, cap, 0, 0u, 0};
    unsigned low = 0u, high = 0xFFFFFFFFu;
    unsigned long long pending = 0;
    const int max_symbol = Lp - 2;
    const int32_t *c = cdf_int + (size_t)b * nsym * Lp;
    const float *q = latent_q + (size_t)b * nsym;
    for (int i = 0; i < nsym; ++i, c += Lp) {
        int s = (int)q[i] + sym_offset;                       // latent_quantized.to(int16) + L//2 (compress.py:135)
        s = s < 0 ? 0 : (s > max_symbol ? max_symbol : s);
        const unsigned long long span = (unsigned long long)high - (unsigned long long)low + 1ull;
        const unsigned c_low = (unsigned)c[s] & 0xFFFFu;
        const unsigned c_high = s == max_symbol ? 0x10000u : ((unsigned)c[s + 1] & 0xFFFFu);
        high = (unsigned)((low - 1u) + (unsigned)((span * c_high) >> 16));
        low = (unsigned)(low + (unsigned)((span * c_low) >> 16));
        for (;;) {
            if (high < 0x80000000u) {
                w.bit_pending(0, pending);
                low <<= 1; high <<= 1; high |= 1u;
            } else if (low >= 0x80000000u) {
                w.bit_pending(1, pending);
                low <<= 1; high <<= 1; high |= 1u;
            } else if (low >= 0x40000000u && high < 0xC0000000u) {
                ++pending;
                low <<= 1; low &= 0x7FFFFFFFu;
                high <<= 1; high |= 0x80000001u;
            } else
                break;
        }
    }
    ++pending;
    if (low < 0x40000000u) w.bit_pending(0, pending);
    else w.bit_pending(1, pending);
    w.flush();
    nbytes[b] = w.n <= cap ? w.n : -w.n;                      // negative: capacity exceeded
}

__global__ void range_decode_kernel(const int32_t *__restrict__ cdf_int, const uint8_t *__restrict__ in, int stride,
                                    const int32_t *__restrict__ nbytes, int B, int nsym, int Lp, int sym_offset,
                                    float *__restrict__ latent_q)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint8_t *buf = in + (size_t)b * stride;
    const int nb = nbytes[b] < 0 ? 0 : (nbytes[b] > stride ? stride : nbytes[b]);
    int pos = 0, cached = 0;
    unsigned cache = 0;
    unsigned low = 0u, high = 0xFFFFFFFFu, value = 0u;
    auto get = [&]() {
        if (cached == 0) {
            if (pos >= nb) { value <<= 1; return; }
            cache = buf[pos++]; cached = 8;
        }
        value = (value << 1) | ((cache >> (cached - 1)) & 1u);
        --cached;
    };
    for (int i = 0; i < 32; ++i) get();
    const int max_symbol = Lp - 2;
    const int32_t *c = cdf_int + (size_t)b * nsym * Lp;
    for (int i = 0; i < nsym; ++i, c += Lp) {
        const unsigned long long span = (unsigned long long)high - (unsigned long long)low + 1ull;
        const unsigned count =
            (unsigned)((((unsigned long long)value - (unsigned long long)low + 1ull) * 0x10000ull - 1ull) / span) & 0xFFFFu;
        int left = 0, right = max_symbol + 1;
        while (left + 1 < right) {
            const int mid = (left + right) >> 1;
            if (((unsigned)c[mid] & 0xFFFFu) <= count) left = mid; else right = mid;
        }
        const int s = left;
        latent_q[(size_t)b * nsym + i] = (float)(s - sym_offset);            // decode - L//2 (decompress.py:93)
        const unsigned c_low = (unsigned)c[s] & 0xFFFFu;
        const unsigned c_high = s == max_symbol ? 0x10000u : ((unsigned)c[s + 1] & 0xFFFFu);
        high = (unsigned)((low - 1u) + (unsigned)((span * c_high) >> 16));
        low = (unsigned)(low + (unsigned)((span * c_low) >> 16));
        for (;;) {
            if (low >= 0x80000000u || high < 0x80000000u) {
                low <<= 1; high <<= 1; high |= 1u; get();
            } else if (low >= 0x40000000u && high < 0xC0000000u) {
                low <<= 1; low &= 0x7FFFFFFFu;
                high <<= 1; high |= 0x80000001u;
                value -= 0x40000000u;
                get();
            } else
                break;
        }
    }
}

#define RC_MAX_LDS_BYTES (60 * 1024)

// ---- bulk renormalisation -----------------------------------------------------------------------
// The E1/E2 loop of the reference coder shifts out one bit per iteration while the top bits of low
// and high agree; that run is clz(low ^ high) bits long and can be emitted at once.  The E3
// (underflow) loop runs while low = 01.., high = 10..: its length is the run of positions, from bit
// 30 down, where low has 1 and high has 0, again one clz.  k consecutive E3 steps give
//   low' = (low << k) & 0x7FFFFFFF,  high' = (high << k) | 0x80000000 | (2^k - 1),  pending += k,
// and on the decoder value' = ((value << k) ^ 0x80000000) | next k bits (each step subtracts 2^30
// before doubling; the k subtractions sum to 2^31 mod 2^32).  Bit-for-bit the same stream as the
// one-bit-at-a-time form (tests compare against the oracle's literal restatement).
// Loop shape: after an E1/E2 run the top bits of low and high differ (0 / 1), an E3 run keeps them so, and
// after an E3 run the E3 condition is false by construction, so the renormalisation "loop" is exactly one
// optional E1/E2 run followed by one optional E3 run; with t = (~low | high) << 1 the E3 run length is
// clz(t) (0 when there is no underflow, because then bit 31 of t is set), which makes both steps branch-free.
// ---- wave-uniform execution -------------------------------------------------------------------
// The recurrence (low, high, value, pending, the bit accumulators) lives in SCALAR registers: every
// quantity below that does not depend on the lane is derived from readlane / ballot / readfirstlane,
// so the compiler keeps it on the SALU (s_mul_hi_u32, s_flbit, s_lshl_b64 ...) and no step of the serial
// chain waits on an LDS or memory round trip:
//   encoder  a parallel pre-pass gathers (cdf[s], cdf[s+1]) of every symbol into LDS; the serial loop
//            takes them 64 symbols at a time into one VGPR pair (next block prefetched) and reads
//            symbol j with v_readlane.  Output bits go through a 64-bit scalar accumulator and leave
//            as whole big-endian words.
//   decoder  the 64 lanes hold the CDF rows of G = 64 / (L+1) consecutive symbols (next block
//            prefetched from LDS).  Per symbol ALL candidates are scaled at once
//            (t = ((span * cdf) >> 16), one v_mad_u64_u32 + shift per lane), compared with value - low,
//            and the ballot mask is walked with the reference's binary search, on scalar bits; the
//            two t's that become the new low/high come back with v_readlane.  Input bits are taken
//            from a 64-bit scalar buffer refilled a word at a time, the next word always already
//            loaded.
__device__ __forceinline__ unsigned ones(int m) { return m >= 32 ? 0xFFFFFFFFu : ((1u << m) - 1u); }
__device__ __forceinline__ unsigned rc_uni(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }
// low 32 bits of (span * c) >> 16 with span = span32 + 1 in [1, 2^32]
__device__ __forceinline__ unsigned rc_scale(unsigned span32, unsigned c) { return (unsigned)(((unsigned long long)span32 * c + c) >> 16); }

__global__ __launch_bounds__(64) void range_encode_wave_kernel(const int32_t *__restrict__ cdf_int, const float *__restrict__ latent_q,
                                                               int nsym, int Lp, int sym_offset, uint8_t *__restrict__ out, int cap,
                                                               int32_t *__restrict__ nbytes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char rc_smem[];
    uint2 *sc = (uint2 *)rc_smem;                                      // [nsym] (c_low, c_high)
    unsigned *sout = (unsigned *)(sc + nsym);                          // [capw] output words (stream order = big endian)
    const int b = blockIdx.x, lane = threadIdx.x;
    const int capw = (cap + 3) >> 2;
    const int32_t *c = cdf_int + (size_t)b * nsym * Lp;
    const int max_symbol = Lp - 2;
    for (int i = lane; i < nsym; i += 64) {
        int s = (int)latent_q[(size_t)b * nsym + i] + sym_offset;     // latent_quantized.to(int16) + L//2 (compress.py:135)
        s = s < 0 ? 0 : (s > max_symbol ? max_symbol : s);
        const int32_t *ci = c + (size_t)i * Lp;
        sc[i] = make_uint2((unsigned)ci[s] & 0xFFFFu, s == max_symbol ? 0x10000u : ((unsigned)ci[s + 1] & 0xFFFFu));
    }
    __syncthreads();
    unsigned low = 0u, high = 0xFFFFFFFFu;
    unsigned long long pending = 0, acc = 0;
    int na = 0, nw = 0;
    auto put = [&](unsigned v, int cnt) {                              // cnt <= 32; na < 32 on entry
        acc = (acc << cnt) | (unsigned long long)v;
        na += cnt;
        if (na >= 32) {
            const unsigned w = (unsigned)(acc >> (na - 32));
            if (nw < capw && lane == 0) sout[nw] = __builtin_bswap32(w);
            ++nw;
            na -= 32;
        }
    };
    auto run = [&](int bit, unsigned long long cnt) {
        while (cnt > 0) {
            const int m = cnt > 32 ? 32 : (int)cnt;
            put(bit ? ones(m) : 0u, m);
            cnt -= m;
        }
    };
    uint2 nxt = lane < nsym ? sc[lane] : make_uint2(0u, 0x10000u);
    for (int i0 = 0; i0 < nsym; i0 += 64) {
        const uint2 cur = nxt;
        if (i0 + 64 + lane < nsym) nxt = sc[i0 + 64 + lane];
        const int cnt = nsym - i0 < 64 ? nsym - i0 : 64;
        for (int j = 0; j < cnt; ++j) {
            const unsigned c_low = __builtin_amdgcn_readlane(cur.x, j), c_high = __builtin_amdgcn_readlane(cur.y, j);
            const unsigned span32 = high - low;
            high = (low - 1u) + rc_scale(span32, c_high);
            low = low + rc_scale(span32, c_low);
            // Renormalisation, branch-free (see "loop shape" above): one E1/E2 run of m bits, then one E3 run of k.
            const unsigned x = low ^ high;
            const int m = (int)x < 0 ? 0 : (x ? __clz(x) : 32);
            if (m > 0) {
                const unsigned top = (unsigned)(((unsigned long long)low << m) >> 32);     // the m agreeing bits
                if (pending == 0)
                    put(top, m);
                else {
                    const unsigned b0 = low >> 31;
                    put(b0, 1);
                    run(!b0, pending);
                    pending = 0;
                    if (m > 1) put(top & ones(m - 1), m - 1);
                }
                low = (unsigned)((unsigned long long)low << m);
                high = (unsigned)(((unsigned long long)high << m) | (unsigned long long)ones(m));
            }
            const unsigned t = (~low | high) << 1;                       // top bit set <=> no underflow pending
            const int k = t ? __clz(t) : 31;
            const unsigned fix = k ? 0x80000000u : 0u;                  // k = 0 must leave low/high alone (also when a zero-width
            low = (low << k) & ~fix;                                     // symbol has driven high below low)
            high = (high << k) | fix | ones(k);
            pending += k;
        }
    }
    ++pending;
    const int fb = low < 0x40000000u ? 0 : 1;
    put(fb, 1);
    run(!fb, pending);
    int n = nw * 4 + ((na + 7) >> 3);
    if (na > 0 && nw < capw && lane == 0) sout[nw] = __builtin_bswap32((unsigned)(acc << (32 - na)));   // zero-padded tail
    if (lane == 0) nbytes[b] = n <= cap ? n : -n;                      // negative: capacity exceeded
    __syncthreads();
    if (n > cap) n = cap;
    const uint8_t *sb = (const uint8_t *)sout;
    for (int i = lane; i < n; i += 64) out[(size_t)b * cap + i] = sb[i];
}

__global__ __launch_bounds__(64) void range_decode_wave_kernel(const int32_t *__restrict__ cdf_int, const uint8_t *__restrict__ in,
                                                               int stride, const int32_t *__restrict__ nbytes, int nsym, int Lp,
                                                               int sym_offset, float *__restrict__ latent_q)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char rc_smem[];
    const int ncdf = nsym * Lp;
    unsigned short *scdf = (unsigned short *)rc_smem;                 // [ncdf]
    unsigned *sin = (unsigned *)(rc_smem + (((size_t)ncdf * 2 + 3) & ~(size_t)3));   // [nwin] stream words, zero padded
    unsigned char *ssym = (unsigned char *)(sin + ((stride + 3) >> 2));               // [nsym]
    const int b = blockIdx.x, lane = threadIdx.x;
    const int32_t *c = cdf_int + (size_t)b * ncdf;
    const int nb = nbytes[b] < 0 ? 0 : (nbytes[b] > stride ? stride : nbytes[b]);
    const int nwin = (nb + 3) >> 2;
    for (int i = lane; i < ncdf; i += 64) scdf[i] = (unsigned short)c[i];
    for (int i = lane; i < nwin * 4; i += 64) ((uint8_t *)sin)[i] = i < nb ? in[(size_t)b * stride + i] : (uint8_t)0;
    __syncthreads();
    // bit reader: `buf` holds `have` unread bits in its low end, `nxt` is the word after them; words past
    // the stream read as 0
    auto word = [&](int p) -> unsigned { return p < nwin ? __builtin_bswap32(rc_uni(sin[p])) : 0u; };
    unsigned long long buf = 0;
    int have = 0, pos = 1;
    unsigned nxt = word(0);
    auto getbits = [&](int m) -> unsigned {                            // m <= 32
        if (have < m) {
            buf = (buf << 32) | (unsigned long long)nxt;
            have += 32;
            nxt = word(pos);
            ++pos;
        }
        have -= m;
        return (unsigned)(buf >> have) & ones(m);
    };
    unsigned low = 0u, high = 0xFFFFFFFFu, value = getbits(32);
    const int max_symbol = Lp - 2;
    const int G = 64 / Lp;                                             // symbols per block of lanes
    unsigned cn = lane < G * Lp && lane < ncdf ? scdf[lane] : 0u;
    for (int i0 = 0; i0 < nsym; i0 += G) {
        const unsigned cv = cn;
        const int inext = (i0 + G) * Lp + lane;
        if (lane < G * Lp && inext < ncdf) cn = scdf[inext];
        const int cnt = nsym - i0 < G ? nsym - i0 : G;
        for (int j = 0; j < cnt; ++j) {
            const unsigned span32 = high - low, off = value - low;
            // largest s with cdf[s] <= ((value-low+1)*2^16 - 1) / span  <=>  (span*cdf[s]) >> 16 <= value - low
            const unsigned t = rc_scale(span32, cv);
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(t <= off) >> (j * Lp);
            // A monotone table gives a run of ones from bit 1 up and the search result is its length; any other
            // pattern walks the reference's binary search literally.
            const unsigned long long mr = (mask >> 1) & (~0ull >> (64 - max_symbol));
            int s = __popcll(mr);
            if ((mr + 1ull) & mr) {
                int left = 0, right = max_symbol + 1;
                while (left + 1 < right) {
                    const int mid = (left + right) >> 1;
                    if ((mask >> mid) & 1ull) left = mid; else right = mid;
                }
                s = left;
            }
            if (lane == 0) ssym[i0 + j] = (unsigned char)s;
            const unsigned t_low = __builtin_amdgcn_readlane(t, j * Lp + s);
            if (s != max_symbol) high = (low - 1u) + __builtin_amdgcn_readlane(t, j * Lp + s + 1);   // c_high = 2^16 leaves high as is
            low = low + t_low;
            const unsigned x = low ^ high;
            const int m = (int)x < 0 ? 0 : (x ? __clz(x) : 32);
            low = (unsigned)((unsigned long long)low << m);
            high = (unsigned)(((unsigned long long)high << m) | (unsigned long long)ones(m));
            const unsigned tt = (~low | high) << 1;
            const int k = tt ? __clz(tt) : 31;
            const unsigned fix = k ? 0x80000000u : 0u;
            low = (low << k) & ~fix;
            high = (high << k) | fix | ones(k);
            // value' = ((value << m | bits_m) << k ^ 2^31) | bits_k  =  (value << (m+k) | bits_(m+k)) ^ (k ? 2^31 : 0)
            const int n = m + k;
            if (n <= 32) {
                const unsigned nbits = getbits(n);
                value = ((unsigned)((unsigned long long)value << n) | nbits) ^ fix;
            } else {                                                   // a long E1/E2 run followed by a long E3 run (m, k <= 31)
                const unsigned v1 = (value << m) | getbits(m);
                value = ((v1 << k) ^ 0x80000000u) | getbits(k);
            }
        }
    }
    __syncthreads();
    for (int i = lane; i < nsym; i += 64) latent_q[(size_t)b * nsym + i] = (float)((int)ssym[i] - sym_offset);
}

extern "C" int pccx_range_encode(const int32_t *cdf_int, const float *latent_q, int B, int nsym, int L, uint8_t *out, int cap,
                                 int32_t *nbytes, void *stream)
{
    if (B == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(cdf_int && latent_q && out && nbytes, "pccx_range_encode: null pointer");
    PCCX_CHECK_ARG(B >= 0 && nsym >= 0 && L >= 1 && cap >= 8, "pccx_range_encode: bad shape");
    const size_t lds = (size_t)nsym * 8 + (size_t)((cap + 3) / 4) * 4;
    if (lds <= RC_MAX_LDS_BYTES) {
        hipLaunchKernelGGL(range_encode_wave_kernel, dim3(B), dim3(64), lds, (hipStream_t)stream, cdf_int, latent_q, nsym, L + 1,
                           L / 2, out, cap, nbytes);
        PCCX_CHECK_LAUNCH();
        return PCCX_OK;
    }
    hipLaunchKernelGGL(range_encode_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, cdf_int, latent_q, B, nsym,
                       L + 1, L / 2, out, cap, nbytes);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

extern "C" int pccx_range_decode(const int32_t *cdf_int, const uint8_t *in, int stride, const int32_t *nbytes, int B, int nsym,
                                 int L, float *latent_q, void *stream)
{
    if (B == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(cdf_int && in && nbytes && latent_q, "pccx_range_decode: null pointer");
    PCCX_CHECK_ARG(B >= 0 && nsym >= 0 && L >= 1 && stride >= 1, "pccx_range_decode: bad shape");
    const size_t lds = (((size_t)nsym * (L + 1) * 2 + 3) & ~(size_t)3) + (size_t)((stride + 3) / 4) * 4 + (size_t)nsym;
    if (lds <= RC_MAX_LDS_BYTES && L >= 2 && L + 1 <= 64) {
        hipLaunchKernelGGL(range_decode_wave_kernel, dim3(B), dim3(64), lds, (hipStream_t)stream, cdf_int, in, stride, nbytes, nsym,
                           L + 1, L / 2, latent_q);
        PCCX_CHECK_LAUNCH();
        return PCCX_OK;
    }
    hipLaunchKernelGGL(range_decode_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, cdf_int, in, stride, nbytes,
                       B, nsym, L + 1, L / 2, latent_q);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// torchac's float -> 16-bit CDF conversion (torchac 0.9.3 _convert_to_int_and_normalize, needs_normalization=True; the
// same formula as prob_forward_kernel's epilogue): cdf_int[l] = (round(cdf[l] * (2^16 - (Lp - 1))) + l) mod 2^16.
__global__ void cdf_float_to_int_kernel(const float *__restrict__ cdf, int64_t n, int Lp, int32_t *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int l = (int)(i % Lp);
    out[i] = ((int)rintf(__fmul_rn(cdf[i], (float)(65536 - (Lp - 1)))) + l) & 0xFFFF;
}

extern "C" int pccx_cdf_float_to_int(const float *cdf, int64_t nrows, int Lp, int32_t *cdf_int, void *stream)
{
    if (nrows == 0) return PCCX_OK;
    PCCX_CHECK_ARG(cdf && cdf_int, "pccx_cdf_float_to_int: null pointer");
    PCCX_CHECK_ARG(nrows > 0 && Lp >= 2 && Lp <= 65536, "pccx_cdf_float_to_int: bad shape");
    const int64_t n = nrows * Lp;
    hipLaunchKernelGGL(cdf_float_to_int_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cdf, n, Lp, cdf_int);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}
