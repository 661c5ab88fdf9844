// rangecoder.hip -- torchac-style range coder (compress.py:136 encode_float_cdf, decompress.py:93
// decode_float_cdf) on the GPU: 32-bit low/high, 16-bit CDFs, E1/E2/E3 renormalisation with pending
// bits, MSB-first bit packing, zero-padded last byte.  Byte layout PARITY UNPINNED (torchac is not in
// the image); the oracle (orc_range_encode / orc_range_decode) restates the same published algorithm
// and must agree byte-for-byte.
//
// The coder is inherently serial per stream (S*d = 1024 symbols per cloud).  One wave per cloud:
// the 64 lanes stage the cloud's CDF table (as uint16), symbols and output bytes through LDS with
// coalesced transfers, and lane 0 runs the serial recurrence against LDS (~64-cycle reads instead of
// dependent ~1 us global loads).  Streams larger than the LDS budget fall back to one lane per cloud
// working from global memory.
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
struct BulkWriter {
    uint8_t *buf;
    int cap, n;
    unsigned long long acc;     // low `na` bits are pending output
    int na;
    __device__ __forceinline__ void put(unsigned v, int cnt)          // cnt <= 32
    {
        if (cnt == 0) return;
        acc = (acc << cnt) | (unsigned long long)v;
        na += cnt;
        while (na >= 8) {
            if (n < cap) buf[n] = (uint8_t)(acc >> (na - 8));
            ++n;
            na -= 8;
        }
    }
    __device__ __forceinline__ void run(int bit, unsigned long long cnt)
    {
        while (cnt > 0) {
            const int c = cnt > 32 ? 32 : (int)cnt;
            put(bit ? (c == 32 ? 0xFFFFFFFFu : ((1u << c) - 1u)) : 0u, c);
            cnt -= c;
        }
    }
    __device__ __forceinline__ void flush()
    {
        if (na > 0) {
            if (n < cap) buf[n] = (uint8_t)(acc << (8 - na));
            ++n;
            na = 0;
        }
    }
};

__device__ __forceinline__ unsigned ones(int m) { return m >= 32 ? 0xFFFFFFFFu : ((1u << m) - 1u); }

__global__ __launch_bounds__(64) void range_encode_lds_kernel(const int32_t *__restrict__ cdf_int, const float *__restrict__ latent_q,
                                                              int nsym, int Lp, int sym_offset, uint8_t *__restrict__ out, int cap,
                                                              int32_t *__restrict__ nbytes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char rc_smem[];
    unsigned short *scdf = (unsigned short *)rc_smem;                 // [nsym*Lp]
    unsigned char *ssym = (unsigned char *)(scdf + (size_t)nsym * Lp); // [nsym]
    uint8_t *sout = ssym + ((nsym + 15) & ~15);                        // [cap]
    const int b = blockIdx.x, lane = threadIdx.x;
    const int32_t *c = cdf_int + (size_t)b * nsym * Lp;
    const int max_symbol = Lp - 2;
    for (int i = lane; i < nsym * Lp; i += 64) scdf[i] = (unsigned short)c[i];
    for (int i = lane; i < nsym; i += 64) {
        int s = (int)latent_q[(size_t)b * nsym + i] + sym_offset;
        ssym[i] = (unsigned char)(s < 0 ? 0 : (s > max_symbol ? max_symbol : s));
    }
    __syncthreads();
    int n_out = 0;
    if (lane == 0) {
        BulkWriter w{sout, cap, 0, 0ull, 0};
        unsigned low = 0u, high = 0xFFFFFFFFu;
        unsigned long long pending = 0;
        for (int i = 0; i < nsym; ++i) {
            const int s = ssym[i];
            const unsigned short *ci = scdf + (size_t)i * Lp;
            const unsigned long long span = (unsigned long long)high - (unsigned long long)low + 1ull;
            const unsigned c_low = ci[s];
            const unsigned c_high = s == max_symbol ? 0x10000u : (unsigned)ci[s + 1];
            high = (unsigned)((low - 1u) + (unsigned)((span * c_high) >> 16));
            low = (unsigned)(low + (unsigned)((span * c_low) >> 16));
            for (;;) {
                const unsigned x = low ^ high;
                if ((int)x >= 0) {                                     // E1/E2 run: top bits agree
                    const int m = x ? __clz(x) : 32;
                    const unsigned b0 = low >> 31;
                    w.put(b0, 1);
                    w.run(!b0, pending);
                    pending = 0;
                    if (m > 1) w.put((low << 1) >> (32 - (m - 1)), m - 1);
                    low = m >= 32 ? 0u : low << m;
                    high = m >= 32 ? 0xFFFFFFFFu : ((high << m) | ones(m));
                } else if ((low & 0x40000000u) && !(high & 0x40000000u)) {   // E3 run
                    const unsigned t = (~low | high) << 1;
                    const int k = t ? __clz(t) : 31;
                    low = (low << k) & 0x7FFFFFFFu;
                    high = (high << k) | 0x80000000u | ones(k);
                    pending += k;
                } else
                    break;
            }
        }
        ++pending;
        const int fb = low < 0x40000000u ? 0 : 1;
        w.put(fb, 1);
        w.run(!fb, pending);
        w.flush();
        n_out = w.n;
        nbytes[b] = w.n <= cap ? w.n : -w.n;
    }
    n_out = __shfl(n_out, 0);
    __syncthreads();
    if (n_out > cap) n_out = cap;
    for (int i = lane; i < n_out; i += 64) out[(size_t)b * cap + i] = sout[i];
}

__global__ __launch_bounds__(64) void range_decode_lds_kernel(const int32_t *__restrict__ cdf_int, const uint8_t *__restrict__ in,
                                                              int stride, const int32_t *__restrict__ nbytes, int nsym, int Lp,
                                                              int sym_offset, float *__restrict__ latent_q)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char rc_smem[];
    unsigned short *scdf = (unsigned short *)rc_smem;                 // [nsym*Lp]
    unsigned char *ssym = (unsigned char *)(scdf + (size_t)nsym * Lp); // [nsym]
    uint8_t *sin = ssym + ((nsym + 15) & ~15);                         // [nb]
    const int b = blockIdx.x, lane = threadIdx.x;
    const int32_t *c = cdf_int + (size_t)b * nsym * Lp;
    const int nb = nbytes[b] < 0 ? 0 : (nbytes[b] > stride ? stride : nbytes[b]);
    for (int i = lane; i < nsym * Lp; i += 64) scdf[i] = (unsigned short)c[i];
    for (int i = lane; i < nb; i += 64) sin[i] = in[(size_t)b * stride + i];
    __syncthreads();
    if (lane == 0) {
        // bit reader: `buf` holds `have` unread bits in its low end; bytes past the stream read as 0
        unsigned long long buf = 0;
        int have = 0, pos = 0;
        auto getbits = [&](int m) -> unsigned {                        // m <= 32
            if (m == 0) return 0u;
            while (have < m) {
                buf = (buf << 8) | (unsigned long long)(pos < nb ? sin[pos] : 0);
                ++pos;
                have += 8;
            }
            have -= m;
            return (unsigned)((buf >> have) & (m >= 32 ? 0xFFFFFFFFull : ((1ull << m) - 1ull)));
        };
        unsigned low = 0u, high = 0xFFFFFFFFu, value = getbits(32);
        const int max_symbol = Lp - 2;
        for (int i = 0; i < nsym; ++i) {
            const unsigned short *ci = scdf + (size_t)i * Lp;
            const unsigned long long span = (unsigned long long)high - (unsigned long long)low + 1ull;
            // largest s with cdf[s] <= ((value-low+1)*2^16 - 1) / span  <=>  (span*cdf[s]) >> 16 <= value - low
            const unsigned off = value - low;
            int left = 0, right = max_symbol + 1;
            while (left + 1 < right) {
                const int mid = (left + right) >> 1;
                if ((unsigned)((span * (unsigned)ci[mid]) >> 16) <= off) left = mid; else right = mid;
            }
            const int s = left;
            ssym[i] = (unsigned char)s;
            const unsigned c_low = ci[s];
            const unsigned c_high = s == max_symbol ? 0x10000u : (unsigned)ci[s + 1];
            high = (unsigned)((low - 1u) + (unsigned)((span * c_high) >> 16));
            low = (unsigned)(low + (unsigned)((span * c_low) >> 16));
            for (;;) {
                const unsigned x = low ^ high;
                if ((int)x >= 0) {
                    const int m = x ? __clz(x) : 32;
                    const unsigned nbits = getbits(m);
                    low = m >= 32 ? 0u : low << m;
                    high = m >= 32 ? 0xFFFFFFFFu : ((high << m) | ones(m));
                    value = m >= 32 ? nbits : ((value << m) | nbits);
                } else if ((low & 0x40000000u) && !(high & 0x40000000u)) {
                    const unsigned t = (~low | high) << 1;
                    const int k = t ? __clz(t) : 31;
                    const unsigned nbits = getbits(k);
                    low = (low << k) & 0x7FFFFFFFu;
                    high = (high << k) | 0x80000000u | ones(k);
                    value = ((value << k) ^ 0x80000000u) | nbits;
                } else
                    break;
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
    if (B == 0) return PCCX_OK;
    const size_t lds = (size_t)nsym * (L + 1) * 2 + ((nsym + 15) & ~15) + (size_t)cap;
    if (lds <= RC_MAX_LDS_BYTES && L + 1 <= 256) {
        hipLaunchKernelGGL(range_encode_lds_kernel, dim3(B), dim3(64), lds, (hipStream_t)stream, cdf_int, latent_q, nsym, L + 1,
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
    if (B == 0) return PCCX_OK;
    const size_t lds = (size_t)nsym * (L + 1) * 2 + ((nsym + 15) & ~15) + (size_t)stride;
    if (lds <= RC_MAX_LDS_BYTES && L + 1 <= 256) {
        hipLaunchKernelGGL(range_decode_lds_kernel, dim3(B), dim3(64), lds, (hipStream_t)stream, cdf_int, in, stride, nbytes, nsym,
                           L + 1, L / 2, latent_q);
        PCCX_CHECK_LAUNCH();
        return PCCX_OK;
    }
    hipLaunchKernelGGL(range_decode_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, cdf_int, in, stride, nbytes,
                       B, nsym, L + 1, L / 2, latent_q);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}
