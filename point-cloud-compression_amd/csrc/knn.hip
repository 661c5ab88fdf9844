// knn.hip -- neighbour search kernels for gfx950: exact K-nearest (register-resident histogram select; radix
// select in LDS for large N / K), ball query
// (ordered wave compaction) and one-directional nearest-neighbour distances (LDS-tiled all-pairs
// min-reduce, the building block of Chamfer distance and D1-PSNR).
//
// These are HBM/LDS-bound float32 + integer selection kernels, not GEMMs.  Every distance is the
// fp32 expression ((dx*dx)+dy*dy)+dz*dz with no FMA contraction, so selections are bit-identical
// to the oracle (oracle/pcc_oracle.c: orc_knn / orc_ball_query / orc_nn_dist).
#include <math.h>

#include "common.h"

// ------------------------------------------------------------------------------------------
// exact kNN: one 256-thread workgroup per query.
//   1. N squared distances -> LDS (as uint keys; d >= 0 so uint order == float order)
//   2. radix select (8 bits per pass, MSB first) of the K-th smallest composite key
//      (distance bits, index): 4 passes over the distance, plus 2 over the index only when the
//      K-th distance is tied beyond what K can take -- ties resolve to the LOWER index.
//   3. unordered compaction of the K selected (key,index) pairs, bitonic sort in LDS
//   4. write dists / idx / gathered (optionally centred + scaled) neighbours
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int block_scan_find(int h, int remaining, int *s_wsum, int *s_found, int tid)
{
    // inclusive scan of h over 256 threads; the thread whose bin crosses `remaining` publishes
    // {bin, remaining - exclusive, h}.
    const int lane = tid & 63, w = tid >> 6;
    int incl = h;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) s_wsum[w] = incl;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < w; ++k) base += s_wsum[k];
    incl += base;
    const int excl = incl - h;
    if (excl < remaining && remaining <= incl) {
        s_found[0] = tid;
        s_found[1] = remaining - excl;
        s_found[2] = h;
    }
    __syncthreads();
    return 0;
}

__global__ __launch_bounds__(256) void knn_kernel(const float *__restrict__ q, int M, const float *__restrict__ ref,
                                                  int N, int K, int Kp, float *__restrict__ dists,
                                                  int64_t *__restrict__ idx, float *__restrict__ nn, float patch_scale)
{
    extern __shared__ unsigned char smem_raw[];
    unsigned long long *sel = (unsigned long long *)smem_raw;   // [Kp]
    unsigned *keys = (unsigned *)(sel + Kp);                    // [N]
    int *hist = (int *)(keys + N);                              // [256]
    int *s_wsum = hist + 256;                                   // [4]
    int *s_found = s_wsum + 4;                                  // [3]
    int *s_cnt = s_found + 3;                                   // [1]

    const int tid = threadIdx.x;
    const int m = blockIdx.x, b = blockIdx.y;
    const float *rp = ref + (size_t)b * N * 3;
    const float qx = q[((size_t)b * M + m) * 3], qy = q[((size_t)b * M + m) * 3 + 1], qz = q[((size_t)b * M + m) * 3 + 2];

    for (int i = tid; i < N; i += 256)
        keys[i] = __float_as_uint(pccx_sqdist(qx, qy, qz, rp[3 * i], rp[3 * i + 1], rp[3 * i + 2]));
    if (tid == 0) *s_cnt = 0;

    // ---- radix select over the distance bits
    unsigned prefix = 0;
    int remaining = K, ties = 0;
    for (int p = 3; p >= 0; --p) {
        hist[tid] = 0;
        __syncthreads();
        const int sh = 8 * p;
        for (int i = tid; i < N; i += 256) {
            unsigned k = keys[i];
            bool match = (p == 3) || ((k >> (sh + 8)) == (prefix >> (sh + 8)));
            if (match) atomicAdd(&hist[(k >> sh) & 255u], 1);
        }
        __syncthreads();
        block_scan_find(hist[tid], remaining, s_wsum, s_found, tid);
        prefix |= (unsigned)s_found[0] << sh;
        remaining = s_found[1];
        ties = s_found[2];
        __syncthreads();
    }
    // prefix == bits of the K-th smallest distance; `remaining` of the `ties` equal ones are taken.
    int idx_thresh = 0x7fffffff;
    if (ties != remaining) {
        unsigned ipre = 0;
        for (int p = 1; p >= 0; --p) {
            hist[tid] = 0;
            __syncthreads();
            const int sh = 8 * p;
            for (int i = tid; i < N; i += 256) {
                bool match = keys[i] == prefix && (p == 1 || ((unsigned)i >> 8) == (ipre >> 8));
                if (match) atomicAdd(&hist[((unsigned)i >> sh) & 255u], 1);
            }
            __syncthreads();
            block_scan_find(hist[tid], remaining, s_wsum, s_found, tid);
            ipre |= (unsigned)s_found[0] << sh;
            remaining = s_found[1];
            __syncthreads();
        }
        idx_thresh = (int)ipre;    // N <= 65536: two index bytes suffice
    }

    // ---- compaction (order irrelevant: sorted next)
    for (int i = tid; i < N; i += 256) {
        unsigned k = keys[i];
        if (k < prefix || (k == prefix && i <= idx_thresh)) {
            int pos = atomicAdd(s_cnt, 1);
            if (pos < Kp) sel[pos] = ((unsigned long long)k << 32) | (unsigned)i;
        }
    }
    for (int i = K + tid; i < Kp; i += 256) sel[i] = ~0ull;
    __syncthreads();

    // ---- bitonic sort of sel[0..Kp)
    for (int size = 2; size <= Kp; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (Kp >> 1); t += 256) {
                int lo = 2 * t - (t & (stride - 1));
                int hi = lo + stride;
                bool up = ((lo & size) == 0);
                unsigned long long a = sel[lo], c = sel[hi];
                if ((a > c) == up) { sel[lo] = c; sel[hi] = a; }
            }
            __syncthreads();
        }
    }

    // ---- outputs
    const size_t ob = ((size_t)b * M + m) * K;
    for (int k = tid; k < K; k += 256) {
        unsigned long long e = sel[k];
        int i = (int)(e & 0xffffffffu);
        if (dists) dists[ob + k] = __uint_as_float((unsigned)(e >> 32));
        if (idx) idx[ob + k] = i;
        if (nn) {
            float x = rp[3 * i], y = rp[3 * i + 1], z = rp[3 * i + 2];
            if (patch_scale != 0.f) {
                // grouped_xyz -= centre (compress.py:72); x_patches * (N/N0)^(1/3) (compress.py:108)
                x = __fmul_rn(__fsub_rn(x, qx), patch_scale);
                y = __fmul_rn(__fsub_rn(y, qy), patch_scale);
                z = __fmul_rn(__fsub_rn(z, qz), patch_scale);
            }
            nn[(ob + k) * 3] = x; nn[(ob + k) * 3 + 1] = y; nn[(ob + k) * 3 + 2] = z;
        }
    }
}

// ------------------------------------------------------------------------------------------
// exact kNN, fast path (N <= 8192, K <= 256): one 256-thread workgroup per query, keys in REGISTERS.
//   1. 32 squared distances per thread (uint keys)
//   2. a histogram pass over the top 11 key bits (exponent + 3 mantissa bits: 1/8-octave bins) and a scan
//      find the bin holding the K-th smallest: every key in a lower bin is selected (set A), the keys of
//      that bin (set B, typically a few dozen) compete for the remaining r = K - |A| places.  Only when B
//      is larger than KNN_CAPB is the bin refined by the next 11 and then the last 9 key bits.
//   3. A and B are compacted into LDS; each B key counts the B keys below it (composite (distance,
//      index) order, so ties resolve to the lower index) and joins A when its rank is < r.  If B is still
//      too large after all 31 bits, its keys are all EQUAL and the r lowest indices are taken by an
//      ordered pass (the index order is the register order).
//   4. the K winners, one per thread, are sorted by a bitonic network in registers: partners inside a
//      wave by lane exchange, the three cross-wave steps through LDS
//   5. dists / idx / gathered (optionally centred + scaled) neighbours
// ------------------------------------------------------------------------------------------
#define KNN_PPT 32
#define KNN_BINS 2048
#define KNN_CAPB 256

template <int X>
__device__ __forceinline__ unsigned knn_xor_lane(unsigned v)             // value of lane (lane ^ X), X < 64
{
    if (X == 1) return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);       // quad_perm [1,0,3,2]
    if (X == 2) return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);       // quad_perm [2,3,0,1]
    if (X == 3) return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x1B, 0xF, 0xF, true);       // quad_perm [3,2,1,0]
    if (X == 7) return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true);      // row_half_mirror
    if (X == 15) return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xF, 0xF, true);     // row_mirror
    if (X < 32) return (unsigned)__builtin_amdgcn_ds_swizzle((int)v, (X << 10) | 0x1F);        // bitmask mode: lane ^ X within 32
    return (unsigned)__shfl_xor((int)v, X);
}

__device__ __forceinline__ unsigned long long knn_exchange(unsigned long long v, int x)
{
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    switch (x) {                                                          // x is a compile-time constant after unrolling
    case 1: lo = knn_xor_lane<1>(lo); hi = knn_xor_lane<1>(hi); break;
    case 2: lo = knn_xor_lane<2>(lo); hi = knn_xor_lane<2>(hi); break;
    case 3: lo = knn_xor_lane<3>(lo); hi = knn_xor_lane<3>(hi); break;
    case 4: lo = knn_xor_lane<4>(lo); hi = knn_xor_lane<4>(hi); break;
    case 7: lo = knn_xor_lane<7>(lo); hi = knn_xor_lane<7>(hi); break;
    case 8: lo = knn_xor_lane<8>(lo); hi = knn_xor_lane<8>(hi); break;
    case 15: lo = knn_xor_lane<15>(lo); hi = knn_xor_lane<15>(hi); break;
    case 16: lo = knn_xor_lane<16>(lo); hi = knn_xor_lane<16>(hi); break;
    case 31: lo = knn_xor_lane<31>(lo); hi = knn_xor_lane<31>(hi); break;
    default: lo = __shfl_xor(lo, x); hi = __shfl_xor(hi, x); break;
    }
    return ((unsigned long long)hi << 32) | lo;
}

// One level of the MSB-first search: histogram of bits [PSH-1 : SH] over the keys whose higher bits equal P,
// scan for the bin where the running count reaches r.  On return P includes the bin, r is the rank wanted inside
// it (1 <= r <= cnt) and cnt its population.
template <int PSH, int SH, int WIDTH>
__device__ __forceinline__ void knn_level(const unsigned (&key)[KNN_PPT], int *hist, int *s_wsum, int *s_found, int tid,
                                          unsigned &P, int &r, int &cnt)
{
    const int lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int t = 0; t < KNN_BINS / 256; ++t) hist[t * 256 + tid] = 0;
    __syncthreads();
#pragma unroll
    for (int t = 0; t < KNN_PPT; ++t)
        if ((key[t] >> PSH) == P) atomicAdd(&hist[(key[t] >> SH) & ((1u << WIDTH) - 1u)], 1);
    __syncthreads();
    int own[KNN_BINS / 256], s = 0;                                       // thread t owns bins 8t .. 8t+7
#pragma unroll
    for (int t = 0; t < KNN_BINS / 256; ++t) { own[t] = hist[tid * (KNN_BINS / 256) + t]; s += own[t]; }
    int incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    if (lane == 63) s_wsum[w] = incl;
    __syncthreads();
    for (int k = 0; k < w; ++k) incl += s_wsum[k];
    int excl = incl - s;
    if (excl < r && r <= incl) {
#pragma unroll
        for (int t = 0; t < KNN_BINS / 256; ++t) {
            if (excl < r && r <= excl + own[t]) { s_found[0] = tid * (KNN_BINS / 256) + t; s_found[1] = excl; s_found[2] = own[t]; }
            excl += own[t];
        }
    }
    __syncthreads();
    P = (P << WIDTH) | (unsigned)s_found[0];
    r -= s_found[1];
    cnt = s_found[2];
}

__global__ __launch_bounds__(256, 4) void knn_fast_kernel(const float *__restrict__ q, int M, const float *__restrict__ ref,
                                                       int N, int K, float *__restrict__ dists, int64_t *__restrict__ idx,
                                                       float *__restrict__ nn, float patch_scale)
{
    __shared__ int hist[KNN_BINS];
    __shared__ unsigned long long selA[256], selB[KNN_CAPB];
    __shared__ int s_wsum[4], s_found[3], s_cnt[2];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int m = blockIdx.x, b = blockIdx.y;
    const float *rp = ref + (size_t)b * N * 3;
    const size_t qo = ((size_t)b * M + m) * 3;
    const float qx = q[qo], qy = q[qo + 1], qz = q[qo + 2];
    const size_t ob = ((size_t)b * M + m) * K;

    unsigned key[KNN_PPT];
#pragma unroll
    for (int t = 0; t < KNN_PPT; ++t) {
        const int i = t * 256 + tid;
        key[t] = i < N ? __float_as_uint(pccx_sqdist(qx, qy, qz, rp[3 * i], rp[3 * i + 1], rp[3 * i + 2])) : 0x7FFFFFFFu;
    }
    if (tid < 2) s_cnt[tid] = 0;

    // ---- the K-th smallest key, MSB first: keys with (key >> sh) < P are in, those with (key >> sh) == P compete
    unsigned P = 0;
    int sh = 20, r = K, cnt_b = 0;
    knn_level<31, 20, 11>(key, hist, s_wsum, s_found, tid, P, r, cnt_b);
    if (cnt_b > KNN_CAPB) {                                               // uniform, rare: refine the bin
        sh = 9;
        knn_level<20, 9, 11>(key, hist, s_wsum, s_found, tid, P, r, cnt_b);
        if (cnt_b > KNN_CAPB) {
            sh = 0;
            knn_level<9, 0, 9>(key, hist, s_wsum, s_found, tid, P, r, cnt_b);
        }
    }
    const int nA = K - r;

    // ---- compaction (order irrelevant) without per-element atomics: count, one packed wave scan and one atomic per
    // wave for the bases, then plain stores.  (A in the low half of the packed counters, B in the high half.)
    const bool use_b = cnt_b <= KNN_CAPB;
    int cnt = 0;
#pragma unroll
    for (int t = 0; t < KNN_PPT; ++t) {
        const unsigned pre = key[t] >> sh;
        cnt += (pre < P ? 1 : 0) + (pre == P ? 0x10000 : 0);
    }
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    int base = 0;
    if (lane == 63) base = atomicAdd(&s_cnt[0], incl);
    base = __shfl(base, 63) + incl - cnt;
    int posA = base & 0xFFFF, posB = base >> 16;
#pragma unroll
    for (int t = 0; t < KNN_PPT; ++t) {
        const unsigned pre = key[t] >> sh;
        const unsigned long long e = ((unsigned long long)key[t] << 32) | (unsigned)(t * 256 + tid);
        if (pre < P) selA[posA++] = e;
        else if (pre == P && use_b) selB[posB++] = e;
    }
    __syncthreads();
    if (cnt_b <= KNN_CAPB) {
        if (tid < cnt_b) {
            const unsigned long long e = selB[tid];
            int rank = 0;
            for (int j = 0; j < cnt_b; ++j) rank += selB[j] < e ? 1 : 0;
            if (rank < r) selA[nA + rank] = e;
        }
    } else {
        // all 31 bits resolved (sh == 0): the B keys equal P.  Index order = (t, tid) order.
        int base = 0;
#pragma unroll 1
        for (int t = 0; t < KNN_PPT && base < r; ++t) {
            const bool tie = key[t] == P;
            const unsigned long long bal = __ballot(tie);
            if (lane == 0) s_wsum[w] = __popcll(bal);
            __syncthreads();
            int off = base, row = 0;
            for (int k = 0; k < 4; ++k) { off += k < w ? s_wsum[k] : 0; row += s_wsum[k]; }
            const int rank = off + __popcll(bal & ((1ull << lane) - 1ull));
            if (tie && rank < r) selA[nA + rank] = ((unsigned long long)key[t] << 32) | (unsigned)(t * 256 + tid);
            base += row;
            __syncthreads();
        }
    }
    __syncthreads();

    // ---- bitonic sort of the K winners, one per thread (padding sorts last).  Merge of two sorted runs of
    // size/2: first compare i with its mirror i ^ (size-1), then with i ^ stride for stride = size/4 .. 1; the lower
    // lane keeps the smaller key.  Partners inside a wave come by DPP (xor 1, 2, 3, 7, 15), ds_swizzle (xor 4, 8,
    // 16, 31) or ds_bpermute (xor 32, 63); the three cross-wave steps go through LDS.
    unsigned long long v = tid < K ? selA[tid] : ~0ull;
    __syncthreads();                                                      // selA is reused as the exchange buffer
#pragma unroll
    for (int size = 2; size <= 256; size <<= 1) {
#pragma unroll
        for (int step = 0, stride = size >> 1; stride > 0; ++step, stride >>= 1) {
            const int x = step == 0 ? size - 1 : stride;                  // partner = tid ^ x
            unsigned long long o;
            if (x >= 64) {
                selA[tid] = v;
                __syncthreads();
                o = selA[tid ^ x];
                __syncthreads();
            } else
                o = knn_exchange(v, x);
            const bool lower = (tid & (step == 0 ? size >> 1 : stride)) == 0;
            const bool o_less = o < v;
            v = (o_less == lower) ? o : v;
        }
    }

    // ---- outputs
    if (tid < K) {
        const int i = (int)(v & 0xffffffffu);
        if (dists) dists[ob + tid] = __uint_as_float((unsigned)(v >> 32));     // the codec's patching step takes the patches only:
        if (idx) idx[ob + tid] = i;                                             // 12 of the 24 bytes per neighbour stay unwritten
        if (nn) {
            float x = rp[3 * i], y = rp[3 * i + 1], z = rp[3 * i + 2];
            if (patch_scale != 0.f) {
                // grouped_xyz -= centre (compress.py:72); x_patches * (N/N0)^(1/3) (compress.py:108)
                x = __fmul_rn(__fsub_rn(x, qx), patch_scale);
                y = __fmul_rn(__fsub_rn(y, qy), patch_scale);
                z = __fmul_rn(__fsub_rn(z, qz), patch_scale);
            }
            nn[(ob + tid) * 3] = x; nn[(ob + tid) * 3 + 1] = y; nn[(ob + tid) * 3 + 2] = z;
        }
    }
}

extern "C" int pccx_knn(const float *q, int B, int M, const float *ref, int N, int K, float *dists, int64_t *idx,
                        float *nn, float patch_scale, void *stream)
{
    if (B == 0 || M == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(q && ref && (dists || idx || nn), "pccx_knn: null pointer (q, ref and at least one of dists / idx / nn are needed)");
    PCCX_CHECK_ARG(B >= 0 && M >= 0 && N >= 1, "pccx_knn: bad shape B=%d M=%d N=%d", B, M, N);
    PCCX_CHECK_ARG(K >= 1 && K <= N && K <= 1024, "pccx_knn: need 1 <= K <= min(N,1024), got K=%d N=%d", K, N);
    PCCX_CHECK_ARG(N <= 32768, "pccx_knn: N=%d > 32768 unsupported", N);
    PCCX_CHECK_ARG(B <= 65535, "pccx_knn: B=%d > 65535 unsupported", B);
    int Kp = 2;
    while (Kp < K) Kp <<= 1;
    size_t shmem = (size_t)Kp * 8 + (size_t)N * 4 + (256 + 4 + 3 + 1) * 4;
    PCCX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&knn_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    if (N <= 256 * KNN_PPT && K <= 256)
        hipLaunchKernelGGL(knn_fast_kernel, dim3(M, B), dim3(256), 0, (hipStream_t)stream, q, M, ref, N, K, dists, idx, nn,
                           patch_scale);
    else
        hipLaunchKernelGGL(knn_kernel, dim3(M, B), dim3(256), shmem, (hipStream_t)stream, q, M, ref, N, K, Kp, dists, idx, nn,
                           patch_scale);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// ------------------------------------------------------------------------------------------
// ball query: one wave per query, ordered compaction of the first K in-radius indices.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ball_query_kernel(const float *__restrict__ q, int M, const float *__restrict__ ref,
                                                         int N, int K, float r2, float *__restrict__ dists,
                                                         int64_t *__restrict__ idx)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int m = blockIdx.x * 4 + w, b = blockIdx.y;
    if (m >= M) return;                         // whole wave exits together
    const float *rp = ref + (size_t)b * N * 3;
    const size_t qo = ((size_t)b * M + m) * 3;
    const float qx = q[qo], qy = q[qo + 1], qz = q[qo + 2];
    const size_t ob = ((size_t)b * M + m) * K;
    int count = 0;
    for (int base = 0; base < N && count < K; base += 64) {
        int i = base + lane;
        float d = 0.f;
        bool in = false;
        if (i < N) {
            d = pccx_sqdist(qx, qy, qz, rp[3 * i], rp[3 * i + 1], rp[3 * i + 2]);
            in = d < r2;
        }
        unsigned long long mask = __ballot(in);
        int pos = count + pccx_ballot_rank(mask);
        if (in && pos < K) { dists[ob + pos] = d; idx[ob + pos] = i; }
        count += __popcll(mask);
    }
    if (count > K) count = K;
    for (int k = count + lane; k < K; k += 64) { dists[ob + k] = 0.f; idx[ob + k] = -1; }
}

extern "C" int pccx_ball_query(const float *q, int B, int M, const float *ref, int N, int K, float radius, float *dists,
                               int64_t *idx, void *stream)
{
    if (B == 0 || M == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(q && ref && dists && idx, "pccx_ball_query: null pointer");
    PCCX_CHECK_ARG(B >= 0 && M >= 0 && N >= 1 && K >= 1, "pccx_ball_query: bad shape");
    PCCX_CHECK_ARG(B <= 65535, "pccx_ball_query: B=%d > 65535 unsupported", B);
    hipLaunchKernelGGL(ball_query_kernel, dim3((M + 3) / 4, B), dim3(256), 0, (hipStream_t)stream, q, M, ref, N, K,
                       radius * radius, dists, idx);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// ------------------------------------------------------------------------------------------
// ball query through a uniform grid hash (north_star: "ball-query as a coalesced grid-hash neighbour search";
// pointnet_sa_module.py:18).  For large candidate sets and small radii the scan above visits every candidate up to the K-th hit;
// here the candidates of a cloud are binned once into cells of side >= r (bq_grid_build_kernel: bounding box, cell histogram and
// exclusive scan in LDS, scatter into a per-cloud cell-ordered index list), and a query visits only the 27 cells around its own
// (9 contiguous runs of the list: the three z-neighbours of a column are adjacent).  pytorch3d's result is the first K hits IN
// INDEX ORDER, which a cell walk does not produce: every hit sets its bit in a per-wave LDS bitmap over the N candidates, and
// the bitmap is then read out in order (popcount prefix across the lanes), which restores the index order for free.
// Results are identical to the scan (same pccx_sqdist, same strict d2 < r2 test); pccx.ops.ball_query picks the grid when the
// box is at least four cells wide and N >= 4096.
// ------------------------------------------------------------------------------------------
#define BQ_GMAX 16                                   // cells per axis (<= 4096 cells: the histogram lives in LDS)
#define BQ_NMAX 32768                                // candidates per cloud (4 KiB of bitmap per wave)

// workspace per cloud (int32): [0..7] grid parameters (lo xyz and 1/cell as float bits, G), [8 .. 8+4096] cell starts, then N sorted indices
extern "C" size_t pccx_ball_query_grid_workspace_ints(int B, int N) { return (size_t)(B > 0 ? B : 0) * (size_t)(8 + BQ_GMAX * BQ_GMAX * BQ_GMAX + 1 + (N > 0 ? N : 0)); }

__device__ __forceinline__ int bq_cell(float v, float lo, float inv, int G)
{
    int c = (int)floorf((v - lo) * inv);
    return c < 0 ? 0 : (c > G - 1 ? G - 1 : c);
}

__global__ __launch_bounds__(256) void bq_grid_build_kernel(const float *__restrict__ ref, int N, float radius, int32_t *__restrict__ ws,
                                                            size_t ws_stride)
{
    __shared__ float red[6][256];
    __shared__ int hist[BQ_GMAX * BQ_GMAX * BQ_GMAX + 1];
    __shared__ int scan_tmp[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float *rp = ref + (size_t)b * N * 3;
    int32_t *w = ws + (size_t)b * ws_stride;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = tid; i < N; i += 256)
#pragma unroll
        for (int a = 0; a < 3; ++a) { const float v = rp[3 * i + a]; lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v); }
#pragma unroll
    for (int a = 0; a < 3; ++a) { red[a][tid] = lo[a]; red[3 + a][tid] = hi[a]; }
    __syncthreads();
    for (int o = 128; o; o >>= 1) {
        if (tid < o)
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                red[a][tid] = fminf(red[a][tid], red[a][tid + o]);
                red[3 + a][tid] = fmaxf(red[3 + a][tid], red[3 + a][tid + o]);
            }
        __syncthreads();
    }
    const float ext = fmaxf(fmaxf(red[3][0] - red[0][0], red[4][0] - red[1][0]), red[5][0] - red[2][0]);
    int G = (int)floorf(ext / (radius * 1.0001f));       // cell = ext / G >= 1.0001 radius: the margin absorbs the rounding of the cell index
    G = G < 1 ? 1 : (G > BQ_GMAX ? BQ_GMAX : G);
    const float cell = ext > 0.f ? ext / (float)G : 1.f, inv = 1.f / cell;
    const float lx = red[0][0], ly = red[1][0], lz = red[2][0];
    const int cells = G * G * G;
    for (int c = tid; c <= cells; c += 256) hist[c] = 0;
    __syncthreads();
    for (int i = tid; i < N; i += 256) {
        const int c = (bq_cell(rp[3 * i], lx, inv, G) * G + bq_cell(rp[3 * i + 1], ly, inv, G)) * G + bq_cell(rp[3 * i + 2], lz, inv, G);
        atomicAdd(&hist[c], 1);
    }
    __syncthreads();
    // exclusive scan of hist[0..cells): each thread owns a contiguous slice
    const int per = (cells + 255) / 256;
    int local = 0;
    for (int k = 0; k < per; ++k) { const int c = tid * per + k; if (c < cells) local += hist[c]; }
    scan_tmp[tid] = local;
    __syncthreads();
    if (tid == 0) { int run = 0; for (int t = 0; t < 256; ++t) { const int v = scan_tmp[t]; scan_tmp[t] = run; run += v; } }
    __syncthreads();
    int run = scan_tmp[tid];
    for (int k = 0; k < per; ++k) { const int c = tid * per + k; if (c < cells) { const int v = hist[c]; hist[c] = run; run += v; } }
    if (tid == 0) hist[cells] = N;
    __syncthreads();
    for (int c = tid; c <= cells; c += 256) w[8 + c] = hist[c];
    if (tid == 0) { w[0] = __float_as_int(lx); w[1] = __float_as_int(ly); w[2] = __float_as_int(lz); w[3] = __float_as_int(inv); w[4] = G; }
    __syncthreads();                                      // hist now serves as the scatter cursors
    int32_t *sorted = w + 8 + BQ_GMAX * BQ_GMAX * BQ_GMAX + 1;
    for (int i = tid; i < N; i += 256) {
        const int c = (bq_cell(rp[3 * i], lx, inv, G) * G + bq_cell(rp[3 * i + 1], ly, inv, G)) * G + bq_cell(rp[3 * i + 2], lz, inv, G);
        sorted[atomicAdd(&hist[c], 1)] = i;               // order inside a cell is irrelevant: the bitmap restores index order
    }
}

__global__ __launch_bounds__(256) void bq_grid_query_kernel(const float *__restrict__ q, int M, const float *__restrict__ ref, int N, int K,
                                                            float r2, const int32_t *__restrict__ ws, size_t ws_stride,
                                                            float *__restrict__ dists, int64_t *__restrict__ idx)
{
    extern __shared__ unsigned bq_bits[];                 // [4 waves][words]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int m = blockIdx.x * 4 + wv, b = blockIdx.y;
    if (m >= M) return;                                   // whole wave exits together (no block barrier below)
    const int words = (N + 31) >> 5;
    unsigned *bits = bq_bits + (size_t)wv * words;
    for (int t = lane; t < words; t += 64) bits[t] = 0u;
    const int32_t *w = ws + (size_t)b * ws_stride;
    const float lx = __int_as_float(w[0]), ly = __int_as_float(w[1]), lz = __int_as_float(w[2]), inv = __int_as_float(w[3]);
    const int G = w[4];
    const int32_t *cstart = w + 8, *sorted = w + 8 + BQ_GMAX * BQ_GMAX * BQ_GMAX + 1;
    const float *rp = ref + (size_t)b * N * 3;
    const size_t qo = ((size_t)b * M + m) * 3;
    const float qx = q[qo], qy = q[qo + 1], qz = q[qo + 2];
    const int cx = bq_cell(qx, lx, inv, G), cy = bq_cell(qy, ly, inv, G), cz = bq_cell(qz, lz, inv, G);
    for (int dx = -1; dx <= 1; ++dx)
        for (int dy = -1; dy <= 1; ++dy) {
            const int ix = cx + dx, iy = cy + dy;
            if (ix < 0 || ix >= G || iy < 0 || iy >= G) continue;             // wave-uniform
            const int z0 = cz > 0 ? cz - 1 : 0, z1 = cz < G - 1 ? cz + 1 : G - 1;
            const int col = (ix * G + iy) * G;
            const int p0 = cstart[col + z0], p1 = cstart[col + z1 + 1];       // one contiguous run of the cell-ordered list
            for (int p = p0 + lane; p < p1; p += 64) {
                const int i = sorted[p];
                if (pccx_sqdist(qx, qy, qz, rp[3 * i], rp[3 * i + 1], rp[3 * i + 2]) < r2) atomicOr(&bits[i >> 5], 1u << (i & 31));
            }
        }
    // (LDS operations of one wave complete in order; the atomics above are visible to the reads below)
    const size_t ob = ((size_t)b * M + m) * K;
    int count = 0;
    for (int base = 0; base < words && count < K; base += 64) {
        const int t = base + lane;
        unsigned wbits = t < words ? bits[t] : 0u;
        const int pc = __popc(wbits);
        int incl = pc;                                    // inclusive prefix sum of the popcounts across the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        int pos = count + incl - pc;
        while (wbits && pos < K) {
            const int bit = __ffs(wbits) - 1;
            wbits &= wbits - 1;
            const int i = 32 * t + bit;
            dists[ob + pos] = pccx_sqdist(qx, qy, qz, rp[3 * i], rp[3 * i + 1], rp[3 * i + 2]);
            idx[ob + pos] = i;
            ++pos;
        }
        count += __shfl(incl, 63);
    }
    if (count > K) count = K;
    for (int k = count + lane; k < K; k += 64) { dists[ob + k] = 0.f; idx[ob + k] = -1; }
}

extern "C" int pccx_ball_query_grid(const float *q, int B, int M, const float *ref, int N, int K, float radius, int32_t *workspace,
                                    float *dists, int64_t *idx, void *stream)
{
    if (B == 0 || M == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(q && ref && dists && idx && workspace, "pccx_ball_query_grid: null pointer");
    PCCX_CHECK_ARG(B >= 0 && M >= 0 && N >= 1 && N <= BQ_NMAX && K >= 1 && radius > 0.f, "pccx_ball_query_grid: bad shape (N <= %d)", BQ_NMAX);
    PCCX_CHECK_ARG(B <= 65535, "pccx_ball_query_grid: B=%d > 65535 unsupported", B);
    const size_t stride = (size_t)8 + BQ_GMAX * BQ_GMAX * BQ_GMAX + 1 + (size_t)N;
    hipLaunchKernelGGL(bq_grid_build_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, ref, N, radius, workspace, stride);
    PCCX_CHECK_LAUNCH();
    const size_t lds = (size_t)4 * ((N + 31) / 32) * 4;
    hipLaunchKernelGGL(bq_grid_query_kernel, dim3((M + 3) / 4, B), dim3(256), lds, (hipStream_t)stream, q, M, ref, N, K, radius * radius,
                       workspace, stride, dists, idx);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// ------------------------------------------------------------------------------------------
// nn_dist: LDS-tiled all-pairs min-reduce.  A 256-thread workgroup owns 1024 query points
// (4 per thread, registers) and streams the other cloud through LDS in tiles of 1024 points
// (12 KiB); every lane reads the same LDS address (broadcast, conflict-free), so one 12-byte LDS
// read feeds 4 distance evaluations.  Algorithmic HBM bytes: 12(P+Q) + 8P per cloud; the P*Q pair
// work is VALU-bound (8 flops/pair).
// ------------------------------------------------------------------------------------------
#define NND_TILE 1024
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Two reference points per step: the three differences, squares and the two adds run as packed fp32
// (v_pk_add_f32 / v_pk_mul_f32, no contraction: the file is built with -ffp-contract=off), the same
// operation sequence per pair as pccx_sqdist.  Without the argmin the update is one v_min3_f32.
// XPT queries per thread: 4 (1024 per workgroup) for the evaluation batches; 1 when the batch is too small to fill the chip that way
// (the training step's 4 clouds: 32 workgroups become 128).  A query's scan over Y is the same sequence either way.
template <bool WANT_NN, int XPT>
// gridDim.z > 1 (pccx_nn_dist_split): workgroup z scans only the references [z qchunk, (z + 1) qchunk) and writes its partial
// (distance, index) to slice z of d2 / nn (B P entries each); nn_merge_kernel then takes the first minimum over the slices.
__global__ __launch_bounds__(256) void nn_dist_kernel(const float *__restrict__ X, int P, const float *__restrict__ Y, int Q,
                                                      float *__restrict__ d2, int32_t *__restrict__ nn, int qchunk)
{
    __shared__ __attribute__((aligned(8))) float tyx[NND_TILE], tyy[NND_TILE], tyz[NND_TILE];
    const int b = blockIdx.y, tid = threadIdx.x;
    const float *xp = X + (size_t)b * P * 3;
    const float *yp = Y + (size_t)b * Q * 3;
    float x[XPT][3], best[XPT];
    int bi[XPT];
#pragma unroll
    for (int j = 0; j < XPT; ++j) {
        int i = blockIdx.x * (256 * XPT) + j * 256 + tid;
        int ii = i < P ? i : P - 1;
        x[j][0] = xp[3 * ii]; x[j][1] = xp[3 * ii + 1]; x[j][2] = xp[3 * ii + 2];
        best[j] = INFINITY; bi[j] = -1;
    }
    const int q_begin = blockIdx.z * qchunk, q_end = q_begin + qchunk < Q ? q_begin + qchunk : Q;
    const size_t slice = (size_t)blockIdx.z * gridDim.y * P;
    for (int base = q_begin; base < q_end; base += NND_TILE) {
        const int cnt = q_end - base < NND_TILE ? q_end - base : NND_TILE;
        const int cnt2 = (cnt + 1) & ~1;                  // an odd tail repeats its last point: no effect on min / first argmin
        __syncthreads();
        for (int t = tid; t < cnt2; t += 256) {
            const size_t src = (size_t)(base + (t < cnt ? t : cnt - 1)) * 3;
            tyx[t] = yp[src]; tyy[t] = yp[src + 1]; tyz[t] = yp[src + 2];
        }
        __syncthreads();
        for (int t = 0; t < cnt2; t += 2) {
            const f32x2 yx = *(const f32x2 *)&tyx[t], yy = *(const f32x2 *)&tyy[t], yz = *(const f32x2 *)&tyz[t];
#pragma unroll
            for (int j = 0; j < XPT; ++j) {
                const f32x2 dx = x[j][0] - yx, dy = x[j][1] - yy, dz = x[j][2] - yz;
                f32x2 d = dx * dx;
                d = d + dy * dy;
                d = d + dz * dz;
                if (WANT_NN) {
                    if (d.x < best[j]) { best[j] = d.x; bi[j] = base + t; }
                    if (d.y < best[j]) { best[j] = d.y; bi[j] = base + t + 1; }
                } else
                    best[j] = fminf(fminf(best[j], d.x), d.y);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < XPT; ++j) {
        int i = blockIdx.x * (256 * XPT) + j * 256 + tid;
        if (i < P) {
            d2[slice + (size_t)b * P + i] = best[j];
            if (WANT_NN) nn[slice + (size_t)b * P + i] = bi[j];
        }
    }
}

extern "C" int pccx_nn_dist(const float *X, int B, int P, const float *Y, int Q, float *d2, int32_t *nn, void *stream)
{
    if (B == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(X && Y && d2, "pccx_nn_dist: null pointer");
    PCCX_CHECK_ARG(B >= 0 && P >= 1 && Q >= 1, "pccx_nn_dist: bad shape");
    PCCX_CHECK_ARG(B <= 65535, "pccx_nn_dist: B=%d > 65535 unsupported", B);
    const bool small = (long long)B * ((P + 1023) / 1024) < 512;
    const int xpt = small ? 1 : 4;
    const int gx = (P + 256 * xpt - 1) / (256 * xpt);
    hipStream_t st = (hipStream_t)stream;
    if (nn) {
        if (small) hipLaunchKernelGGL((nn_dist_kernel<true, 1>), dim3(gx, B), dim3(256), 0, st, X, P, Y, Q, d2, nn, Q);
        else hipLaunchKernelGGL((nn_dist_kernel<true, 4>), dim3(gx, B), dim3(256), 0, st, X, P, Y, Q, d2, nn, Q);
    } else {
        if (small) hipLaunchKernelGGL((nn_dist_kernel<false, 1>), dim3(gx, B), dim3(256), 0, st, X, P, Y, Q, d2, nn, Q);
        else hipLaunchKernelGGL((nn_dist_kernel<false, 4>), dim3(gx, B), dim3(256), 0, st, X, P, Y, Q, d2, nn, Q);
    }
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// first minimum over the `split` partial results of pccx_nn_dist_split: slice z holds the references of chunk z in ascending order, so
// taking a later slice only on a strictly smaller distance keeps the lowest index among equal distances, as the unsplit scan does
__global__ void nn_merge_kernel(const float *__restrict__ pd, const int32_t *__restrict__ pn, long n, int split, float *__restrict__ d2,
                                int32_t *__restrict__ nn)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float best = pd[i];
        int bi = pn ? pn[i] : -1;
        for (int z = 1; z < split; ++z) {
            const float v = pd[(size_t)z * n + i];
            if (v < best) { best = v; bi = pn ? pn[(size_t)z * n + i] : -1; }
        }
        d2[i] = best;
        if (nn) nn[i] = bi;
    }
}

// How many reference chunks pccx_nn_dist_split should use for this shape: 1 when the plain launch already fills the chip (one
// 256-query workgroup per CU), else enough chunks for about two workgroups per CU, each chunk a multiple of the LDS tile
extern "C" int pccx_nn_dist_split_count(int B, int P, int Q)
{
    const long long wgs = (long long)B * ((P + 255) / 256);
    if (wgs <= 0 || wgs >= 256 || Q < 2 * NND_TILE) return 1;
    int split = (int)((512 + wgs - 1) / wgs);
    const int max_split = Q / NND_TILE;
    if (split > max_split) split = max_split;
    if (split > 16) split = 16;
    return split < 1 ? 1 : split;
}

// pccx_nn_dist for batches too small to fill the chip (the training step's 4 clouds of 8192 points: 128 workgroups): the reference
// cloud is cut into `split` chunks scanned by different workgroups, the partial results (split x B x P distances and, when nn is
// asked for, indices: `scratch_d` / `scratch_nn` from the caller) are merged by a second small kernel.  Same results as pccx_nn_dist.
extern "C" int pccx_nn_dist_split(const float *X, int B, int P, const float *Y, int Q, int split, float *scratch_d, int32_t *scratch_nn,
                                  float *d2, int32_t *nn, void *stream)
{
    if (B == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(X && Y && d2 && scratch_d && (!nn || scratch_nn), "pccx_nn_dist_split: null pointer");
    PCCX_CHECK_ARG(B >= 0 && B <= 65535 && P >= 1 && Q >= 1 && split >= 1 && split <= 64, "pccx_nn_dist_split: bad shape (split=%d)", split);
    int qchunk = (Q + split - 1) / split;
    qchunk = (qchunk + NND_TILE - 1) / NND_TILE * NND_TILE;
    split = (Q + qchunk - 1) / qchunk;
    hipStream_t st = (hipStream_t)stream;
    const int gx = (P + 255) / 256;
    if (nn) hipLaunchKernelGGL((nn_dist_kernel<true, 1>), dim3(gx, B, split), dim3(256), 0, st, X, P, Y, Q, scratch_d, scratch_nn, qchunk);
    else hipLaunchKernelGGL((nn_dist_kernel<false, 1>), dim3(gx, B, split), dim3(256), 0, st, X, P, Y, Q, scratch_d, (int32_t *)nullptr, qchunk);
    const long n = (long)B * P;
    long blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(nn_merge_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const float *)scratch_d, (const int32_t *)(nn ? scratch_nn : nullptr), n, split, d2, nn);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// ------------------------------------------------------------------------------------------
// D2 (point-to-plane) support for eval.py:58-60,73-81.
//   normals_pca_kernel : per point, PCA of its K nearest neighbours (open3d estimate_normals with
//                        KDTreeSearchParamKNN): fp64 covariance, analytic symmetric 3x3 eigen-solve,
//                        eigenvector of the smallest eigenvalue.  The sign is arbitrary (as open3d's
//                        unoriented normals); D2 squares the projection.  PARITY UNPINNED vs open3d.
//   plane_err_kernel   : ((p - q) . n_q)^2 for every reconstructed p and its nearest original q.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void cross3(const double *a, const double *b, double *o)
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

__global__ void normals_pca_kernel(const float *__restrict__ xyz, int N, const int64_t *__restrict__ nbr, int K,
                                   float *__restrict__ normals)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float *p = xyz + (size_t)b * N * 3;
    const int64_t *nb = nbr + ((size_t)b * N + i) * K;
    double m[3] = {0, 0, 0};
    for (int k = 0; k < K; ++k)
        for (int a = 0; a < 3; ++a) m[a] += (double)p[3 * nb[k] + a];
    for (int a = 0; a < 3; ++a) m[a] /= K;
    double c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0;
    for (int k = 0; k < K; ++k) {
        const double x = (double)p[3 * nb[k]] - m[0], y = (double)p[3 * nb[k] + 1] - m[1], z = (double)p[3 * nb[k] + 2] - m[2];
        c00 += x * x; c01 += x * y; c02 += x * z; c11 += y * y; c12 += y * z; c22 += z * z;
    }
    c00 /= K; c01 /= K; c02 /= K; c11 /= K; c12 /= K; c22 /= K;
    // smallest eigenvalue of the symmetric matrix (trigonometric closed form), on a scale-normalised copy
    const double scale = fmax(fmax(fabs(c00), fabs(c11)), fmax(fabs(c22), fmax(fabs(c01), fmax(fabs(c02), fabs(c12)))));
    double n[3] = {0, 0, 1};
    if (scale > 0) {
        const double a00 = c00 / scale, a01 = c01 / scale, a02 = c02 / scale, a11 = c11 / scale, a12 = c12 / scale, a22 = c22 / scale;
        const double q = (a00 + a11 + a22) / 3;
        const double p1 = a01 * a01 + a02 * a02 + a12 * a12;
        const double p2 = (a00 - q) * (a00 - q) + (a11 - q) * (a11 - q) + (a22 - q) * (a22 - q) + 2 * p1;
        double lam = q;
        if (p2 > 0) {
            const double pp = sqrt(p2 / 6);
            const double b00 = (a00 - q) / pp, b01 = a01 / pp, b02 = a02 / pp, b11 = (a11 - q) / pp, b12 = a12 / pp, b22 = (a22 - q) / pp;
            double r = (b00 * (b11 * b22 - b12 * b12) - b01 * (b01 * b22 - b12 * b02) + b02 * (b01 * b12 - b11 * b02)) / 2;
            r = r < -1 ? -1 : (r > 1 ? 1 : r);
            const double phi = acos(r) / 3;
            lam = q + 2 * pp * cos(phi + 2.0943951023931953);       // smallest eigenvalue (phi + 2*pi/3)
        }
        // eigenvector: the largest cross product of two rows of (A - lam I)
        const double r0[3] = {a00 - lam, a01, a02}, r1[3] = {a01, a11 - lam, a12}, r2[3] = {a02, a12, a22 - lam};
        double c0[3], c1[3], c2[3];
        cross3(r0, r1, c0); cross3(r0, r2, c1); cross3(r1, r2, c2);
        const double d0 = c0[0] * c0[0] + c0[1] * c0[1] + c0[2] * c0[2];
        const double d1 = c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2];
        const double d2 = c2[0] * c2[0] + c2[1] * c2[1] + c2[2] * c2[2];
        const double *best = d0 >= d1 && d0 >= d2 ? c0 : (d1 >= d2 ? c1 : c2);
        const double dm = fmax(d0, fmax(d1, d2));
        if (dm > 0) {
            const double inv = 1.0 / sqrt(dm);
            n[0] = best[0] * inv; n[1] = best[1] * inv; n[2] = best[2] * inv;
        }
    }
    float *o = normals + ((size_t)b * N + i) * 3;
    o[0] = (float)n[0]; o[1] = (float)n[1]; o[2] = (float)n[2];
}

extern "C" int pccx_estimate_normals(const float *xyz, int B, int N, const int64_t *nbr, int K, float *normals, void *stream)
{
    if (B == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(xyz && nbr && normals, "pccx_estimate_normals: null pointer");
    PCCX_CHECK_ARG(B >= 0 && N >= 1 && K >= 1 && B <= 65535, "pccx_estimate_normals: bad shape");
    hipLaunchKernelGGL(normals_pca_kernel, dim3((N + 127) / 128, B), dim3(128), 0, (hipStream_t)stream, xyz, N, nbr, K, normals);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

__global__ void plane_err_kernel(const float *__restrict__ X, int P, const float *__restrict__ Y, const float *__restrict__ nY,
                                 int Q, const int32_t *__restrict__ nn, float *__restrict__ err)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const float *x = X + ((size_t)b * P + i) * 3;
    const int j = nn[(size_t)b * P + i];
    const float *y = Y + ((size_t)b * Q + j) * 3, *n = nY + ((size_t)b * Q + j) * 3;
    const double d = (double)(x[0] - y[0]) * n[0] + (double)(x[1] - y[1]) * n[1] + (double)(x[2] - y[2]) * n[2];
    err[(size_t)b * P + i] = (float)(d * d);
}

extern "C" int pccx_point_plane_err(const float *X, int B, int P, const float *Y, const float *normals_Y, int Q,
                                    const int32_t *nn, float *err, void *stream)
{
    if (B == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(X && Y && normals_Y && nn && err, "pccx_point_plane_err: null pointer");
    PCCX_CHECK_ARG(B >= 0 && P >= 1 && Q >= 1 && B <= 65535, "pccx_point_plane_err: bad shape");
    hipLaunchKernelGGL(plane_err_kernel, dim3((P + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, X, P, Y, normals_Y, Q, nn, err);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// ------------------------------------------------------------------------------------------
// Gradient of pytorch3d-style chamfer_distance (AE.py:57-70 get_loss, pppe_pcd_ae.py:817-838):
//   L = mean_b [ mean_i min_j |x_i - y_j|^2 + mean_j min_i |x_i - y_j|^2 ]
// With the argmins fixed, dL/dx_i = 2 wx (x_i - y_nn(i)) + sum_{j: nn'(j) = i} 2 wy (x_i - y_j),
// wx = g/(B P), wy = g/(B Q), and symmetrically for y.  The second term is a scatter: fp32 atomics
// (one 12-byte row per lane; order-dependent in the last bits, as any atomic sum).
// ------------------------------------------------------------------------------------------
// scatter of one 3-vector per lane into row `j` of `dst`, with the wave's heavy hitters combined first: up to four rounds take the row of
// the first remaining lane, sum the vectors of every lane that targets the same row (a masked butterfly) and issue ONE atomic per
// component; lanes left after that add their own.  Why: with an untrained decoder the reconstruction is a blob, thousands of points share
// a handful of nearest neighbours, and the plain form serialised ~25 k same-address atomics per cloud (198 us per call for 4 clouds of
// 8192 points); rows that are all distinct pay four ballots.
__device__ __forceinline__ void scatter3_combined(float *dst, int j, bool on, float d0, float d1, float d2)
{
    unsigned long long todo = __ballot(on);
    for (int round = 0; round < 4 && todo; ++round) {
        const int leader = __ffsll((long long)todo) - 1;
        const int jl = __shfl(j, leader);
        const bool mine = on && j == jl;
        const unsigned long long grp = __ballot(mine);
        if (__popcll(grp) > 1) {
            float s0 = mine ? d0 : 0.f, s1 = mine ? d1 : 0.f, s2 = mine ? d2 : 0.f;
#pragma unroll
            for (int o = 32; o; o >>= 1) {
                s0 += __shfl_xor(s0, o);
                s1 += __shfl_xor(s1, o);
                s2 += __shfl_xor(s2, o);
            }
            if ((int)(threadIdx.x & 63) == leader) {
                atomicAdd(&dst[3 * jl + 0], s0);
                atomicAdd(&dst[3 * jl + 1], s1);
                atomicAdd(&dst[3 * jl + 2], s2);
            }
            if (mine) on = false;
        } else if (mine) {                                    // a row of its own: nothing to combine
            atomicAdd(&dst[3 * j + 0], d0);
            atomicAdd(&dst[3 * j + 1], d1);
            atomicAdd(&dst[3 * j + 2], d2);
            on = false;
        }
        todo &= ~grp;
    }
    if (on) {
        atomicAdd(&dst[3 * j + 0], d0);
        atomicAdd(&dst[3 * j + 1], d1);
        atomicAdd(&dst[3 * j + 2], d2);
    }
}

__global__ void chamfer_grad_kernel(const float *__restrict__ X, int P, const float *__restrict__ Y, int Q,
                                    const int32_t *__restrict__ nn_xy, const int32_t *__restrict__ nn_yx, float wx, float wy,
                                    const float *__restrict__ g_dev, float *__restrict__ gX, float *__restrict__ gY)
{
    if (g_dev) { wx *= *g_dev; wy *= *g_dev; }        // upstream gradient read on the device (no host sync; hipGraph capture)
    const int b = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const float *x = X + (size_t)b * P * 3, *y = Y + (size_t)b * Q * 3;
    float *gx = gX + (size_t)b * P * 3, *gy = gY + (size_t)b * Q * 3;
    {
        const bool on = t < P;
        int j = 0;
        float d[3] = {0.f, 0.f, 0.f};
        if (on) {
            j = nn_xy[(size_t)b * P + t];
            for (int a = 0; a < 3; ++a) {
                d[a] = 2.f * wx * (x[3 * t + a] - y[3 * j + a]);
                atomicAdd(&gx[3 * t + a], d[a]);
            }
        }
        scatter3_combined(gy, j, on, -d[0], -d[1], -d[2]);
    }
    {
        const bool on = t < Q;
        int i = 0;
        float d[3] = {0.f, 0.f, 0.f};
        if (on) {
            i = nn_yx[(size_t)b * Q + t];
            for (int a = 0; a < 3; ++a) {
                d[a] = 2.f * wy * (y[3 * t + a] - x[3 * i + a]);
                atomicAdd(&gy[3 * t + a], d[a]);
            }
        }
        scatter3_combined(gx, i, on, -d[0], -d[1], -d[2]);
    }
}

// chamfer_distance's value from the two nearest-neighbour passes (pytorch3d defaults: point_reduction = batch_reduction = "mean"):
//   out = (1/B) sum_b [ (1/P) sum_p dxy[b,p] + (1/Q) sum_q dyx[b,q] ]   accumulated in double, stored as float.
// One workgroup, one launch (round 3 formed it with six torch kernels: two casts, two means, an add, a mean).
__global__ __launch_bounds__(1024) void chamfer_mean_kernel(const float *__restrict__ dxy, const float *__restrict__ dyx, int B, int P, int Q,
                                                            float *__restrict__ out)
{
    __shared__ double red[16];
    double tot = 0;
    for (int b = 0; b < B; ++b) {
        double sx = 0, sy = 0;
        for (int i = threadIdx.x; i < P; i += 1024) sx += (double)dxy[(size_t)b * P + i];
        for (int i = threadIdx.x; i < Q; i += 1024) sy += (double)dyx[(size_t)b * Q + i];
        tot += sx / (double)P + sy / (double)Q;
    }
    for (int o = 32; o; o >>= 1) tot += __shfl_xor(tot, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = tot;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0;
        for (int w = 0; w < 16; ++w) t += red[w];
        out[0] = (float)(t / (double)B);
    }
}

extern "C" int pccx_chamfer_mean(const float *dxy, const float *dyx, int B, int P, int Q, float *out, void *stream)
{
    PCCX_CHECK_ARG(dxy && dyx && out && B >= 1 && P >= 1 && Q >= 1, "pccx_chamfer_mean: bad arguments");
    hipLaunchKernelGGL(chamfer_mean_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, dxy, dyx, B, P, Q, out);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

static int chamfer_grad_launch(const float *X, int B, int P, const float *Y, int Q, const int32_t *nn_xy, const int32_t *nn_yx,
                               float grad_out, const float *g_dev, float *gX, float *gY, void *stream, bool prezeroed = false);

extern "C" int pccx_chamfer_grad(const float *X, int B, int P, const float *Y, int Q, const int32_t *nn_xy,
                                 const int32_t *nn_yx, float grad_out, float *gX, float *gY, void *stream)
{
    return chamfer_grad_launch(X, B, P, Y, Q, nn_xy, nn_yx, grad_out, nullptr, gX, gY, stream);
}

// the same with the upstream gradient (a scalar) read from device memory
extern "C" int pccx_chamfer_grad_dev(const float *X, int B, int P, const float *Y, int Q, const int32_t *nn_xy,
                                     const int32_t *nn_yx, const float *grad_out_dev, float *gX, float *gY, void *stream)
{
    PCCX_CHECK_ARG(grad_out_dev, "pccx_chamfer_grad_dev: null pointer");
    return chamfer_grad_launch(X, B, P, Y, Q, nn_xy, nn_yx, 1.0f, grad_out_dev, gX, gY, stream);
}

// the same into gX / gY the caller has cleared (flags & 4): no clearing launches
extern "C" int pccx_chamfer_grad_dev_acc(const float *X, int B, int P, const float *Y, int Q, const int32_t *nn_xy,
                                         const int32_t *nn_yx, const float *grad_out_dev, float *gX, float *gY, int flags, void *stream)
{
    PCCX_CHECK_ARG(grad_out_dev, "pccx_chamfer_grad_dev_acc: null pointer");
    return chamfer_grad_launch(X, B, P, Y, Q, nn_xy, nn_yx, 1.0f, grad_out_dev, gX, gY, stream, (flags & 4) != 0);
}

static int chamfer_grad_launch(const float *X, int B, int P, const float *Y, int Q, const int32_t *nn_xy, const int32_t *nn_yx,
                               float grad_out, const float *g_dev, float *gX, float *gY, void *stream, bool prezeroed)
{
    if (B == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(X && Y && nn_xy && nn_yx && gX && gY, "pccx_chamfer_grad: null pointer");
    PCCX_CHECK_ARG(P >= 1 && Q >= 1 && B <= 65535, "pccx_chamfer_grad: bad shape");
    if (!prezeroed) {
        PCCX_CHECK_HIP(pccx_zero_async(gX, sizeof(float) * (size_t)B * P * 3, (hipStream_t)stream));
        PCCX_CHECK_HIP(pccx_zero_async(gY, sizeof(float) * (size_t)B * Q * 3, (hipStream_t)stream));
    }
    const int n = P > Q ? P : Q;
    hipLaunchKernelGGL(chamfer_grad_kernel, dim3((n + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, X, P, Y, Q, nn_xy, nn_yx,
                       grad_out / ((float)B * P), grad_out / ((float)B * Q), g_dev, gX, gY);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}
