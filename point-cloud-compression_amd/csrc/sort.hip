// sort.hip -- the block partition of room-scale clouds (BASELINE configs[3]; pccx/large.py) without library sorts:
//   * pccx_sort_keys_u64: a STABLE least-significant-digit radix sort of (63-bit Morton key, point index) pairs -- the order
//     torch.sort(keys, stable=True).indices gives, which round 3 used (rocPRIM merge sort + four torch index kernels per room);
//   * pccx_gather_blocks / pccx_scatter_blocks: the rows of a room in that order cut into blocks of `block` points (the last block
//     completed with copies of its final point) and the inverse (decoded blocks written back at the original row indices, padding
//     dropped), for the blocks first, first + stride, ... of a rank.
// HBM-bound integer work: per pass every key (8 B) and index (4 B) is read twice and written once; 8 passes of 8 bits.
#include "common.h"

#define SORT_ITEMS 16
#define SORT_TILE (256 * SORT_ITEMS)

// counts of the pass's digit per tile: hist[tile][256]
__global__ __launch_bounds__(256) void radix_hist_kernel(const unsigned long long *__restrict__ keys, long long n, int shift,
                                                         unsigned *__restrict__ hist)
{
    __shared__ unsigned h[256];
    const int tid = threadIdx.x;
    h[tid] = 0u;
    __syncthreads();
    const long long base = (long long)blockIdx.x * SORT_TILE;
#pragma unroll 4
    for (int i = 0; i < SORT_ITEMS; ++i) {
        const long long p = base + (long long)i * 256 + tid;
        if (p < n) atomicAdd(&h[(unsigned)(keys[p] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(size_t)blockIdx.x * 256 + tid] = h[tid];
}

// One pass: every tile finds where its keys of each digit start (all tiles' counts of lower digits + earlier tiles' counts of the
// same digit: summed here from the table, thread d = digit d, instead of a scan kernel of its own), then places its keys round by
// round (256 consecutive keys per round, so earlier keys come first): rank among equal digits of the round = match-any inside the
// wave by eight ballots + the counts of the lower waves.
__global__ __launch_bounds__(256) void radix_scatter_kernel(const unsigned long long *__restrict__ keys_in, const unsigned *__restrict__ idx_in,
                                                            long long n, int shift, const unsigned *__restrict__ hist, int ntiles,
                                                            unsigned long long *__restrict__ keys_out, unsigned *__restrict__ idx_out,
                                                            long long *__restrict__ idx_out64)
{
    __shared__ unsigned start[256];
    __shared__ unsigned wcnt[4][256];
    __shared__ unsigned scan[256];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int tile = blockIdx.x;
    {
        unsigned below = 0u, total = 0u;
        for (int t = 0; t < ntiles; ++t) {
            const unsigned v = hist[(size_t)t * 256 + tid];
            total += v;
            below += t < tile ? v : 0u;
        }
        scan[tid] = total;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {                // inclusive scan over the digits
            const unsigned v = tid >= off ? scan[tid - off] : 0u;
            __syncthreads();
            scan[tid] += v;
            __syncthreads();
        }
        start[tid] = scan[tid] - total + below;
#pragma unroll
        for (int q = 0; q < 4; ++q) wcnt[q][tid] = 0u;
    }
    __syncthreads();
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int i = 0; i < SORT_ITEMS; ++i) {
        const long long p = (long long)tile * SORT_TILE + (long long)i * 256 + tid;
        const bool valid = p < n;
        const unsigned long long key = valid ? keys_in[p] : 0ull;
        const unsigned idx = valid ? (idx_in ? idx_in[p] : (unsigned)p) : 0u;
        const unsigned dig = (unsigned)(key >> shift) & 255u;
        unsigned long long m = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (dig >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            m &= bit ? bal : ~bal;
        }
        const unsigned rank_w = (unsigned)__popcll(m & lt);
        if (valid && rank_w == 0u) wcnt[w][dig] = (unsigned)__popcll(m);
        __syncthreads();
        if (valid) {
            unsigned pos = start[dig] + rank_w;
            for (int q = 0; q < w; ++q) pos += wcnt[q][dig];
            keys_out[pos] = key;
            if (idx_out64) idx_out64[pos] = (long long)idx;
            else idx_out[pos] = idx;
        }
        __syncthreads();
        start[tid] += wcnt[0][tid] + wcnt[1][tid] + wcnt[2][tid] + wcnt[3][tid];
#pragma unroll
        for (int q = 0; q < 4; ++q) wcnt[q][tid] = 0u;
        __syncthreads();
    }
}

extern "C" size_t pccx_sort_keys_workspace_bytes(int64_t n)
{
    if (n <= 0) return 0;
    const size_t ntiles = ((size_t)n + SORT_TILE - 1) / SORT_TILE;
    return (size_t)n * 8 + 2 * (((size_t)n * 4 + 15) / 16 * 16) + ntiles * 256 * 4;
}

// keys (n) are sorted IN PLACE (ascending, stable: equal keys keep their input order); order (n) = the input position of the key at
// each sorted position, i.e. torch.sort(keys, stable=True).indices.  key_bits: significant low bits of the keys (63 for the Morton keys).
extern "C" int pccx_sort_keys_u64(int64_t *keys, int64_t n, int key_bits, int64_t *order, void *workspace, void *stream)
{
    if (n == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(keys && order && workspace && n > 0 && n < ((int64_t)1 << 32), "pccx_sort_keys_u64: bad arguments (n=%lld)", (long long)n);
    PCCX_CHECK_ARG(key_bits >= 1 && key_bits <= 64 && ((uintptr_t)workspace & 15) == 0, "pccx_sort_keys_u64: key_bits=%d / workspace alignment", key_bits);
    hipStream_t st = (hipStream_t)stream;
    const int ntiles = (int)((n + SORT_TILE - 1) / SORT_TILE);
    int passes = (key_bits + 7) / 8;
    if (passes & 1) ++passes;                                    // an even number of passes: the sorted keys land in `keys` again
    unsigned long long *k0 = (unsigned long long *)keys, *k1 = (unsigned long long *)workspace;
    const size_t ib = ((size_t)n * 4 + 15) / 16 * 16;
    unsigned *i0 = (unsigned *)((char *)workspace + (size_t)n * 8), *i1 = (unsigned *)((char *)i0 + ib);
    unsigned *hist = (unsigned *)((char *)i1 + ib);
    for (int p = 0; p < passes; ++p) {
        const int shift = 8 * p;
        const unsigned long long *kin = (p & 1) ? k1 : k0;
        unsigned long long *kout = (p & 1) ? k0 : k1;
        const unsigned *iin = p == 0 ? nullptr : ((p & 1) ? i0 : i1);
        unsigned *iout = (p & 1) ? i1 : i0;
        hipLaunchKernelGGL(radix_hist_kernel, dim3(ntiles), dim3(256), 0, st, kin, (long long)n, shift < 64 ? shift : 63, hist);
        hipLaunchKernelGGL(radix_scatter_kernel, dim3(ntiles), dim3(256), 0, st, kin, iin, (long long)n, shift < 64 ? shift : 63, (const unsigned *)hist, ntiles, kout,
                           iout, p == passes - 1 ? (long long *)order : (long long *)nullptr);
    }
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// blocks[b][i] = pc[order[min((first + b * stride) * block + i, n - 1)]]   for b < count
__global__ __launch_bounds__(256) void gather_blocks_kernel(const float *__restrict__ pc, const long long *__restrict__ order, long long n, int block,
                                                            long long first, long long stride, long long rows, float *__restrict__ out)
{
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (long long)gridDim.x * blockDim.x) {
        const long long b = r / block, i = r - b * block;
        long long pos = (first + b * stride) * block + i;
        pos = pos < n ? pos : n - 1;
        const long long src = order[pos];
        out[3 * r] = pc[3 * src];
        out[3 * r + 1] = pc[3 * src + 1];
        out[3 * r + 2] = pc[3 * src + 2];
    }
}

// out[order[(first + b * stride) * block + i]] = rows[b][i] wherever that position is < n (the padding rows of the last block are dropped)
__global__ __launch_bounds__(256) void scatter_blocks_kernel(const float *__restrict__ rowsv, const long long *__restrict__ order, long long n, int block,
                                                             long long first, long long stride, long long rows, float *__restrict__ out)
{
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (long long)gridDim.x * blockDim.x) {
        const long long b = r / block, i = r - b * block;
        const long long pos = (first + b * stride) * block + i;
        if (pos < n) {
            const long long dst = order[pos];
            out[3 * dst] = rowsv[3 * r];
            out[3 * dst + 1] = rowsv[3 * r + 1];
            out[3 * dst + 2] = rowsv[3 * r + 2];
        }
    }
}

static int blocks_args_ok(const void *a, const void *b, const void *c, int64_t n, int block, int64_t first, int64_t stride, int64_t count, const char *who)
{
    PCCX_CHECK_ARG(a && b && c && n > 0 && block > 0 && first >= 0 && stride >= 1 && count > 0, "%s: bad arguments", who);
    PCCX_CHECK_ARG((first + (count - 1) * stride) * (int64_t)block < n, "%s: block %lld starts beyond the cloud (n=%lld, block=%d)", who,
                   (long long)(first + (count - 1) * stride), (long long)n, block);
    return PCCX_OK;
}

extern "C" int pccx_gather_blocks(const float *pc, const int64_t *order, int64_t n, int block, int64_t first, int64_t stride, int64_t count,
                                  float *blocks_out, void *stream)
{
    if (count == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    const int rc = blocks_args_ok(pc, order, blocks_out, n, block, first, stride, count, "pccx_gather_blocks");
    if (rc != PCCX_OK) return rc;
    const long long rows = (long long)count * block;
    long long grid = (rows + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(gather_blocks_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, pc, (const long long *)order, (long long)n, block,
                       (long long)first, (long long)stride, rows, blocks_out);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

extern "C" int pccx_scatter_blocks(const float *rows_in, const int64_t *order, int64_t n, int block, int64_t first, int64_t stride, int64_t count,
                                   float *pc_out, void *stream)
{
    if (count == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    const int rc = blocks_args_ok(rows_in, order, pc_out, n, block, first, stride, count, "pccx_scatter_blocks");
    if (rc != PCCX_OK) return rc;
    const long long rows = (long long)count * block;
    long long grid = (rows + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(scatter_blocks_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, rows_in, (const long long *)order, (long long)n, block,
                       (long long)first, (long long)stride, rows, pc_out);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}
