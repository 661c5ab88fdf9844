// geometry.hip -- normalize / denormalize / gather / farthest-point sampling for gfx950.
//
// All four are latency- or HBM-bound integer/float32 selection work; nothing here is
// GEMM-shaped.  One workgroup owns one cloud: a 8192x3 fp32 cloud is 96 KiB, so it lives in
// LDS (160 KiB/CU) and in registers for the whole of FPS; throughput comes from running one
// cloud per CU across the 256 CUs, not from splitting a cloud.
#include <math.h>

#include "common.h"

// ------------------------------------------------------------------------------------------
// normalize (pn_kit.py:47-60) / denormalize (pn_kit.py:62-66)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_min(float v)
{
#pragma unroll
    for (int o = 32; o; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

__global__ __launch_bounds__(1024) void normalize_kernel(const float *__restrict__ pc, int N, float one_minus_margin,
                                                         float *__restrict__ out, float *__restrict__ center,
                                                         float *__restrict__ longest)
{
    __shared__ float red[6][16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float *p = pc + (size_t)b * N * 3;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = tid; i < N; i += 1024) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float v = p[3 * i + a];
            mn[a] = fminf(mn[a], v);
            mx[a] = fmaxf(mx[a], v);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float lo = wave_min(mn[a]), hi = wave_max(mx[a]);
        if (lane == 0) { red[a][w] = lo; red[3 + a][w] = hi; }
    }
    __syncthreads();
    float c[3], lg = -INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float lo = red[a][0], hi = red[3 + a][0];
        for (int k = 1; k < 16; ++k) { lo = fminf(lo, red[a][k]); hi = fmaxf(hi, red[3 + a][k]); }
        c[a] = __fdiv_rn(__fadd_rn(hi, lo), 2.0f);   // (max+min)/2          (:53)
        lg = fmaxf(lg, __fsub_rn(hi, lo));           // max(range)          (:54)
    }
    if (tid < 3) center[3 * b + tid] = c[tid];
    if (tid == 0) longest[b] = lg;
    float *o = out + (size_t)b * N * 3;
    for (int i = tid; i < 3 * N; i += 1024) {
        const int a = i % 3;
        const float ca = a == 0 ? c[0] : (a == 1 ? c[1] : c[2]);
        float v = __fsub_rn(p[i], ca);                             // pc - center      (:56)
        v = __fdiv_rn(__fmul_rn(v, one_minus_margin), lg);         // *(1-m)/longest   (:57)
        o[i] = __fadd_rn(v, 0.5f);                                 // + 0.5            (:58)
    }
}

__global__ void denormalize_kernel(const float *__restrict__ pc, int N, float one_minus_margin,
                                   const float *__restrict__ center, const float *__restrict__ longest,
                                   float *__restrict__ out)
{
    const int b = blockIdx.y;
    const float lg = longest[b];
    const float *p = pc + (size_t)b * N * 3;
    float *o = out + (size_t)b * N * 3;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 3 * N; i += gridDim.x * blockDim.x) {
        float v = __fsub_rn(p[i], 0.5f);
        v = __fdiv_rn(__fmul_rn(v, lg), one_minus_margin);
        o[i] = __fadd_rn(v, center[3 * b + i % 3]);
    }
}

extern "C" int pccx_normalize(const float *pc, int B, int N, double margin, float *out, float *center, float *longest,
                              void *stream)
{
    if (B == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(pc && out && center && longest, "pccx_normalize: null pointer");
    PCCX_CHECK_ARG(B >= 0 && N >= 1, "pccx_normalize: bad shape B=%d N=%d", B, N);
    hipLaunchKernelGGL(normalize_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, pc, N, (float)(1.0 - margin), out,
                       center, longest);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

extern "C" int pccx_denormalize(const float *pc, int B, int N, double margin, const float *center, const float *longest,
                                float *out, void *stream)
{
    if (B == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(pc && out && center && longest, "pccx_denormalize: null pointer");
    PCCX_CHECK_ARG(B >= 0 && N >= 1, "pccx_denormalize: bad shape B=%d N=%d", B, N);
    int gx = (3 * N + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(denormalize_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, pc, N, (float)(1.0 - margin),
                       center, longest, out);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// ------------------------------------------------------------------------------------------
// gather (pn_kit.index_points, pn_kit.py:332-360; pytorch3d knn_gather)
// ------------------------------------------------------------------------------------------
__global__ void gather_kernel(const float *__restrict__ points, int N, int C, const int64_t *__restrict__ idx, int M,
                              float *__restrict__ out)
{
    const int b = blockIdx.y;
    const size_t total = (size_t)M * C;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        int m = (int)(e / C), c = (int)(e % C);
        int64_t i = idx[(size_t)b * M + m];
        if (i < 0) i = 0;            // clamp(min=0) of pointnet_sa_module.py:27
        if (i >= N) i = N - 1;       // never fault on a bad index
        out[(size_t)b * total + e] = points[((size_t)b * N + (size_t)i) * C + c];
    }
}

extern "C" int pccx_gather(const float *points, int B, int N, int C, const int64_t *idx, int M, float *out, void *stream)
{
    if (B == 0 || M == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(points && idx && out, "pccx_gather: null pointer");
    PCCX_CHECK_ARG(B >= 0 && N >= 1 && C >= 1 && M >= 0, "pccx_gather: bad shape");
    size_t total = (size_t)M * C;
    int gx = (int)((total + 255) / 256);
    if (gx > 2048) gx = 2048;
    hipLaunchKernelGGL(gather_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, points, N, C, idx, M, out);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// ------------------------------------------------------------------------------------------
// farthest point sampling (pn_kit.py:309-330)
//
// One 1024-thread workgroup per cloud.  Thread t owns points t, t+1024, ...: coordinates and the
// running min-distance stay in registers for all npoint rounds; a copy of the cloud sits in LDS
// so the next centroid is one LDS read.  A round is: PPT distance evaluations per thread ->
// thread-local argmax (lowest index on ties) -> wave64 butterfly max-reduce on (value,index) ->
// 16 wave partials through LDS (double-buffered, so ONE barrier per round) -> every wave reduces
// the 16 partials redundantly.  Selection is bit-identical to the reference: distance is
// (dx*dx+dy*dy)+dz*dz in fp32, update on strict '<', argmax returns the first maximum.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void argmax_merge(float &v, int &i, float ov, int oi)
{
    // selects, not a branch: as an `if` this compiled to an exec-mask region with a scalar branch per merge -- 12 s_and_saveexec and six
    // s_cbranch in every round of FPS, on the round's serial path
    const bool take = (ov > v) | ((ov == v) & (oi < i));
    v = take ? ov : v;
    i = take ? oi : i;
}

// Wave-wide argmax of (value, index) pairs with torch.max's tie rule (largest value, then smallest index), without LDS traffic:
// four DPP butterfly steps make every lane of a 16-lane row hold its row's winner (row_mirror, row_half_mirror, two quad
// permutes: a few cycles each, where ds_bpermute -- what __shfl_xor compiles to -- costs an LDS round trip per step and made a
// round of FPS ~2 us), then the four row winners are read as scalars and merged.  Returns the winner in every lane.
__device__ __forceinline__ void row16_argmax(float &v, int &i)
{
#define PCCX_DPP_STEP(CTRL)                                                                              \
    {                                                                                                    \
        const float ov = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true)); \
        const int oi = __builtin_amdgcn_update_dpp(0, i, CTRL, 0xf, 0xf, true);                          \
        argmax_merge(v, i, ov, oi);                                                                      \
    }
    PCCX_DPP_STEP(0x140)   // row_mirror
    PCCX_DPP_STEP(0x141)   // row_half_mirror
    PCCX_DPP_STEP(0x1B)    // quad_perm [3,2,1,0]
    PCCX_DPP_STEP(0xB1)    // quad_perm [1,0,3,2]
#undef PCCX_DPP_STEP
}

// The same reductions on ONE 64-bit key per candidate: (distance bits << 32) | (0x7fffffff - index).  Distances are non-negative floats
// (their bit patterns order like the values; a padding lane's -inf becomes 0), so the larger key is the larger distance and, among equal
// distances, the SMALLER index: torch.max's rule with one 64-bit compare and two selects per merge instead of three compares, two
// scalar mask operations and two selects -- these merges are the serial path of an FPS round.
__device__ __forceinline__ unsigned long long argmax_key(float v, int i)
{
    const unsigned hi = v < 0.f ? 0u : __float_as_uint(v);
    return ((unsigned long long)hi << 32) | (unsigned)(0x7fffffff - i);
}
__device__ __forceinline__ int argmax_key_index(unsigned long long k) { return 0x7fffffff - (int)(unsigned)(k & 0xffffffffu); }

__device__ __forceinline__ unsigned long long row16_argmax_key(unsigned long long k)
{
#define PCCX_DPP_STEP64(CTRL)                                                                                          \
    {                                                                                                                  \
        const unsigned olo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)k, CTRL, 0xf, 0xf, true);         \
        const unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(k >> 32), CTRL, 0xf, 0xf, true); \
        const unsigned long long ok = ((unsigned long long)ohi << 32) | olo;                                           \
        k = ok > k ? ok : k;                                                                                           \
    }
    PCCX_DPP_STEP64(0x140)   // row_mirror
    PCCX_DPP_STEP64(0x141)   // row_half_mirror
    PCCX_DPP_STEP64(0x1B)    // quad_perm [3,2,1,0]
    PCCX_DPP_STEP64(0xB1)    // quad_perm [1,0,3,2]
#undef PCCX_DPP_STEP64
    return k;
}

__device__ __forceinline__ unsigned long long wave_argmax_key(unsigned long long k)
{
    k = row16_argmax_key(k);
    unsigned long long best = 0ull;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)k, 16 * r);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(k >> 32), 16 * r);
        const unsigned long long rk = ((unsigned long long)hi << 32) | lo;
        best = rk > best ? rk : best;
    }
    return best;
}

__device__ __forceinline__ void wave_argmax(float &v, int &i)
{
    row16_argmax(v, i);
    float rv[4];
    int ri[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        rv[r] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16 * r));
        ri[r] = __builtin_amdgcn_readlane(i, 16 * r);
    }
    v = rv[0]; i = ri[0];
#pragma unroll
    for (int r = 1; r < 4; ++r) argmax_merge(v, i, rv[r], ri[r]);
}

// The winner's COORDINATES travel with its key.  Rounds 2-4 kept a copy of the cloud in LDS (96 KB for 8192 points) only to read the
// coordinates of the point a round selects, which held the kernel to one workgroup per CU.  Here every lane carries the coordinates of
// its own best point through the lane's scan (three more selects per point), the lane that owns the wave's winner parks them beside the
// wave's key, and after the barrier the winning wave's slot is read -- the same depth of dependent LDS reads as before (key, then
// coordinates), 768 B of LDS per workgroup, so TWO 1024-thread workgroups share a CU (57 registers: 8 waves per SIMD) and one cloud's
// barrier waits run under the other's arithmetic.  Measured (1024 clouds of 8192 points, 64 samples): 0.377 -> 0.361 ms -- a round turns
// out to be bound by the EXECUTION of its vector instructions (16 waves x ~100 instructions x 4 cycles on 4 SIMDs ~ 0.7 us of a 1.47 us
// round), not by its latencies, so the second workgroup buys only the barrier waits.  Indices: the same arithmetic per point, the same
// first-maximum rule (bit-identical).  The second template parameter is unused (kept for the instantiation names in the profiles).
template <int PPT, bool USE_LDS>
__global__ __launch_bounds__(1024) void fps_kernel(const float *__restrict__ xyz, int N, int npoint,
                                                   const int32_t *__restrict__ start, int64_t *__restrict__ out)
{
    __shared__ unsigned long long part_k[2][16];              // wave winners as 64-bit keys (argmax_key)
    __shared__ float part_c[2][16][4];                        // ... and their coordinates
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float *p = xyz + (size_t)b * N * 3;

    // Points live in registers in PAIRS (point tid + 2j*1024 and tid + (2j+1)*1024): the three differences, squares and the two adds
    // of a round run as packed fp32 (v_pk_add_f32 / v_pk_mul_f32, no contraction: -ffp-contract=off), i.e. the operation sequence
    // of pccx_sqdist per point at half the instruction count.
    constexpr int PP = (PPT + 1) / 2;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 px[PP], py[PP], pz[PP], md[PP];
#pragma unroll
    for (int j = 0; j < PP; ++j)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int i = tid + (2 * j + e) * 1024;
            if (2 * j + e < PPT && i < N) {
                px[j][e] = p[3 * i]; py[j][e] = p[3 * i + 1]; pz[j][e] = p[3 * i + 2];
                md[j][e] = 1e10f;              // distance = ones * 1e10 (:320)
            } else {
                px[j][e] = py[j][e] = pz[j][e] = 0.f;
                md[j][e] = -INFINITY;          // never selected
            }
        }

    int far = start ? start[b] : 0;
    if (far < 0 || far >= N) far = 0;
    float cx = p[3 * far], cy = p[3 * far + 1], cz = p[3 * far + 2];     // the first centroid's coordinates: one read from memory
    int par = 0;
    for (int s = 0; s < npoint; ++s) {
        if (tid == 0) out[(size_t)b * npoint + s] = far;             // centroids[:, i] = farthest (:324)
        float best = -INFINITY, bx = 0.f, by = 0.f, bz = 0.f;
        int bi = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < PP; ++j) {
            const f32x2 dx = px[j] - cx, dy = py[j] - cy, dz = pz[j] - cz;   // (:326): (x - c)^2 summed x, y, z
            f32x2 d = dx * dx;
            d = d + dy * dy;
            d = d + dz * dz;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                md[j][e] = fminf(md[j][e], d[e]);                             // (:327-328) distance[mask] = dist[mask]: one v_min_f32
                const bool up = md[j][e] > best;                              // ascending index, strict '>'
                best = up ? md[j][e] : best;
                bi = up ? tid + (2 * j + e) * 1024 : bi;
                bx = up ? px[j][e] : bx;
                by = up ? py[j][e] : by;
                bz = up ? pz[j][e] : bz;
            }
        }
        const unsigned long long mk = argmax_key(best, bi);
        const unsigned long long wk = wave_argmax_key(mk);
        if (mk == wk) {                        // the owner of the wave's winner (indices are distinct; lanes with no point tie on -inf and write zeros)
            part_k[par][w] = wk;
            part_c[par][w][0] = bx; part_c[par][w][1] = by; part_c[par][w][2] = bz;
        }
        __syncthreads();
        const unsigned long long k16 = part_k[par][lane & 15];       // the 16 wave winners, one per lane of every row
        const unsigned long long vk = row16_argmax_key(k16);
        far = __builtin_amdgcn_readfirstlane(argmax_key_index(vk));   // torch.max(distance,-1)[1] (:329)
        const int ww = __builtin_amdgcn_readfirstlane(__ffsll((long long)__ballot(k16 == vk)) - 1) & 15;
        cx = part_c[par][ww][0]; cy = part_c[par][ww][1]; cz = part_c[par][ww][2];
        par ^= 1;
    }
}

// Small clouds (N <= 512: the patches of the PointNet++ families, pointnet_sa_module.py:66-68): one WAVE per cloud, up to 8 points
// per lane in registers, no workgroup barrier in the round -- argmax by DPP, the winner's coordinates from the wave's LDS copy.
// Same arithmetic and the same first-maximum rule as fps_kernel (point i = lane + 64 j: ascending j in the lane, lower index on ties
// across lanes), so the indices are identical.
template <int PPL>
__global__ __launch_bounds__(256) void fps_wave_kernel(const float *__restrict__ xyz, int B, int N, int npoint,
                                                       const int32_t *__restrict__ start, int64_t *__restrict__ out)
{
    __shared__ float sxyz[4][3 * 64 * PPL];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + w;
    if (b >= B) return;                                     // whole wave; no barriers below
    const float *p = xyz + (size_t)b * N * 3;
    float *sx = sxyz[w];
    for (int i = lane; i < 3 * N; i += 64) sx[i] = p[i];
    constexpr int PP = PPL / 2;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 px[PP], py[PP], pz[PP], md[PP];
#pragma unroll
    for (int j = 0; j < PP; ++j)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int i = lane + (2 * j + e) * 64;
            if (i < N) {
                px[j][e] = p[3 * i]; py[j][e] = p[3 * i + 1]; pz[j][e] = p[3 * i + 2];
                md[j][e] = 1e10f;
            } else {
                px[j][e] = py[j][e] = pz[j][e] = 0.f;
                md[j][e] = -INFINITY;
            }
        }
    int far = start ? start[b] : 0;
    if (far < 0 || far >= N) far = 0;
    far = __builtin_amdgcn_readfirstlane(far);
    __builtin_amdgcn_s_waitcnt(0);                           // the wave's own LDS stores (one wave: no barrier needed)
    for (int s = 0; s < npoint; ++s) {
        if (lane == 0) out[(size_t)b * npoint + s] = far;
        const float cx = sx[3 * far], cy = sx[3 * far + 1], cz = sx[3 * far + 2];
        float best = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < PP; ++j) {
            const f32x2 dx = px[j] - cx, dy = py[j] - cy, dz = pz[j] - cz;
            f32x2 d = dx * dx;
            d = d + dy * dy;
            d = d + dz * dz;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                md[j][e] = fminf(md[j][e], d[e]);
                const bool up = md[j][e] > best;                     // ascending index, strict '>'
                best = up ? md[j][e] : best;
                bi = up ? lane + (2 * j + e) * 64 : bi;
            }
        }
        far = __builtin_amdgcn_readfirstlane(argmax_key_index(wave_argmax_key(argmax_key(best, bi))));
    }
}

template <int PPL>
static int launch_fps_wave(const float *xyz, int B, int N, int npoint, const int32_t *start, int64_t *out, hipStream_t st)
{
    hipLaunchKernelGGL(fps_wave_kernel<PPL>, dim3((B + 3) / 4), dim3(256), 0, st, xyz, B, N, npoint, start, out);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// N > 16384: running min-distance lives in a global workspace (L2-resident), same selection rule.
__global__ __launch_bounds__(1024) void fps_big_kernel(const float *__restrict__ xyz, int N, int npoint,
                                                       const int32_t *__restrict__ start, int64_t *__restrict__ out,
                                                       float *__restrict__ work)
{
    __shared__ float part_v[2][16];
    __shared__ int part_i[2][16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float *p = xyz + (size_t)b * N * 3;
    float *md = work + (size_t)b * N;
    for (int i = tid; i < N; i += 1024) md[i] = 1e10f;
    int far = start ? start[b] : 0;
    if (far < 0 || far >= N) far = 0;
    int par = 0;
    for (int s = 0; s < npoint; ++s) {
        if (tid == 0) out[(size_t)b * npoint + s] = far;
        const float cx = p[3 * far], cy = p[3 * far + 1], cz = p[3 * far + 2];
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int i = tid; i < N; i += 1024) {
            float d = pccx_sqdist(p[3 * i], p[3 * i + 1], p[3 * i + 2], cx, cy, cz);
            float m = md[i];
            if (d < m) { m = d; md[i] = d; }
            if (m > best) { best = m; bi = i; }
        }
        wave_argmax(best, bi);
        if (lane == 0) { part_v[par][w] = best; part_i[par][w] = bi; }
        __syncthreads();
        float v = part_v[par][lane & 15];
        int vi = part_i[par][lane & 15];
        row16_argmax(v, vi);
        far = __builtin_amdgcn_readfirstlane(vi);
        par ^= 1;
    }
}

template <int PPT>
static int launch_fps(const float *xyz, int B, int N, int npoint, const int32_t *start, int64_t *out, hipStream_t st)
{
    hipLaunchKernelGGL((fps_kernel<PPT, true>), dim3(B), dim3(1024), 0, st, xyz, N, npoint, start, out);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

extern "C" int pccx_fps(const float *xyz, int B, int N, int npoint, const int32_t *start_idx, int64_t *idx_out,
                        float *workspace, void *stream)
{
    if (B == 0 || npoint == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(xyz && idx_out, "pccx_fps: null pointer");
    PCCX_CHECK_ARG(B >= 0 && N >= 1 && npoint >= 0, "pccx_fps: bad shape B=%d N=%d npoint=%d", B, N, npoint);
    hipStream_t st = (hipStream_t)stream;
    if (N <= 128) return launch_fps_wave<2>(xyz, B, N, npoint, start_idx, idx_out, st);
    if (N <= 256) return launch_fps_wave<4>(xyz, B, N, npoint, start_idx, idx_out, st);
    if (N <= 512) return launch_fps_wave<8>(xyz, B, N, npoint, start_idx, idx_out, st);
    if (N <= 1024) return launch_fps<1>(xyz, B, N, npoint, start_idx, idx_out, st);
    if (N <= 2048) return launch_fps<2>(xyz, B, N, npoint, start_idx, idx_out, st);
    if (N <= 4096) return launch_fps<4>(xyz, B, N, npoint, start_idx, idx_out, st);
    if (N <= 8192) return launch_fps<8>(xyz, B, N, npoint, start_idx, idx_out, st);
    if (N <= 16384) return launch_fps<16>(xyz, B, N, npoint, start_idx, idx_out, st);
    PCCX_CHECK_ARG(workspace, "pccx_fps: N=%d > 16384 needs a workspace of B*N floats", N);
    hipLaunchKernelGGL(fps_big_kernel, dim3(B), dim3(1024), 0, st, xyz, N, npoint, start_idx, idx_out, workspace);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

// ------------------------------------------------------------------------------------------
// Morton keys for block-partitioning large clouds (BASELINE configs[3]: S3DIS rooms chunked into
// 8192-point blocks so that S = N*ALPHA/K stays 64).  21 bits per axis over the cloud's bounding box.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long spread3_21(unsigned v)
{
    unsigned long long x = v & 0x1fffffull;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__global__ void morton_keys_kernel(const float *__restrict__ xyz, long long n, float lox, float loy, float loz, float inv,
                                   int64_t *__restrict__ keys)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float fx = fminf(fmaxf((xyz[3 * i] - lox) * inv, 0.f), 2097151.f);
        const float fy = fminf(fmaxf((xyz[3 * i + 1] - loy) * inv, 0.f), 2097151.f);
        const float fz = fminf(fmaxf((xyz[3 * i + 2] - loz) * inv, 0.f), 2097151.f);
        keys[i] = (int64_t)((spread3_21((unsigned)fx) << 2) | (spread3_21((unsigned)fy) << 1) | spread3_21((unsigned)fz));
    }
}

// The same keys with the bounding box found on the device (no host round trip): ordered-int atomics into bbox[0..5]
// (min x, y, z, max x, y, z), then lo = min, extent = max over the axes of (max - min) in fp32, exactly what the host path passes.
__device__ __forceinline__ int pccx_ordered_int(float f)
{
    const int b = __float_as_int(f);
    return b >= 0 ? b : b ^ 0x7fffffff;
}
__device__ __forceinline__ float pccx_ordered_float(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7fffffff); }

__global__ void bbox_init_kernel(int *__restrict__ bbox)
{
    if (threadIdx.x < 3) bbox[threadIdx.x] = 0x7fffffff;
    else if (threadIdx.x < 6) bbox[threadIdx.x] = (int)0x80000000;
}

__global__ __launch_bounds__(256) void bbox_kernel(const float *__restrict__ xyz, long long n, int *__restrict__ bbox)
{
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = xyz[3 * i + a];
            lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v);
        }
    // wave reduction, then the four waves through LDS: ONE atomic per component per workgroup.  (One per WAVE from 2048 workgroups was
    // 49 152 atomics on six addresses -- same-address atomics serialise at L2 -- and made this kernel 0.56 ms per room, 9 % of the
    // room-scale step; with at most 512 workgroups it is 3072.)
    __shared__ float red[4][6];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], off));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off));
        }
        if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][a] = lo[a]; red[threadIdx.x >> 6][3 + a] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        atomicMin(bbox + a, pccx_ordered_int(fminf(fminf(red[0][a], red[1][a]), fminf(red[2][a], red[3][a]))));
        atomicMax(bbox + 3 + a, pccx_ordered_int(fmaxf(fmaxf(red[0][3 + a], red[1][3 + a]), fmaxf(red[2][3 + a], red[3][3 + a]))));
    }
}

__global__ void morton_keys_auto_kernel(const float *__restrict__ xyz, long long n, const int *__restrict__ bbox,
                                        int64_t *__restrict__ keys)
{
    const float lox = pccx_ordered_float(bbox[0]), loy = pccx_ordered_float(bbox[1]), loz = pccx_ordered_float(bbox[2]);
    float ext = fmaxf(fmaxf(__fsub_rn(pccx_ordered_float(bbox[3]), lox), __fsub_rn(pccx_ordered_float(bbox[4]), loy)),
                      __fsub_rn(pccx_ordered_float(bbox[5]), loz));
    ext = fmaxf(ext, 1e-30f);
    const float inv = __fdiv_rn(2097151.f, ext);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float fx = fminf(fmaxf((xyz[3 * i] - lox) * inv, 0.f), 2097151.f);
        const float fy = fminf(fmaxf((xyz[3 * i + 1] - loy) * inv, 0.f), 2097151.f);
        const float fz = fminf(fmaxf((xyz[3 * i + 2] - loz) * inv, 0.f), 2097151.f);
        keys[i] = (int64_t)((spread3_21((unsigned)fx) << 2) | (spread3_21((unsigned)fy) << 1) | spread3_21((unsigned)fz));
    }
}

extern "C" int pccx_morton_keys_auto(const float *xyz, int64_t n, int64_t *keys, int32_t *bbox_workspace, void *stream)
{
    if (n == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(xyz && keys && bbox_workspace && n > 0, "pccx_morton_keys_auto: bad arguments");
    long long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bbox_init_kernel, dim3(1), dim3(64), 0, st, bbox_workspace);
    hipLaunchKernelGGL(bbox_kernel, dim3((unsigned)(blocks < 512 ? blocks : 512)), dim3(256), 0, st, xyz, (long long)n, bbox_workspace);
    hipLaunchKernelGGL(morton_keys_auto_kernel, dim3((unsigned)blocks), dim3(256), 0, st, xyz, (long long)n, bbox_workspace, keys);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}

extern "C" int pccx_morton_keys(const float *xyz, int64_t n, const float *lo_host, float extent, int64_t *keys, void *stream)
{
    if (n == 0) return PCCX_OK;   // empty batch: nothing to do, pointers may be null
    PCCX_CHECK_ARG(xyz && lo_host && keys && extent > 0.f, "pccx_morton_keys: bad arguments");
    long long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(morton_keys_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, xyz, (long long)n,
                       lo_host[0], lo_host[1], lo_host[2], 2097151.f / extent, keys);
    PCCX_CHECK_LAUNCH();
    return PCCX_OK;
}
