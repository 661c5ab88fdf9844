/*
 * oracle/pcc_oracle.c -- CPU restatement of the reference's integer / selection
 * path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the checker, never the product: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The shipped path
 * (point-cloud-compression_amd/) never links or calls anything in here and fails
 * loudly when its HIP library is missing.
 *
 * Every function cites the reference lines (relative to /root/reference) whose
 * algorithm it restates.  The restatements deliberately keep the reference's
 * *algorithm* (explicit DFS stack, O(nodes*S) masks, sequential depth search) so
 * that agreement with the closed-form HIP kernels is a cross-check of two
 * different derivations.
 *
 * Parity pins (see tests/test_oracle_golden.py):
 *   - octree encode / getDecodeFromPc / decode / depth search / bit packing are
 *     pinned bit-for-bit against outputs of the reference's own octree_np.py and
 *     pn_kit.py run in the build container (tests/golden/make_golden.py).
 *   - FPS is pinned on index equality against pn_kit.farthest_point_sample_batch.
 *   - kNN / ball query / Chamfer / range coder restate third-party semantics
 *     (pytorch3d, torchac -- absent from the image): PARITY UNPINNED for tie
 *     order and byte layout; definitions are stated at each function.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (oracle/Makefile).  FP contraction
 * is off so that a*a+b*b is two roundings, as in the reference's torch/numpy ops.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* numpy float32 floor_divide (npy_divmodf), used by octree_np.py:130          */
/* ------------------------------------------------------------------------- */
static float np_floor_divide_f32(float a, float b)
{
    float mod, div, floordiv;
    if (b == 0.0f) return a / b;
    mod = fmodf(a, b);
    div = (a - mod) / b;
    if (mod != 0.0f) {
        if ((b < 0) != (mod < 0)) { mod += b; div -= 1.0f; }
    }
    if (div != 0.0f) {
        floordiv = floorf(div);
        if (div - floordiv > 0.5f) floordiv += 1.0f;
    } else {
        floordiv = copysignf(0.0f, a / b);
    }
    return floordiv;
}

static float nan_to_num_f32(float v)
{
    if (isnan(v)) return 0.0f;
    if (isinf(v)) return v > 0 ? 3.4028234663852886e+38f : -3.4028234663852886e+38f;
    return v;
}

static int cmp_row3(const void *pa, const void *pb)
{
    const float *a = (const float *)pa, *b = (const float *)pb;
    for (int i = 0; i < 3; ++i) {
        if (a[i] < b[i]) return -1;
        if (a[i] > b[i]) return 1;
    }
    return 0;
}

/*
 * octree_np.getDecodeFromPc (octree_np.py:114-133), 2-D branch (:130-132):
 *   pc_octree = (pc // cube_reso * cube_reso) + cube_reso/2 ; nan_to_num ; np.unique(axis=0)
 * cube_reso is a Python float; under NumPy-2 promotion it is applied as float32.
 * Returns the number of unique rows written to out (sorted lexicographically, as
 * np.unique(axis=0) does).  out must hold n*3 floats.
 */
ORC_API int orc_get_decode_from_pc(const float *pc, int n, double resolution, int depth, float *out)
{
    int capped = depth < 30 ? depth : 30;
    double divisor = pow(2.0, (double)capped);
    if (divisor < 1.0) divisor = 1.0;
    double cube = resolution / divisor;
    if (cube < 1e-6) cube = 1e-6;
    float cf = (float)cube;
    float half = (float)(cube / 2);
    for (int i = 0; i < n * 3; ++i) {
        float q = np_floor_divide_f32(pc[i], cf);
        float v = q * cf;
        v = v + half;
        out[i] = nan_to_num_f32(v);
    }
    if (n == 0) return 0;
    qsort(out, (size_t)n, 3 * sizeof(float), cmp_row3);
    int u = 1;
    for (int i = 1; i < n; ++i) {
        if (cmp_row3(out + 3 * i, out + 3 * (u - 1)) != 0) {
            if (u != i) memcpy(out + 3 * u, out + 3 * i, 3 * sizeof(float));
            ++u;
        }
    }
    return u;
}

/*
 * octree_np.encode (octree_np.py:10-45).  LIFO stack DFS; per visited node one bit
 * appended to the list of its level (:28,:42); children pushed in the order
 * 000,001,010,...,111 of (x,y,z) offsets (:31-40); final stream = levels
 * concatenated (:43).  Masks are inclusive on both sides (:22-26).
 * bits must hold at least 1 + 8*n*depth bytes (one byte per bit, values 0/1).
 * Returns number of bits, or -1 if cap is too small.
 */
typedef struct { double x, y, z; int d; } orc_node;

ORC_API int orc_octree_encode(const float *pc_in, int n, double resolution, int depth,
                              uint8_t *bits, int cap)
{
    float *pc = (float *)malloc((size_t)(n > 0 ? n : 1) * 3 * sizeof(float));
    int m = orc_get_decode_from_pc(pc_in, n, resolution, depth, pc);
    int levels = depth + 1;
    size_t per = (size_t)8 * (size_t)(m > 0 ? m : 1) + 8;
    uint8_t *lv = (uint8_t *)malloc(per * (size_t)levels);
    int *cnt = (int *)calloc((size_t)levels, sizeof(int));
    size_t stack_cap = (size_t)8 * (size_t)levels + 16;
    orc_node *stack = (orc_node *)malloc(stack_cap * sizeof(orc_node));
    size_t sp = 0;
    stack[sp++] = (orc_node){0.0, 0.0, 0.0, 0};
    int overflow = 0;
    while (sp) {
        orc_node nd = stack[--sp];
        double reso = resolution / pow(2.0, (double)nd.d);
        int any = 0;
        for (int i = 0; i < m && !any; ++i) {
            double px = pc[3 * i], py = pc[3 * i + 1], pz = pc[3 * i + 2];
            if (nd.x <= px && px <= nd.x + reso && nd.y <= py && py <= nd.y + reso &&
                nd.z <= pz && pz <= nd.z + reso)
                any = 1;
        }
        if ((size_t)cnt[nd.d] >= per) { overflow = 1; break; }
        lv[(size_t)nd.d * per + (size_t)cnt[nd.d]++] = (uint8_t)any;
        if (any && nd.d < depth) {
            double h = reso / 2;
            if (sp + 8 > stack_cap) {
                stack_cap *= 2;
                stack = (orc_node *)realloc(stack, stack_cap * sizeof(orc_node));
            }
            stack[sp++] = (orc_node){nd.x, nd.y, nd.z, nd.d + 1};
            stack[sp++] = (orc_node){nd.x, nd.y, nd.z + h, nd.d + 1};
            stack[sp++] = (orc_node){nd.x, nd.y + h, nd.z, nd.d + 1};
            stack[sp++] = (orc_node){nd.x, nd.y + h, nd.z + h, nd.d + 1};
            stack[sp++] = (orc_node){nd.x + h, nd.y, nd.z, nd.d + 1};
            stack[sp++] = (orc_node){nd.x + h, nd.y, nd.z + h, nd.d + 1};
            stack[sp++] = (orc_node){nd.x + h, nd.y + h, nd.z, nd.d + 1};
            stack[sp++] = (orc_node){nd.x + h, nd.y + h, nd.z + h, nd.d + 1};
        }
    }
    int total = 0;
    if (!overflow) {
        for (int l = 0; l < levels; ++l) total += cnt[l];
        if (total > cap) overflow = 1;
    }
    if (!overflow) {
        int o = 0;
        for (int l = 0; l < levels; ++l) {
            memcpy(bits + o, lv + (size_t)l * per, (size_t)cnt[l]);
            o += cnt[l];
        }
    }
    free(stack); free(cnt); free(lv); free(pc);
    return overflow ? -1 : total;
}

/*
 * pn_kit.encode_sampled_np (pn_kit.py:380-401), one cloud (the body of the
 * `for i, pc` loop): DEPTH = 1; up to 16 attempts; accept when
 * len(code)/N > min_bpp and getDecodeFromPc(pc).shape == pc.shape; otherwise the
 * LAST attempt's code is kept (DEPTH ends at 17 but code is depth 16's).
 * Returns nbits; *depth_out = DEPTH as the reference leaves it.
 */
ORC_API int orc_encode_sampled(const float *pc, int S, double scale, int N, double min_bpp,
                               uint8_t *bits, int cap, int *depth_out)
{
    int DEPTH = 1, nbits = 0;
    float *tmp = (float *)malloc((size_t)(S > 0 ? S : 1) * 3 * sizeof(float));
    for (int attempt = 0; attempt < 16; ++attempt) {
        nbits = orc_octree_encode(pc, S, scale, DEPTH, bits, cap);
        if (nbits < 0) break;
        double bpp = (double)nbits / (double)N;
        int u = orc_get_decode_from_pc(pc, S, scale, DEPTH, tmp);
        if (bpp > min_bpp && u == S) break;
        DEPTH += 1;
    }
    free(tmp);
    if (depth_out) *depth_out = DEPTH;
    return nbits;
}

/*
 * octree_np.decode AS WRITTEN (octree_np.py:47-112), bug-compatible:
 *   - the parse loop overwrites `bits` with the first group (:61), so exactly one
 *     group of <= 8 bits is consumed and depth == 1;
 *   - the DFS reads that group for the 8 level-1 children popped 111..000 (:74-96);
 *   - output is padded to S=64 with the last point, or zeros when empty (:100-107).
 * out holds 64*3 floats.  Returns the number of distinct decoded points (<= 8).
 */
ORC_API int orc_octree_decode_reference(const uint8_t *bits_in, int nbits, double resolution, float *out)
{
    /* parse loop (:50-69), restated literally */
    uint8_t grp[8];
    int glen = 0;
    {
        const uint8_t *bits = bits_in;
        int len = nbits, n = 8, idx = 0, ngroups = 0;
        uint8_t cur[8];
        while (idx < len) {
            int g = len - idx < n ? len - idx : n;
            if (g < 0) g = 0;
            /* bits_ls.append(bits_group); only the first group can ever be appended */
            if (g == 0) break;
            if (g > 8) g = 8; /* first n is 8; later iterations are unreachable */
            memcpy(cur, bits + idx, (size_t)g);
            if (ngroups == 0) { memcpy(grp, cur, (size_t)g); glen = g; }
            ++ngroups;
            bits = cur; len = g;            /* bits = bits_group (:61) */
            int s = 0;
            for (int i = 0; i < g; ++i) s += cur[i];
            n = s * 8;
            idx += len;                     /* idx += len(bits) (:67) */
            if (n == 0) break;
        }
        if (ngroups == 0) { glen = 0; }
        /* depth = len(bits_ls) - 1 : 1 if a group was appended, else 0 */
        if (ngroups == 0) {
            /* bits empty: bits_ls == [[1]], depth 0: root is a leaf at depth 0 */
            float c = (float)(resolution / 2);
            for (int i = 0; i < 64; ++i) { out[3*i] = c; out[3*i+1] = c; out[3*i+2] = c; }
            return 1;
        }
    }
    /* DFS (:71-96) with depth == 1 */
    float pts[8 * 3];
    int np_ = 0, ptr = 0;
    double h = resolution / 2;          /* next_cube_reso for the root */
    double creso = resolution / 2;      /* curr_cube_reso at depth 1 */
    /* children pushed 000..111 (z fastest), popped in reverse */
    for (int c = 7; c >= 0; --c) {
        if (ptr >= glen) break;         /* guard (:78-79) */
        uint8_t b = grp[ptr++];
        if (b == 1) {
            double sx = (c & 4) ? h : 0.0, sy = (c & 2) ? h : 0.0, sz = (c & 1) ? h : 0.0;
            pts[3*np_]   = (float)(sx + creso / 2);
            pts[3*np_+1] = (float)(sy + creso / 2);
            pts[3*np_+2] = (float)(sz + creso / 2);
            ++np_;
        }
    }
    if (np_ == 0) { memset(out, 0, 64 * 3 * sizeof(float)); return 0; }
    for (int i = 0; i < 64; ++i) {
        int s = i < np_ ? i : np_ - 1;
        out[3*i] = pts[3*s]; out[3*i+1] = pts[3*s+1]; out[3*i+2] = pts[3*s+2];
    }
    return np_;
}

/*
 * "full" decode: the level-by-level decode octree_np.decode evidently intends
 * (SURVEY Appendix B).  NOT in the reference; defined here as: skip the root bit,
 * level l has 8*popcount(level l-1) bits, DFS identical to encode's; emits the
 * centres of the deepest level's set bits in DFS (descending-Morton) order.
 * Returns the number of points (<= cap_pts) or -1 on a malformed stream.
 */
ORC_API int orc_octree_decode_full(const uint8_t *bits, int nbits, double resolution,
                                   float *out, int cap_pts, int *depth_out)
{
    if (nbits < 1 || bits[0] != 1) { if (depth_out) *depth_out = 0; return 0; }
    /* level offsets */
    int off[32], cnt[32], depth = 0, pos = 1, prev_pop = 1;
    off[0] = 0; cnt[0] = 1;
    while (pos < nbits && prev_pop > 0 && depth < 30) {
        int n = 8 * prev_pop;
        if (pos + n > nbits) n = nbits - pos;       /* tolerate byte-padding tail */
        if (n < 8 * prev_pop) break;
        ++depth; off[depth] = pos; cnt[depth] = n;
        int pop = 0;
        for (int i = 0; i < n; ++i) pop += bits[pos + i];
        prev_pop = pop; pos += n;
    }
    if (depth_out) *depth_out = depth;
    int ptr[32]; memset(ptr, 0, sizeof ptr);
    size_t stack_cap = (size_t)8 * (size_t)(depth + 1) + 16, sp = 0;
    orc_node *stack = (orc_node *)malloc(stack_cap * sizeof(orc_node));
    stack[sp++] = (orc_node){0, 0, 0, 0};
    int np_ = 0;
    while (sp) {
        orc_node nd = stack[--sp];
        double reso = resolution / pow(2.0, (double)nd.d);
        if (ptr[nd.d] >= cnt[nd.d]) break;
        uint8_t b = bits[off[nd.d] + ptr[nd.d]++];
        if (b != 1) continue;
        if (nd.d == depth) {
            if (np_ >= cap_pts) { free(stack); return -1; }
            out[3*np_] = (float)(nd.x + reso / 2); out[3*np_+1] = (float)(nd.y + reso / 2);
            out[3*np_+2] = (float)(nd.z + reso / 2); ++np_;
        } else {
            double h = reso / 2;
            stack[sp++] = (orc_node){nd.x, nd.y, nd.z, nd.d + 1};
            stack[sp++] = (orc_node){nd.x, nd.y, nd.z + h, nd.d + 1};
            stack[sp++] = (orc_node){nd.x, nd.y + h, nd.z, nd.d + 1};
            stack[sp++] = (orc_node){nd.x, nd.y + h, nd.z + h, nd.d + 1};
            stack[sp++] = (orc_node){nd.x + h, nd.y, nd.z, nd.d + 1};
            stack[sp++] = (orc_node){nd.x + h, nd.y, nd.z + h, nd.d + 1};
            stack[sp++] = (orc_node){nd.x + h, nd.y + h, nd.z, nd.d + 1};
            stack[sp++] = (orc_node){nd.x + h, nd.y + h, nd.z + h, nd.d + 1};
        }
    }
    free(stack);
    return np_;
}

/*
 * pn_kit.binary_array_to_byte_array (pn_kit.py:463-467): groups of 8 bits parsed
 * as a base-2 string, so a final partial group is RIGHT-aligned in its byte.
 */
ORC_API int orc_pack_bits(const uint8_t *bits, int nbits, uint8_t *bytes)
{
    int nb = 0;
    for (int i = 0; i < nbits; i += 8) {
        unsigned v = 0;
        for (int j = i; j < i + 8 && j < nbits; ++j) v = (v << 1) | (bits[j] & 1u);
        bytes[nb++] = (uint8_t)v;
    }
    return nb;
}

/* pn_kit.byte_array_to_binary_array (pn_kit.py:469-475): f'{b:08b}' per byte. */
ORC_API int orc_unpack_bits(const uint8_t *bytes, int nbytes, uint8_t *bits)
{
    for (int i = 0; i < nbytes; ++i)
        for (int j = 0; j < 8; ++j) bits[8 * i + j] = (bytes[i] >> (7 - j)) & 1u;
    return 8 * nbytes;
}

/*
 * pn_kit.farthest_point_sample_batch (pn_kit.py:309-330), one cloud, explicit
 * start index instead of torch.randint (:321).  distance starts at 1e10 (:320);
 * dist = sum((xyz-c)**2, -1) in float32 (:326); strict `<` update (:327-328);
 * torch.max(...)[1] -> first index of the maximum (:329).
 */
ORC_API void orc_fps(const float *xyz, int N, int npoint, int start, int64_t *idx_out)
{
    float *distance = (float *)malloc((size_t)N * sizeof(float));
    for (int i = 0; i < N; ++i) distance[i] = 1e10f;
    int far = start;
    for (int s = 0; s < npoint; ++s) {
        idx_out[s] = far;
        float cx = xyz[3 * far], cy = xyz[3 * far + 1], cz = xyz[3 * far + 2];
        float best = -INFINITY; int bi = 0;
        for (int i = 0; i < N; ++i) {
            float dx = xyz[3 * i] - cx, dy = xyz[3 * i + 1] - cy, dz = xyz[3 * i + 2] - cz;
            float d = dx * dx + dy * dy;
            d = d + dz * dz;
            if (d < distance[i]) distance[i] = d;
            if (distance[i] > best) { best = distance[i]; bi = i; }
        }
        far = bi;
    }
    free(distance);
}

/*
 * pytorch3d.ops.knn_points semantics (call sites compress.py:71, pn_kit.py:190):
 * squared L2, K smallest, ascending.  PARITY UNPINNED (pytorch3d absent): defined
 * here as dist = ((dx*dx)+dy*dy)+dz*dz in float32, ordered by (dist, index).
 * dists/idx are (M,K).  K <= N required.
 */
typedef struct { float d; int i; } orc_di;
static int cmp_di(const void *a, const void *b)
{
    const orc_di *x = (const orc_di *)a, *y = (const orc_di *)b;
    if (x->d < y->d) return -1;
    if (x->d > y->d) return 1;
    return (x->i > y->i) - (x->i < y->i);
}
ORC_API void orc_knn(const float *q, int M, const float *ref, int N, int K, float *dists, int64_t *idx)
{
    orc_di *buf = (orc_di *)malloc((size_t)N * sizeof(orc_di));
    for (int m = 0; m < M; ++m) {
        float qx = q[3 * m], qy = q[3 * m + 1], qz = q[3 * m + 2];
        for (int i = 0; i < N; ++i) {
            float dx = qx - ref[3 * i], dy = qy - ref[3 * i + 1], dz = qz - ref[3 * i + 2];
            float d = dx * dx;
            d = d + dy * dy;
            d = d + dz * dz;
            buf[i].d = d; buf[i].i = i;
        }
        qsort(buf, (size_t)N, sizeof(orc_di), cmp_di);
        for (int k = 0; k < K; ++k) { dists[(size_t)m * K + k] = buf[k].d; idx[(size_t)m * K + k] = buf[k].i; }
    }
    free(buf);
}

/*
 * pytorch3d.ops.ball_query semantics (call site pointnet_sa_module.py:18):
 * first K reference indices (index order) with d2 < r*r, padded with -1; dists
 * padded with 0.  PARITY UNPINNED (pytorch3d absent).
 */
ORC_API void orc_ball_query(const float *q, int M, const float *ref, int N, int K, float radius,
                            float *dists, int64_t *idx)
{
    float r2 = radius * radius;
    for (int m = 0; m < M; ++m) {
        int c = 0;
        float qx = q[3 * m], qy = q[3 * m + 1], qz = q[3 * m + 2];
        for (int i = 0; i < N && c < K; ++i) {
            float dx = qx - ref[3 * i], dy = qy - ref[3 * i + 1], dz = qz - ref[3 * i + 2];
            float d = dx * dx;
            d = d + dy * dy;
            d = d + dz * dz;
            if (d < r2) { dists[(size_t)m * K + c] = d; idx[(size_t)m * K + c] = i; ++c; }
        }
        for (; c < K; ++c) { dists[(size_t)m * K + c] = 0.0f; idx[(size_t)m * K + c] = -1; }
    }
}

/*
 * One-directional nearest-neighbour squared distances: for every x in X (P,3) the
 * min over Y (Q,3) of |x-y|^2 (float32, same summation order as orc_knn) and its
 * index.  Building block for pytorch3d chamfer_distance (AE.py:67, eval.py:204)
 * and for eval.py's D1 loop (eval.py:73-81).
 */
ORC_API void orc_nn_dist(const float *X, int P, const float *Y, int Q, float *d2, int32_t *nn)
{
    for (int p = 0; p < P; ++p) {
        float best = INFINITY; int bi = -1;
        float px = X[3 * p], py = X[3 * p + 1], pz = X[3 * p + 2];
        for (int j = 0; j < Q; ++j) {
            float dx = px - Y[3 * j], dy = py - Y[3 * j + 1], dz = pz - Y[3 * j + 2];
            float d = dx * dx;
            d = d + dy * dy;
            d = d + dz * dz;
            if (d < best) { best = d; bi = j; }
        }
        d2[p] = best; if (nn) nn[p] = bi;
    }
}

/*
 * torchac (==0.9.3, requirements_cpu.txt:14) range coder, restated from its
 * published algorithm (L3C "torchac" backend): 32-bit low/high, 16-bit CDF
 * precision, E1/E2/E3 renormalisation with pending bits, bits packed MSB-first,
 * final byte zero-padded.  Call sites compress.py:136, decompress.py:93.
 * PARITY UNPINNED for the byte layout (torchac absent from the image); the pinned
 * properties are lossless round trip and size ~ sum(-log2 p).
 *
 * cdf: (nsym, Lp) uint16-valued entries stored as int32 (entry Lp-1 is implicitly
 * 0x10000, as torchac treats the last symbol), sym: (nsym) in [0, Lp-2].
 */
typedef struct { uint8_t *buf; int cap; int n; uint8_t cache; int count; int overflow; } orc_bitw;
static void bw_bit(orc_bitw *w, int bit)
{
    w->cache = (uint8_t)((w->cache << 1) | (bit & 1));
    if (++w->count == 8) {
        if (w->n < w->cap) w->buf[w->n++] = w->cache; else w->overflow = 1;
        w->count = 0; w->cache = 0;
    }
}
static void bw_bit_pending(orc_bitw *w, int bit, uint64_t *pending)
{
    bw_bit(w, bit);
    while (*pending > 0) { bw_bit(w, !bit); --*pending; }
}
static void bw_flush(orc_bitw *w)
{
    if (w->count > 0) {
        for (int i = w->count; i < 8; ++i) w->cache = (uint8_t)(w->cache << 1);
        if (w->n < w->cap) w->buf[w->n++] = w->cache; else w->overflow = 1;
        w->count = 0; w->cache = 0;
    }
}

ORC_API int orc_range_encode(const int32_t *cdf, int nsym, int Lp, const int16_t *sym,
                             uint8_t *out, int cap)
{
    orc_bitw w = {out, cap, 0, 0, 0, 0};
    uint32_t low = 0, high = 0xFFFFFFFFu;
    uint64_t pending = 0;
    const int max_symbol = Lp - 2;
    for (int i = 0; i < nsym; ++i) {
        const int32_t *c = cdf + (size_t)i * Lp;
        int s = sym[i];
        uint64_t span = (uint64_t)high - (uint64_t)low + 1;
        uint32_t c_low = (uint32_t)(c[s] & 0xFFFF);
        uint32_t c_high = s == max_symbol ? 0x10000u : (uint32_t)(c[s + 1] & 0xFFFF);
        high = (uint32_t)((low - 1) + ((span * c_high) >> 16));
        low = (uint32_t)(low + ((span * c_low) >> 16));
        for (;;) {
            if (high < 0x80000000u) {
                bw_bit_pending(&w, 0, &pending);
                low <<= 1; high <<= 1; high |= 1;
            } else if (low >= 0x80000000u) {
                bw_bit_pending(&w, 1, &pending);
                low <<= 1; high <<= 1; high |= 1;
            } else if (low >= 0x40000000u && high < 0xC0000000u) {
                ++pending;
                low <<= 1; low &= 0x7FFFFFFFu;
                high <<= 1; high |= 0x80000001u;
            } else break;
        }
    }
    ++pending;
    if (low < 0x40000000u) bw_bit_pending(&w, 0, &pending);
    else bw_bit_pending(&w, 1, &pending);
    bw_flush(&w);
    return w.overflow ? -1 : w.n;
}

typedef struct { const uint8_t *buf; int n; int pos; uint8_t cache; int cached; } orc_bitr;
static void br_get(orc_bitr *r, uint32_t *value)
{
    if (r->cached == 0) {
        if (r->pos >= r->n) { *value <<= 1; return; }
        r->cache = r->buf[r->pos++]; r->cached = 8;
    }
    *value <<= 1;
    *value |= (uint32_t)((r->cache >> (r->cached - 1)) & 1u);
    --r->cached;
}

ORC_API void orc_range_decode(const int32_t *cdf, int nsym, int Lp, const uint8_t *in, int nbytes,
                              int16_t *sym_out)
{
    orc_bitr r = {in, nbytes, 0, 0, 0};
    uint32_t low = 0, high = 0xFFFFFFFFu, value = 0;
    const int max_symbol = Lp - 2;
    for (int i = 0; i < 32; ++i) br_get(&r, &value);
    for (int i = 0; i < nsym; ++i) {
        const int32_t *c = cdf + (size_t)i * Lp;
        uint64_t span = (uint64_t)high - (uint64_t)low + 1;
        uint16_t count = (uint16_t)((((uint64_t)value - (uint64_t)low + 1) * 0x10000u - 1) / span);
        /* binary search: largest s with cdf[s] <= count, s in [0, max_symbol] */
        int left = 0, right = max_symbol + 1;
        while (left + 1 < right) {
            int mid = (left + right) / 2;
            if ((uint16_t)(c[mid] & 0xFFFF) <= count) left = mid; else right = mid;
        }
        int s = left;
        sym_out[i] = (int16_t)s;
        uint32_t c_low = (uint32_t)(c[s] & 0xFFFF);
        uint32_t c_high = s == max_symbol ? 0x10000u : (uint32_t)(c[s + 1] & 0xFFFF);
        high = (uint32_t)((low - 1) + ((span * c_high) >> 16));
        low = (uint32_t)(low + ((span * c_low) >> 16));
        for (;;) {
            if (low >= 0x80000000u || high < 0x80000000u) {
                low <<= 1; high <<= 1; high |= 1; br_get(&r, &value);
            } else if (low >= 0x40000000u && high < 0xC0000000u) {
                low <<= 1; low &= 0x7FFFFFFFu;
                high <<= 1; high |= 0x80000001u;
                value -= 0x40000000u;
                br_get(&r, &value);
            } else break;
        }
    }
}
