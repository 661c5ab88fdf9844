/*
 * pccx.h -- C ABI of libpccx.so, the MI355X (gfx950) implementation of the
 * patch-based compress / decompress hot path of rhmes/point-cloud-compression.
 *
 * The reference has no FFI of its own: its hot path is a set of Python callables
 * (pn_kit.py, octree_np.py, AE.py) plus pytorch3d / torchac functions they import.
 * Each entry point below replaces one of those callables (cited as file:line
 * relative to the reference root).  INTEGRATION.md shows the ctypes binding a
 * maintainer adds on the reference side.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless the name ends in _host;
 *   - tensors are dense, row-major, fp32 / int32 / int64 / uint8 as declared;
 *   - `stream` is a hipStream_t (NULL = default stream); calls only enqueue work,
 *     they never synchronise;
 *   - return value: 0 on success, otherwise a negative pccx_status; the text of the
 *     last error on the calling thread is returned by pccx_last_error().
 */
#ifndef PCCX_H
#define PCCX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCCX_API __attribute__((visibility("default")))

typedef enum {
    PCCX_OK = 0,
    PCCX_ERR_ARG = -1,        /* bad shape / unsupported size */
    PCCX_ERR_HIP = -2,        /* a HIP runtime call failed */
    PCCX_ERR_UNSUPPORTED = -3
} pccx_status;

PCCX_API const char *pccx_last_error(void);
PCCX_API int pccx_version(void);

/* ---- geometry ---------------------------------------------------------------------------- */

/* pn_kit.normalize (pn_kit.py:47-60), batched: per cloud b, centre on the bounding-box
 * middle, scale the longest side to (1-margin), shift by +0.5.
 * pc,out: (B,N,3) f32 (may alias); center: (B,3); longest: (B). */
PCCX_API int pccx_normalize(const float *pc, int B, int N, double margin, float *out, float *center,
                            float *longest, void *stream);

/* pn_kit.denormalize (pn_kit.py:62-66). pc,out: (B,N,3). */
PCCX_API int pccx_denormalize(const float *pc, int B, int N, double margin, const float *center,
                              const float *longest, float *out, void *stream);

/* pn_kit.farthest_point_sample_batch (pn_kit.py:309-330) with the random start (:321) made an
 * explicit argument; also pytorch3d sample_farthest_points (pointnet_sa_module.py:12) with
 * start 0.  xyz: (B,N,3); start_idx: (B) int32 device, or NULL for start 0; idx_out: (B,npoint)
 * int64.  workspace: B*N floats, required only when N > 16384 (else may be NULL). */
PCCX_API int pccx_fps(const float *xyz, int B, int N, int npoint, const int32_t *start_idx,
                      int64_t *idx_out, float *workspace, void *stream);

/* 63-bit Morton keys over the bounding box [lo, lo+extent]^3 (lo_host: 3 floats on the HOST), used to
 * cut clouds larger than one block into spatially compact 8192-point blocks (BASELINE configs[3]).
 * xyz: (n,3) f32; keys: (n) int64. */
PCCX_API int pccx_morton_keys(const float *xyz, int64_t n, const float *lo_host, float extent, int64_t *keys,
                              void *stream);
/* The same keys over the cloud's own bounding box, found on the device (bbox_workspace: 6 int32 on the device, scratch): no
 * host round trip, so the block partition of a large cloud can be queued without a synchronisation. */
PCCX_API int pccx_morton_keys_auto(const float *xyz, int64_t n, int64_t *keys, int32_t *bbox_workspace, void *stream);

/* The block partition of a room-scale cloud (BASELINE configs[3]; the reference has no such path: octree_np.decode hard-codes S = 64,
 * octree_np.py:100, so a cloud of N points is cut into blocks of 8192).  pccx_sort_keys_u64: STABLE ascending radix sort of the keys IN
 * PLACE; order (n) int64 = the input position of the key at each sorted position (torch.sort(keys, stable=True).indices).  key_bits:
 * significant low bits (63 for the Morton keys).  workspace: pccx_sort_keys_workspace_bytes(n) bytes, 16-byte aligned; n < 2^32. */
PCCX_API size_t pccx_sort_keys_workspace_bytes(int64_t n);
PCCX_API int pccx_sort_keys_u64(int64_t *keys, int64_t n, int key_bits, int64_t *order, void *workspace, void *stream);
/* blocks_out (count, block, 3): block b = rows order[(first + b * stride) * block + i], i < block, of pc (n,3); positions >= n repeat
 * row order[n - 1] (the last block is completed with copies of its final point).  first / stride select a rank's blocks. */
PCCX_API int pccx_gather_blocks(const float *pc, const int64_t *order, int64_t n, int block, int64_t first, int64_t stride,
                                int64_t count, float *blocks_out, void *stream);
/* The inverse: pc_out[order[(first + b * stride) * block + i]] = rows_in[b][i] for every position < n (padding rows are dropped). */
PCCX_API int pccx_scatter_blocks(const float *rows_in, const int64_t *order, int64_t n, int block, int64_t first, int64_t stride,
                                 int64_t count, float *pc_out, void *stream);

/* pn_kit.index_points (pn_kit.py:332-360) / pytorch3d knn_gather (pointnet_sa_module.py:28):
 * out[b,m,:] = points[b, idx[b,m], :].  points: (B,N,C); idx: (B,M) int64 (negative -> row 0,
 * the clamp of pointnet_sa_module.py:27); out: (B,M,C). */
PCCX_API int pccx_gather(const float *points, int B, int N, int C, const int64_t *idx, int M,
                         float *out, void *stream);

/* pytorch3d knn_points(p1=q, p2=ref, K, return_nn) (compress.py:71, pn_kit.py:190, eval.py:132):
 * squared L2, K smallest ascending, ties by lower index.  q: (B,M,3); ref: (B,N,3);
 * dists: (B,M,K) f32 or NULL; idx: (B,M,K) int64 or NULL; nn: (B,M,K,3) or NULL (at least one of the three).  K <= 1024, K <= N <= 32768.
 * If patch_scale != 0, nn instead receives (ref[idx]-q)*patch_scale: the fused form of
 * compress.py:72 (subtract centre) and compress.py:108 (scale by (N/N0)^(1/3)). */
PCCX_API int pccx_knn(const float *q, int B, int M, const float *ref, int N, int K, float *dists,
                      int64_t *idx, float *nn, float patch_scale, void *stream);

/* pytorch3d ball_query(p1=q, p2=ref, K, radius) (pointnet_sa_module.py:18): first K indices in
 * index order with d2 < radius^2, padded with -1 (dists padded with 0). */
PCCX_API int pccx_ball_query(const float *q, int B, int M, const float *ref, int N, int K,
                             float radius, float *dists, int64_t *idx, void *stream);
/* The same query through a uniform grid hash of the candidates (cells of side >= radius, 27-cell walk, index order restored by
 * a bitmap): identical results, for large N / small radius.  workspace: pccx_ball_query_grid_workspace_ints(B, N) int32; N <= 32768. */
PCCX_API size_t pccx_ball_query_grid_workspace_ints(int B, int N);
PCCX_API int pccx_ball_query_grid(const float *q, int B, int M, const float *ref, int N, int K, float radius,
                                  int32_t *workspace, float *dists, int64_t *idx, void *stream);

/* One-directional nearest neighbour: for each x in X (B,P,3) the min over Y (B,Q,3) of |x-y|^2.
 * d2: (B,P); nn: (B,P) int32 or NULL.  Building block of pytorch3d chamfer_distance (AE.py:67,
 * eval.py:204) and of eval.py's D1 loop (eval.py:73-81). */
PCCX_API int pccx_nn_dist(const float *X, int B, int P, const float *Y, int Q, float *d2, int32_t *nn,
                          void *stream);
/* pccx_nn_dist for batches too small to fill the chip (configs[4]: 4 clouds): the reference cloud is scanned in `split` chunks by
 * different workgroups (partials in scratch_d / scratch_nn: split * B * P entries each; scratch_nn may be NULL when nn is) and merged
 * by a second kernel; identical results.  pccx_nn_dist_split_count: the split this shape should use (1 = call pccx_nn_dist). */
PCCX_API int pccx_nn_dist_split_count(int B, int P, int Q);
PCCX_API int pccx_nn_dist_split(const float *X, int B, int P, const float *Y, int Q, int split, float *scratch_d, int32_t *scratch_nn,
                                float *d2, int32_t *nn, void *stream);

/* chamfer_distance's value from the two pccx_nn_dist passes dxy (B,P), dyx (B,Q) (pytorch3d defaults: point and batch mean, both
 * directions summed; AE.py:67): out[0] = batch mean of (mean_p dxy + mean_q dyx), accumulated in double */
PCCX_API int pccx_chamfer_mean(const float *dxy, const float *dyx, int B, int P, int Q, float *out, void *stream);
/* Backward of chamfer_distance (batch mean of both directions' point means; AE.py:57-70,
 * pppe_pcd_ae.py:817-838) for fixed argmins nn_xy (B,P), nn_yx (B,Q) from pccx_nn_dist:
 * gX (B,P,3), gY (B,Q,3) receive grad_out * dL/dX, dL/dY (buffers are zeroed here). */
PCCX_API int pccx_chamfer_grad(const float *X, int B, int P, const float *Y, int Q, const int32_t *nn_xy,
                               const int32_t *nn_yx, float grad_out, float *gX, float *gY, void *stream);
/* pccx_chamfer_grad with the upstream gradient scalar read from device memory (no host sync: usable under hipGraph capture). */
PCCX_API int pccx_chamfer_grad_dev(const float *X, int B, int P, const float *Y, int Q, const int32_t *nn_xy,
                                   const int32_t *nn_yx, const float *grad_out_dev, float *gX, float *gY, void *stream);
/* ... into gX / gY the caller cleared (flags & 4) */
PCCX_API int pccx_chamfer_grad_dev_acc(const float *X, int B, int P, const float *Y, int Q, const int32_t *nn_xy,
                                       const int32_t *nn_yx, const float *grad_out_dev, float *gX, float *gY, int flags,
                                       void *stream);

/* D2 (point-to-plane) PSNR support (eval.py:58-60,73-81).  pccx_estimate_normals: PCA normal of
 * every point over its K neighbours nbr (B,N,K) int64 (e.g. pccx_knn with K=30, as open3d's
 * estimate_normals(KDTreeSearchParamKNN(knn=30))); unoriented.  pccx_point_plane_err:
 * err[b,i] = ((X[b,i] - Y[b,nn[b,i]]) . normals_Y[b,nn[b,i]])^2. */
PCCX_API int pccx_estimate_normals(const float *xyz, int B, int N, const int64_t *nbr, int K, float *normals,
                                   void *stream);
PCCX_API int pccx_point_plane_err(const float *X, int B, int P, const float *Y, const float *normals_Y, int Q,
                                  const int32_t *nn, float *err, void *stream);

/* ---- integer path: octree of the S sampled centres ------------------------------------------- */

/* Bytes needed per cloud for the bit array (one byte per bit) of pccx_octree_encode. */
PCCX_API int pccx_octree_bits_capacity(int S);

/* pn_kit.encode_sampled_np (pn_kit.py:380-401) = depth search over octree_np.encode
 * (octree_np.py:10-45) + octree_np.getDecodeFromPc (octree_np.py:114-133), scale = 1, followed by
 * pn_kit.binary_array_to_byte_array (pn_kit.py:463-467).
 * centres: (B,S,3) f32, S <= 1024; N: points per cloud (bpp denominator); min_bpp as
 * pn_kit.OCTREE_BPP_DICT[K].
 * bits:   (B,cap) uint8, one byte per bit, cap = pccx_octree_bits_capacity(S);
 * nbits:  (B) int32 stream length in bits;  depth: (B) int32 (DEPTH as the reference leaves it);
 * bytes:  (B,(cap+7)/8) uint8 packed stream, last partial byte right-aligned; nbytes: (B) int32. */
PCCX_API int pccx_octree_encode(const float *centres, int B, int S, int N, double min_bpp,
                                uint8_t *bits, int32_t *nbits, int32_t *depth, uint8_t *bytes,
                                int32_t *nbytes, void *stream);

/* pn_kit.decode_sampled_np -> octree_np.decode (octree_np.py:47-112) from the packed stream
 * (decompress.py:80-83 unpacks with pn_kit.byte_array_to_binary_array first).
 * mode 0 = "reference": bug-compatible with octree_np.decode as written (only the first 8 bits are
 *          consumed; output padded to 64 points);  S_out must be 64.
 * mode 1 = "full": level-by-level decode (the build's extension), descending-Morton order; the
 *          first min(count,S_out) points are written, the rest repeat the last point.
 * bytes: (B,stride) uint8; nbytes: (B) int32; out: (B,S_out,3) f32; count: (B) int32 decoded points. */
PCCX_API int pccx_octree_decode(const uint8_t *bytes, int stride, const int32_t *nbytes, int B,
                                int mode, int S_out, float *out, int32_t *count, void *stream);

/* ---- neural transforms (fp32 on the matrix cores) --------------------------------------------- */

/* Packed weight blobs.  pccx_pack_* run on the HOST, once per model load: every pointer they take
 * is a HOST pointer to a dense row-major fp32 tensor with the shape of the reference's state_dict
 * entry named in the comment; the caller uploads the resulting blob.
 * Limits: bottleneck d <= 16, d*L <= 128; layer widths are those of AE.py. */
PCCX_API size_t pccx_ae_encoder_blob_floats(void);
/* AE.AE.sa / AE.AE.pn (AE.py:16-17): sa.conv0 (32,3) sa.conv1 (64,32) sa.conv2 (128,64)
 * pn.mlp_Modules.{0,1,2,3}.0 (128,131) (256,128) (512,256) (d,512), each with its bias. */
PCCX_API int pccx_pack_ae_encoder(const float *sa_w0, const float *sa_b0, const float *sa_w1,
                                  const float *sa_b1, const float *sa_w2, const float *sa_b2,
                                  const float *pn_w0, const float *pn_b0, const float *pn_w1,
                                  const float *pn_b1, const float *pn_w2, const float *pn_b2,
                                  const float *pn_w3, const float *pn_b3, int d, float *blob_host);
PCCX_API size_t pccx_ae_decoder_blob_floats(int k);
/* AE.AE.inv_pool / inv_mlp (AE.py:19-27): inv_pool.{0,2,4} (256,d) (1024,256) (k*128,1024);
 * inv_mlp.mlp_Modules.{0,1,2,3}.0 (128,128+d) (64,128) (32,64) (3,32). */
PCCX_API int pccx_pack_ae_decoder(const float *ip_w0, const float *ip_b0, const float *ip_w1,
                                  const float *ip_b1, const float *ip_w2, const float *ip_b2,
                                  const float *m_w0, const float *m_b0, const float *m_w1,
                                  const float *m_b1, const float *m_w2, const float *m_b2,
                                  const float *m_w3, const float *m_b3, int k, int d, float *blob_host);
PCCX_API size_t pccx_prob_blob_floats(void);
/* AE.ConditionalProbabilityModel (AE.py:96-105): model_pn.mlp_Modules.{0,1,2}.0 (64,3) (128,64)
 * (256,128); model_mlp.{0,2,4} (512,259) (512,512) (d*L,512). */
PCCX_API int pccx_pack_prob(const float *p_w0, const float *p_b0, const float *p_w1, const float *p_b1,
                            const float *p_w2, const float *p_b2, const float *m_w0, const float *m_b0,
                            const float *m_w1, const float *m_b1, const float *m_w2, const float *m_b2,
                            int d, int L, float *blob_host);

/* Analysis transform of AE.AE.forward (AE.py:37-45) = the two per-patch loops of
 * compress.py:113-127: ae.sa (pn_kit.py:164-211), ae.pn (pn_kit.py:124-144), sigmoid spread, round.
 * patches: (P,K,3) f32, already centred and scaled (compress.py:105-108); K % 16 == 0, K <= 1024.
 * feat_ws: workspace of P*K*128 floats.  Outputs (P,d) f32: latent_raw (PointNet output),
 * latent (after sigmoid spread), latent_q (rounded).  Spread is L - 0.2. */
PCCX_API int pccx_ae_encode(const float *patches, int P, int K, const float *enc_blob, int d, int L,
                            float *feat_ws, float *latent_raw, float *latent, float *latent_q,
                            void *stream);

/* The two halves of pccx_ae_encode, as the reference calls them (compress.py:114 ae.sa,
 * compress.py:121 ae.pn).  feat: (P,8,K,16) f32 = the (P,128,K) SetAbstraction feature map with
 * channels grouped by 16 (feat[p][c/16][i][c%16]), the layout pccx_pn_forward consumes. */
PCCX_API int pccx_sa_forward(const float *patches, int P, int K, const float *enc_blob, float *feat,
                             void *stream);
PCCX_API int pccx_pn_forward(const float *patches, const float *feat, int P, int K,
                             const float *enc_blob, int d, int L, float *latent_raw, float *latent,
                             float *latent_q, void *stream);

PCCX_API size_t pccx_ae_decode_workspace_floats(int P);
/* Synthesis transform (AE.py:48-53 = decompress.py:97-102) and, optionally, the reassembly of
 * decompress.py:104-116.  latent_q: (P,d) f32; workspace: pccx_ae_decode_workspace_floats(P).
 * patches_out (P,k,3) or NULL: raw decoder output (new_xyz.transpose(2,1)).
 * pc_out (P*k,3) or NULL: ((patch / scale) + centres[patch] - 0.5) * longest[b] / (1-margin) +
 * center[b] with b = patch / S; needs centres (P,3), nrm_center (B,3), nrm_longest (B). */
PCCX_API int pccx_ae_decode(const float *latent_q, int P, int d, int k, const float *dec_blob,
                            float *workspace, float *patches_out, float scale, const float *centres,
                            const float *nrm_center, const float *nrm_longest, int S, double margin,
                            float *pc_out, void *stream);

/* The bf16x3 arithmetic mode (the default of the host layer since round 2, validated against every
 * oracle / golden parity test at the f32 tolerances): the same synthesis transform with the K = 1024 Linear and inv_mlp evaluated as
 * fp32 products of three bf16 pieces per operand on the bf16 matrix cores (six v_mfma_f32_16x16x32_bf16 passes,
 * fp32 accumulate; error at the level of an fp32 summation reorder, not bit-identical to pccx_ae_decode).
 * b3_blob: pccx_dec_b3_blob_floats(k) floats on the device, filled once from the packed decoder blob (already on
 * the device) by pccx_pack_ae_decoder_b3.  workspace: pccx_ae_decode_b3_workspace_floats(P).  Replaces the same
 * reference lines as pccx_ae_decode (AE.py:48-53, decompress.py:97-116). */
PCCX_API size_t pccx_sa_b3_blob_floats(void);
PCCX_API int pccx_pack_sa_b3(const float *enc_blob_dev, float *sa_b3_blob_dev, void *stream);
/* pccx_sa_forward (pn_kit.py:164-211) with conv1 / conv2 on bf16x3 operands. */
PCCX_API int pccx_sa_forward_b3(const float *patches, int P, int K, const float *enc_blob, const float *sa_b3_blob,
                                float *feat, void *stream);
PCCX_API size_t pccx_pn_b3_blob_floats(void);
PCCX_API int pccx_pack_pn_b3(const float *enc_blob_dev, float *pn_b3_blob_dev, void *stream);
/* pccx_pn_forward (pn_kit.py:124-144, AE.py:43-45) on bf16x3 operands. */
PCCX_API int pccx_pn_forward_b3(const float *patches, const float *feat, int P, int K, const float *enc_blob,
                                const float *pn_b3_blob, int d, int L, float *latent_raw, float *latent,
                                float *latent_q, void *stream);
/* pccx_sa_forward_b3 + pccx_pn_forward_b3 in ONE kernel (compress.py:113-127: ae.sa, ae.pn, sigmoid spread, round): the
 * (P,128,K) feature map of ae.sa stays inside the CU instead of making a round trip through HBM.  Same blobs, same arithmetic
 * and results as the two calls.  pccx_ae_encode_b3_fused_ok(K) tells whether a K-point patch fits the kernel's LDS budget
 * (K <= 512); otherwise run the two calls through a feature workspace. */
PCCX_API int pccx_ae_encode_b3_fused_ok(int K);
PCCX_API int pccx_ae_encode_b3(const float *patches, int P, int K, const float *enc_blob, const float *sa_b3_blob,
                               const float *pn_b3_blob, int d, int L, float *latent_raw, float *latent,
                               float *latent_q, void *stream);
/* The in-patch neighbour selection SetAbstraction starts with (pn_kit.py:186-190: knn_points(xyz, xyz, K=16) on each
 * (K,3) patch) as a kernel of its own: pure vector-ALU work that runs at 8 waves per SIMD here instead of the encoder's 2.
 * Table layout nbr[P][K][16], pccx_patch_knn16_index_bytes(K) bytes per index (1 while K <= 256, else 2), 16-byte aligned,
 * pccx_patch_knn16_bytes(P, K) bytes in all.  Rows hold the SET of the 16 nearest points under the oracle's (distance, index)
 * order (orc_knn); the order inside a row is unspecified (a max-pool follows, pn_kit.py:211). */
PCCX_API int pccx_patch_knn16_index_bytes(int K);
PCCX_API size_t pccx_patch_knn16_bytes(int P, int K);
PCCX_API int pccx_patch_knn16(const float *patches, int P, int K, void *nbr, void *stream);
/* pccx_ae_encode_b3 with the neighbour selection taken out into pccx_patch_knn16 (both launched from this call): the form
 * the host layer uses.  workspace: pccx_ae_encode_b3_workspace_bytes(P, K) bytes on the device, 16-byte aligned.  Results are
 * bit-identical to pccx_ae_encode_b3. */
PCCX_API size_t pccx_ae_encode_b3_workspace_bytes(int P, int K);
PCCX_API int pccx_ae_encode_b3_ws(const float *patches, int P, int K, const float *enc_blob, const float *sa_b3_blob,
                                  const float *pn_b3_blob, int d, int L, float *latent_raw, float *latent,
                                  float *latent_q, void *workspace, void *stream);
/* The fused kernel of pccx_ae_encode_b3_ws ALONE, on neighbour tables the caller has filled with pccx_patch_knn16(patches, P, K,
 * tables, stream): the two launches as two calls, for a host that times or schedules them separately (bench.py's stage table).
 * Replaces the same lines (compress.py:113-127). */
PCCX_API int pccx_ae_encode_b3_tables(const float *patches, int P, int K, const float *enc_blob, const float *sa_b3_blob,
                                      const float *pn_b3_blob, int d, int L, float *latent_raw, float *latent,
                                      float *latent_q, const void *tables, void *stream);
PCCX_API size_t pccx_dec_b3_blob_floats(int k);
PCCX_API int pccx_pack_ae_decoder_b3(const float *dec_blob_dev, int k, float *b3_blob_dev, void *stream);
PCCX_API size_t pccx_ae_decode_b3_workspace_floats(int P);
PCCX_API int pccx_ae_decode_b3(const float *latent_q, int P, int d, int k, const float *dec_blob,
                               const float *b3_blob, float *workspace, float *patches_out, float scale,
                               const float *centres, const float *nrm_center, const float *nrm_longest,
                               int S, double margin, float *pc_out, void *stream);

/* The f16x2 arithmetic mode: the same two transforms with every fp32 product formed from TWO fp16 pieces
 * per operand (hi = rn16(x), lo = rn16(x - hi): 22-23 significant bits, the operand precision of "3xTF32") and three
 * v_mfma_f32_16x16x32_f16 passes, fp32 accumulate -- half the matrix instructions of bf16x3.  fp16's narrow exponent is handled by
 * exact power-of-two scales: static ones per layer from rigorous interval bounds of the layers (computed when the weights are
 * packed) and one per patch from the data (largest |coordinate| / head activation), so no operand can overflow and the lo piece
 * keeps its bits; powers of two commute with fp32 rounding, so the scaled chain computes what the unscaled one would.
 * The blobs are built on the HOST from the same state_dict tensors as pccx_pack_ae_encoder / pccx_pack_ae_decoder (HOST pointers,
 * same argument order) and uploaded by the caller.
 * pccx_ae_encode_h2_ws replaces compress.py:113-127 (ae.sa, ae.pn, sigmoid spread, round) like pccx_ae_encode_b3_ws; enc_blob is
 * the fp32 blob (conv0 of SetAbstraction stays an fp32 MFMA), workspace holds the neighbour tables of pccx_patch_knn16.
 * pccx_ae_decode_h2 replaces AE.py:48-53 / decompress.py:97-116 like pccx_ae_decode_b3; dec_blob is the fp32 blob (the two head
 * Linears stay fp32 MFMAs). */
PCCX_API size_t pccx_ae_encoder_h2_blob_floats(void);
PCCX_API int pccx_pack_ae_encoder_h2(const float *sa_w0, const float *sa_b0, const float *sa_w1, const float *sa_b1,
                                     const float *sa_w2, const float *sa_b2, const float *pn_w0, const float *pn_b0,
                                     const float *pn_w1, const float *pn_b1, const float *pn_w2, const float *pn_b2,
                                     const float *pn_w3, const float *pn_b3, int d, float *h2_blob);
PCCX_API int pccx_ae_encode_h2_fused_ok(int K);
PCCX_API size_t pccx_ae_encode_h2_workspace_bytes(int P, int K);
PCCX_API int pccx_ae_encode_h2_ws(const float *patches, int P, int K, const float *enc_blob, const float *h2_blob, int d, int L,
                                  float *latent_raw, float *latent, float *latent_q, void *workspace, void *stream);
/* the fused f16x2 kernel alone on caller-filled tables, as pccx_ae_encode_b3_tables (compress.py:113-127) */
PCCX_API int pccx_ae_encode_h2_tables(const float *patches, int P, int K, const float *enc_blob, const float *h2_blob, int d, int L,
                                      float *latent_raw, float *latent, float *latent_q, const void *tables, void *stream);
PCCX_API size_t pccx_ae_decoder_h2_blob_floats(int k);
PCCX_API int pccx_pack_ae_decoder_h2(const float *ip_w0, const float *ip_b0, const float *ip_w1, const float *ip_b1,
                                     const float *ip_w2, const float *ip_b2, const float *m_w0, const float *m_b0,
                                     const float *m_w1, const float *m_b1, const float *m_w2, const float *m_b2,
                                     const float *m_w3, const float *m_b3, int k, int d, float *h2_blob);
PCCX_API size_t pccx_ae_decode_h2_workspace_floats(int P);
PCCX_API int pccx_ae_decode_h2(const float *latent_q, int P, int d, int k, const float *dec_blob, const float *h2_blob,
                               float *workspace, float *patches_out, float scale, const float *centres,
                               const float *nrm_center, const float *nrm_longest, int S, double margin, float *pc_out,
                               void *stream);

/* AE.ConditionalProbabilityModel.forward (AE.py:107-123) + pn_kit.pmf_to_cdf (pn_kit.py:452-461)
 * + torchac's float-CDF -> 16-bit conversion.  centres: (B,S,3), S % 16 == 0.  Any of the outputs
 * may be NULL: pmf (B,S,d,L) f32; cdf (B,S,d,L+1) f32; cdf_int (B,S,d,L+1) int32 holding uint16. */
PCCX_API int pccx_prob_forward(const float *centres, int B, int S, int d, int L, const float *prob_blob,
                               float *pmf, float *cdf, int32_t *cdf_int, void *stream);

/* torchac.encode_float_cdf (compress.py:134-136) / decode_float_cdf (decompress.py:92-93) on the
 * integer CDFs of pccx_prob_forward.  One independent stream per cloud of nsym = S*d symbols.
 * latent_q: (B,nsym) f32 integer-valued in [-(L/2), L/2]; out: (B,cap) bytes; nbytes: (B) int32
 * (negative = capacity exceeded, |value| bytes were needed). */
PCCX_API int pccx_range_encode(const int32_t *cdf_int, const float *latent_q, int B, int nsym, int L,
                               uint8_t *out, int cap, int32_t *nbytes, void *stream);
PCCX_API int pccx_range_decode(const int32_t *cdf_int, const uint8_t *in, int stride,
                               const int32_t *nbytes, int B, int nsym, int L, float *latent_q,
                               void *stream);

/* torchac's float -> 16-bit integer CDF conversion (torchac 0.9.3 _convert_to_int_and_normalize with
 * needs_normalization=True, the first step of encode_float_cdf / decode_float_cdf, compress.py:136, decompress.py:93):
 * cdf (nrows, Lp) f32 in [0,1] -> cdf_int (nrows, Lp) int32 holding 16-bit values. */
PCCX_API int pccx_cdf_float_to_int(const float *cdf, int64_t nrows, int Lp, int32_t *cdf_int, void *stream);

/* ---- the three files of every cloud (HOST side; no HIP call) ------------------------------------ */

/* compress.py:139-152 writes <name>.p.bin, <name>.s.bin and <name>.c.bin inside its timer and decompress.py:80-91,113 reads them back
 * inside its own, one Python open/write/close per file.  Here the files of a whole batch are cut from / filled into the HOST copy of the
 * batch's packed stream buffer by `threads` host threads (0 = min(8, hardware threads)).  packed_host: the buffer of
 * pccx_streams_packed_bytes(B, s_stride, p_cap) bytes laid out as
 *     s_nbytes (B) int32 | p_nbytes (B) int32 | c (B,4) f32 [cx,cy,cz,longest] | s_bytes (B,s_stride) u8 | p_bytes (B,p_cap) u8
 * (what the octree / range-coder entry points above write when handed views of one device buffer, copied to the host in one piece).
 * File b is <dir>/<names + name_off[b]><ext> (names: NUL-terminated strings; name_off: (B) int64 offsets into it).  Formats are the
 * reference's: .s.bin = the first s_nbytes[b] bytes of row b, .p.bin = the first p_nbytes[b] bytes of row b, .c.bin = 16 bytes.
 * Write refuses (before touching the file system) a byte count that is negative or exceeds its row.  Read fills the counts and rows
 * (tails cleared) and fails on a missing file, a file longer than its row or a .c.bin that is not 16 bytes.  Both block until done. */
PCCX_API size_t pccx_streams_packed_bytes(int B, int s_stride, int p_cap);
PCCX_API int pccx_write_streams_host(const void *packed_host, int B, int s_stride, int p_cap, const char *dir, const char *names,
                                     const int64_t *name_off, int threads);
PCCX_API int pccx_read_streams_host(void *packed_host, int B, int s_stride, int p_cap, const char *dir, const char *names,
                                    const int64_t *name_off, int threads);
/* Sizes in bytes of <name>.s.bin and <name>.p.bin of B clouds (-1 = no such file): what a reader sizes s_stride / p_cap from. */
PCCX_API int pccx_stream_sizes_host(int B, const char *dir, const char *names, const int64_t *name_off, int64_t *s_sizes,
                                    int64_t *p_sizes, int threads);

/* ---- generic layers for the other model families (PPPF_AE.py, pointnet_sa_module.py,
 *      pppe_pcd_ae.py:556-917): correctness-first building blocks ------------------------------- */

/* 1x1 Conv2d / Conv1d / Linear (pointnet_sa_module.py:51, PPPF_AE.py:65-80,122-123,
 * pppe_pcd_ae.py:556-568,697-707) on row-major "channels last" activations, with eval-mode
 * BatchNorm folded into W and b by the caller: out[M][N] = act(x[M][K] . W^T + b).
 * pccx_pack_linear (HOST pointers): W (N,K) row-major -> pccx_packed_linear_floats(N,K) floats.
 * pccx_linear: x (M, ldx>=K), wp packed (device), bias (N) or NULL, out (M, ldo>=N); `relu` is a flag word: bit 0 = ReLU,
 * bit 1 = the autocast form of train_pppe_pcd_ae.py:193-217 (operands and result rounded to bf16, products on the bf16
 * matrix cores, fp32 accumulate; pccx_linear_dw takes the same bit 1 in `flags`). */
PCCX_API size_t pccx_packed_linear_floats(int N, int K);
PCCX_API int pccx_pack_linear(const float *W_host, int N, int K, float *wp_host);
PCCX_API int pccx_linear(const float *x, int M, int K, int ldx, const float *wp, const float *bias,
                         int N, int relu, float *out, int ldo, void *stream);

/* The same layer in the bf16x3 arithmetic: pccx_pack_linear_b3 splits the packed f32 fragments (device)
 * into three bf16 planes per K = 32 block, pccx_linear_b3 forms each fp32 product from six bf16 MFMA products, fp32 accumulate. */
PCCX_API size_t pccx_packed_linear_b3_floats(int N, int K);
PCCX_API int pccx_pack_linear_b3(const float *wp_dev, int N, int K, float *wplanes_dev, void *stream);
PCCX_API int pccx_linear_b3(const float *x, int M, int K, int ldx, const float *wplanes, const float *bias,
                            int N, int relu, float *out, int ldo, void *stream);

/* The wide stacks of the PointNet++ families kept in "planes" between layers (csrc/planes.hip): the activation of M rows x K
 * channels is stored as the three bf16 planes of the next layer's MFMA B operand, [K/32 block][16-row tile][plane][64 lanes] x 16 B,
 * pccx_planes_floats(M, K) floats.
 *   pccx_group_planes : gather + concat + split = index_points(features, idx) ++ index_points(xyz, idx) of
 *       pointnet_sa_module.py:73-83 (idx -1 -> row 0, :27), written as planes.  Row r reads source row
 *       (r / rows_per_batch) * n_src + idx[r] of f0 (C0 channels, row stride ld0) then f1 (C1, ld1); idx NULL: row r itself
 *       (fp32 rows -> planes).  Either source may be absent (C = 0).
 *   pccx_pack_planes_gemm : pccx_pack_linear_b3's planes reordered into per-m-block streams (pccx_planes_gemm_weight_floats).
 *   pccx_planes_gemm  : one Conv1x1 / Linear (+ folded BatchNorm) (+ ReLU, bit 0 of relu) on planes.  epilogue 0: out = planes of
 *       the N output channels; 1: out = fp32 rows (M, ldo); 2: out (M / group, ldo) = max over each `group` consecutive rows
 *       (torch.max over nsample, pointnet_sa_module.py:91; group in {32, 64, 128} dividing M). */
PCCX_API size_t pccx_planes_floats(int64_t M, int K);
PCCX_API int pccx_group_planes(const float *f0, int C0, int ld0, const float *f1, int C1, int ld1, const int64_t *idx, int64_t M,
                               int64_t rows_per_batch, int64_t n_src, float *planes, void *stream);
/* torch.cat([a, b.unsqueeze(1).repeat(1, P, 1)], -1) written directly as planes (FoldingNet's inputs, PPPF_AE.py:99-106): row r =
 * the C0 channels of f0 row (mod0 > 0 ? r % mod0 : r) then the C1 channels of f1 row r / div1. */
PCCX_API int pccx_fold_planes(const float *f0, int C0, int ld0, int64_t mod0, const float *f1, int C1, int ld1, int64_t div1, int64_t M,
                              float *planes, void *stream);
/* act(base[r / div] + x[mod ? r % mod : r] @ w^T), r < M -- pccx_rows_affine_small's values (same fmaf chain) -- written as the operand
 * planes of the next layer (pccx_planes_floats(M, C) floats) instead of fp32 rows: FoldingNet's first layers, PPPF_AE.py:99-107 */
PCCX_API int pccx_rows_affine_planes(const float *base, int C, int64_t div, const float *x, int ldx, int Ks, int64_t mod, const float *w,
                                     int relu, int64_t M, float *planes, void *stream);
PCCX_API size_t pccx_planes_gemm_weight_floats(int N, int K);
PCCX_API int pccx_pack_planes_gemm(const float *wplanes_dev, int N, int K, float *wstream_dev, void *stream);
PCCX_API int pccx_planes_gemm(const float *planes_in, int64_t M, int K, const float *wstream, const float *bias, int N, int relu,
                              int epilogue, int group, float *out, int ldo, void *stream);
/* pccx_planes_gemm with the gather inside the kernel: row r of the input = row (r / rows_per_batch) * n_src + max(idx[r], 0) of src,
 * fp32 rows of ldp = 32 * ceil(K / 32) floats ([features, xyz] zero padded, 16-byte aligned). */
PCCX_API int pccx_planes_gemm_gather(const float *src, int ldp, const int64_t *idx, int64_t rows_per_batch, int64_t n_src, int64_t M,
                                     int K, const float *wstream, const float *bias, int N, int relu, int epilogue, int group,
                                     float *out, int ldo, void *stream);
/* Three wide Conv-BN-ReLU layers (widths 241..256, 241..256, 497..512: the first three of sa3, PPPF_AE.py:32-34) in one kernel, one
 * 16-row tile per wave, activations in registers, the input rows gathered inside (as pccx_planes_gemm_gather; idx NULL: row r itself);
 * the output is the operand planes of the layer that follows.  K0 of 8 or 9 blocks of 32 channels (<= 288).  The weight stream is
 * built by pccx_pack_planes_chain_wide from the three layers' pccx_pack_linear_b3 planes. */
PCCX_API size_t pccx_planes_chain_wide_weight_floats(int K0);
PCCX_API int pccx_pack_planes_chain_wide(const float *wp3_l0, const float *wp3_l1, const float *wp3_l2, int K0, int N0, int N1, int N2,
                                         float *wstream_dev, void *stream);
PCCX_API int pccx_planes_chain_wide(const float *src, int ldp, const int64_t *idx, int64_t rows_per_batch, int64_t n_src, int64_t M,
                                    int K0, const float *wstream, const float *b0, int N0, const float *b1, int N1, const float *b2,
                                    int N2, float *out_planes, void *stream);
/* A whole Conv-BN-ReLU x 4 + max-over-nsample stack (pointnet_sa_module.py:90-91) in one kernel, the activations between the layers
 * staying in registers: out (M / group, ldo) from the planes of the gathered input.  wstream = the four layers'
 * pccx_pack_planes_gemm streams back to back; b0..b3 the (folded) biases.  Supported width patterns: (<=32, 33..64, 33..64, 65..128)
 * and (97..128, 97..128, 97..128, 129..256) -- sa1 and sa2 of PPPF_AE.py:29-34; anything else returns PCCX_ERR_ARG and the caller
 * runs pccx_planes_gemm layer by layer.  group in {32, 64, 128}, or 1 = no reduction: out (M, ldo) holds the stack's output rows
 * (the stacks evaluated once per SOURCE row, the groups taking their maxima with pccx_gather_max). */
PCCX_API int pccx_planes_chain4(const float *planes_in, int64_t M, int K0, const float *wstream, const float *b0, int N0,
                                const float *b1, int N1, const float *b2, int N2, const float *b3, int N3, int group, float *out,
                                int ldo, void *stream);
/* The same stack with the gather inside the kernel: row r of the input is row (r / rows_per_batch) * n_src + max(idx[r], 0) of src,
 * fp32 rows of ldp = 32 * ceil(K0 / 32) floats (the K0 channels [features, xyz] zero padded; 16-byte aligned).  The grouped tensor of
 * pointnet_sa_module.py:73-83 is never materialised. */
PCCX_API int pccx_planes_chain4_gather(const float *src, int ldp, const int64_t *idx, int64_t rows_per_batch, int64_t n_src, int64_t M,
                                       int K0, const float *wstream, const float *b0, int N0, const float *b1, int N1,
                                       const float *b2, int N2, const float *b3, int N3, int group, float *out, int ldo, void *stream);

/* ---- the same layers in the f16x2 arithmetic (f16x2): every fp32 product formed from TWO fp16 pieces per operand, three
 * v_mfma_f32_16x16x32_f16 per fp32 product instead of bf16x3's six, planes of 4 bytes per value instead of 6.  fp16 has five exponent
 * bits, so every operand travels times an exact power of two: activations entering layer l times sigma_l (chosen by the host from
 * rigorous interval bounds of the stack, so that |sigma_l y| <= 2^15), weights times tau_l (max|W| tau <= 2^14); a layer's accumulator
 * is sigma_l tau_l (W y + b s).  The host passes each layer's bias as sigma_l tau_l b and the power-of-two ratios the kernels need:
 * rho = sigma for a kernel that splits fp32 rows, scale_out = sigma_next / (sigma tau) for a planes epilogue, 1 / (sigma tau) for a
 * row / max epilogue.  The bounds assume stack inputs of magnitude <= 1: `dyn` (device, 2 floats from pccx_dyn_scale: s and 1 / s, or
 * NULL for 1) is the stack's dynamic input normalisation -- Conv / ReLU stacks are positively homogeneous in (input, biases), so the
 * kernels multiply gathered inputs and biases by s and the stack's final rows by 1 / s; no input can overflow whatever the data.
 * amax8 (device, 8 floats, or NULL): the row epilogues fold the largest |value| they write into it (atomic max over 8 replicas) --
 * the next stack's input bound.  Same shapes, epilogues and restrictions as the bf16x3 entry points above. */
PCCX_API size_t pccx_planes_floats_h2(int64_t M, int K);
PCCX_API size_t pccx_packed_linear_h2_floats(int N, int K);
/* wp_dev: the f32 fragments of pccx_pack_linear (uploaded) -> the two fp16 planes [t][MT][2] of tau * W */
PCCX_API int pccx_pack_linear_h2(const float *wp_dev, int N, int K, float tau, float *wplanes_dev, void *stream);
PCCX_API size_t pccx_planes_gemm_weight_floats_h2(int N, int K);
PCCX_API int pccx_pack_planes_gemm_h2(const float *wplanes_dev, int N, int K, float *wstream_dev, void *stream);
PCCX_API int pccx_group_planes_h2(const float *f0, int C0, int ld0, const float *f1, int C1, int ld1, const int64_t *idx, int64_t M,
                                  int64_t rows_per_batch, int64_t n_src, float rho, const float *dyn, float *planes, void *stream);
PCCX_API int pccx_fold_planes_h2(const float *f0, int C0, int ld0, int64_t mod0, const float *f1, int C1, int ld1, int64_t div1, int64_t M,
                                 float rho, const float *dyn, float *planes, void *stream);
PCCX_API int pccx_rows_affine_planes_h2(const float *base, int C, int64_t div, const float *x, int ldx, int Ks, int64_t mod, const float *w,
                                        int relu, int64_t M, float rho, const float *dyn, float *planes, void *stream);
PCCX_API int pccx_planes_gemm_h2(const float *planes_in, int64_t M, int K, const float *wstream, const float *bias, int N, int relu,
                                 int epilogue, int group, float scale_out, const float *dyn, float *amax8, float *out, int ldo,
                                 void *stream);
/* The last layer of a set-abstraction stack whose groups are only reduced together (PPPF_AE.py:44 over pointnet_sa_module.py:91: the maximum
 * over the centroids of the maxima over their samples = the maximum over every source row that is a sample of any centroid, because the stack
 * acts on each un-centred row by itself): member = 1 byte per row from pccx_group_members, group = the source rows of one batch element
 * (32, 64 or 128, M % group == 0) -> out (M / group, ldo).  The layer's fp32 rows are never written. */
PCCX_API int pccx_planes_gemm_h2_member_max(const float *planes_in, int64_t M, int K, const float *wstream, const float *bias, int N, int relu,
                                            int group, const unsigned char *member, float scale_out, const float *dyn, float *out, int ldo,
                                            void *stream);
/* member (n_idx / per_batch, n_src) bytes: 1 where the row is named by an entry of idx (per_batch entries per batch element; -1 -> row 0, the
 * clamp of pointnet_sa_module.py:27), else 0.  The table's size and address must be multiples of 4 (cleared in words by a kernel). */
PCCX_API int pccx_group_members(const int64_t *idx, int64_t n_idx, int64_t per_batch, int64_t n_src, unsigned char *member, void *stream);
PCCX_API int pccx_planes_gemm_gather_h2(const float *src, int ldp, const int64_t *idx, int64_t rows_per_batch, int64_t n_src, int64_t M,
                                        int K, const float *wstream, const float *bias, int N, int relu, int epilogue, int group,
                                        float rho_in, float scale_out, const float *dyn, float *amax8, float *out, int ldo,
                                        void *stream);
/* scales5_host (HOST, 5 floats): {rho_in (gather form only), rho_1, rho_2, rho_3, 1 / (sigma_3 tau_3)}, rho_l = sigma_l / (sigma_{l-1} tau_{l-1}) */
PCCX_API int pccx_planes_chain4_h2(const float *planes_in, int64_t M, int K0, const float *wstream, const float *b0, int N0,
                                   const float *b1, int N1, const float *b2, int N2, const float *b3, int N3, int group,
                                   const float *scales5_host, const float *dyn, float *amax8, float *out, int ldo, void *stream);
PCCX_API int pccx_planes_chain4_gather_h2(const float *src, int ldp, const int64_t *idx, int64_t rows_per_batch, int64_t n_src, int64_t M,
                                          int K0, const float *wstream, const float *b0, int N0, const float *b1, int N1,
                                          const float *b2, int N2, const float *b3, int N3, int group, const float *scales5_host,
                                          const float *dyn, float *amax8, float *out, int ldo, void *stream);
/* The dynamic input normalisation: pccx_absmax folds max |x| over n floats into amax8 (8 non-negative floats the caller cleared);
 * pccx_dyn_scale turns one or two such maxima into dyn2 = {s, 1 / s}, s the largest power of two <= 1 with bound * s <= 1 where
 * bound = (combine ? max(a1 m1, a2 m2) : a1 m1 + a2 m2) + add (m2_8 may be NULL). */
PCCX_API int pccx_absmax(const float *x, int64_t n, float *amax8, void *stream);
PCCX_API int pccx_dyn_scale(const float *m1_8, float a1, const float *m2_8, float a2, float add, int combine, float *dyn2, void *stream);

/* The epilogue of pccx_prob_forward as an op of its own, for the generic (any --d / --L, compress.py:30-34) path: softmax over
 * the L levels of each row of logits (rows, L), pmf_to_cdf (pn_kit.py:452-461: cumsum, leading 0, clamp <= 1) and torchac 0.9.3's
 * integer CDF.  Any of pmf (rows, L), cdf (rows, L+1), cdf_int (rows, L+1) may be NULL. */
PCCX_API int pccx_softmax_cdf(const float *logits, int64_t rows, int L, float *pmf, float *cdf, int32_t *cdf_int, void *stream);
/* The reassembly epilogue of pccx_ae_decode as an op of its own (decompress.py:104-116): patches (P, k, 3) raw decoder output ->
 * out (P*k, 3) = ((patch / scale) + centres[patch] - 0.5) * longest[b] / (1 - margin) + center[b], b = patch / S. */
PCCX_API int pccx_reassemble(const float *patches, int64_t P, int k, float scale, const float *centres, const float *nrm_center,
                             const float *nrm_longest, int S, double margin, float *out, void *stream);
/* out[r][c] = act(base[r / div][c] + sum_{k<Ks} x[(mod ? r % mod : r)][k] * w[c][k]), Ks <= 4, C % 4 == 0: the per-point part of a
 * layer whose input rows are [small per-point part | long per-patch part] (FoldingNet's [grid | latent] and [coarse | latent],
 * PPPF_AE.py:99-107); base = the per-patch part's Linear (bias included), one row per patch.  w is (C, Ks) row-major. */
PCCX_API int pccx_rows_affine_small(const float *base, int C, int64_t div, const float *x, int ldx, int Ks, int64_t mod,
                                    const float *w, int relu, int64_t M, float *out, void *stream);
/* torch.max(knn_gather(y, idx.clamp(min=0)), nsample_dim)[0] without the gathered tensor (pointnet_sa_module.py:27-28,91):
 * y (B,N,C) fp32 rows, idx (B,M,ns) int64 with -1 padding -> out (B,M,C).  C % 4 == 0.  With it PointnetSAModule (which gathers
 * un-centred rows, :73-85) runs its Conv-BN-ReLU stack on the N source rows once instead of on M*ns copies of them. */
PCCX_API int pccx_gather_max(const float *y, int B, int N, int C, const int64_t *idx, int M, int ns, float *out, void *stream);
/* The same maxima written as the NEXT level's input rows (pointnet_sa_module.py:83, features first, xyz last): out (B, M, ldo),
 * ldo = 32 * ceil((C + 3) / 32): [C maxima | xyz (B, M, 3) of the centroids | zeros] -- what the gathering planes kernels read. */
PCCX_API int pccx_gather_max_rows(const float *y, int B, int N, int C, const int64_t *idx, int M, int ns, const float *xyz, float *out, int ldo,
                                  void *stream);
/* torch.max(features, neighbour_dim)[0] (pointnet_sa_module.py:91, pppe_pcd_ae.py:610):
 * x (G,Kn,C) -> out (G,C). */
PCCX_API int pccx_group_max(const float *x, int64_t G, int Kn, int C, float *out, void *stream);

/* y = sigmoid(x)*(L-0.2) - (L-0.2)/2 (PPPF_AE.py:136-137, AE.py:43-44), optionally rounded;
 * y = round(x) (AE.STEQuantize, AE.py:79-81). */
PCCX_API int pccx_sigmoid_spread(const float *x, int64_t n, int L, int do_round, float *y, void *stream);
PCCX_API int pccx_round(const float *x, int64_t n, float *y, void *stream);

/* quantize_st forward value and its dequantisation (pppe_pcd_ae.py:719-735, :873). y_deq may be NULL. */
PCCX_API int pccx_quantize_st(const float *x, int64_t n, float qmin, float qmax, int levels, float *y_q,
                              float *y_deq, void *stream);

/* ---- training-step primitives (train_pppe_pcd_ae.py:184-226): train-mode BatchNorm, the
 *      backward of the generic layers, loss pieces, gradient clipping and Adam.  Rows are channels-last.
 *      `sums` arguments are scratch of 2*C doubles. ------------------------------------------------ */
PCCX_API int pccx_pack_linear_device(const float *W, int N, int K, int transpose, float *wp, void *stream);
/* dW (N,K) += dZ^T (M,N) . X (M,K)   (dW must be initialised by the caller) */
PCCX_API int pccx_linear_dw(const float *dZ, const float *X, int64_t M, int N, int K, int ldz, int ldx,
                            float *dW, int flags, void *stream);
/* BatchNorm{1,2}d in training mode (pppe_pcd_ae.py:556-568): batch moments per channel, running stats
 * updated with `momentum` (running_* may be NULL); then y = [relu]((z-mean)*rstd*gamma+beta). */
PCCX_API int pccx_bn_train_stats(const float *Z, int64_t M, int C, float eps, float momentum, double *sums,
                                 float *mean, float *rstd, float *running_mean, float *running_var,
                                 void *stream);
PCCX_API int pccx_bn_relu_forward(const float *Z, int64_t M, int C, const float *mean, const float *rstd,
                                  const float *gamma, const float *beta, int relu, float *Y, void *stream);
/* backward of BN(train)+ReLU: dZ, and g_gamma / g_beta ACCUMULATED into the parameter gradients */
PCCX_API int pccx_bn_relu_backward(const float *dY, const float *Y, const float *Z, int64_t M, int C,
                                   const float *mean, const float *rstd, const float *gamma, double *sums,
                                   float *dZ, float *g_gamma, float *g_beta, void *stream);
/* nn.Linear on 1..8 rows (the pppe model's global, decoder and probability Linears at batch 4, pppe_pcd_ae.py:655-714, :739-802):
 * out (M, N) = act(x (M, K) . W^T + b) and dX (M, K) += dZ (M, N) . W straight from the row-major W (N, K), each W row read once
 * (a weight stream, not matrix work; no packed copy).  flags as pccx_linear (bit 0 ReLU, bit 1 autocast rounding).  K % 4 == 0;
 * dX is zeroed by the caller. */
PCCX_API int pccx_linear_skinny(const float *x, int M, int K, int ldx, const float *W, const float *bias, int N, int flags,
                                float *out, int ldo, void *stream);
PCCX_API int pccx_linear_skinny_dx(const float *dZ, int M, int N, int ldz, const float *W, int K, int flags, float *dX, int ldd,
                                   void *stream);
PCCX_API int pccx_col_sum(const float *dY, int64_t M, int C, double *sums, float *g_bias, void *stream);
/* The training step's forms of the four entry points above (train_pppe_pcd_ae.py:184-226; round 4): the same arithmetic in fewer launches.
 * flags bit 2 (value 4): the accumulation target (`sums`, dF, gX / gY) was CLEARED BY THE CALLER -- pccx/train.py clears one arena per
 * step with pccx_zero_bytes instead of one launch per reduction; pccx_bn_relu_train_forward flags bit 3 (value 8): `sums` already HOLDS the
 * column moments of Z (pccx_linear_moments produced Z), no reduction is launched; likewise pccx_bn_relu_train_backward after pccx_linear_bnback.  pccx_bn_relu_train_forward = pccx_bn_train_stats + pccx_bn_relu_forward
 * (moments, then one kernel that finalises them and applies the layer; mean / rstd / running stats / Y bit-identical);
 * pccx_bn_relu_train_backward = pccx_bn_relu_backward with g_gamma / g_beta WRITTEN (not accumulated); pccx_col_sum_w writes g_bias. */
PCCX_API int pccx_bn_relu_train_forward(const float *Z, int64_t M, int C, float eps, float momentum, double *sums, const float *gamma,
                                        const float *beta, int relu, float *mean, float *rstd, float *running_mean,
                                        float *running_var, float *Y, int flags, void *stream);
PCCX_API int pccx_bn_relu_train_backward(const float *dY, const float *Y, const float *Z, int64_t M, int C, const float *mean,
                                         const float *rstd, const float *gamma, double *sums, float *dZ, float *g_gamma,
                                         float *g_beta, int flags, void *stream);
PCCX_API int pccx_col_sum_w(const float *dY, int64_t M, int C, double *sums, float *g_bias, int flags, void *stream);
/* `sums` of the three entry points above holds pccx_train_sums_doubles(C) doubles (eight replicas of the 2 C column sums: a workgroup adds
 * into replica blockIdx % 8, the consuming kernel adds the replicas up -- the same-address atomics at the end of a reduction are an eighth
 * as deep); the older pccx_bn_train_stats / pccx_bn_relu_backward / pccx_col_sum keep their 2 C doubles. */
PCCX_API size_t pccx_train_sums_doubles(int C);
/* A Conv / Linear without bias whose epilogue accumulates the column moments of its output (sum z, sum z^2 into the eight replicas of
 * `sums`, pccx_train_sums_doubles(N) doubles): the Conv -> BatchNorm pairs of pppe_pcd_ae.py:556-568 in train mode then need no pass of
 * their own over the activation -- pccx_bn_relu_train_forward with flags bit 3 (8) takes `sums` as filled.  flags: bit 1 (2) = autocast
 * (bf16 operands and result; the moments are those of the rounded result), bit 2 (4) = `sums` was cleared by the caller. */
PCCX_API int pccx_linear_moments(const float *x, int M, int K, int ldx, const float *wp, int N, int flags, float *out, int ldo,
                                 double *sums, void *stream);
/* The dX GEMM behind such a pair in the backward pass: out = x . W^T is the BatchNorm's dY, and the epilogue accumulates the two column
 * sums its backward needs (sum d xhat | sum d, d = (Y > 0 ? dY : 0), xhat = (Z - mean) rstd) from the rows it has just produced;
 * pccx_bn_relu_train_backward with flags bit 3 (8) takes `sums` as filled.  Y, Z: the BatchNorm's output and input rows (M, ldo). */
PCCX_API int pccx_linear_bnback(const float *x, int M, int K, int ldx, const float *wp, int N, int flags, float *out, int ldo,
                                const float *Y, const float *Z, const float *mean, const float *rstd, double *sums, void *stream);
/* clear `bytes` (a multiple of 4) with a kernel (as a hipGraph node it is ordered like every other kernel: DESIGN.md section 7) */
PCCX_API int pccx_zero_bytes(void *p, size_t bytes, void *stream);
/* dst[0 .. bytes) = src[0 .. bytes) by a kernel (both 16-byte aligned, bytes % 16 == 0): how a training loop hands the next batch and its
 * FPS / kNN tables (functions of the coordinates only, pppe_pcd_ae.py:596-632) to the fixed buffers a captured step reads. */
PCCX_API int pccx_copy_bytes(const void *src, void *dst, size_t bytes, void *stream);
/* *table[i] += delta for n int64 counters whose device addresses sit in table_dev (BatchNorm's num_batches_tracked of a whole model) */
PCCX_API int pccx_add_i64_table(const int64_t *table_dev, int n, int64_t delta, void *stream);
PCCX_API int pccx_relu_backward(const float *dY, const float *Y, int64_t n, float *dZ, void *stream);
PCCX_API int pccx_group_max_arg(const float *x, int64_t G, int Kn, int C, float *out, int32_t *arg,
                                void *stream);
PCCX_API int pccx_group_max_backward(const float *dOut, const int32_t *arg, int64_t G, int Kn, int C,
                                     float *dX, void *stream);
/* backward of pccx_gather: dF (B,N,C) = scatter-add of dG (B,Mrows, row stride ldg >= C) through idx */
PCCX_API int pccx_gather_backward(const float *dG, int ldg, const int64_t *idx, int B, int Mrows, int N,
                                  int C, float *dF, void *stream);
/* the same, ACCUMULATING into a dF the caller cleared (flags & 4; without the flag dF is cleared here) */
PCCX_API int pccx_gather_backward_acc(const float *dG, int ldg, const int64_t *idx, int B, int Mrows, int N, int C, float *dF,
                                      int flags, void *stream);
/* F.smooth_l1_loss(a, b, reduction="mean") * n summed into value[0] (double); grad = grad_scale * dl/da */
PCCX_API int pccx_smooth_l1(const float *a, const float *b, int64_t n, float grad_scale, double *value,
                            float *grad, void *stream);
PCCX_API int pccx_quantize_st_backward(const float *x, const float *d_ydeq, int64_t n, float qmin, float qmax,
                                       int levels, float *dx, void *stream);
PCCX_API int pccx_rate_from_logits(const float *logits, const float *y_q, int B, int bins, int ld_yq,
                                   float *out, void *stream);
/* clip_grad_norm_ + torch.optim.Adam: accumulate sum g^2 over all tensors into acc (zeroed by the caller),
 * then one pccx_adam_step per tensor reads the norm on the device (gnorm_sq may be NULL = no clipping). */
PCCX_API int pccx_sumsq_accumulate(const float *g, int64_t n, double *acc, void *stream);
PCCX_API int pccx_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n,
                            const double *gnorm_sq, float max_norm, float lr, float beta1, float beta2,
                            float eps, int step, void *stream);
/* The same two steps over ALL parameter tensors in one launch each.  table_dev: ntensors rows of six int64 in device memory,
 * {param, grad, exp_avg, exp_avg_sq (device pointers), n (elements), first_block}; a workgroup handles 1024 consecutive elements
 * of one tensor, first_block = sum over the earlier rows of ceil(n / 1024), total_blocks = that sum over all rows.  hyper_dev
 * (the device state below) replaces lr / step when not NULL.  Element for element the arithmetic of pccx_adam_step. */
PCCX_API int pccx_sumsq_multi(const int64_t *table_dev, int ntensors, int64_t total_blocks, double *acc, void *stream);
PCCX_API int pccx_adam_multi(const int64_t *table_dev, int ntensors, int64_t total_blocks, const double *gnorm_sq,
                             float max_norm, const float *hyper_dev, float lr, int step, float beta1, float beta2,
                             float eps, void *stream);
/* Adam's per-step scalars kept on the device (torch.optim.Adam's `step` / bias_correction1/2, train_pppe_pcd_ae.py:216,220 via
 * optimizer.step()): a 32-byte, 8-byte aligned state
 *     float lr | float 1-beta1^t | float 1-beta2^t | int32 t | double beta1^t | double beta2^t
 * pccx_adam_advance_dev does t += 1 and refreshes the two corrections (one thread, stream-ordered), so a captured training
 * step carries its own step counter and a replay needs no host write. */
PCCX_API int pccx_adam_advance_dev(float *hyper, double beta1, double beta2, void *stream);
/* pccx_adam_step with lr and the bias corrections read from that device state (its first three floats).
 * No per-step launch argument, so the training step can be captured as a hipGraph (pccx.train.GraphedTrainStep). */
PCCX_API int pccx_adam_step_dev(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n,
                                const double *gnorm_sq, float max_norm, const float *hyper, float beta1, float beta2,
                                float eps, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PCCX_H */
