// blobs.h -- layout of the packed weight blobs (device float arrays) the NN kernels read.
//
// A blob is built once at model-load time on the host by pccx_pack_* (pack.hip) from the
// reference's state_dict tensors (SURVEY Appendix C key names) and uploaded by the caller.
// Dense layers are stored as MFMA A-operand fragments (see mfma_chain.h):
//     frag(kt, mt)[lane][r] = W[16*mt + (lane&15)][chan(16*kt + 4*(lane>>4) + r)]
// kt-major ([kt][mt][64 lanes][4]); `chan` is a per-layer input-channel permutation chosen so
// that producer and consumer kernels agree without any data movement.  All offsets are in
// floats and multiples of 4 (16-byte aligned).
#pragma once

// ---- AE encoder: SetAbstraction 3->32->64->128 (AE.py:16) + PointNet 131->128->256->512->d (AE.py:17)
#define ENC_SA_W0B0 0                         // [32][4]: w(3), bias
#define ENC_SA_B1 (ENC_SA_W0B0 + 32 * 4)       // [64]
#define ENC_SA_B2 (ENC_SA_B1 + 64)             // [128]
#define ENC_SA_W1 (ENC_SA_B2 + 128)            // KT=2,  MT=4
#define ENC_SA_W2 (ENC_SA_W1 + 2 * 4 * 256)    // KT=4,  MT=8
#define ENC_PN_B0 (ENC_SA_W2 + 4 * 8 * 256)    // [128]
#define ENC_PN_B1 (ENC_PN_B0 + 128)            // [256]
#define ENC_PN_B2 (ENC_PN_B1 + 256)            // [512]
#define ENC_PN_B3 (ENC_PN_B2 + 512)            // [16]  (d <= 16, zero padded)
#define ENC_PN_W0 (ENC_PN_B3 + 16)             // KT=9 (128 SA features, then xyz), MT=8
#define ENC_PN_W1 (ENC_PN_W0 + 9 * 8 * 256)    // KT=8,  MT=16
#define ENC_PN_W2 (ENC_PN_W1 + 8 * 16 * 256)   // KT=16, MT=32
#define ENC_PN_W3 (ENC_PN_W2 + 16 * 32 * 256)  // KT=32, MT=1
// PointNet fragments again, in the order pn_forward_kernel consumes them (one pass = 744 fragments,
// padded to ENC_PN_STREAM_CHUNKS chunks of WS_CHUNK): L0 [kt][mt], L1 [kt][mt], then for each pair of
// layer-2 output tiles mp: L2 [kt][2mp..2mp+1] followed by L3 k-tiles 2mp, 2mp+1.
#ifndef WS_CHUNK
#define WS_CHUNK 8                        // fragments (KiB) per LDS-ring chunk of the PointNet stream (mfma_chain.h: WStreamT)
#endif
#define ENC_PN_STREAM (ENC_PN_W3 + 32 * 1 * 256)
#define ENC_PN_STREAM_FRAGS 744
#define ENC_PN_STREAM_CHUNKS (2 * ((ENC_PN_STREAM_FRAGS + 2 * WS_CHUNK - 1) / (2 * WS_CHUNK)))      // even
#define ENC_BLOB_FLOATS (ENC_PN_STREAM + ENC_PN_STREAM_CHUNKS * WS_CHUNK * 256)

// PointNet weight stream on bf16x3 operands (encoder.hip: pn_forward_b3_kernel): L0 [5][8][3], L1 [4][16][3],
// then per half h of layer 2's outputs: L2 [8][16][3] and L3 [8][1][3] fragments, [kt32][mt][plane] each (pccx_pack_pn_b3),
// padded to chunks of 24.
#define PN_B3_CHUNK 24
#define PN_B3_STREAM_FRAGS (120 + 192 + 768 + 48)
#define PN_B3_STREAM_CHUNKS (2 * ((PN_B3_STREAM_FRAGS + 2 * PN_B3_CHUNK - 1) / (2 * PN_B3_CHUNK)))
#define PN_B3_BLOB_FLOATS ((size_t)PN_B3_STREAM_CHUNKS * PN_B3_CHUNK * 256)

// ---- AE decoder: inv_pool 16->256->1024->k*128 (AE.py:19-26) + inv_mlp 144->128->64->32->3 (AE.py:27)
#define DEC_H_B1 0                             // [256]
#define DEC_H_B2 (DEC_H_B1 + 256)              // [1024]
#define DEC_H_W1 (DEC_H_B2 + 1024)             // KT=1,  MT=16
#define DEC_H_W2 (DEC_H_W1 + 1 * 16 * 256)     // KT=16, MT=64
#define DEC_M_B0 (DEC_H_W2 + 16 * 64 * 256)    // [128]
#define DEC_M_B1 (DEC_M_B0 + 128)              // [64]
#define DEC_M_B2 (DEC_M_B1 + 64)               // [32]
#define DEC_M_B3 (DEC_M_B2 + 32)               // [16] (3 used)
#define DEC_M_W0 (DEC_M_B3 + 16)               // KT=9 (128 inv_pool channels, then the 16 latents), MT=8
#define DEC_M_W1 (DEC_M_W0 + 9 * 8 * 256)      // KT=8, MT=4
#define DEC_M_W2 (DEC_M_W1 + 8 * 4 * 256)      // KT=4, MT=2
#define DEC_M_W3 (DEC_M_W2 + 4 * 2 * 256)      // KT=2, MT=1
#define DEC_G_B (DEC_M_W3 + 2 * 1 * 256)       // [k*128] bias of inv_pool.4, rows permuted to p*128+c
// followed by one weight stream per point p (what the workgroup of point p consumes, in order):
// 512 inv_pool.4 fragments (rows permuted to o' = p*128 + c, [kt=64][8 m-tiles]), then the 114 inv_mlp
// fragments (L0 [9][8], L1 [8][4], L2 [4][2], L3 [2][1]; the same for every p), padded to DEC_STREAM_CHUNKS chunks of DEC_WS_CHUNK.
#define DEC_STREAM_GEMM_FRAGS 512
#define DEC_STREAM_FRAGS 626
#ifndef DEC_WS_CHUNK
#define DEC_WS_CHUNK 8                    // the decoder's ring: one k-tile (8 m-tiles) per chunk
#endif
#define DEC_STREAM_CHUNKS (2 * ((DEC_STREAM_FRAGS + 2 * DEC_WS_CHUNK - 1) / (2 * DEC_WS_CHUNK)))
#define DEC_G_W(k) (DEC_G_B + (k) * 128)
#define DEC_BLOB_FLOATS(k) (DEC_G_W(k) + (size_t)(k) * DEC_STREAM_CHUNKS * DEC_WS_CHUNK * 256)

// ---- bf16x3 mode: decoder GEMM as fp32 products of three bf16 pieces per operand (decoder.hip, dec_main_kernel<true>).
// Per point p: [32 k-steps of 32][8 m-tiles][3 planes hi/mid/lo] bf16 A fragments of v_mfma_f32_16x16x32_bf16
// (1 KiB each: lane (m = lane%16, kg = lane/16) holds 8 bf16 of k-slots 8*kg + j <-> channel 32t + 16*(j>>2) + 4*kg + (j&3),
// the order in which two fp32 C tiles concatenate), then the inv_mlp layers in the same form (L0 [5][8][3] with the odd ninth
// k-tile zero-padded, L1 [4][4][3], L2 [2][2][3], L3 [1][1][3] = 183 fragments), padded to chunks of 12.
#define DEC_B3_CHUNK 12
#define DEC_B3_GEMM_FRAGS (32 * 8 * 3)
#define DEC_B3_TAIL_FRAGS (120 + 48 + 12 + 3)
#define DEC_B3_STREAM_FRAGS (DEC_B3_GEMM_FRAGS + DEC_B3_TAIL_FRAGS)
#define DEC_B3_STREAM_CHUNKS (2 * ((DEC_B3_STREAM_FRAGS + 2 * DEC_B3_CHUNK - 1) / (2 * DEC_B3_CHUNK)))
#define DEC_B3_BLOB_FLOATS(k) ((size_t)(k) * DEC_B3_STREAM_CHUNKS * DEC_B3_CHUNK * 256)


// ---- ConditionalProbabilityModel (AE.py:87-123): PointNet 3->64->128->256, MLP 259->512->512->d*L
#define PRB_P_B0 0                             // [64]
#define PRB_P_B1 (PRB_P_B0 + 64)               // [128]
#define PRB_P_B2 (PRB_P_B1 + 128)              // [256]
#define PRB_M_B0 (PRB_P_B2 + 256)              // [512]
#define PRB_M_B1 (PRB_M_B0 + 512)              // [512]
#define PRB_M_B2 (PRB_M_B1 + 512)              // [128] (d*L = 112 used)
#define PRB_P_W0 (PRB_M_B2 + 128)              // KT=1,  MT=4
#define PRB_P_W1 (PRB_P_W0 + 1 * 4 * 256)      // KT=4,  MT=8
#define PRB_P_W2 (PRB_P_W1 + 4 * 8 * 256)      // KT=8,  MT=16
#define PRB_M_W0 (PRB_P_W2 + 8 * 16 * 256)     // KT=17 (256 features, then xyz), MT=32
#define PRB_M_W1 (PRB_M_W0 + 17 * 32 * 256)    // KT=32, MT=32
#define PRB_M_W2 (PRB_M_W1 + 32 * 32 * 256)    // KT=32, MT=8
// model_mlp fragments again, in the order prob_forward_kernel consumes them (LDS ring, mfma_chain.h: WStreamT<8, 4>):
// W0 [17 kt][32 mt], then for each pair of layer-1 output tiles mp: W1 [32 kt][2mp..2mp+1] and W2 k-tiles 2mp, 2mp+1 [8 mt].
#define PRB_WS_CHUNK 8
#define PRB_STREAM (PRB_M_W2 + 32 * 8 * 256)
#define PRB_STREAM_FRAGS (1 * 32 + 16 * (32 * 2 + 2 * 8))   // layer 0's xyz k-tile, then layers 1 / 2 (its 16 feature k-tiles are read from PRB_M_W0 once per cloud)
#define PRB_STREAM_CHUNKS (4 * ((PRB_STREAM_FRAGS + 4 * PRB_WS_CHUNK - 1) / (4 * PRB_WS_CHUNK)))      // multiple of the ring depth
#define PRB_BLOB_FLOATS (PRB_STREAM + (size_t)PRB_STREAM_CHUNKS * PRB_WS_CHUNK * 256)


// ---- f16x2 mode (mfma_chain.h: two fp16 pieces per operand, three products; pack_h2.hip builds these blobs on the HOST from the
// state_dict tensors, with the power-of-two scales that keep every operand inside fp16's range).
// Encoder: [meta][biases, pre-multiplied by their layer's sigma * tau][SetAbstraction conv1 [1][4][2], conv2 [2][8][2] fragments]
// [PointNet stream: L0 [5][8][2], L1 [4][16][2], per half h of layer 2's outputs L2 [8][16][2] + L3 [8][1][2]], [kt32][mt][plane] each.
#define ENC_H2_META 0                          // [16] floats, indices H2E_*
#define H2E_RHO0 0                             // multiplier of the split in front of SetAbstraction conv1 (= sigma0)
#define H2E_RHO1 1                             // ... in front of conv2 (= sigma1 / (sigma0 tau1))
#define H2E_INV2 2                             // conv2 accumulator -> feature in patch-scaled units (= 1 / (sigma1 tau2))
#define H2E_RHO_IN 3                           // split of PointNet's input (= sigma_in)
#define H2E_RHO_P1 4                           // splits in front of PointNet layers 1, 2, 3
#define H2E_RHO_P2 5
#define H2E_RHO_P3 6
#define H2E_INV_OUT 7                          // last accumulator -> latent in patch-scaled units
#define ENC_H2_SA_B1 16                        // [64]   b1 * sigma0 * tau1
#define ENC_H2_SA_B2 (ENC_H2_SA_B1 + 64)       // [128]  b2 (unscaled: added after the neighbour max)
#define ENC_H2_PN_B0 (ENC_H2_SA_B2 + 128)      // [128]  b * sigma * tau of its layer, likewise below
#define ENC_H2_PN_B1 (ENC_H2_PN_B0 + 128)      // [256]
#define ENC_H2_PN_B2 (ENC_H2_PN_B1 + 256)      // [512]
#define ENC_H2_PN_B3 (ENC_H2_PN_B2 + 512)      // [16]
#define ENC_H2_PN_BIAS_FLOATS (128 + 256 + 512 + 16)
#define ENC_H2_SA_W (ENC_H2_PN_B3 + 16)        // 8 + 32 fragments
#define ENC_H2_SA_W1_FRAGS (1 * 4 * 2)
#define ENC_H2_SA_W2_FRAGS (2 * 8 * 2)
#define ENC_H2_PN_STREAM (ENC_H2_SA_W + (ENC_H2_SA_W1_FRAGS + ENC_H2_SA_W2_FRAGS) * 256)
#define PN_H2_CHUNK 32
#define PN_H2_STREAM_FRAGS (80 + 128 + 2 * (256 + 16))
#define PN_H2_STREAM_CHUNKS ((PN_H2_STREAM_FRAGS + PN_H2_CHUNK - 1) / PN_H2_CHUNK)
// the same PointNet weights again in the order of the two-tiles-per-wave form (encoder_fused_h2.hip, NT2): L0 [5][8][2] padded to whole
// chunks (it is streamed on its own, once per 128-point half), then L1 [4][16][2] and, for each slice e of 4 of layer 2's output tiles,
// L2 [8][4 tiles 4e..][2] followed by the two k-steps 2e, 2e+1 of layer 3 [2][1][2]
#define ENC_H2_PN_STREAM2 (ENC_H2_PN_STREAM + (size_t)PN_H2_STREAM_CHUNKS * PN_H2_CHUNK * 256)
#define PN_H2_S2_L0_FRAGS 96                   // 80 used
#define PN_H2_S2_MAIN_FRAGS (128 + 8 * (64 + 4))
#define ENC_H2_BLOB_FLOATS (ENC_H2_PN_STREAM2 + (size_t)(PN_H2_S2_L0_FRAGS + PN_H2_S2_MAIN_FRAGS) * 256)

// Decoder: [meta][biases][one weight stream per point p: GEMM [32 k-steps][8 m-tiles][2 planes], then inv_mlp L0 [5][8][2],
// L1 [4][4][2], L2 [2][2][2], L3 [1][1][2], padded to chunks of 8 (= 4 m-tiles x 2 planes: half a k-step)]
#define DEC_H2_META 0
#define H2D_SIG_H 0                            // scale of the (per-patch normalised) head activation planes
#define H2D_RHO0 1                             // split in front of inv_mlp layer 0 (GEMM accumulators)
#define H2D_SIG_Q 2                            // scale of the latent channels of that layer's input (= sigma of layer 0's input)
#define H2D_RHO1 3
#define H2D_RHO2 4
#define H2D_RHO3 5
#define H2D_INV_OUT 6
#define DEC_H2_M_B0 16                         // [128]
#define DEC_H2_M_B1 (DEC_H2_M_B0 + 128)        // [64]
#define DEC_H2_M_B2 (DEC_H2_M_B1 + 64)         // [32]
#define DEC_H2_M_B3 (DEC_H2_M_B2 + 32)         // [16]
#define DEC_H2_G_B (DEC_H2_M_B3 + 16)          // [k*128], rows permuted to p*128+c
#define DEC_H2_CHUNK 8
#define DEC_H2_GEMM_FRAGS (32 * 8 * 2)
#define DEC_H2_TAIL_FRAGS ((40 + 16 + 4 + 1) * 2)
#define DEC_H2_STREAM_FRAGS (DEC_H2_GEMM_FRAGS + 2 * DEC_H2_TAIL_FRAGS)     // the tail twice: inv_mlp runs once per pair of a wave's four patch tiles
#define DEC_H2_STREAM_CHUNKS (4 * ((DEC_H2_STREAM_FRAGS + 4 * DEC_H2_CHUNK - 1) / (4 * DEC_H2_CHUNK)))      // a multiple of the ring depth
#define DEC_H2_G_W(k) (DEC_H2_G_B + (size_t)(k) * 128)
#define DEC_H2_BLOB_FLOATS(k) (DEC_H2_G_W(k) + (size_t)(k) * DEC_H2_STREAM_CHUNKS * DEC_H2_CHUNK * 256)
