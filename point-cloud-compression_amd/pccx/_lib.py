"""ctypes loader for libpccx.so (the C ABI declared in include/pccx.h).

There is NO CPU fallback: if the library is missing or a call fails, an exception is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PCCX_LIB: another build of the same ABI (a sanitizer or experiment build).  PCCX_LIB_PARTIAL=1 accepts a library that exports only
# part of the ABI -- the host-only sanitizer build of the packers (oracle/Makefile `asan`); calling an entry point it lacks raises.
LIB_PATH = os.environ.get("PCCX_LIB") or os.path.join(_HERE, "lib", "libpccx.so")

c_f32p = C.c_void_p   # device pointers travel as integers (tensor.data_ptr())
_P = C.c_void_p

# name -> argtypes; every function returns int (pccx_status) unless listed in _RESTYPES
_SIGNATURES = {
    "pccx_version": [],
    "pccx_normalize": [_P, C.c_int, C.c_int, C.c_double, _P, _P, _P, _P],
    "pccx_denormalize": [_P, C.c_int, C.c_int, C.c_double, _P, _P, _P, _P],
    "pccx_fps": [_P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P],
    "pccx_morton_keys": [_P, C.c_int64, _P, C.c_float, _P, _P],
    "pccx_morton_keys_auto": [_P, C.c_int64, _P, _P, _P],
    "pccx_sort_keys_workspace_bytes": [C.c_int64],
    "pccx_sort_keys_u64": [_P, C.c_int64, C.c_int, _P, _P, _P],
    "pccx_gather_blocks": [_P, _P, C.c_int64, C.c_int, C.c_int64, C.c_int64, C.c_int64, _P, _P],
    "pccx_scatter_blocks": [_P, _P, C.c_int64, C.c_int, C.c_int64, C.c_int64, C.c_int64, _P, _P],
    "pccx_gather": [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, _P],
    "pccx_knn": [_P, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P, _P, _P, C.c_float, _P],
    "pccx_ball_query": [_P, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_float, _P, _P, _P],
    "pccx_ball_query_grid_workspace_ints": [C.c_int, C.c_int],
    "pccx_ball_query_grid": [_P, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_float, _P, _P, _P, _P],
    "pccx_nn_dist": [_P, C.c_int, C.c_int, _P, C.c_int, _P, _P, _P],
    "pccx_nn_dist_split_count": [C.c_int, C.c_int, C.c_int],
    "pccx_nn_dist_split": [_P, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P, _P, _P, _P, _P],
    "pccx_chamfer_grad": [_P, C.c_int, C.c_int, _P, C.c_int, _P, _P, C.c_float, _P, _P, _P],
    "pccx_chamfer_mean": [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P],
    "pccx_chamfer_grad_dev": [_P, C.c_int, C.c_int, _P, C.c_int, _P, _P, _P, _P, _P, _P],
    "pccx_estimate_normals": [_P, C.c_int, C.c_int, _P, C.c_int, _P, _P],
    "pccx_point_plane_err": [_P, C.c_int, C.c_int, _P, _P, C.c_int, _P, _P, _P],
    "pccx_octree_bits_capacity": [C.c_int],
    "pccx_octree_encode": [_P, C.c_int, C.c_int, C.c_int, C.c_double, _P, _P, _P, _P, _P, _P],
    "pccx_octree_decode": [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P],
    "pccx_ae_encoder_blob_floats": [],
    "pccx_pack_ae_encoder": [_P] * 14 + [C.c_int, _P],
    "pccx_ae_decoder_blob_floats": [C.c_int],
    "pccx_pack_ae_decoder": [_P] * 14 + [C.c_int, C.c_int, _P],
    "pccx_prob_blob_floats": [],
    "pccx_pack_prob": [_P] * 12 + [C.c_int, C.c_int, _P],
    "pccx_ae_encode": [_P, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P, _P, _P, _P, _P],
    "pccx_sa_forward": [_P, C.c_int, C.c_int, _P, _P, _P],
    "pccx_pn_forward": [_P, _P, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P, _P, _P, _P],
    "pccx_ae_decode_workspace_floats": [C.c_int],
    "pccx_ae_decode": [_P, C.c_int, C.c_int, C.c_int, _P, _P, _P, C.c_float, _P, _P, _P, C.c_int, C.c_double, _P, _P],
    "pccx_sa_b3_blob_floats": [],
    "pccx_pack_sa_b3": [_P, _P, _P],
    "pccx_sa_forward_b3": [_P, C.c_int, C.c_int, _P, _P, _P, _P],
    "pccx_pn_b3_blob_floats": [],
    "pccx_pack_pn_b3": [_P, _P, _P],
    "pccx_pn_forward_b3": [_P, _P, C.c_int, C.c_int, _P, _P, C.c_int, C.c_int, _P, _P, _P, _P],
    "pccx_ae_encode_b3_fused_ok": [C.c_int],
    "pccx_ae_encode_b3": [_P, C.c_int, C.c_int, _P, _P, _P, C.c_int, C.c_int, _P, _P, _P, _P],
    "pccx_patch_knn16_index_bytes": [C.c_int],
    "pccx_patch_knn16_bytes": [C.c_int, C.c_int],
    "pccx_patch_knn16": [_P, C.c_int, C.c_int, _P, _P],
    "pccx_ae_encode_b3_workspace_bytes": [C.c_int, C.c_int],
    "pccx_ae_encode_b3_ws": [_P, C.c_int, C.c_int, _P, _P, _P, C.c_int, C.c_int, _P, _P, _P, _P, _P],
    "pccx_ae_encode_b3_tables": [_P, C.c_int, C.c_int, _P, _P, _P, C.c_int, C.c_int, _P, _P, _P, _P, _P],
    "pccx_dec_b3_blob_floats": [C.c_int],
    "pccx_pack_ae_decoder_b3": [_P, C.c_int, _P, _P],
    "pccx_ae_decode_b3_workspace_floats": [C.c_int],
    "pccx_ae_decode_b3": [_P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, C.c_float, _P, _P, _P, C.c_int, C.c_double, _P, _P],
    "pccx_ae_encoder_h2_blob_floats": [],
    "pccx_pack_ae_encoder_h2": [_P] * 14 + [C.c_int, _P],
    "pccx_ae_encode_h2_fused_ok": [C.c_int],
    "pccx_ae_encode_h2_workspace_bytes": [C.c_int, C.c_int],
    "pccx_ae_encode_h2_ws": [_P, C.c_int, C.c_int, _P, _P, C.c_int, C.c_int, _P, _P, _P, _P, _P],
    "pccx_ae_encode_h2_tables": [_P, C.c_int, C.c_int, _P, _P, C.c_int, C.c_int, _P, _P, _P, _P, _P],
    "pccx_ae_decoder_h2_blob_floats": [C.c_int],
    "pccx_pack_ae_decoder_h2": [_P] * 14 + [C.c_int, C.c_int, _P],
    "pccx_ae_decode_h2_workspace_floats": [C.c_int],
    "pccx_ae_decode_h2": [_P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, C.c_float, _P, _P, _P, C.c_int, C.c_double, _P, _P],
    "pccx_prob_forward": [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P],
    "pccx_range_encode": [_P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, _P],
    "pccx_range_decode": [_P, _P, C.c_int, _P, C.c_int, C.c_int, C.c_int, _P, _P],
    "pccx_cdf_float_to_int": [_P, C.c_int64, C.c_int, _P, _P],
    "pccx_streams_packed_bytes": [C.c_int, C.c_int, C.c_int],
    "pccx_write_streams_host": [_P, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p, _P, C.c_int],
    "pccx_read_streams_host": [_P, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p, _P, C.c_int],
    "pccx_stream_sizes_host": [C.c_int, C.c_char_p, C.c_char_p, _P, _P, _P, C.c_int],
    "pccx_packed_linear_floats": [C.c_int, C.c_int],
    "pccx_pack_linear": [_P, C.c_int, C.c_int, _P],
    "pccx_linear": [_P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, C.c_int, _P, C.c_int, _P],
    "pccx_packed_linear_b3_floats": [C.c_int, C.c_int],
    "pccx_pack_linear_b3": [_P, C.c_int, C.c_int, _P, _P],
    "pccx_linear_b3": [_P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, C.c_int, _P, C.c_int, _P],
    "pccx_softmax_cdf": [_P, C.c_int64, C.c_int, _P, _P, _P, _P],
    "pccx_reassemble": [_P, C.c_int64, C.c_int, C.c_float, _P, _P, _P, C.c_int, C.c_double, _P, _P],
    "pccx_rows_affine_small": [_P, C.c_int, C.c_int64, _P, C.c_int, C.c_int, C.c_int64, _P, C.c_int, C.c_int64, _P, _P],
    "pccx_rows_affine_planes": [_P, C.c_int, C.c_int64, _P, C.c_int, C.c_int, C.c_int64, _P, C.c_int, C.c_int64, _P, _P],
    "pccx_gather_max": [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P, _P],
    "pccx_gather_max_rows": [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P, _P, C.c_int, _P],
    "pccx_group_max": [_P, C.c_int64, C.c_int, C.c_int, _P, _P],
    "pccx_planes_floats": [C.c_int64, C.c_int],
    "pccx_group_planes": [_P, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P, C.c_int64, C.c_int64, C.c_int64, _P, _P],
    "pccx_fold_planes": [_P, C.c_int, C.c_int, C.c_int64, _P, C.c_int, C.c_int, C.c_int64, C.c_int64, _P, _P],
    "pccx_planes_gemm_weight_floats": [C.c_int, C.c_int],
    "pccx_pack_planes_gemm": [_P, C.c_int, C.c_int, _P, _P],
    "pccx_planes_gemm": [_P, C.c_int64, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P],
    "pccx_planes_gemm_gather": [_P, C.c_int, _P, C.c_int64, C.c_int64, C.c_int64, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int,
                                _P],
    "pccx_planes_chain_wide_weight_floats": [C.c_int],
    "pccx_pack_planes_chain_wide": [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P],
    "pccx_planes_chain_wide": [_P, C.c_int, _P, C.c_int64, C.c_int64, C.c_int64, C.c_int, _P, _P, C.c_int, _P, C.c_int, _P, C.c_int, _P, _P],
    "pccx_planes_chain4": [_P, C.c_int64, C.c_int, _P, _P, C.c_int, _P, C.c_int, _P, C.c_int, _P, C.c_int, C.c_int, _P, C.c_int, _P],
    "pccx_planes_chain4_gather": [_P, C.c_int, _P, C.c_int64, C.c_int64, C.c_int64, C.c_int, _P, _P, C.c_int, _P, C.c_int, _P, C.c_int, _P, C.c_int,
                                  C.c_int, _P, C.c_int, _P],
    "pccx_planes_floats_h2": [C.c_int64, C.c_int],
    "pccx_packed_linear_h2_floats": [C.c_int, C.c_int],
    "pccx_pack_linear_h2": [_P, C.c_int, C.c_int, C.c_float, _P, _P],
    "pccx_planes_gemm_weight_floats_h2": [C.c_int, C.c_int],
    "pccx_pack_planes_gemm_h2": [_P, C.c_int, C.c_int, _P, _P],
    "pccx_group_planes_h2": [_P, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P, C.c_int64, C.c_int64, C.c_int64, C.c_float, _P, _P, _P],
    "pccx_fold_planes_h2": [_P, C.c_int, C.c_int, C.c_int64, _P, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_float, _P, _P, _P],
    "pccx_rows_affine_planes_h2": [_P, C.c_int, C.c_int64, _P, C.c_int, C.c_int, C.c_int64, _P, C.c_int, C.c_int64, C.c_float, _P, _P, _P],
    "pccx_planes_gemm_h2": [_P, C.c_int64, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _P, _P, _P, C.c_int, _P],
    "pccx_planes_gemm_h2_member_max": [_P, C.c_int64, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_float, _P, _P, C.c_int, _P],
    "pccx_group_members": [_P, C.c_int64, C.c_int64, C.c_int64, _P, _P],
    "pccx_planes_gemm_gather_h2": [_P, C.c_int, _P, C.c_int64, C.c_int64, C.c_int64, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_float, C.c_float, _P, _P, _P, C.c_int, _P],
    "pccx_planes_chain4_h2": [_P, C.c_int64, C.c_int, _P, _P, C.c_int, _P, C.c_int, _P, C.c_int, _P, C.c_int, C.c_int, _P, _P, _P, _P, C.c_int, _P],
    "pccx_planes_chain4_gather_h2": [_P, C.c_int, _P, C.c_int64, C.c_int64, C.c_int64, C.c_int, _P, _P, C.c_int, _P, C.c_int, _P, C.c_int, _P, C.c_int,
                                     C.c_int, _P, _P, _P, _P, C.c_int, _P],
    "pccx_absmax": [_P, C.c_int64, _P, _P],
    "pccx_dyn_scale": [_P, C.c_float, _P, C.c_float, C.c_float, C.c_int, _P, _P],
    "pccx_sigmoid_spread": [_P, C.c_int64, C.c_int, C.c_int, _P, _P],
    "pccx_round": [_P, C.c_int64, _P, _P],
    "pccx_pack_linear_device": [_P, C.c_int, C.c_int, C.c_int, _P, _P],
    "pccx_linear_skinny": [_P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, C.c_int, _P, C.c_int, _P],
    "pccx_linear_skinny_dx": [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P, C.c_int, _P],
    "pccx_linear_dw": [_P, _P, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P],
    "pccx_bn_train_stats": [_P, C.c_int64, C.c_int, C.c_float, C.c_float, _P, _P, _P, _P, _P, _P],
    "pccx_bn_relu_forward": [_P, C.c_int64, C.c_int, _P, _P, _P, _P, C.c_int, _P, _P],
    "pccx_bn_relu_backward": [_P, _P, _P, C.c_int64, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P],
    "pccx_col_sum": [_P, C.c_int64, C.c_int, _P, _P, _P],
    "pccx_bn_relu_train_forward": [_P, C.c_int64, C.c_int, C.c_float, C.c_float, _P, _P, _P, C.c_int, _P, _P, _P, _P, _P, C.c_int, _P],
    "pccx_bn_relu_train_backward": [_P, _P, _P, C.c_int64, C.c_int, _P, _P, _P, _P, _P, _P, _P, C.c_int, _P],
    "pccx_col_sum_w": [_P, C.c_int64, C.c_int, _P, _P, C.c_int, _P],
    "pccx_train_sums_doubles": [C.c_int],
    "pccx_linear_moments": [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P, C.c_int, _P, _P],
    "pccx_linear_bnback": [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P, C.c_int, _P, _P, _P, _P, _P, _P],
    "pccx_zero_bytes": [_P, C.c_size_t, _P],
    "pccx_copy_bytes": [_P, _P, C.c_size_t, _P],
    "pccx_add_i64_table": [_P, C.c_int, C.c_int64, _P],
    "pccx_gather_backward_acc": [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P],
    "pccx_chamfer_grad_dev_acc": [_P, C.c_int, C.c_int, _P, C.c_int, _P, _P, _P, _P, _P, C.c_int, _P],
    "pccx_relu_backward": [_P, _P, C.c_int64, _P, _P],
    "pccx_group_max_arg": [_P, C.c_int64, C.c_int, C.c_int, _P, _P, _P],
    "pccx_group_max_backward": [_P, _P, C.c_int64, C.c_int, C.c_int, _P, _P],
    "pccx_gather_backward": [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P],
    "pccx_smooth_l1": [_P, _P, C.c_int64, C.c_float, _P, _P, _P],
    "pccx_quantize_st_backward": [_P, _P, C.c_int64, C.c_float, C.c_float, C.c_int, _P, _P],
    "pccx_rate_from_logits": [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P],
    "pccx_sumsq_accumulate": [_P, C.c_int64, _P, _P],
    "pccx_adam_step": [_P, _P, _P, _P, C.c_int64, _P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, _P],
    "pccx_sumsq_multi": [_P, C.c_int, C.c_int64, _P, _P],
    "pccx_adam_multi": [_P, C.c_int, C.c_int64, _P, C.c_float, _P, C.c_float, C.c_int, C.c_float, C.c_float, C.c_float, _P],
    "pccx_adam_advance_dev": [_P, C.c_double, C.c_double, _P],
    "pccx_adam_step_dev": [_P, _P, _P, _P, C.c_int64, _P, C.c_float, _P, C.c_float, C.c_float, C.c_float, _P],
    "pccx_quantize_st": [_P, C.c_int64, C.c_float, C.c_float, C.c_int, _P, _P, _P],
}
_RESTYPES = {"pccx_planes_floats_h2": C.c_size_t, "pccx_packed_linear_h2_floats": C.c_size_t, "pccx_planes_gemm_weight_floats_h2": C.c_size_t,
             "pccx_streams_packed_bytes": C.c_size_t, "pccx_sort_keys_workspace_bytes": C.c_size_t, "pccx_train_sums_doubles": C.c_size_t, "pccx_ae_encoder_h2_blob_floats": C.c_size_t, "pccx_ae_decoder_h2_blob_floats": C.c_size_t,
             "pccx_ae_encode_h2_workspace_bytes": C.c_size_t, "pccx_ae_decode_h2_workspace_floats": C.c_size_t,
             "pccx_patch_knn16_bytes": C.c_size_t, "pccx_ae_encode_b3_workspace_bytes": C.c_size_t, "pccx_ae_encoder_blob_floats": C.c_size_t, "pccx_ae_decoder_blob_floats": C.c_size_t,
             "pccx_prob_blob_floats": C.c_size_t, "pccx_ae_decode_workspace_floats": C.c_size_t,
             "pccx_packed_linear_floats": C.c_size_t, "pccx_ball_query_grid_workspace_ints": C.c_size_t, "pccx_packed_linear_b3_floats": C.c_size_t, "pccx_dec_b3_blob_floats": C.c_size_t, "pccx_sa_b3_blob_floats": C.c_size_t, "pccx_pn_b3_blob_floats": C.c_size_t,
             "pccx_ae_decode_b3_workspace_floats": C.c_size_t, "pccx_planes_floats": C.c_size_t,
             "pccx_planes_gemm_weight_floats": C.c_size_t, "pccx_planes_chain_wide_weight_floats": C.c_size_t}

_lib = None


class PccxError(RuntimeError):
    pass


def load():
    """Load libpccx.so; raises PccxError when it has not been built (python -m pccx.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PccxError(
                f"{LIB_PATH} not found: build the HIP library first (python -m pccx.build). "
                "pccx has no CPU fallback.")
        # torch first: its bundled HIP runtime must be the one (and only) libamdhip64 in the process,
        # so that libpccx's kernels and torch's allocator / streams share devices and contexts.
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        lib.pccx_last_error.restype = C.c_char_p
        lib.pccx_last_error.argtypes = []
        partial = os.environ.get("PCCX_LIB_PARTIAL") == "1"
        for name, args in _SIGNATURES.items():
            if partial and not hasattr(lib, name):
                continue
            fn = getattr(lib, name)   # AttributeError if the ABI and the header disagree
            fn.argtypes = args
            fn.restype = _RESTYPES.get(name, C.c_int)
        _lib = lib
    return _lib


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise PccxError(f"{name} failed ({rc}): {lib.pccx_last_error().decode()}")


def declared_symbols():
    return ["pccx_last_error"] + list(_SIGNATURES)
