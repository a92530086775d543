"""ctypes binding of the C-ABI library ``libdva_hip.so`` (declared in ``include/dva.h``).

This is the only place Python touches native code.  There is deliberately NO fallback: if the
library is missing, or a tensor is not on a HIP device, the ops raise.  (The CPU oracle under
``oracle/`` is test infrastructure and is never imported from this package.)
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DVA_LIB_PATH: another build of the same sources (compile-time A/B variants, csrc/Makefile LIB= / EXTRA=) -- tools only
LIB_PATH = os.environ.get("DVA_LIB_PATH") or os.path.join(_HERE, "csrc", "libdva_hip.so")

DVA_F32, DVA_BF16 = 0, 1
DVA_SUM, DVA_MEAN, DVA_MAX, DVA_MIN = 0, 1, 2, 3
REDUCE_CODE = {"sum": DVA_SUM, "add": DVA_SUM, "mean": DVA_MEAN, "max": DVA_MAX, "min": DVA_MIN}
CAMERA_CODE = {
    "s3dis_equirectangular": 0,
    "scannet": 1,
    "kitti360_perspective": 2,
    "kitti360_fisheye": 3,
}

_ERRORS = {
    -1: "DVA_ERR_INVALID (bad argument)",
    -2: "DVA_ERR_UNSUPPORTED",
    -3: "DVA_ERR_LAUNCH (HIP runtime error)",
    -4: "DVA_ERR_OVERFLOW (composite key does not fit int64)",
}


class DvaError(RuntimeError):
    """A C-ABI entry returned a DVA_ERR_* code (``.code``; include/dva.h)."""

    def __init__(self, message, code=None):
        super().__init__(message)
        self.code = code


class DvaCamera(ctypes.Structure):
    """Mirror of ``struct dva_camera`` (include/dva.h)."""
    _fields_ = [
        ("model", ctypes.c_int32),
        ("img_w", ctypes.c_int32),
        ("img_h", ctypes.c_int32),
        ("crop_top", ctypes.c_int32),
        ("crop_bottom", ctypes.c_int32),
        ("r_min", ctypes.c_float),
        ("r_max", ctypes.c_float),
        ("img_xyz", ctypes.c_float * 3),
        ("rot", ctypes.c_float * 9),
        ("trans", ctypes.c_float * 3),
        ("fx", ctypes.c_float),
        ("fy", ctypes.c_float),
        ("mx", ctypes.c_float),
        ("my", ctypes.c_float),
        ("fisheye", ctypes.c_float * 7),
        ("r_min_d", ctypes.c_double),
        ("r_max_d", ctypes.c_double),
        ("voxel", ctypes.c_double),
        ("k_swell", ctypes.c_double),
        ("d_swell", ctypes.c_double),
        ("exact", ctypes.c_int32),
    ]


_vp, _i64, _i32, _f32, _f64 = (ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_float,
                               ctypes.c_double)

# name -> (restype, argtypes); every symbol include/dva.h declares must be listed here
SIGNATURES = {
    "dva_version": (ctypes.c_int, []),
    "dva_device_count": (ctypes.c_int, []),
    "dva_segment_csr_fwd": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    "dva_segment_csr_bwd": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    "dva_gather_csr": (ctypes.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "dva_segment_softmax_csr_fwd": (ctypes.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _f32, _vp]),
    "dva_segment_softmax_csr_bwd": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "dva_pack_gather_index": (ctypes.c_int, [_vp, _vp, _vp, _i32, _f64, _i64, _i64, _vp, _vp]),
    "dva_gather_nearest_fwd": (ctypes.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp]),
    "dva_gather_nearest_bwd": (ctypes.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp]),
    "dva_gather_bilinear_fwd": (ctypes.c_int,
                                [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp]),
    "dva_gather_bilinear_bwd": (ctypes.c_int,
                                [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp]),
    "dva_view_attention_fwd": (ctypes.c_int,
                               [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32,
                                _i32, _f32, _i32, _i32, _vp]),
    "dva_view_attention_bwd": (ctypes.c_int,
                               [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64,
                                _i64, _i32, _i32, _i32, _i32, _i32, _vp]),
    "dva_gather_row_index": (ctypes.c_int, [_vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "dva_view_gather_attention_fwd": (ctypes.c_int,
                                      [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64,
                                       _i32, _i32, _i32, _f32, _i32, _i32, _vp]),
    "dva_view_gather_attention_bwd": (ctypes.c_int,
                                      [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                       _vp, _vp, _i32, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _vp]),
    "dva_row_plan_workspace_bytes": (ctypes.c_int64, [_i64, _i64]),
    "dva_row_plan": (ctypes.c_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _vp]),
    "dva_view_gather_rows_grad": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i64, _i64,
                                                 _i32, _i32, _i32, _vp]),
    "dva_gather_rows_sum": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i32, _vp, _i64, _i64, _i32, _i32, _vp]),
    "dva_gather_bilinear_taps": (ctypes.c_int, [_vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp]),
    "dva_csr_expand": (ctypes.c_int, [_vp, _i64, _vp, _vp]),
    "dva_deepset_fwd_first": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp]),
    "dva_deepset_segmax": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp]),
    "dva_deepset_fwd_layer": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "dva_deepset_fwd_score": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _i32, _i32, _vp]),
    "dva_deepset_bwd_score": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _i32,
                                             _i32, _vp]),
    "dva_deepset_bwd_layer": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                             _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp]),
    "dva_deepset_bwd_max": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "dva_deepset_bwd_first": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "dva_rowbn_stats": (ctypes.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "dva_rowbn_apply": (ctypes.c_int, [_vp, _vp, _vp, _i64, _i32, _f32, _i32, _vp]),
    "dva_rowbn_bwd_stats": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _f32, _i32, _vp]),
    "dva_rowbn_bwd_apply": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _f32, _i32, _vp]),
    "dva_bn_finalize": (ctypes.c_int, [_vp, ctypes.c_double, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _i32, _i32, _vp,
                                       _vp]),
    "dva_scale_f64": (ctypes.c_int, [_vp, ctypes.c_double, _vp, _i32, _vp]),
    "dva_chain_prep": (ctypes.c_int, [_vp, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp]),
    "dva_chain_tile_count": (ctypes.c_int, [_vp, _vp, _i32, _vp, _vp]),
    "dva_chain_tile_build": (ctypes.c_int, [_vp, _vp, _i32, _vp, _vp, _vp]),
    "dva_chain_moments": (ctypes.c_int, [_vp, _i64, _vp, _i32, _vp, _vp, _vp]),
    "dva_chain_stats2": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "dva_chain_pooled": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _vp]),
    "dva_chain_stats": (ctypes.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp]),
    "dva_chain_stats_a2": (ctypes.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp]),
    "dva_chain_score_stats_a2": (ctypes.c_int, [_vp] * 12 + [_i32, _i64, _i64, _vp]),
    "dva_chain_bwd_layer6_a2": (ctypes.c_int, [_vp] * 13 + [_i32, _i64, _i64, _vp]),
    "dva_chain_attn_fwd": (ctypes.c_int, [_vp] * 18 + [_i64, _i64, _i64, _i32, _i32, _i32, _f32, _vp]),
    "dva_chain_attn_bwd": (ctypes.c_int, [_vp] * 14 + [_i64, _i64, _i64, _i32, _i32, _i32, _f32, _vp]),
    "dva_chain_attn_bwd_f32": (ctypes.c_int, [_vp] * 14 + [_i64, _i64, _i64, _i32, _i32, _i32, _f32, _vp]),
    "dva_gather_segment_max_fwd": (ctypes.c_int, [_vp] * 5 + [_i64, _i64, _i64, _i32, _i32, _vp]),
    "dva_gather_segment_max_bwd": (ctypes.c_int, [_vp] * 8 + [_i64, _i64, _i64, _i32, _i32, _vp]),
    "dva_chain_keys": (ctypes.c_int, [_vp] * 12 + [_i64, _i64, _vp]),
    "dva_chain_keys_compat": (ctypes.c_int, [_vp] * 14 + [_i32, _f32, _i64, _i64, _vp]),
    "dva_chain_attn_fwd_keys": (ctypes.c_int, [_vp] * 12 + [_f32] + [_vp] * 8 + [_i64, _i64, _i64, _i32, _i32, _i32, _f32, _vp]),
    "dva_qkv_dquery": (ctypes.c_int, [_vp, _i32, _vp, _vp, _vp, _i64, _i64, _i32, _f32, _vp]),
    "dva_chain_score_stats_keys": (ctypes.c_int, [_vp] * 15 + [_i32, _f32, _i64, _i64, _vp]),
    "dva_chain_bwd_layer6_keys": (ctypes.c_int, [_vp] * 16 + [_i32, _f32, _i64, _i64, _vp]),
    "dva_qkv_compat": (ctypes.c_int, [_vp] * 4 + [_i64, _i32, _f32, _vp]),
    "dva_qkv_compat_bwd": (ctypes.c_int, [_vp] * 7 + [_i64, _i64, _i32, _f32, _vp]),
    "dva_chain_attn_bwd_planrec": (ctypes.c_int, [_vp] * 15 + [_i64, _i64, _i64, _i32, _i32, _i32, _f32, _vp]),
    "dva_plan_inverse": (ctypes.c_int, [_vp, _vp, _i64, _vp]),
    "dva_plan_split_table_bytes": (ctypes.c_int64, [_i64, _i64]),
    "dva_plan_split_build": (ctypes.c_int, [_vp, _i64, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _vp]),
    "dva_plan_split_sort_records": (ctypes.c_int, [_vp, _vp, _i64, _i64, _vp, _vp, _i64, _vp, _vp, _vp]),
    "dva_plan_split_sort_records32": (ctypes.c_int, [_vp, _vp, _i64, _i64, _vp, _i64, _vp, _vp]),
    "dva_plan_split_rows_grad": (ctypes.c_int, [_vp, _vp, _i64, _i64, _vp, _i64, _vp, _i32, _i32, _i32, _i32, _vp]),
    "dva_chain_score_stats": (ctypes.c_int, [_vp] * 14 + [_i32, _i64, _i64, _vp]),
    "dva_chain_bwd_layer": (ctypes.c_int, [_i32] + [_vp] * 22 + [_i32, _i64, _i64, _vp]),
    "dva_gather_bilinear_taps_anchor": (ctypes.c_int, [_vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "dva_bilinear_taps_cat": (ctypes.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "dva_anchor_rows_sum": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp]),
    "dva_anchor_combine": (ctypes.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "dva_anchor_fixup": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp]),
    "dva_anchor_rows_sum_bn": (ctypes.c_int, [_vp] * 10 + [_i64, _i64, _i32, _vp]),
    "dva_anchor_fixup_bn": (ctypes.c_int, [_vp] * 8 + [_i64, _i32, _i32, _i32, _i32, _vp]),
    "dva_emod_prep": (ctypes.c_int, [_vp, _i32, _vp, _vp]),
    "dva_emod_stats": (ctypes.c_int, [_i32] + [_vp] * 9 + [_i64, _i64, _i32, _vp]),
    "dva_emod_attn_fwd": (ctypes.c_int, [_vp] * 23 + [_i64, _i64, _i64, _i32, _i32, _i32, _f32, _vp]),
    "dva_emod_attn_bwd": (ctypes.c_int, [_vp] * 20 + [_i64, _i64, _i64, _i32, _i32, _i32, _f32, _vp]),
    "dva_emod_stats1_plan": (ctypes.c_int, [_vp] * 6 + [_i64, _i64, _i32, _vp]),
    "dva_emod_bwd": (ctypes.c_int, [_i32] + [_vp] * 16 + [_i64, _i64, _i64, _i32, _i32, _vp]),
    "dva_chain_score_l6_stats": (ctypes.c_int, [_vp] * 16 + [_i32, _i64, _i64, _vp]),
    "dva_chain_l6_consts": (ctypes.c_int, [_vp] * 7),
    "dva_chain_bwd_layer5_merged": (ctypes.c_int, [_vp] * 18 + [_i32, _i64, _i64, _vp]),
    "dva_chain_route_stats": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "dva_copy_ceiling": (ctypes.c_int, [_vp, _vp, _i64, _vp]),
    "dva_zero_unseen_rows": (ctypes.c_int, [_vp, _vp, _i64, _i64, _vp]),
    "dva_chain_bn_consts": (ctypes.c_int, [_vp, ctypes.c_double, _vp, _vp, _vp, _vp, _vp, ctypes.c_float, ctypes.c_float,
                                           _i32, _vp, _i32, _i32, _vp, _vp, _vp]),
    "dva_concat_cast_fwd": (ctypes.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    "dva_concat_cast_bwd": (ctypes.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    "dva_mapping_row_index": (ctypes.c_int, [_vp, _vp, _vp, _i32, ctypes.c_double, _i64, _i64, _i32, _i32, _i32, _vp, _vp]),
    "dva_view_gather_rows_grad_rec16": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _vp]),
    "dva_view_gather_rows_grad_rec16_to": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _i64, _i32, _i32, _i32, _vp]),
    "dva_chain_tile_chunks": (ctypes.c_int, [_vp, _i64, _i64, _i32, _vp, _vp]),
    "dva_chain_tile_offsets": (ctypes.c_int, [_vp, _i32, _vp, _vp, _vp]),
    "dva_bn_bwd_consts": (ctypes.c_int, [_vp, _vp, ctypes.c_double, _i32, _vp, _vp, _vp, _i32, _vp]),
    "dva_chain3_prep": (ctypes.c_int, [_vp, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp]),
    "dva_chain3_stats2": (ctypes.c_int, [_vp] * 11 + [_i64, _vp]),
    "dva_chain3_stats": (ctypes.c_int, [_i32] + [_vp] * 9 + [_i64, _i64, _vp]),
    "dva_chain3_scores": (ctypes.c_int, [_vp] * 7 + [_i32, _vp, _i64, _vp]),
    "dva_chain3_score_stats": (ctypes.c_int, [_vp] * 10 + [_i32, _i64, _vp]),
    "dva_chain3_bwd_layer": (ctypes.c_int, [_i32] + [_vp] * 19 + [_i64, _i64, _vp]),
    "dva_chain_stats1": (ctypes.c_int, [_vp, _vp, _i32, _vp, _vp]),
    "dva_chain_dw1": (ctypes.c_int, [_vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "dva_chain_set_prep": (ctypes.c_int, [_vp, _i32, _vp, _vp, _i32, _vp, _vp]),
    "dva_chain_set_fwd": (ctypes.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "dva_chain_set_bwd": (ctypes.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp,
                                         _i64, _vp]),
    "dva_chain3_set_prep": (ctypes.c_int, [_vp, _i32, _vp, _vp, _i32, _vp, _vp]),
    "dva_chain3_set_fwd": (ctypes.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "dva_chain3_set_bwd": (ctypes.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp,
                                         _i64, _vp]),
    "dva_voxel_parent_workspace_bytes": (ctypes.c_int64, [_i64]),
    "dva_voxel_parent_index": (ctypes.c_int, [_vp, _i64, _vp, _i64, _i32, _i32, _vp, _vp, _i64, _vp]),
    "dva_voxel_kernel_map": (ctypes.c_int, [_vp, _i64, _vp, _i64, _vp, _i32, _vp, _vp, _i64, _vp]),
    "dva_sparse_conv_workspace_bytes": (ctypes.c_int64, [_i32, _i32, _i32, _i32]),
    "dva_sparse_conv_apply": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _vp,
                                             _i64, _vp]),
    "dva_sparse_conv_wgrad": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp]),
    "dva_knn_workspace_bytes": (ctypes.c_int64, [_i64]),
    "dva_knn": (ctypes.c_int, [_vp, _i64, _vp, _f32, _i32, _i32, _vp, _vp, _vp, _vp, _i64, _vp]),
    "dva_view_occlusion": (ctypes.c_int, [_vp, _vp, _i64, _i64, _i32, _vp, _i32, _vp, _i32, _vp, _vp, _vp]),
    "dva_lex_workspace_bytes": (ctypes.c_int64, [_i64]),
    "dva_argsort_i64": (ctypes.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _vp]),
    "dva_argunique_i64": (ctypes.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _vp]),
    "dva_visibility_workspace_bytes": (ctypes.c_int64, [ctypes.POINTER(DvaCamera), _i64]),
    "dva_visibility": (ctypes.c_int,
                       [_vp, _i64, ctypes.POINTER(DvaCamera), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                        _vp, _i64, _vp]),
    "dva_camera_projection": (ctypes.c_int,
                              [_vp, _i64, ctypes.POINTER(DvaCamera), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "dva_visibility_batch_workspace_bytes": (ctypes.c_int64, [ctypes.POINTER(DvaCamera), _i64, _i32]),
    "dva_visibility_batch": (ctypes.c_int,
                             [_vp, _i64, ctypes.POINTER(DvaCamera), _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                              _vp, _vp, _i64, _vp]),
    "dva_mapping_features_batch": (ctypes.c_int,
                                   [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _vp, _vp,
                                    ctypes.POINTER(ctypes.c_int32), _vp]),
    "dva_mapping_features": (ctypes.c_int,
                             [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.POINTER(DvaCamera),
                              _i64, _vp, ctypes.POINTER(ctypes.c_int32), _vp]),
}

_lib = None


def load():
    """Load libdva_hip.so (once) and attach the prototypes. Raises DvaError if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DvaError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (or `make -C deepviewagg_amd/csrc`). deepviewagg_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here == header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def source_sha256():
    """sha256 over the kernel sources the library is built from (csrc/*.hip, csrc/*.h, include/dva.h; names + bytes,
    sorted): what ties a measurement file (profiles/pmc_traffic_latest.json) to the build it was taken on."""
    import glob
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(_HERE, "csrc")
    files = sorted(glob.glob(os.path.join(src, "*.hip")) + glob.glob(os.path.join(src, "*.h")))
    files.append(os.path.join(os.path.dirname(_HERE), "include", "dva.h"))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def check(rc, what):
    if rc != 0:
        raise DvaError(f"{what} failed: {_ERRORS.get(rc, rc)}", code=rc)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream_of(t):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def dtype_code(t):
    import torch
    if t.dtype == torch.float32:
        return DVA_F32
    if t.dtype == torch.bfloat16:
        return DVA_BF16
    raise TypeError(f"deepviewagg_amd kernels take float32 or bfloat16 features, got {t.dtype}")


def require_device(*tensors):
    """The kernels only exist for HIP devices; refuse anything else loudly."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise DvaError(
                "deepviewagg_amd ops run on a HIP device only (got a CPU tensor); "
                "there is no CPU fallback in the product path")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise DvaError(f"tensors on different devices: {dev} vs {t.device}")
    return dev
