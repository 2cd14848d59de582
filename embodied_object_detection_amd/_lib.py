"""ctypes binding of libeod_hip.so (C ABI declared in include/eod_hip.h).

Fails loudly: there is no CPU fallback.  If the shared library is missing or a call returns a negative
EOD_ERR_* code, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EOD_LIB_PATH") or os.path.join(HERE, "libeod_hip.so")     # EOD_LIB_PATH: A/B of two builds (tools/)

c_f32p = C.c_void_p
c_i32p = C.c_void_p
c_void_p = C.c_void_p


class EodError(RuntimeError):
    pass


_ERR = {-1: "EOD_ERR_BAD_DIMS", -2: "EOD_ERR_ALIGN", -3: "EOD_ERR_LAUNCH", -4: "EOD_ERR_NULL", -5: "EOD_ERR_CAPACITY"}


MAX_BATCH = 8       # EOD_MAX_BATCH
MAX_LEVELS = 40     # EOD_MAX_LEVELS


class EodConvDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("res", C.c_void_p), ("y", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("m_count", C.c_void_p), ("m_unit", C.c_int32),
        ("m_segments", C.c_int32),
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32), ("OH", C.c_int32), ("OW", C.c_int32),
        ("Cout", C.c_int32), ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("Kpad", C.c_int32), ("relu", C.c_int32), ("res_mode", C.c_int32), ("in_relu", C.c_int32),
        ("out_mode", C.c_int32), ("tap4", C.c_int32), ("force_tile", C.c_int32), ("force_splitk", C.c_int32),
        ("out_scale", C.c_float), ("levels", C.c_int32), ("level_off", C.c_int32 * (MAX_LEVELS + 1)), ("level_h", C.c_int32 * MAX_LEVELS),
        ("level_w", C.c_int32 * MAX_LEVELS), ("fuse_w", C.c_void_p), ("out_units", C.c_void_p), ("fuse_b", C.c_float), ("w_split", C.c_void_p),
        ("plan_rows", C.c_int32), ("lds_reserve", C.c_int32), ("gn_partial", C.c_void_p), ("gn_groups", C.c_int32),
        ("y2", C.c_void_p), ("split_n", C.c_int32), ("prefetch2", C.c_int32), ("gate", C.c_void_p),
    ]


class EodBoxRefine(C.Structure):
    _fields_ = [("deltas", C.c_void_p), ("ld", C.c_int32), ("wx", C.c_float), ("wy", C.c_float), ("ww", C.c_float), ("wh", C.c_float),
                ("clip", C.c_int32), ("img_w", C.c_float), ("img_h", C.c_float), ("boxes_out", C.c_void_p)]


class EodCenterNetLossDesc(C.Structure):
    _fields_ = [
        ("head_out", C.c_void_p), ("head_stride", C.c_int32), ("P", C.c_int32), ("levels", C.c_int32), ("level_off", C.c_int32 * 9),
        ("level_scale", C.c_float * 8), ("agn_heatmap", C.c_void_p), ("reg_targets", C.c_void_p), ("pos_inds", C.c_void_p),
        ("n_pos", C.c_int32), ("hm_focal_alpha", C.c_float), ("hm_focal_beta", C.c_float), ("loss_gamma", C.c_float),
        ("sigmoid_clamp", C.c_float), ("ignore_high_fp", C.c_float), ("pos_weight", C.c_float), ("neg_weight", C.c_float),
        ("reg_weight", C.c_float), ("num_pos_avg", C.c_float), ("reg_norm", C.c_float), ("counts_local", C.c_void_p),
        ("counts_total", C.c_void_p), ("world_size", C.c_float), ("d_head_out", C.c_void_p),
        ("losses", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
    ]


class EodCenterNetTargetDesc(C.Structure):
    _fields_ = [
        ("gt_boxes", C.c_void_p), ("n_boxes", C.c_int32), ("levels", C.c_int32), ("level_off", C.c_int32 * 9), ("level_w", C.c_int32 * 8),
        ("level_stride", C.c_int32 * 8), ("soi_lo", C.c_float * 8), ("soi_hi", C.c_float * 8), ("hm_min_overlap", C.c_double),
        ("min_radius", C.c_double), ("agn_heatmap", C.c_void_p), ("reg_targets", C.c_void_p), ("pos_inds", C.c_void_p),
        ("counts", C.c_void_p),
    ]


class EodProposalDesc(C.Structure):
    _fields_ = [
        ("head_out", C.c_void_p), ("head_stride", C.c_int32), ("levels", C.c_int32), ("level_off", C.c_int32 * 6),
        ("level_w", C.c_int32 * 5), ("level_stride", C.c_int32 * 5), ("level_scale", C.c_float * 5),
        ("score_thresh", C.c_float), ("pre_nms_topk", C.c_int32), ("post_nms_topk", C.c_int32), ("nms_thresh", C.c_float),
        ("cap", C.c_int32), ("out_boxes", C.c_void_p), ("out_scores", C.c_void_p), ("out_count", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("batch", C.c_int32),
    ]


class EodDetDesc(C.Structure):
    _fields_ = [
        ("boxes", C.c_void_p), ("scores", C.c_void_p), ("count", C.c_void_p), ("R_cap", C.c_int32), ("C1", C.c_int32),
        ("img_w", C.c_float), ("img_h", C.c_float), ("score_thresh", C.c_float), ("nms_thresh", C.c_float),
        ("topk", C.c_int32), ("out_boxes", C.c_void_p), ("out_scores", C.c_void_p), ("out_classes", C.c_void_p),
        ("out_rows", C.c_void_p), ("out_count", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("out_unique_rows", C.c_void_p), ("out_unique_count", C.c_void_p), ("unique_cap", C.c_int32),
        ("out_rep_of", C.c_void_p), ("out_rep_list", C.c_void_p), ("out_rep_count", C.c_void_p), ("batch", C.c_int32),
    ]


class EodStageTailDesc(C.Structure):
    _fields_ = [("feat", C.c_void_p), ("zs", C.c_void_p), ("prob_acc", C.c_void_p), ("accumulate", C.c_int32), ("feat_norm_out", C.c_void_p),
                ("count", C.c_void_p), ("R_cap", C.c_int32), ("D", C.c_int32), ("C1", C.c_int32), ("temp", C.c_float), ("zs_mem", C.c_void_p),
                ("prop_scores", C.c_void_p), ("mem_scores_out", C.c_void_p), ("final_inv_stages", C.c_float), ("batch", C.c_int32),
                ("hb", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p), ("hb_dim", C.c_int32), ("w2_ld", C.c_int32),
                ("boxes_in", C.c_void_p), ("boxes_out", C.c_void_p), ("deltas_out", C.c_void_p), ("wx", C.c_float), ("wy", C.c_float),
                ("ww", C.c_float), ("wh", C.c_float), ("clip", C.c_int32), ("img_w", C.c_float), ("img_h", C.c_float)]


class EodAdamWTensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("n", C.c_size_t),
                ("lr", C.c_double), ("weight_decay", C.c_double), ("step", C.c_int32), ("folded_out", C.c_void_p), ("row_scale", C.c_void_p),
                ("cols", C.c_int32), ("ld_out", C.c_int32), ("grad_of_folded", C.c_int32)]


class EodRotateTensor(C.Structure):
    _fields_ = [("w", C.c_void_p), ("out", C.c_void_p), ("Cout", C.c_int32), ("KH", C.c_int32), ("KW", C.c_int32), ("Cin", C.c_int32),
                ("ld_in", C.c_int32), ("ld_out", C.c_int32)]


class EodMemWriteDesc(C.Structure):
    _fields_ = [
        ("featn", C.c_void_p), ("prop_boxes", C.c_void_p), ("prop_masks", C.c_void_p), ("det_rows", C.c_void_p),
        ("det_count", C.c_void_p), ("K_cap", C.c_int32), ("R_cap", C.c_int32), ("proj", C.c_void_p),
        ("H", C.c_int32), ("W", C.c_int32), ("D", C.c_int32), ("n_cells", C.c_int32), ("mask_thresh", C.c_float),
        ("mem", C.c_void_p), ("obs", C.c_void_p), ("k_out", C.c_void_p), ("workspace", C.c_void_p),
        ("workspace_bytes", C.c_size_t), ("dirty", C.c_void_p), ("err_flags", C.c_void_p), ("snapshot_f16", C.c_void_p),
        ("batch", C.c_int32),
    ]


# name -> (restype, argtypes); every symbol include/eod_hip.h declares
SIGNATURES = {
    "eod_abi_version": (C.c_int, []),
    "eod_conv2d": (C.c_int, [C.POINTER(EodConvDesc), C.c_void_p]),
    "eod_conv2d_workspace_bytes": (C.c_size_t, [C.POINTER(EodConvDesc)]),
    "eod_conv2d_gn_fused": (C.c_int, [C.POINTER(EodConvDesc)]),
    "eod_groupnorm_partial_offset": (C.c_size_t, [C.c_int, C.c_int]),
    "eod_set_conv_math": (C.c_int, [C.c_int]),
    "eod_get_conv_math": (C.c_int, []),
    "eod_conv_split_weights_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "eod_conv_split_weights_bf16x3": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "eod_preprocess_image": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p]),
    "eod_maxpool3x3s2": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.c_void_p]),
    "eod_groupnorm_workspace_bytes": (C.c_size_t, [C.POINTER(C.c_int32), C.c_int, C.c_int]),
    "eod_groupnorm_relu": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_int, C.c_int,
                                     C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_void_p]),
    "eod_groupnorm_backward_workspace_bytes": (C.c_size_t, [C.POINTER(C.c_int32), C.c_int, C.c_int]),
    "eod_groupnorm_relu_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int,
                                              C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "eod_mask_predictor_sigmoid": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                             C.c_int, C.c_void_p, C.c_void_p]),
    "eod_roi_align": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "eod_roi_align_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                         C.c_int, C.c_void_p, C.c_void_p]),
    "eod_centernet_loss_workspace_bytes": (C.c_size_t, []),
    "eod_centernet_loss": (C.c_int, [C.POINTER(EodCenterNetLossDesc), C.c_void_p]),
    "eod_fast_rcnn_loss_workspace_bytes": (C.c_size_t, [C.c_int]),
    "eod_fast_rcnn_loss": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                     C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_size_t, C.c_void_p]),
    "eod_match_label": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p]),
    "eod_match_label_proposals": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "eod_sample_proposals": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "eod_zs_logits": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "eod_zs_logits_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p,
                                         C.c_void_p]),
    "eod_centernet_targets": (C.c_int, [C.POINTER(EodCenterNetTargetDesc), C.c_void_p]),
    "eod_unique_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "eod_proposals_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "eod_centernet_proposals": (C.c_int, [C.POINTER(EodProposalDesc), C.c_void_p]),
    "eod_zs_classify": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                  C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p]),
    "eod_cascade_stage_tail": (C.c_int, [C.POINTER(EodStageTailDesc), C.c_void_p]),
    "eod_apply_deltas": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float,
                                   C.c_float, C.c_float, C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p]),
    "eod_cascade_scores": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "eod_detections_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "eod_fast_rcnn_inference": (C.c_int, [C.POINTER(EodDetDesc), C.c_void_p]),
    "eod_detector_postprocess": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float,
                                           C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_int, C.c_void_p]),
    "eod_concat_lists": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "eod_paste_masks": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float,
                                  C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "eod_unproject_grid_index": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float,
                                           C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_int, C.c_int,
                                           C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "eod_memory_normalize_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "eod_memory_normalize_dirty_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "eod_memory_gather_pool": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                         C.c_int, C.c_int, C.c_void_p]),
    "eod_memory_pooled_halves": (C.c_size_t, [C.c_int, C.c_int]),
    "eod_memory_project_weights_bytes": (C.c_size_t, []),
    "eod_memory_project_prepare": (C.c_int, [C.c_void_p] * 8),
    "eod_memory_project_fuse": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p]),
    "eod_memory_project_backward_weights": (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_float] + [C.c_void_p] * 7),
    "eod_memory_project_backward_weights_workspace_bytes": (C.c_size_t, []),
    "eod_memory_project_backward_weights_ws": (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_float] + [C.c_void_p] * 7 +
                                               [C.c_size_t, C.c_void_p]),
    "eod_memory_pool_backward": (C.c_int, [C.c_void_p] * 3 + [C.c_int, C.c_int] + [C.c_void_p] * 5),
    "eod_conv2d_backward_weights": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 9 + [C.c_void_p, C.c_void_p, C.c_void_p]),
    "eod_conv2d_backward_weights_levels_workspace_bytes": (C.c_size_t, [C.c_int] * 5),
    "eod_conv2d_backward_weights_levels": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 5 +
                                           [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "eod_conv2d_backward_weights_workspace_bytes": (C.c_size_t, [C.c_int] * 9),
    "eod_conv2d_backward_weights_ws": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 9 + [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                                                                          C.c_void_p]),
    "eod_conv2d_backward_input": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 10 + [C.c_void_p, C.c_void_p]),
    "eod_upsample2_sum_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "eod_maxpool3x3s2_backward": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 6 + [C.c_void_p]),
    "eod_relu_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "eod_adamw_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_double, C.c_double, C.c_double, C.c_double,
                                 C.c_double, C.c_int, C.c_double, C.c_void_p]),
    "eod_conv_rotate_weights": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "eod_conv_rotate_weights_multi": (C.c_int, [C.POINTER(EodRotateTensor), C.c_int, C.c_void_p]),
    "eod_adamw_step_multi": (C.c_int, [C.POINTER(EodAdamWTensor), C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p]),
    "eod_memory_scores": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                    C.c_void_p]),
    "eod_memory_write_workspace_bytes": (C.c_size_t, [C.c_int] * 6),
    "eod_memory_write_init": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "eod_memory_write": (C.c_int, [C.POINTER(EodMemWriteDesc), C.c_void_p]),
    "eod_semmap_labels": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p,
                                    C.c_void_p, C.c_void_p]),
    "eod_fill_f32": (C.c_int, [C.c_void_p, C.c_float, C.c_size_t, C.c_void_p]),
    "eod_fill_i32": (C.c_int, [C.c_void_p, C.c_int32, C.c_size_t, C.c_void_p]),
}

_lib = None


def load():
    """Load the HIP library; raise if it is not built (there is no fallback path)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EodError(
            f"{LIB_PATH} is missing: the HIP extension is not built. Run `python -m embodied_object_detection_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the product path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int, what: str):
    if status != 0:
        raise EodError(f"{what} failed: {_ERR.get(status, status)}")
