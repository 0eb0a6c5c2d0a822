"""ctypes binding of libdeeplio_hip.so (the C-ABI declared in include/deeplio_hip.h).

The product path has NO fallback: if the shared library is missing or a symbol is absent,
import fails loudly.  Build it with ``python -m deeplio_amd.build`` (or
``__graft_entry__.build()``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdeeplio_hip.so")

DLIO_OK, DLIO_EINVAL, DLIO_EUNSUP, DLIO_ELAUNCH, DLIO_EWS = 0, -1, -2, -3, -4


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "N", "Cin", "H", "W", "in_ctot", "in_coff", "Cout", "OH", "OW", "out_ctot", "out_coff",
        "KH", "KW", "SH", "SW", "PH", "PW", "res_ctot", "res_coff", "in_relu")]


_p = C.c_void_p
_i = C.c_int
_i64 = C.c_int64
_u64 = C.c_uint64
_f = C.c_float
_d = C.c_double
_sz = C.c_size_t
_cd = C.POINTER(ConvDesc)

# name -> (restype, argtypes); mirrors include/deeplio_hip.h one to one
SIGNATURES = {
    "dlio_version": (_i, []),
    "dlio_abi_hash": (C.c_uint32, []),
    "dlio_arch": (C.c_char_p, []),
    "dlio_build_probes": (_i, []),
    "dlio_strerror": (C.c_char_p, [_i]),
    "dlio_streams_share_queue": (_i, [_p, _p, C.POINTER(_i)]),
    "dlio_last_hip_error_string": (C.c_char_p, []),
    "dlio_prof_enable": (_i, [_i]),
    "dlio_prof_sample": (_i, [_i]),
    "dlio_prof_reset": (_i, []),
    "dlio_prof_release": (_i, []),
    "dlio_prof_collect": (_i, [_i, C.POINTER(_d), C.POINTER(_d), C.POINTER(_d), C.POINTER(_i64)]),
    "dlio_conv2d_prep_weight_floats": (_sz, [_i, _i, _i, _i, _i]),
    "dlio_conv2d_prep_weight": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "dlio_conv2d_prep_weights_batched": (_i, [_p, _i, _i64, _p]),
    "dlio_conv2d_fwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _cd, _p]),
    "dlio_conv_bx3_prep_floats": (_sz, [_i, _i, _i, _i]),
    "dlio_conv_bx3_prep": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "dlio_conv1x1_bx3_fwd": (_i, [_p, _p, _p, _p, _p, _cd, _p]),
    "dlio_conv1x1_bx3_fwd_aff": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _cd, _p]),
    "dlio_conv3x5s2_bx3_fwd": (_i, [_p, _p, _p, _p, _p, _cd, _p]),
    "dlio_conv_bx3_fwd_taps": (_i, [_p, _p, _p, _p, _p, _cd, _p]),
    "dlio_conv_h2_fwd_strided": (_i, [_p, _p, _p, _p, _p, _p, _cd, _p]),
    "dlio_conv_h2_fwd_taps": (_i, [_p, _p, _p, _p, _p, _p, _cd, _p]),
    "dlio_conv1x1_bx3_ws_bytes": (_sz, [_cd]),
    "dlio_conv1x1_bx3_fwd_ws": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _cd, _p]),
    "dlio_fire_expand_dgrad": (_i, [_p, _p, _p, _p, _i, _p, _p, _p, _sz, _cd, _p]),
    "dlio_conv3x3_bx3_prep_floats": (_sz, [_i, _i, _i]),
    "dlio_conv3x3_bx3_prep": (_i, [_p, _p, _i, _i, _i, _p]),
    "dlio_conv3x3_bx3_prep_batched": (_i, [_p, _i, _i64, _p]),
    "dlio_conv3x3_bx3_fwd": (_i, [_p, _p, _p, _p, _p, _cd, _p]),
    "dlio_conv3x3_h2_ok": (_i, [_cd]),
    "dlio_conv1x1_h2_fwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _sz, _cd, _p]),
    "dlio_conv3x3_h2_fwd": (_i, [_p, _p, _p, _p, _p, _p, _cd, _p]),
    "dlio_conv3x3_bx3_ws_bytes": (_sz, [_cd]),
    "dlio_conv3x3_bx3_fwd_ws": (_i, [_p, _p, _p, _p, _p, _p, _sz, _cd, _p]),
    "dlio_fire_planes_bytes": (_sz, [_i, _i, _i, _i]),
    "dlio_bn_split16": (_i, [_p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _f, _f, _p, _p, _p, _p, _p, _p, _i, _i, _p, _p, _sz,
                             _i, _d, _p, _p]),
    "dlio_fire_expand_fwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "dlio_conv_h2_prep_floats": (_sz, [_i, _i, _i, _i]),
    "dlio_conv_h2_prep": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "dlio_conv_h2_prep_batched": (_i, [_p, _i, _i64, _p]),
    "dlio_fire_expand_stats_ws_bytes": (_sz, [_i, _i, _i, _i]),
    "dlio_fire_expand_fwd_stats": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i] + [_p] * 8 + [_f, _f, _p, _p, _p, _p,
                                        _p, _sz, _i, _p]),
    "dlio_bn_small_ok": (_i, [_i, _i]),
    "dlio_bn_small_fwd": (_i, [_p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _f, _f, _p, _p, _p, _p,
                               _p, _i, _i, _p, _p, _p, _p, _i, _i, _p, _i, _i, _p, _p]),
    "dlio_bn_small_bwd": (_i, [_p, _i, _i, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i,
                               _p, _p]),
    "dlio_bn_aff_apply": (_i, [_p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _i, _i, _p, _p, _p, _p, _i, _i, _p, _i, _i, _p]),
    "dlio_bn_aff_pool_ok": (_i, [_i, _i, _i]),
    "dlio_bn_aff_pool_fwd": (_i, [_p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _i, _i, _p, _p, _p, _p, _p, _p, _i, _i, _p]),
    "dlio_bn_coop_ok": (_i, [_i, _i]),
    "dlio_bn_bf16_coop_fwd": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _p, _f, _f, _p, _p, _p, _p, _p, _p, _i, _i, _p, _i, _i,
                                   _p, _i, _i, _p, _p, _p]),
    "dlio_bn_bf16_coop_bwd": (_i, [_p, _i, _i, _p, _i, _i, _p, _p, _p, _p, _p, _i, _i, _p, _p, _i, _i, _i, _i, _i, _p, _p,
                                   _p]),
    "dlio_bn_coop_set_cus": (_i, [_i]),
    "dlio_bn_coop_set_mode": (_i, [_i]),
    "dlio_bn_coop_get_mode": (_i, []),
    "dlio_bn_coop_one_item": (_i, [_i, _i]),
    "dlio_bn_coop_parts": (_i, [_i, _i]),
    "dlio_bn_coop_gap_ok": (_i, [_i, _i]),
    "dlio_bn_coop_ws_bytes": (_sz, [_i, _i]),
    "dlio_bn_coop_empty": (C.c_uint64, []),
    "dlio_bn_coop_fwd": (_i, [_p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _f, _f, _p, _p, _p,
                              _p, _i, _i, _p, _p, _p, _p, _i, _i, _p, _i, _i, _p, _p, _p, _p]),
    "dlio_bn_coop_bwd": (_i, [_p, _i, _i, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i,
                              _p, _p, _p, _p]),
    "dlio_bn_coop_pool_ok": (_i, [_i, _i, _i, _i]),
    "dlio_bn_coop_bwd_pool": (_i, [_p, _i, _i, _p, _p, _p, _p, _i, _i, _i, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                                   _p, _i, _i, _i, _i, _i, _p, _p, _p, _p]),
    "dlio_zero_upsample2d": (_i, [_p, _p, _i64, _i, _i, _i, _i, _i, _i, _p]),
    "dlio_phase_interleave2d": (_i, [_p, _i, _i, _p, _i, _i, _p, _i, _i, _i, _i, _i, _i, _p]),
    "dlio_conv2d_dgrad_strided": (_i, [_p, _p, _p, _cd, _p]),
    "dlio_conv2d_wgrad_ws_bytes": (_sz, [_cd]),
    "dlio_conv2d_wgrad": (_i, [_p, _p, _p, _p, _p, _p, _p, _sz, _i, _cd, _p]),
    "dlio_conv3x3_wgrad_h2_ok": (_i, [_cd]),
    "dlio_conv3x3_wgrad_h2": (_i, [_p, _p, _p, _p, _p, _p, _sz, _i, _cd, _p]),
    "dlio_chan_stats_ws_bytes": (_sz, [_i, _i, _i]),
    "dlio_chan_stats": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _sz, _p]),
    "dlio_bn_finalize": (_i, [_p, _p, _i, _d, _p, _f, _f, _p, _p, _p, _p, _p, _p]),
    "dlio_bn_eval_params": (_i, [_p, _p, _p, _f, _i, _p, _p, _p, _p]),
    "dlio_bn_apply": (_i, [_p, _i, _i, _p, _p, _p, _p, _i, _i, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "dlio_bn_train_stats": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _f, _f, _p, _p, _p, _p, _p, _p, _sz, _p, _p, _i, _d,
                                 _p]),
    "dlio_bn_bwd_reduce": (_i, [_p, _i, _i, _p, _i, _i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p,
                               _p, _p, _i, _p, _sz, _p]),
    "dlio_bn_bwd_apply": (_i, [_p, _i, _i, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _p, _p,
                              _i, _i, _i, _i, _i, _i, _p]),
    "dlio_chan_sum": (_i, [_p, _i, _i, _i, _i, _i, _p, _i, _p, _sz, _p]),
    "dlio_maxpool2d_fwd": (_i, [_p, _p, _p, _p] + [_i] * 11 + [_p]),
    "dlio_maxpool2d_fwd_aff": (_i, [_p, _p, _p, _p] + [_i] * 11 + [_p]),
    "dlio_bn_bwd_pool": (_i, [_p] * 10 + [_i] * 8 + [_p, _sz, _p]),
    "dlio_maxpool2d_bwd": (_i, [_p, _p, _p, _p, _p] + [_i] * 11 + [_p]),
    "dlio_maxpool2d_bwd_dot": (_i, [_p, _p, _p, _p] + [_i] * 11 + [_p]),
    "dlio_plane_dot": (_i, [_p, _p, _p, _p, _i, _i, _p]),
    "dlio_pair_fuse_fc_ws_bytes": (_sz, [_i, _i, _i]),
    "dlio_pair_fuse_fc_fwd": (_i, [_p, _p, _i, _i, _i, _i, _p, _p, _i, _i, _p, _p, _p, _sz, _p, _p]),
    "dlio_pair_fuse_bwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "dlio_se_fc_ok": (_i, [_i, _i, _i]),
    "dlio_se_fc_fwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "dlio_se_fc_bwd": (_i, [_p] * 9 + [_f, _p, _p, _i, _i, _i, _i, _p]),
    "dlio_gap_fwd": (_i, [_p, _i, _i, _p, _i, _i, _i, _p]),
    "dlio_gap_bwd": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "dlio_chan_scale_fwd": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "dlio_chan_scale_bwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "dlio_linear_fwd": (_i, [_p, _i, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _p]),
    "dlio_act_bwd": (_i, [_p, _p, _p, _i64, _i, _p]),
    "dlio_linear_bwd_data_ws_bytes": (_sz, [_i, _i, _i]),
    "dlio_linear_bwd_data": (_i, [_p, _i, _p, _p, _i, _i, _i, _i, _i, _p, _sz, _p]),
    "dlio_linear_bwd_weight": (_i, [_p, _i, _p, _i, _p, _p, _i, _i, _i, _i, _p]),
    "dlio_abs_max": (_i, [_p, _i64, _p, _p]),
    "dlio_ew_binary": (_i, [_p, _p, _p, _i64, _i, _p, _p]),
    "dlio_seg_sum_fwd": (_i, [_p, _p, _i, _i, _i, _p]),
    "dlio_seg_sum_bwd": (_i, [_p, _p, _i, _i, _i, _p]),
    "dlio_ew_scale": (_i, [_p, _f, _p, _i64, _p]),
    "dlio_copy2d": (_i, [_p, _i, _p, _i, _i, _i, _i, _p]),
    "dlio_dropout_fwd": (_i, [_p, _p, _p, _i64, _f, _u64, _u64, _p]),
    "dlio_dropout_bwd": (_i, [_p, _p, _p, _i64, _f, _p]),
    "dlio_nonfinite_flag": (_i, [_p, _i64, _p, _p]),
    "dlio_rnn_ws_bytes": (_sz, [_i, _i, _i]),
    "dlio_lstm_seq_fwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i,
                              _i, _p, _sz, _p]),
    "dlio_lstm_seq_bwd": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i,
                              _p, _sz, _p]),
    "dlio_gru_seq_fwd": (_i, [_p, _p, _p, _p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _sz,
                             _p]),
    "dlio_gru_seq_bwd": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _sz,
                             _p]),
    "dlio_lstm_layer_ok": (_i, [_i, _i, _i, _i, _i]),
    "dlio_lstm_layer_ws_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "dlio_lstm_layer_fwd": (_i, [_p, _i] + [_p] * 8 + [_p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _p, _sz, _p]),
    "dlio_lstm_layer_wgrad": (_i, [_p, _p, _i, _p] + [_p] * 8 + [_i, _i, _i, _i, _i, _i, _p]),
    "dlio_lstm_layer_bwd": (_i, [_p, _i, _p, _i, _p, _p, _p] + [_p] * 4 + [_p] + [_p] * 8 + [_i, _p, _i, _i, _i, _i, _i, _i, _p,
                                 _sz, _p]),
    "dlio_optim_set_max_blocks": (_i, [_i]),
    "dlio_soft_fusion_ok": (_i, [_i, _i, _i]),
    "dlio_soft_fusion_fwd": (_i, [_p, _i, _p, _i] + [_p] * 6 + [_i, _i, _i, _p]),
    "dlio_soft_fusion_bwd": (_i, [_p, _p, _i, _p, _i] + [_p] * 9 + [_i, _i, _i, _i, _p]),
    "dlio_heads_ok": (_i, [_i, _i, _i]),
    "dlio_heads_fwd": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _f, _u64, _u64, _p]),
    "dlio_heads_bwd": (_i, [_p, _p, _p, _i, _p, _p, _p, _p, _i, _p, _p, _p, _p, _i, _i, _f, _i, _p]),
    "dlio_se3_chain_fwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "dlio_se3_chain_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "dlio_so3_project": (_i, [_p, _p, _p, _i, _p]),
    "dlio_so3_project_bwd": (_i, [_p, _p, _p, _i, _p]),
    "dlio_pose_loss_fwd": (_i, [C.POINTER(_p), C.POINTER(_p), C.POINTER(C.c_int32), _p, _p, _f, _i,
                               _p, _p]),
    "dlio_pose_loss_bwd": (_i, [C.POINTER(_p), C.POINTER(_p), C.POINTER(C.c_int32), _p, _p, _f, _i,
                               _p, _p, C.POINTER(_p), _p, _p, _p]),
    "dlio_pose_tail_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _f, _i, _i, _p, _p, _p, _p, _p, _p, _p]),
    "dlio_pose_tail_bwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _f, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p,
                               _p, _p, _i, _p]),
    "dlio_pose_tail_ws_floats": (_sz, [_i, _i, _i, _i]),
    "dlio_pair_stack": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "dlio_gt_relative": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "dlio_chan_stats_splits": (_i, [_i, _i, _i]),
    "dlio_bn_train_apply": (_i, [_p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _f, _f, _p, _p, _p, _p, _p, _p, _i, _i,
                                 _p, _i, _i, _p, _i, _i, _p, _sz, _i, _d, _p, _p, _p, _p, _p]),
    "dlio_bn_bwd": (_i, [_p, _i, _i, _p, _i, _i, _p, _p, _p, _p, _p, _i, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i,
                         _p, _sz, _i, _d, _p, _p, _p]),
    "dlio_scan_project_ws_bytes": (_sz, [_i, _i]),
    "dlio_scan_project": (_i, [_p, _p, _i, _i, _i, _d, _d, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "dlio_scan_normals": (_i, [_p, _p, _p, _i, _i, _p]),
    "dlio_velo_image": (_i, [_p, _p, _p, _p, _f, _p, _p, _i, _i, _i, _i, _i, _p, _p]),
    "dlio_conv_bf16_prep_elems": (_sz, [_i, _i, _i, _i]),
    "dlio_conv_bf16_prep": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "dlio_conv_bf16_prep_batched": (_i, [_p, _i, _i64, _p]),
    "dlio_conv3x3_bf16_fwd": (_i, [_p, _p, _p, _p, _p, _cd, _p]),
    "dlio_conv1x1_bf16_fwd": (_i, [_p, _p, _p, _p, _p, _cd, _p]),
    "dlio_conv2d_wgrad_bf16": (_i, [_p, _p, _p, _p, _sz, _i, _cd, _p]),
    "dlio_bf16_stats_splits": (_i, [_i, _i, _i]),
    "dlio_bf16_stats_ws_bytes": (_sz, [_i, _i, _i]),
    "dlio_bn_bf16_apply": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _p, _f, _f, _p, _p, _p, _p, _p, _p, _i, _i, _p, _i, _i,
                                _p, _i, _i, _i, _p, _sz, _i, _d, _p]),
    "dlio_bn_bf16_bwd": (_i, [_p, _i, _i, _p, _i, _i, _p, _p, _p, _p, _p, _i, _i, _p, _p, _i, _i, _i, _i, _i, _i,
                              _p, _sz, _i, _d, _p, _p]),
    "dlio_maxpool_bf16_fwd": (_i, [_p, _p, _p, _p] + [_i] * 11 + [_p]),
    "dlio_maxpool_bf16_bwd": (_i, [_p, _p, _p, _p, _p] + [_i] * 11 + [_p]),
    "dlio_maxpool_bf16_bwd_dot": (_i, [_p, _p, _p, _p] + [_i] * 11 + [_p]),
    "dlio_gap_bf16_fwd": (_i, [_p, _i, _i, _p, _i, _i, _i, _p]),
    "dlio_gap_bf16_bwd": (_i, [_p, _p, _i, _i, _i, _p]),
    "dlio_cast_bf16": (_i, [_p, _p, _i64, _i, _p]),
    "dlio_adam_step": (_i, [_p, _p, _p, _p, _i64, _f, _f, _f, _f, _f, _i, _f, _p]),
    "dlio_sgd_step": (_i, [_p, _p, _p, _i64, _f, _f, _f, _i, _f, _p]),
    "dlio_rmsprop_step": (_i, [_p, _p, _p, _p, _p, _i64, _f, _f, _f, _f, _f, _f, _p]),
    "dlio_adadelta_step": (_i, [_p, _p, _p, _p, _i64, _f, _f, _f, _f, _f, _p]),
    "dlio_sumsq": (_i, [_p, _i64, _p, _p]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "deeplio_amd: %s is missing -- the HIP extension is mandatory (no CPU fallback). "
            "Run `python -m deeplio_amd.build`." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ImportError("deeplio_amd: symbol %s missing from %s" % (name, LIB_PATH)) from e
        fn.restype = res
        fn.argtypes = args
    # a stale library (built from another revision of include/deeplio_hip.h) must not be called
    # through this table: same symbol names do not mean same signatures
    from ._header import abi_hash, abi_version
    if lib.dlio_version() != abi_version() or lib.dlio_abi_hash() != abi_hash():
        raise ImportError("deeplio_amd: %s was built from a different include/deeplio_hip.h (library ABI %d / "
                          "%08x, header %d / %08x) -- rebuild with `python -m deeplio_amd.build`"
                          % (LIB_PATH, lib.dlio_version(), lib.dlio_abi_hash(), abi_version(), abi_hash()))
    # ... and neither must a library built with a timing probe (kernels that skip work on purpose: tools/variant_lib.py)
    probes = lib.dlio_build_probes()
    if probes and os.environ.get("DLIO_ALLOW_PROBES", "0") != "1":
        raise ImportError("deeplio_amd: %s was built with timing-probe macros (mask %d: DLIO_SPLIT_Q0 / BX3_ABLATE / "
                          "W1_COAL_PROBE) and computes wrong results -- rebuild with `python -m deeplio_amd.build --force` "
                          "(DLIO_ALLOW_PROBES=1 loads it for a timing experiment)" % (LIB_PATH, probes))
    return lib


lib = _load()


def strerror(code):
    return lib.dlio_strerror(int(code)).decode()


def check(code, what=""):
    """Map C-ABI status to the reference's error style: ValueError for bad configuration
    (nets/__init__.py:103,145 raise ValueError on unknown names), RuntimeError otherwise."""
    if code == DLIO_OK:
        return
    msg = "deeplio_hip %s failed: %s (%d)" % (what, strerror(code), code)
    if code == DLIO_ELAUNCH:
        msg += " [%s]" % lib.dlio_last_hip_error_string().decode()
    if code in (DLIO_EINVAL, DLIO_EUNSUP):
        raise ValueError(msg)
    raise RuntimeError(msg)
