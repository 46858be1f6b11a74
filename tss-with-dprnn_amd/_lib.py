"""ctypes binding of libdprnn_mi355x.so (the C ABI in include/dprnn_mi355x.h).

The library is the product: there is NO fallback.  If it is missing or fails to load,
``load_library`` raises, and every model forward goes through it.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import threading
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "libdprnn_mi355x.so"
_lock = threading.Lock()
_lib = None

c_fp = ctypes.c_void_p      # device pointers travel as integers (tensor.data_ptr())
c_i = ctypes.c_int
c_i64 = ctypes.c_int64
c_f = ctypes.c_float
c_d = ctypes.c_double

# name -> argtypes (all return int unless listed in _VOID / _SPECIAL); mirrors include/dprnn_mi355x.h
SIGNATURES = {
    "dprnn_encoder_stats": [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_fp],
    "dprnn_bottleneck_prep": [c_fp, c_fp, c_fp, c_f, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_fp],
    "dprnn_bottleneck_fwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_fp],
    "dprnn_bottleneck_att_fwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_fp],
    "dprnn_att_row_scale": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fp],
    "dprnn_segment_fwd": [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp],
    "dprnn_lstm_pack": [c_fp] * 10 + [c_fp],
    "dprnn_bilstm_fwd": [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i64, c_i64, c_i64, c_i, c_i, c_fp],
    "dprnn_bilstm_norm_fwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_f, c_i, c_i64, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i64, c_i64,
                              c_i64, c_i, c_i, c_fp],
    "dprnn_rnn_pack": [c_i] + [c_fp] * 10 + [c_fp],
    "dprnn_rnn_fwd": [c_i, c_fp, c_fp, c_fp, c_fp, c_fp, c_f, c_i, c_i64, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i64, c_i64,
                      c_i64, c_i, c_i, c_fp],
    "dprnn_rnn_pack_ex": [c_i, c_i] + [c_fp] * 10 + [c_fp],
    "dprnn_rnn_fwd_ex": [c_i, c_i, c_fp, c_fp, c_fp, c_fp, c_fp, c_f, c_i, c_i64, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i64, c_i64,
                         c_i64, c_i, c_i, c_fp],
    "dprnn_rnn_fwd_hp_ex": [c_i, c_i, c_fp, c_fp, c_fp, c_fp, c_fp, c_f, c_i, c_i64, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i64, c_i64,
                            c_i64, c_i, c_i, c_fp],
    "dprnn_proj_stats_hp_fwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_i64, c_i64, c_fp, c_i, c_i, c_fp],
    "dprnn_lin_pack_f16": [c_fp, c_i, c_fp, c_fp],
    "dprnn_rnn_lin_fwd": [c_i, c_i, c_fp, c_fp, c_fp, c_fp, c_fp, c_f, c_i, c_i64, c_fp, c_fp, c_fp, c_i, c_fp, c_fp, c_fp, c_fp, c_fp,
                          c_i, c_i, c_i, c_i64, c_i64, c_i64, c_i, c_fp],
    "dprnn_rnn_lin_split_fwd": [c_i, c_fp, c_fp, c_fp, c_fp, c_fp, c_f, c_i, c_i64, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp,
                                c_i, c_i, c_i, c_i64, c_i64, c_i64, c_i, c_fp],
    "dprnn_rnn_lin_split_ragged_fwd": [c_i, c_fp, c_fp, c_fp, c_fp, c_fp, c_f, c_fp, c_fp, c_fp, c_i, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp,
                                       c_i, c_i, c_fp],
    "dprnn_rnn_fwd_ragged_hp_ex": [c_i, c_fp, c_fp, c_fp, c_fp, c_fp, c_f, c_fp, c_fp, c_fp, c_i, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_fp],
    "dprnn_lstm_pack_split": [c_fp] * 10 + [c_fp],
    "dprnn_lstm_pack_f16": [c_fp] * 10 + [c_fp],
    "dprnn_bilstm_fwd_stamped": [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i64, c_i64, c_i64, c_i, c_i, c_fp, c_fp],
    "dprnn_proj_stats_fwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_i64, c_i64, c_i, c_fp],
    "dprnn_proj_stats_ragged_fwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_i64, c_fp, c_i, c_i, c_fp],
    "dprnn_rnn_fwd_ragged": [c_fp, c_fp, c_fp, c_fp, c_fp, c_f, c_fp, c_fp, c_fp, c_i, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_fp],
    "dprnn_rnn_fwd_ragged_ex": [c_i, c_fp, c_fp, c_fp, c_fp, c_fp, c_f, c_fp, c_fp, c_fp, c_i, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_fp],
    "dprnn_saturation_begin": [c_fp, c_fp],
    "dprnn_saturation_flags": [c_fp, c_fp],
    "dprnn_rows_affine_fwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_i64, c_i, c_i64, c_fp],
    "dprnn_att_row_scale_rows": [c_fp, c_i, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp],
    "dprnn_scale_init": [c_fp, c_fp],
    "dprnn_scale_update": [c_fp, c_fp],
    "dprnn_scale_fork": [c_fp, c_fp, c_fp],
    "dprnn_scale_join": [c_fp, c_fp, c_fp],
    "dprnn_norm_residual_fwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_f, c_fp, c_i, c_i64, c_fp],
    "dprnn_head_conv_ola_fwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp],
    "dprnn_head_conv_ola_norm_fwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_f, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp],
    "dprnn_head_conv_ola_ragged_fwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_f, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i64, c_i, c_i, c_fp],
    "dprnn_head_gate_fwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i64, c_fp],
    "dprnn_head_mask_decode_fwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fp],
    "dprnn_decoder_fwd": [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_fp],
    "dprnn_pairwise_sisdr": [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp],
    "dprnn_pairwise_sisdr_metric": [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp],
    "dprnn_pointwise_fwd": [c_fp, c_i, c_i, c_fp, c_fp, c_fp, c_i, c_i64, c_fp, c_fp, c_i, c_fp, c_fp, c_i, c_i, c_fp, c_i64, c_fp],
    "dprnn_bn_prep": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i64, c_f, c_f, c_fp, c_fp, c_fp, c_i, c_fp],
    "dprnn_resblock_pool_fwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_fp],
    "dprnn_time_mean_fwd": [c_fp, c_fp, c_i, c_fp, c_i, c_i, c_i, c_i, c_fp],
    "dprnn_linear_small_fwd": [c_fp, c_i, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_i, c_i, c_i, c_fp],
    "dprnn_rows_stats": [c_fp, c_fp, c_i, c_i64, c_i, c_fp],
    # ---- training path (include/dprnn_mi355x_train.h)
    "dprnn_lstm_fwd_train": [c_fp, c_fp, c_fp, c_fp, c_fp, c_f, c_i, c_i64, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i,
                             c_i64, c_i64, c_i64, c_i, c_fp],
    "dprnn_lstm_pack_bwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp],
    "dprnn_rnn_fwd_train": [c_i, c_fp, c_fp, c_fp, c_fp, c_fp, c_f, c_i, c_i64, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i,
                            c_i64, c_i64, c_i64, c_i, c_fp],
    "dprnn_rnn_pack_bwd": [c_i, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp],
    "dprnn_rnn_fwd_train_ex": [c_i, c_i, c_fp, c_fp, c_fp, c_fp, c_fp, c_f, c_i, c_i64, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i,
                               c_i64, c_i64, c_i64, c_i, c_fp],
    "dprnn_rnn_pack_bwd_ex": [c_i, c_i, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp],
    "dprnn_rnn_bwd_ex": [c_i, c_fp, c_i, c_fp, c_fp, c_fp, c_fp, c_fp, c_i64, c_i, c_i, c_i, c_i64, c_i64, c_i64, c_i, c_fp],
    "dprnn_wgrad_ex": [c_fp, c_fp, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_i64, c_i64, c_i, c_fp, c_i64, c_i64,
                       c_fp, c_fp, c_i, c_fp, c_fp],
    "dprnn_dgrad_ex": [c_fp, c_fp, c_fp, c_i, c_fp, c_i, c_i, c_fp, c_i, c_fp, c_i, c_i, c_i64, c_fp],
    "dprnn_gn_bwd_dz_ex": [c_fp] * 6 + [c_f, c_i, c_i64, c_fp, c_fp],
    "dprnn_absmax": [c_fp, c_i64, c_fp, c_fp],
    "dprnn_wgrad_lstm_f16": [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i64, c_i64, c_i64, c_fp, c_fp, c_fp, c_fp, c_fp],
    "dprnn_rnn_bwd": [c_i, c_fp, c_fp, c_fp, c_fp, c_fp, c_i64, c_i, c_i, c_i, c_i64, c_i64, c_i64, c_i, c_fp],
    "dprnn_lstm_bwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_i64, c_i, c_i, c_i, c_i64, c_i64, c_i64, c_i, c_fp],
    "dprnn_wgrad": [c_fp, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_i64, c_i64, c_i, c_fp, c_i64, c_i64,
                    c_fp, c_fp, c_i, c_fp, c_fp],
    "dprnn_dgrad": [c_fp, c_i, c_fp, c_i, c_i, c_fp, c_i, c_fp, c_i, c_i, c_i64, c_fp],
    "dprnn_head_gate_fwd_train": [c_fp] * 7 + [c_i64, c_fp],
    "dprnn_head_mask_decode_fwd_train": [c_fp] * 8 + [c_i] * 7 + [c_fp],
    "dprnn_head_conv_bwd": [c_fp] * 6 + [c_i] * 5 + [c_fp],
    "dprnn_gn_bwd_stats": [c_fp] * 7 + [c_f, c_i, c_i64, c_fp, c_fp, c_fp],
    "dprnn_gn_bwd_stats_ex": [c_fp] * 7 + [c_f, c_i, c_i64, c_fp, c_fp, c_fp, c_fp],
    "dprnn_lin_bwd_f16": [c_fp] * 6 + [c_f, c_i, c_i64] + [c_fp] * 8,
    "dprnn_gn_bwd_dz": [c_fp] * 5 + [c_f, c_i, c_i64, c_fp, c_fp],
    "dprnn_gate_bwd": [c_fp, c_fp, c_fp, c_i64, c_fp],
    "dprnn_mask_decode_bwd": [c_fp] * 5 + [c_i, c_fp, c_fp, c_fp, c_fp] + [c_i] * 6 + [c_fp],
    "dprnn_rows64_affine": [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_fp],
    "dprnn_rows64_norm_bwd": [c_fp, c_fp, c_fp, c_fp, c_f, c_fp, c_fp, c_fp, c_i, c_i, c_fp],
    "dprnn_rows64_norm_bwd_ex": [c_fp, c_fp, c_fp, c_fp, c_f, c_fp, c_fp, c_fp, c_i, c_i, c_d, c_fp],
    "dprnn_ola_prelu_sum": [c_fp] * 5 + [c_i] * 5 + [c_fp],
    "dprnn_segment_bwd": [c_fp] * 4 + [c_i] * 5 + [c_fp],
    "dprnn_encoder_norm_rows": [c_fp] * 5 + [c_i] * 6 + [c_fp],
    "dprnn_encoder_norm_bwd_stats": [c_fp] * 5 + [c_f, c_fp, c_fp] + [c_i] * 6 + [c_fp],
    "dprnn_encoder_bwd": [c_fp] * 5 + [c_f, c_fp, c_fp, c_fp] + [c_i] * 6 + [c_fp],
    "dprnn_scratch_to_grad": [c_fp, c_fp, c_i, c_i, c_fp],
    "dprnn_pit_sisdr_fwd": [c_fp] * 7 + [c_i, c_i, c_i, c_fp],
    "dprnn_pit_sisdr_bwd": [c_fp] * 6 + [c_i, c_i, c_i, c_fp],
    "dprnn_multi_sumsq": [c_fp, c_i, c_fp, c_fp],
    "dprnn_time_mean_bwd": [c_fp, c_fp, c_i, c_fp, c_i, c_i, c_i, c_i, c_fp],
    "dprnn_resblock_pool_bwd": [c_fp] * 9 + [c_i, c_i, c_i, c_fp],
    "dprnn_bn_moments": [c_fp, c_fp, c_fp, c_i64, c_f, c_fp, c_fp, c_i, c_fp],
    "dprnn_bn_bwd": [c_fp] * 5 + [c_i, c_fp, c_fp, c_i64, c_i, c_fp],
    "dprnn_affine_prelu_rows": [c_fp] * 6 + [c_i64, c_i, c_fp],
    "dprnn_affine_prelu_bwd": [c_fp] * 8 + [c_i64, c_i, c_fp],
    "dprnn_add_rows": [c_fp, c_fp, c_fp, c_i64, c_fp],
    "dprnn_utt_colsum": [c_fp, c_fp, c_i, c_i, c_i, c_fp],
    "dprnn_linear_small_bwd": [c_fp, c_i, c_fp, c_i, c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_fp, c_fp, c_i, c_i, c_fp],
    "dprnn_fusion_bwd": [c_fp] * 12 + [c_i] * 7 + [c_fp],
    "dprnn_att_scale_bwd": [c_fp] * 7 + [c_i64, c_fp],
    "dprnn_att_softmax_bwd": [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_fp],
    "dprnn_adam_multi": [c_fp, c_i, c_f, c_d, c_d, c_f, c_f, c_i, c_f, c_fp, c_fp],
    # ---- any-width forward (include/dprnn_mi355x_wide.h)
    "dprnn_wide_encoder_fwd": [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_fp],
    "dprnn_wide_flat_stats": [c_fp, c_fp, c_i, c_i64, c_fp],
    "dprnn_wide_norm_prep": [c_fp, c_fp, c_fp, c_f, c_fp, c_fp, c_i, c_i64, c_i, c_fp],
    "dprnn_wide_segment_fwd": [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_fp],
    "dprnn_wide_rnn_pack": [c_i, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_fp, c_fp, c_fp],
    "dprnn_wide_rnn_fwd": [c_i, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i64, c_i64, c_i64, c_i64, c_fp],
    "dprnn_wide_rnn_fwd_tiles": [c_i, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_fp, c_i, c_i64, c_fp],
    "dprnn_wide_absmax": [c_fp, c_i64, c_fp, c_fp],
    "dprnn_wide_lstm_pack_f16": [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_fp, c_fp],
    "dprnn_wide_lstm_f16_fwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i64, c_i64, c_i64, c_i64, c_fp, c_i, c_fp],
    "dprnn_wide_linear_pack_f16": [c_fp, c_i, c_i, c_i, c_fp, c_f, c_fp, c_fp, c_fp],
    "dprnn_wide_linear_f16": [c_fp, c_i64, c_fp, c_fp, c_fp, c_fp, c_i64, c_fp, c_i64, c_i64, c_i, c_i, c_fp],
    "dprnn_wide_gemm": [c_fp, c_i64, c_fp, c_fp, c_i64, c_fp, c_fp, c_i, c_fp, c_fp, c_fp, c_i, c_fp, c_i64, c_i64, c_i, c_fp, c_i64, c_fp, c_i64,
                        c_i, c_i, c_fp],
    "dprnn_wide_norm_residual_fwd": [c_fp, c_fp, c_fp, c_fp, c_i64, c_i, c_i64, c_fp, c_fp],
    "dprnn_wide_ola_fwd": [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fp],
}
_VOID = {
    "dprnn_set_fast_activations": [c_i],
    "dprnn_set_fast_head_activations": [c_i],
    "dprnn_set_gemm_mode": [c_i],
    "dprnn_set_bwd_gemm_mode": [c_i],
    "dprnn_wgrad_use_f16_pieces": [c_fp],
    "dprnn_dgrad_use_f16_pieces": [c_fp],
    "dprnn_gn_bwd_dz_store_absmax": [c_fp],
    "dprnn_dgrad_store_absmax": [c_fp],
    "dprnn_rnn_bwd_use_gradmax": [c_fp],
    "dprnn_debug_bwd_stamps": [c_fp],
    "dprnn_pack_forget": [c_fp],
    "dprnn_set_rowgemm_grid": [c_i],
}
_SPECIAL = {
    "dprnn_last_error": ([], ctypes.c_char_p),
    "dprnn_abi_version": ([], c_i),
    "dprnn_num_cus": ([], c_i),
    "dprnn_get_gemm_mode": ([], c_i),
    "dprnn_get_rowgemm_grid": ([], c_i),
    "dprnn_get_fast_activations": ([], c_i),
    "dprnn_get_fast_head_activations": ([], c_i),
    "dprnn_lstm_tape_floats": ([c_i, c_i, c_i], c_i64),
    "dprnn_rnn_tape_floats": ([c_i, c_i, c_i, c_i], c_i64),
    "dprnn_lin_pack_floats": ([], c_i64),
    "dprnn_wide_rnn_pack_floats": ([c_i, c_i, c_i], c_i64),
    "dprnn_wide_lstm_pack_f16_words": ([c_i, c_i], c_i64),
    "dprnn_wide_linear_pack_f16_words": ([c_i, c_i], c_i64),
    "dprnn_rnn_lin_state_floats": ([c_i, c_i], c_i64),
    "dprnn_rnn_pack_bwd_floats": ([c_i], c_i64),
    "dprnn_rnn_bwd_gradmax_offset": ([c_i], c_i64),
    "dprnn_rnn_bwd_dg_floats": ([c_i, c_i, c_i, c_i, c_i], c_i64),
    "dprnn_get_bwd_gemm_mode": ([], c_i),
    "dprnn_scale_begin": ([c_fp], c_i),
    "dprnn_scale_slot": ([c_i], c_i),
    "dprnn_wgrad_workspace_floats": ([c_i, c_i, c_i64], c_i64),
    "dprnn_lin_bwd_workspace_floats": ([c_i64], c_i64),
    "dprnn_workspace_bytes": ([c_i] * 9 + [ctypes.POINTER(c_i64)], c_i64),
}
ALL_SYMBOLS = sorted(list(SIGNATURES) + list(_VOID) + list(_SPECIAL))
ABI_VERSION = 3             # DPRNN_ABI_VERSION of include/dprnn_mi355x.h this binding is written against
SAT_SLOTS = 8               # DPRNN_SAT_SLOTS
SCALE_SLOTS = 32            # DPRNN_SCALE_SLOTS
SAT_NAMES = ("recurrence input rows", "rows of the Linear after a recurrence", "rows of a row GEMM (bottleneck / mask head / data gradient)",
             "BPTT gate gradients", "weight-gradient operands", "non-finite input (NaN / Inf samples)", "", "")


class DprnnError(RuntimeError):
    pass


def library_path() -> str:
    # DPRNN_LIB: alternative build of the same ABI (A/B timing of kernel variants inside one process pool / one GPU box)
    return os.environ.get("DPRNN_LIB") or os.path.join(_HERE, _LIB_NAME)


def build_library(force: bool = False, verbose: bool = False, build_dir: Optional[str] = None) -> str:
    """Compile csrc/*.hip for gfx950 into tss-with-dprnn_amd/libdprnn_mi355x.so (hipcc cross-compiles
    without a GPU).  force = `make clean` first.  build_dir: put the objects AND the library there instead (a from-clean
    build that leaves the in-tree library alone); returns the path of the library built."""
    csrc = os.path.join(_HERE, "csrc")
    extra = []
    out = library_path() if build_dir is None else os.path.join(build_dir, _LIB_NAME)
    if build_dir is not None:
        os.makedirs(build_dir, exist_ok=True)
        extra = [f"BUILD={build_dir}", f"OUT={out}"]
    if force:
        subprocess.run(["make", "-C", csrc, "clean"] + extra, check=True, capture_output=not verbose)
    jobs = max(2, min(8, os.cpu_count() or 2))
    res = subprocess.run(["make", "-C", csrc, f"-j{jobs}"] + extra, capture_output=True, text=True)
    if res.returncode != 0:
        raise DprnnError("building libdprnn_mi355x.so failed:\n" + res.stdout + res.stderr)
    if verbose:
        print(res.stdout)
    return out


def load_library() -> ctypes.CDLL:
    """Load the HIP library; raise loudly if it is not there (no CPU / PyTorch fallback exists)."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = library_path()
        if not os.path.exists(path):
            raise DprnnError(
                f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"or `make -C {os.path.join(_HERE, 'csrc')}`. The DPRNN MI355X path has no fallback.")
        try:
            lib = ctypes.CDLL(path)
        except OSError as e:
            raise DprnnError(f"cannot load {path}: {e}") from e
        try:
            lib.dprnn_abi_version.restype = c_i
            have = int(lib.dprnn_abi_version())
        except AttributeError as e:
            raise DprnnError(f"{path} does not export dprnn_abi_version: not a libdprnn_mi355x build") from e
        if have != ABI_VERSION:
            raise DprnnError(f"{path} has ABI version {have}, this host code is written against {ABI_VERSION}: rebuild the library "
                             f"(`make -C {os.path.join(_HERE, 'csrc')}`) or point DPRNN_LIB at a matching build")
        for name, args in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = c_i
        for name, args in _VOID.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = None
        for name, (args, res) in _SPECIAL.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = res
        _lib = lib
        return lib


def check(lib, rc: int, what: str) -> None:
    if rc != 0:
        msg = lib.dprnn_last_error()
        raise DprnnError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")
