"""ctypes binding of libbsi_hip.so (the C ABI declared in include/bsi_hip.h).

There is NO fallback: if the library is missing or a tensor is not on a HIP device the
functions raise.  torch is used only for device memory and the current stream.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must be imported first: it loads the HIP runtime libbsi_hip.so binds to)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BSI_HIP_LIB") or os.path.join(_HERE, "lib", "libbsi_hip.so")  # BSI_HIP_LIB: an experimental build for A/B runs

_lib = None


class BSIParams(C.Structure):
    _fields_ = [("lambda_0", C.c_float), ("alpha_M", C.c_float), ("alpha_R", C.c_float),
                ("ln_low", C.c_float), ("delta", C.c_float)]


class GemmArgs(C.Structure):
    _fields_ = [("A", C.c_void_p), ("W", C.c_void_p), ("bias", C.c_void_p), ("out", C.c_void_p),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("lda", C.c_int), ("ldw", C.c_int), ("ldo", C.c_int), ("epilogue", C.c_int),
                ("gate", C.c_void_p), ("gate_rows", C.c_int), ("gate_stride", C.c_int),
                ("tokens", C.c_int), ("pos", C.c_void_p), ("aux", C.c_void_p), ("out2", C.c_void_p), ("colsum_rows", C.c_void_p)]


class CastDesc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("dst_t", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int), ("ld", C.c_int),
                ("ld_t", C.c_int), ("tile0", C.c_int), ("reserved", C.c_int)]


class FwdGate(C.Structure):  # bsi_fwd_gate (include/bsi_hip.h)
    _fields_ = [("event", C.c_void_p), ("cast", C.c_void_p), ("n_cast", C.c_int), ("cast_tiles", C.c_int)]


class CopyDesc(C.Structure):  # bsi_copy_desc (include/bsi_hip.h)
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("len", C.c_size_t), ("tile0", C.c_uint), ("reserved", C.c_uint)]


class Seg(C.Structure):  # bsi_seg (include/bsi_hip.h)
    _fields_ = [("p_off", C.c_size_t), ("g_off", C.c_size_t), ("len", C.c_size_t), ("my_chunk", C.c_size_t), ("out_chunk", C.c_size_t)]


class ColsumJob(C.Structure):
    _fields_ = [("src", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int), ("ld", C.c_int), ("out", C.c_void_p)]


class ConvArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("x2", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("zeros", C.c_void_p),
                ("out", C.c_void_p), ("film", C.c_void_p), ("resid", C.c_void_p), ("film_rows", C.c_int),
                ("film_stride", C.c_int), ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Cin", C.c_int),
                ("Cin2", C.c_int), ("Cout", C.c_int), ("taps", C.c_int), ("ldo", C.c_int), ("epilogue", C.c_int),
                ("gn_partial", C.c_void_p)]


CONV_BIAS_BF16, CONV_FILM_SILU_BF16, CONV_BIAS_RESID_F32 = range(3)


class ConvPackDesc(C.Structure):
    _fields_ = [("w", C.c_void_p), ("out", C.c_void_p), ("Cout", C.c_int), ("Cin", C.c_int), ("taps", C.c_int), ("cin_pad", C.c_int),
                ("ld", C.c_int), ("col0", C.c_int)]


class UNetConfig(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("C", "H", "W", "dim", "levels", "heads", "ff_nmin", "ff_nmax", "emb_size", "c_dim")]


class UNetResBlockWeights(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("gn_w", "gn_b", "conv1_w", "conv1_b", "conv2_w", "conv2_b")]


class UNetWeights(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("enc_w", "enc_b", "dec_w", "dec_b", "pe_scale", "pe_bias", "pm1_w", "pm1_b", "pm3_w",
                                          "pm3_b", "film_w", "film_b")] + \
               [("blocks", C.POINTER(UNetResBlockWeights))] + \
               [(k, C.c_void_p) for k in ("agn_w", "agn_b", "aqkv_w", "aqkv_b", "aout_w", "aout_b")]


class UNetResBlockWeightsT(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("conv1_wT", "conv2_wT", "skip_wT")]


class UNetWeightsT(C.Structure):
    _fields_ = [("blocks", C.POINTER(UNetResBlockWeightsT))] + [(k, C.c_void_p) for k in ("aqkv_wT", "aout_wT", "film_wT", "pm3_wT")]


class UNetResBlockGrads(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("gn_w", "gn_b", "conv1_w", "conv1_b", "conv2_w", "conv2_b", "skip_w")]


class UNetGrads(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("enc_w", "enc_b", "dec_w", "dec_b", "pm1_w_padded", "pm1_b", "pm3_w", "pm3_b",
                                          "film_w", "film_b")] + \
               [("blocks", C.POINTER(UNetResBlockGrads))] + \
               [(k, C.c_void_p) for k in ("agn_w", "agn_b", "aqkv_w", "aqkv_b", "aout_w", "aout_b")]


class DitConfig(C.Structure):
    _fields_ = [("C", C.c_int), ("H", C.c_int), ("W", C.c_int), ("patch", C.c_int),
                ("dim", C.c_int), ("depth", C.c_int), ("heads", C.c_int),
                ("ff_nmin", C.c_int), ("ff_nmax", C.c_int)]


class DitBlockWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("qkv_w", "out_w", "fc1_w", "fc2_w", "ada0_w", "ada2_w",
                 "qkv_b", "out_b", "fc1_b", "fc2_b", "ada0_b", "ada2_b")]


class DitWeights(C.Structure):
    _fields_ = [("enc_w", C.c_void_p), ("enc_b", C.c_void_p), ("pos", C.c_void_p),
                ("t_scale", C.c_void_p), ("t_bias", C.c_void_p),
                ("dec_ln_w", C.c_void_p), ("dec_ln_b", C.c_void_p), ("dec_w", C.c_void_p), ("dec_b", C.c_void_p),
                ("blocks", C.POINTER(DitBlockWeights))]


class DitBlockWeightsT(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("qkv_wT", "out_wT", "fc1_wT", "fc2_wT", "ada2_wT")]


class DitWeightsT(C.Structure):
    _fields_ = [("blocks", C.POINTER(DitBlockWeightsT))]


class DitBlockGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("qkv_w", "qkv_b", "out_w", "out_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b",
                                          "ada0_w", "ada0_b", "ada2_w", "ada2_b")]


class DitGrads(C.Structure):
    _fields_ = [("enc_w_padded", C.c_void_p), ("enc_b", C.c_void_p), ("dec_ln_w", C.c_void_p),
                ("dec_ln_b", C.c_void_p), ("dec_w", C.c_void_p), ("dec_b", C.c_void_p),
                ("blocks", C.POINTER(DitBlockGrads))]


(EPI_BIAS_F32, EPI_BIAS_BF16, EPI_BIAS_GELU_BF16, EPI_BIAS_SILU_BF16, EPI_GATE_RESID, EPI_BIAS_POS_F32,
 EPI_BIAS_GELU_DUAL, EPI_MUL_GELUGRAD_BF16) = range(8)

_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
_PROTOS = {
    "bsi_version": (C.c_int, []),
    "bsi_last_error": (C.c_char_p, []),
    "bsi_edm_coeffs": (_i, [C.POINTER(BSIParams), _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "bsi_lambda_to_t": (_i, [C.POINTER(BSIParams), _vp, _i, _vp, _vp, _vp]),
    "bsi_schedule": (_i, [C.POINTER(BSIParams), _vp, _i, _vp, _vp, _vp]),
    "bsi_lambda_grid": (_i, [C.POINTER(BSIParams), _vp, _vp, _i, _vp, _vp]),
    "bsi_q_sample": (_i, [C.POINTER(BSIParams), _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "bsi_sample_init": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "bsi_fourier_features": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "bsi_scale_rows": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "bsi_predict_combine": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "bsi_predict_combine_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "bsi_refine_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "bsi_refine_step_philox": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "bsi_philox_normal": (_i, [_vp, C.c_uint, _sz, _vp, _vp]),
    "bsi_philox4x32_10": (_i, [_vp, _sz, _vp, _vp]),
    "bsi_philox_uint32": (_i, [_vp, C.c_uint, _sz, _vp, _vp]),
    "bsi_sqerr_rows": (_i, [_vp, _vp, _vp, _f, _i, _i, _i, _i, _vp, _vp]),
    "bsi_sqerr_rows_bwd": (_i, [_vp, _vp, _vp, _vp, _f, _i, _i, _i, _i, _vp, _vp]),
    "bsi_recon_nll": (_i, [_vp, _vp, _f, _vp, _f, _f, _i, _i, _i, _i, _vp, _vp]),
    "bsi_tgrid": (_i, [_vp, _vp, _i, _vp, _vp]),
    "bsi_affine_noise": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "bsi_axpbypcz": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _sz, _vp, _vp]),
    "bsi_clip": (_i, [_vp, _f, _f, _sz, _vp, _vp]),
    "bsi_clip_bwd": (_i, [_vp, _vp, _f, _f, _sz, _vp, _vp]),
    "bsi_vdm_coeffs": (_i, [_vp, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "bsi_vdm_step_coeffs": (_i, [_vp, _i, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "bsi_vdm_recon_nll": (_i, [_vp, _vp, _f, _vp, _f, _f, _i, _i, _i, _i, _vp, _vp]),
    "bsi_vdm_prior": (_i, [_vp, _f, _i, _i, _vp, _vp]),
    "bsi_bfn_coeffs": (_i, [_vp, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "bsi_bfn_schedule": (_i, [_vp, _i, _f, _vp, _vp, _vp, _vp]),
    "bsi_to_uint8": (_i, [_vp, _f, _f, _sz, _vp, _vp]),
    "bsi_cast_bf16": (_i, [_vp, _i, _i, _vp, _i, _vp]),
    "bsi_nyquist_embed": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "bsi_gemm_bf16": (_i, [C.POINTER(GemmArgs), _vp]),
    "bsi_gemm_splitk_workspace_bytes": (C.c_size_t, [_i, _i, _i]),
    "bsi_gemm_splitk_f32_workspace_bytes": (C.c_size_t, [_i, _i, _i]),
    "bsi_gemm_bf16_grouped": (_i, [C.POINTER(GemmArgs), _i, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, _vp]),
    "bsi_colsum_rows_scratch_bytes": (C.c_size_t, [_i, _i]),
    "bsi_colsum_rows_f32": (_i, [C.POINTER(ColsumJob), _i, _vp, _vp]),
    "bsi_gemm_bf16_ws": (_i, [C.POINTER(GemmArgs), _vp, C.c_size_t, _vp]),
    "bsi_gemm_set_variant": (_i, [_i]),
    "bsi_conv_set_grid_limit": (_i, [_i]),
    "bsi_conv_set_ablation": (_i, [_i]),
    "bsi_gemm_tn_workspace_bytes": (_sz, [_i, _i, _i]),
    "bsi_gemm_tn_bf16": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp]),
    "bsi_gemm_tn_pair_bf16": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _vp, _i, _i, _i, _vp, _vp]),
    "bsi_gemm_tn_bias_bf16": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp]),
    "bsi_colsum_workspace_bytes": (_sz, [_i]),
    "bsi_colsum_bf16": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _vp]),
    "bsi_ln_modulate": (_i, [_vp, _i, _i, _f, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "bsi_resid_ln_modulate": (_i, [_vp, _i, _i, _f, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "bsi_resid2_ln_modulate": (_i, [_vp, _i, _i, _f, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "bsi_attention_fwd": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "bsi_attention_fwd_lse": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp]),
    "bsi_attention_bwd": (_i, [_vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "bsi_attention_bwd_long": (_i, [_vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "bsi_gate_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp]),
    "bsi_ln_mod_bwd": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _f, _vp]),
    "bsi_ln_gate_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _i, _i, _i, _vp]),
    "bsi_silu_bwd_bf16": (_i, [_vp, _vp, _sz, _vp, _vp]),
    "bsi_cast_transpose_bf16": (_i, [_vp, _i, _i, _vp, _i, _vp]),
    "bsi_cast_batch_tiles": (_i, [_i, _i, _i]),
    "bsi_cast_batch_bf16": (_i, [_vp, _i, _i, _vp]),
    "bsi_cast_rows_bf16": (_i, [_vp, _i, _i, _i, _vp, _i, _vp]),
    "bsi_silu_bf16": (_i, [_vp, _sz, _vp, _vp]),
    "bsi_dit_tape_bytes": (_sz, [C.POINTER(DitConfig), _i]),
    "bsi_dit_backward_workspace_bytes": (_sz, [C.POINTER(DitConfig), _i]),
    "bsi_dit_train_forward": (_i, [C.POINTER(DitConfig), C.POINTER(DitWeights), _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                   _f, C.c_ulonglong, _vp]),
    "bsi_dit_backward": (_i, [C.POINTER(DitConfig), C.POINTER(DitWeights), C.POINTER(DitWeightsT), C.POINTER(DitGrads), _i,
                              _vp, _vp, _vp, _vp, _f, C.c_ulonglong, _vp]),
    "bsi_dropout_mask": (_i, [_f, C.c_ulonglong, C.c_uint, C.c_uint, C.c_uint, _vp, _vp]),
    "bsi_attention_dropout_words": (_i, [_f, C.c_ulonglong, C.c_uint, _i, _vp, _vp]),
    "bsi_attention_fwd_dropout": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _f, C.c_ulonglong, C.c_uint, _vp, _vp]),
    "bsi_attention_bwd_dropout": (_i, [_vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _f, C.c_ulonglong, C.c_uint, _vp, _vp]),
    "bsi_dit_backward_set_events": (_i, [_vp, _i]),
    "bsi_dit_train_forward_set_gates": (_i, [_vp, _i]),
    "bsi_sqnorm_workspace_bytes": (_sz, []),
    "bsi_grad_sqnorm": (_i, [_vp, _sz, _vp, _vp, _vp]),
    "bsi_clip_adamw_ema": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _vp, _f, _f, _f, _f, _f, _f, _f, _i, _f, _vp]),
    "bsi_conv_nhwc_bf16": (_i, [C.POINTER(ConvArgs), _vp]),
    "bsi_conv_weight_pack": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "bsi_conv_weight_pack_t": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "bsi_conv_weight_pack_batch": (_i, [_vp, _i, _i, _vp]),
    "bsi_conv_wgrad_workspace_bytes": (C.c_size_t, [_i, _i, _i, _i, _i]),
    "bsi_conv_wgrad_nhwc_bf16": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp]),
    "bsi_conv_wgrad_bias_nhwc_bf16": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp]),
    "bsi_conv_wgrad_unpack": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "bsi_groupnorm_nhwc": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _vp, _f, _i, _vp, _vp, _vp]),
    "bsi_groupnorm_stats_nhwc": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _vp, _f, _i, _vp, _vp, _vp, _vp]),
    "bsi_groupnorm_apply_nhwc": (_i, [_vp, _i, _vp, _vp, _i, _vp, _i, _i, _vp, _vp, _f, _i, _vp, _vp, _vp, _vp]),
    "bsi_groupnorm_bwd_nhwc": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "bsi_groupnorm_bwd_cast_nhwc": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "bsi_film_silu": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _f, C.c_ulonglong, C.c_uint, _vp, _vp]),
    "bsi_film_silu_bwd": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _i, _f, C.c_ulonglong, C.c_uint, _vp, _vp, _i, _vp]),
    "bsi_unet_decode_bwd": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    "bsi_unet_decode": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "bsi_unet_cin_pad": (_i, [C.POINTER(UNetConfig)]),
    "bsi_unet_workspace_bytes": (_sz, [C.POINTER(UNetConfig), _i]),
    "bsi_unet_film_scratch_bytes": (_sz, [C.POINTER(UNetConfig), _i]),
    "bsi_unet_film": (_i, [C.POINTER(UNetConfig), C.POINTER(UNetWeights), _vp, _i, _vp, _vp, _vp]),
    "bsi_unet_tape_bytes": (_sz, [C.POINTER(UNetConfig), _i]),
    "bsi_unet_backward_workspace_bytes": (_sz, [C.POINTER(UNetConfig), _i]),
    "bsi_unet_train_forward": (_i, [C.POINTER(UNetConfig), C.POINTER(UNetWeights), _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f,
                                    C.c_ulonglong, _vp]),
    "bsi_unet_backward": (_i, [C.POINTER(UNetConfig), C.POINTER(UNetWeights), C.POINTER(UNetWeightsT), C.POINTER(UNetGrads), _i,
                               _vp, _vp, _vp, _vp, _f, C.c_ulonglong, _vp]),
    "bsi_unet_forward": (_i, [C.POINTER(UNetConfig), C.POINTER(UNetWeights), _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "bsi_dit_kpad": (_i, [C.POINTER(DitConfig)]),
    "bsi_dit_tokens": (_i, [C.POINTER(DitConfig)]),
    "bsi_dit_workspace_bytes": (_sz, [C.POINTER(DitConfig), _i]),
    "bsi_dit_adaln_scratch_bytes": (_sz, [C.POINTER(DitConfig), _i]),
    "bsi_dit_adaln": (_i, [C.POINTER(DitConfig), C.POINTER(DitWeights), _vp, _i, _vp, _vp, _vp]),
    "bsi_dit_forward": (_i, [C.POINTER(DitConfig), C.POINTER(DitWeights), _i, _vp, _vp, _i, _vp, _vp, _vp, _i,
                             _vp, _vp, _vp, _vp]),
    "bsi_cu_pair_create": (_i, [_i, C.POINTER(_vp)]),
    "bsi_cu_pair_destroy": (_i, [_vp]),
    "bsi_cu_pair_streams": (_i, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i)]),
    "bsi_dit_forward_pair": (_i, [C.POINTER(DitConfig), C.POINTER(DitWeights), _i, _vp, _vp, _i, _vp, _vp, _vp, _i,
                                  _vp, _vp, _vp, _i, _vp]),
    "bsi_sqnorm_segments": (_i, [_vp, _vp, _i, _sz, _vp, _vp]),
    "bsi_sqnorm_finish": (_i, [_vp, _sz, _vp, _vp]),
    "bsi_copy_batch_tiles": (_i, [_sz]),
    "bsi_copy_batch_f32": (_i, [_vp, _i, _i, _vp]),
    "bsi_clip_adamw_ema_segments": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _sz, _vp, _f, _f, _f, _f, _f, _f, _f, _i, _f, _vp]),
    "bsi_clock_probe": (_i, [_vp, _i, _vp]),
    "bsi_set_cu_reserve": (_i, [_i]),
    "bsi_compute_cus": (_i, []),
    "bsi_set_ln_stream_cus": (_i, [_i]),
    "bsi_set_tile_queue": (_i, [_i]),
    "bsi_set_attention_bwd_skew": (_i, [_i]),
    "bsi_mfma_probe_workspace_bytes": (_sz, []),
    "bsi_mfma_probe": (_i, [_i, _vp, C.POINTER(C.c_int), C.POINTER(C.c_double), _vp]),
    "bsi_prof_enable": (_i, [C.c_uint]),
    "bsi_prof_read": (_i, [_i, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
}

EXPORTS = tuple(_PROTOS)
PROF_CLASSES = {"gemm_qkv": 0, "gemm_out": 1, "gemm_fc1": 2, "gemm_fc2": 3, "attention": 4, "ln_modulate": 5,
                "prologue": 6, "final": 7, "gemm_enc": 8, "adaln": 9}


def fc1_kernel_name():
    """Name (as rocprofv3 prints it) of the kernel the DiT engine launches for fc1 = the benchmark's dominant kernel."""
    return "gemm_bf16_k64r_kernel<BSI_EPI_BIAS_GELU_BF16=2, DYN=false> (fc1)"


def mfma_probe(iters=50000, device=None):
    """Run the box yardstick (bsi_mfma_probe) on the current stream and wait for it: {"tflops", "mhz", "ms"}."""
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    ws = torch.zeros(lib().bsi_mfma_probe_workspace_bytes() // 8, dtype=torch.int64, device=dev)
    nwg, flop = C.c_int(0), C.c_double(0.0)
    check(lib().bsi_mfma_probe(iters, ptr(ws), C.byref(nwg), C.byref(flop), stream()))
    torch.cuda.synchronize(dev)
    w = ws[: 4 * 8 * nwg.value].reshape(nwg.value * 8, 4).cpu()
    ticks = float(int(w[:, 3].max()) - int(w[:, 2].min()))  # first wave in to last wave out, on the chip-wide 100 MHz counter
    return {"tflops": nwg.value * flop.value / (ticks * 1e-8) / 1e12, "mhz": float((100.0 * w[:, 0].double() / w[:, 1].double()).mean()),
            "ms": ticks * 1e-5, "iters": iters}


def prof_enable(names=()):
    mask = 0
    for n in names:
        mask |= 1 << PROF_CLASSES[n]
    check(lib().bsi_prof_enable(mask))


def prof_read(name):
    cnt, tot = C.c_int(0), C.c_double(0.0)
    check(lib().bsi_prof_read(PROF_CLASSES[name], C.byref(cnt), C.byref(tot)))
    return cnt.value, tot.value


def lib():
    """The loaded library; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()' or make -C bsi_amd/csrc)")
        _lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(_lib, name)
            fn.restype = res
            fn.argtypes = args
        if os.environ.get("BSI_GEMM_VARIANT"):  # kernel experiments (tools/gemm_bench.py documents the encoding)
            _lib.bsi_gemm_set_variant(int(os.environ["BSI_GEMM_VARIANT"]))
        if os.environ.get("BSI_CONV_ABL"):  # convolution kernel choice / ablations (bsi_conv_set_ablation)
            _lib.bsi_conv_set_ablation(int(os.environ["BSI_CONV_ABL"]))
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().bsi_last_error()
        raise RuntimeError(f"bsi_hip error {rc}: {msg.decode() if msg else '?'}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Refuses CPU tensors: there is no CPU path."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("bsi_amd: tensor is not on a HIP device; the native path has no CPU fallback")
    if not t.is_contiguous():
        raise RuntimeError("bsi_amd: tensor must be contiguous")
    return C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
