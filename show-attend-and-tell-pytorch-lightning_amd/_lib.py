"""ctypes binding of libsat_hip.so (include/sat_hip.h).  Fails loudly: no fallback."""
import ctypes as C
import weakref
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsat_hip.so")
_lib = None

f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int32)


class GemmDesc(C.Structure):
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int64), ("a_rows", C.c_void_p),
                ("B", C.c_void_p), ("ldb", C.c_int64),
                ("C", C.c_void_p), ("ldc", C.c_int64), ("c_rows", C.c_void_p),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("amode", C.c_int32), ("bmode", C.c_int32), ("accumulate", C.c_int32), ("epi", C.c_int32),
                ("bias", C.c_void_p), ("e0", C.c_void_p), ("lde0", C.c_int64), ("c0", C.c_int32), ("c1", C.c_int32),
                ("slab", C.c_void_p), ("slab_elems", C.c_int64)]


class DecoderDims(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("B", "R", "T", "L", "D", "A", "m", "n", "V", "P", "deep_output", "padding_idx", "precision", "layers")] + \
               [("embed_max_norm", C.c_float), ("dropout", C.c_float), ("embedding_dropout", C.c_float), ("dropout_seed", C.c_uint64)]


PARAM_FIELDS = ("embedding", "init_f_w", "init_f_b", "init_i_w", "init_i_b", "w_ih", "w_hh", "b_ih", "b_hh",
                "att_enc", "att_dec", "att_f", "beta_w", "beta_b", "out_hidden", "out_context", "out_w", "out_b")
#: C field -> reference state-dict key
PARAM_KEYS = dict(embedding="embedding.weight", init_f_w="init_lstm.factorize.weight", init_f_b="init_lstm.factorize.bias",
                  init_i_w="init_lstm.init.weight", init_i_b="init_lstm.init.bias", w_ih="lstm.weight_ih_l0",
                  w_hh="lstm.weight_hh_l0", b_ih="lstm.bias_ih_l0", b_hh="lstm.bias_hh_l0",
                  att_enc="attention.encoder_att.weight", att_dec="attention.decoder_att.weight", att_f="attention.f_att.weight",
                  beta_w="beta.0.weight", beta_b="beta.0.bias", out_hidden="output.hidden.weight",
                  out_context="output.context.weight", out_w="output.output.weight", out_b="output.output.bias")


MAX_LSTM_LAYERS = 4
#: stacked LSTM layers l >= 1 (lstm.weight_ih_l{l}, ...): pointer arrays indexed l-1
UP_FIELDS = ("up_w_ih", "up_w_hh", "up_b_ih", "up_b_hh")
UP_KEYS = dict(up_w_ih="lstm.weight_ih_l%d", up_w_hh="lstm.weight_hh_l%d", up_b_ih="lstm.bias_ih_l%d", up_b_hh="lstm.bias_hh_l%d")


def param_names(layers=1):
    """Argument order of the decoder parameter list: the 18 base tensors, then 4 per stacked layer."""
    return list(PARAM_FIELDS) + ["%s_l%d" % (k[3:], l) for l in range(1, layers) for k in UP_FIELDS]


def param_key(name):
    """decoder parameter name -> reference state-dict key"""
    if name in PARAM_KEYS:
        return PARAM_KEYS[name]
    base, l = name.rsplit("_l", 1)
    return UP_KEYS["up_" + base] % int(l)


class DecoderParams(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in PARAM_FIELDS] + [(k, C.c_void_p * (MAX_LSTM_LAYERS - 1)) for k in UP_FIELDS]


class OptTensor(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("n", C.c_int64), ("lr", C.c_float),
                ("weight_decay", C.c_float), ("shadow_bf16", C.c_void_p)]


class OptHyper(C.Structure):
    _fields_ = [("kind", C.c_int32), ("nesterov", C.c_int32), ("first_step", C.c_int32), ("beta1", C.c_float), ("beta2", C.c_float),
                ("eps", C.c_float), ("bias_correction1", C.c_float), ("bias_correction2_sqrt", C.c_float), ("momentum", C.c_float),
                ("clip_value", C.c_float)]


class BeamSampling(C.Structure):
    _fields_ = [("method", C.c_int32), ("sample_topk", C.c_int32), ("seed", C.c_uint64), ("gumbel", C.c_void_p), ("decoder_noise", C.c_float),
                ("reserved", C.c_int32), ("normals", C.c_void_p)]


class ImageDesc(C.Structure):
    _fields_ = [("offset", C.c_int64)] + [(k, C.c_int32) for k in ("height", "width", "crop_top", "crop_left", "crop_h", "crop_w", "resized_h",
                                                                  "resized_w", "out_top", "out_left", "flip", "reserved")]


class DecoderBatch(C.Structure):
    _fields_ = [("ann", C.c_void_p), ("caps", C.c_void_p), ("lengths", C.c_void_p), ("prow", C.c_void_p), ("src_row", C.c_void_p),
                ("step_offsets_host", C.c_void_p), ("teacher_host", C.c_void_p)]


#: every symbol include/sat_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "sat_abi_version": (C.c_int, []),
    "sat_debug_trace_launches": (C.c_int, [C.c_int32]),
    "sat_debug_option": (C.c_int, [C.c_char_p, C.c_int32]),
    "sat_last_error": (C.c_char_p, []),
    "sat_gemm_f32": (C.c_int, [C.POINTER(GemmDesc), C.c_void_p]),
    "sat_decoder_workspace_bytes": (C.c_size_t, [C.POINTER(DecoderDims)]),
    "sat_decoder_train_fwd": (C.c_int, [C.POINTER(DecoderDims), C.POINTER(DecoderParams), C.POINTER(DecoderBatch),
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "sat_decoder_train_bwd": (C.c_int, [C.POINTER(DecoderDims), C.POINTER(DecoderParams), C.POINTER(DecoderBatch),
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(DecoderParams), C.c_void_p,
                                        C.c_void_p, C.c_size_t, C.c_void_p]),
    "sat_ce_label_smooth_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    "sat_ce_label_smooth_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p,
                                          C.c_void_p, C.c_void_p]),
    "sat_doubly_stochastic_fwd": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p]),
    "sat_doubly_stochastic_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]),
    "sat_attention_precompute": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "sat_attention_step_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                         C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                         C.c_int32, C.c_int32, C.c_void_p]),
}


class GemmTypes(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("a_bf16", "b_bf16", "c_bf16", "bf16_mfma")]


class ConvGeom(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("N", "H", "W", "C", "K", "R", "S", "stride", "pad", "stride_w")]


class BlockDesc(C.Structure):
    """sat_block_desc (include/sat_hip.h): one residual block of the trunk for sat_encoder_blocks_fwd / _bwd"""
    _fields_ = ([(k, C.c_int32) for k in ("kind", "stride", "cin", "mid", "cout", "has_ds", "N", "H", "W", "fwd_res_bn", "dgrad_join", "bn_bwd_epilogue")] +
                [(k, C.c_void_p) for k in ("w1", "w2", "w3", "wd")] +
                [("gamma", C.c_void_p * 4), ("beta", C.c_void_p * 4), ("running_mean", C.c_void_p * 4), ("running_var", C.c_void_p * 4),
                 ("eps", C.c_float * 4), ("momentum", C.c_float * 4)] +
                [(k, C.c_void_p) for k in ("dw1", "dw2", "dw3", "dwd")] + [("dgamma", C.c_void_p * 4), ("dbeta", C.c_void_p * 4)])


_vp, _i32, _i64, _f = C.c_void_p, C.c_int32, C.c_int64, C.c_float
SYMBOLS.update({
    "sat_encoder_blocks_arena_bytes": (C.c_size_t, [C.POINTER(BlockDesc), _i32]),
    "sat_encoder_blocks_fwd": (C.c_int, [C.POINTER(BlockDesc), _i32, _vp, _vp, C.c_size_t, _vp, C.POINTER(C.c_void_p), _vp]),
    "sat_encoder_blocks_bwd": (C.c_int, [C.POINTER(BlockDesc), _i32, _i32, _i32, _vp, _vp, C.c_size_t, _vp, _vp, _i32, _vp, _vp, _i64, _vp, _i64, _vp, _vp,
                                         C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int32), _vp]),
    "sat_conv2d_fwd": (C.c_int, [_vp, _vp, _vp, _vp, C.POINTER(ConvGeom), _vp]),
    "sat_conv2d_dgrad": (C.c_int, [_vp, _vp, _vp, C.POINTER(ConvGeom), _i32, _vp]),
    "sat_conv2d_wgrad": (C.c_int, [_vp, _vp, _vp, C.POINTER(ConvGeom), _vp, _i64, _vp]),
    "sat_conv2d_wgrad_slab_bytes": (C.c_size_t, [C.POINTER(ConvGeom)]),
    "sat_conv2d_fwd_bf16": (C.c_int, [_vp, _vp, _vp, _vp, C.POINTER(ConvGeom), _vp]),
    "sat_conv2d_dgrad_bf16": (C.c_int, [_vp, _vp, _vp, C.POINTER(ConvGeom), _i32, _vp]),
    "sat_conv2d_wgrad_bf16": (C.c_int, [_vp, _vp, _vp, C.POINTER(ConvGeom), _vp, _i64, _vp]),
    "sat_gemm_ex": (C.c_int, [C.POINTER(GemmDesc), C.POINTER(GemmTypes), _vp]),
    "sat_image_normalize_nhwc4": (C.c_int, [_vp, _vp, _i32, _i32, _i32, C.POINTER(C.c_float), C.POINTER(C.c_float), _vp]),
    "sat_pad_channels_3to4": (C.c_int, [_vp, _vp, _i64, _i32, _vp]),
    "sat_bn_scratch_bytes": (C.c_size_t, [_i64, _i32]),
    "sat_bn_train_fwd": (C.c_int, [_vp, _i64, _i32, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp]),
    "sat_bn_eval_fwd": (C.c_int, [_vp, _i64, _i32, _vp, _vp, _f, _vp, _vp, _vp, _i32, _vp, _vp]),
    "sat_colsum": (C.c_int, [_vp, _i64, _i64, _i32, _vp, _vp, _vp]),
    "sat_bn_train_bwd": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp]),
    "sat_bn_train_fwd_t": (C.c_int, [_i32, _vp, _i64, _i32, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "sat_conv2d_fwd_stats_bytes": (C.c_size_t, [C.POINTER(ConvGeom)]),
    "sat_conv2d_fwd_bf16_stats": (C.c_int, [_vp, _vp, _vp, C.POINTER(ConvGeom), _vp, C.POINTER(C.c_int32), _vp]),
    "sat_bn_train_fwd_tiles_bf16": (C.c_int, [_vp, _i64, _i32, _vp, _i32, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "sat_bn_eval_fwd_t": (C.c_int, [_i32, _vp, _i64, _i32, _vp, _vp, _f, _vp, _vp, _vp, _i32, _vp, _vp]),
    "sat_conv2d_dgrad_stats_bytes": (C.c_size_t, [C.POINTER(ConvGeom)]),
    "sat_conv2d_dgrad_bf16_bnstats": (C.c_int, [_vp, _vp, _vp, C.POINTER(ConvGeom), _i32, _vp, _vp, _vp, _vp, _vp, C.POINTER(C.c_int32), _vp]),
    "sat_bn_train_fwd_tiles_bf16_resbn": (C.c_int, [_vp, _i64, _i32, _vp, _i32, _vp, _vp, C.c_float, C.c_float, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                                   _i32, _vp, _vp, _vp, _vp]),
    "sat_conv2d_dgrad_bf16_fused": (C.c_int, [_vp, _vp, _vp, C.POINTER(ConvGeom), _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(C.c_int32), _vp]),
    "sat_bn_train_bwd_tiles_bf16": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp]),
    "sat_bn_train_bwd_t": (C.c_int, [_i32, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp]),
    "sat_maxpool3x3s2_fwd_t": (C.c_int, [_i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "sat_maxpool3x3s2_bwd_t": (C.c_int, [_i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "sat_cast_f32_to_bf16": (C.c_int, [_vp, _vp, _i64, _vp]),
    "sat_image_normalize_nhwc8_bf16": (C.c_int, [_vp, _vp, _i32, _i32, _i32, C.POINTER(C.c_float), C.POINTER(C.c_float), _vp]),
    "sat_image_normalize_nhwc4_padded_bf16": (C.c_int, [_vp, _vp, _i32, _i32, _i32, C.POINTER(C.c_float), C.POINTER(C.c_float), _vp]),
    "sat_stem_filter_pairs": (C.c_int, [_vp, _vp, _i32, _vp]),
    "sat_dwconv3x3_fwd_t": (C.c_int, [_i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "sat_dwconv3x3_dgrad_t": (C.c_int, [_i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "sat_dwconv3x3_wgrad_scratch_bytes": (C.c_size_t, [_i32, _i32, _i32, _i32, _i32]),
    "sat_dwconv3x3_wgrad_t": (C.c_int, [_i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "sat_shuffle_join_t": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "sat_shuffle_split_t": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "sat_cast_bf16_to_f32": (C.c_int, [_vp, _vp, _i64, _vp]),
    "sat_grouped_filter_expand": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "sat_grouped_filter_grad_extract": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "sat_stem_filter_grad_unpairs": (C.c_int, [_vp, _vp, _i32, _vp]),
    "sat_stem_filter_pad": (C.c_int, [_vp, _vp, _i64, _vp]),
    "sat_stem_filter_grad_unpad": (C.c_int, [_vp, _vp, _i64, _vp]),
    "sat_maxpool3x3s2_fwd": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "sat_maxpool3x3s2_bwd": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "sat_resize_fwd": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "sat_resize_bwd": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
})


class ProfileEntry(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("launches", C.c_int64), ("total_ms", C.c_double), ("flops", C.c_double), ("bytes", C.c_double)]


SYMBOLS.update({
    "sat_decoder_infer_workspace_bytes": (C.c_size_t, [C.POINTER(DecoderDims), _i32]),
    "sat_decoder_infer_begin": (C.c_int, [C.POINTER(DecoderDims), C.POINTER(DecoderParams), _vp, _i32, _i32, _vp, _vp, _vp, C.c_size_t, _vp]),
    "sat_decoder_infer_step": (C.c_int, [C.POINTER(DecoderDims), C.POINTER(DecoderParams), _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp,
                                         C.c_size_t, _vp]),
    "sat_beam_search_workspace_bytes": (C.c_size_t, [C.POINTER(DecoderDims), _i32]),
    "sat_beam_search_batched": (C.c_int, [C.POINTER(DecoderDims), C.POINTER(DecoderParams), _vp, _i32, _i32, C.POINTER(C.c_float), _i32, C.POINTER(C.c_int32),
                                          _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "sat_beam_search_sampled": (C.c_int, [C.POINTER(DecoderDims), C.POINTER(DecoderParams), _vp, _i32, _i32, C.POINTER(C.c_float), _i32, C.POINTER(C.c_int32),
                                          C.POINTER(BeamSampling), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "sat_beam_scores": (C.c_int, [_vp, _i32, _i32, _f, _vp, _i32, _vp, _vp, _vp]),
    "sat_topk": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp, _vp]),
})
SYMBOLS.update({"sat_optimizer_chunk_elems": (C.c_int32, []),
                "sat_grad_clip_coef": (C.c_int, [_vp, _vp, _i32, _f, _vp, _vp, _vp]),
                "sat_optimizer_step": (C.c_int, [_vp, _vp, _i32, C.POINTER(OptHyper), _vp, _vp]),
                "sat_optimizer_step_dev": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _vp])})
SYMBOLS.update({"sat_stem_tail_fwd_t": (C.c_int, [_i32, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
                "sat_stem_tail_bwd_t": (C.c_int, [_i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp])})
_u64 = C.c_uint64
SYMBOLS.update({
    "sat_attention_step_bwd": (C.c_int, [_vp, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp,
                                         _i32, _i32, _i32, _i32, _i32, _vp]),
    "sat_attention_context_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "sat_lstm_cell_fwd": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "sat_lstm_cell_bwd": (C.c_int, [_vp, _i32] + [_vp] * 17 + [_i32, _i32, _vp]),
    "sat_deep_output_fwd": (C.c_int, [_vp] * 7 + [_f, _u64, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "sat_deep_output_bwd": (C.c_int, [_vp] * 8 + [_f, _u64] + [_vp] * 8 + [_i32, _i32, _i32, _i32, _i32, _vp]),
    "sat_init_lstm_fwd": (C.c_int, [_vp] * 5 + [_f, _u64, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "sat_init_lstm_bwd": (C.c_int, [_vp] * 5 + [_f, _u64] + [_vp] * 8 + [_i32, _i32, _i32, _i32, _i32, _vp]),
    "sat_embedding_fwd": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _f, _vp, _vp]),
    "sat_embedding_bwd": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "sat_sigmoid_bwd": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
})
SYMBOLS.update({"sat_image_batch_workspace_bytes": (C.c_size_t, [_vp, _i32, _i32, _i32]),
                "sat_image_batch_transform": (C.c_int, [_vp, _i64, _vp, _vp, _i32, _i32, _i32, _vp, _f, _vp, _vp, _vp, C.c_size_t, _vp])})
SYMBOLS.update({"sat_profile_start": (C.c_int, []),
                "sat_profile_start_only": (C.c_int, [C.c_char_p]),
                "sat_profile_pause": (C.c_int, [_i32]),
                "sat_profile_stop": (C.c_int, [C.POINTER(ProfileEntry), _i32, C.POINTER(C.c_int32)])})


def profile_start(only=None):
    """record every instrumented scope, or only the family ``only``"""
    if only:
        check(lib().sat_profile_start_only(only.encode()), "sat_profile_start_only")
    else:
        check(lib().sat_profile_start(), "sat_profile_start")


def profile_pause(paused=True):
    check(lib().sat_profile_pause(1 if paused else 0), "sat_profile_pause")


def profile_stop(max_entries=256):
    buf = (ProfileEntry * max_entries)()
    n = C.c_int32(0)
    check(lib().sat_profile_stop(buf, max_entries, C.byref(n)), "sat_profile_stop")
    return [dict(name=buf[i].name.decode(), launches=buf[i].launches, total_ms=buf[i].total_ms, flops=buf[i].flops, bytes=buf[i].bytes)
            for i in range(n.value)]


class SatHipError(RuntimeError):
    pass


def lib():
    """The loaded library; raises if it was not built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SatHipError("libsat_hip.so is missing at %s: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950).  There is no CPU/PyTorch fallback." % LIB_PATH)
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)          # AttributeError if the export is missing
            fn.restype, fn.argtypes = res, args
        if handle.sat_abi_version() != 23:
            raise SatHipError("libsat_hip.so ABI version %d != 23 (rebuild: make -C csrc)" % handle.sat_abi_version())
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        raise SatHipError("%s failed (%d): %s" % (what, rc, lib().sat_last_error().decode(errors="replace")))


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise SatHipError("sat_amd computes on the GPU only: got a %s tensor (no CPU fallback)" % t.device)


# ----------------------------------------------------------------------------- gradient destinations (dist.GradSync)
_grad_sinks = {}


def register_grad_sink(param, make_view):
    """``make_view()`` returns a fresh view of the memory where ``param``'s gradient should be written (a slice of a
    data-parallel bucket); the backward wrappers ask ``grad_buffer`` for it."""
    _grad_sinks[id(param)] = make_view


def unregister_grad_sink(param):
    _grad_sinks.pop(id(param), None)


_handed_out = {}          # id(param) -> weak reference to the slice view handed out and not yet adopted as param.grad


def grad_buffer(param):
    """Where a backward kernel writes ``param``'s gradient: the registered bucket slice when the parameter holds no
    gradient yet (autograd then adopts the slice as ``param.grad``), else new memory with the parameter's layout
    (autograd adds it to the existing gradient).  The slice goes out ONCE per backward pass: a second backward node of the same
    parameter (the encoder or decoder Function applied twice before one ``backward()``) would otherwise overwrite the first node's
    gradient in the shared slice and autograd would add the slice to itself.  "Still out" = the view handed out is alive and the
    parameter has not adopted it (``AccumulateGrad`` runs only after every use of the parameter)."""
    make = _grad_sinks.get(id(param))
    if make is not None and param.grad is None:
        ref = _handed_out.get(id(param))
        if ref is None or ref() is None:
            v = make()
            _handed_out[id(param)] = weakref.ref(v)
            return v
    return torch.empty_like(param)


def ptr(t):
    """raw device address for a ``c_void_p`` argument / struct field (a plain int: ctypes converts it; wrapping it in ``c_void_p`` here cost
    0.2 us x ~2100 arguments per train step)"""
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr():
    """the current stream of the current device as a raw handle.  (``torch.cuda.current_stream().cuda_stream`` builds a Stream object through
    several Python layers: 8 us a call, ~330 calls per C2 train step = a quarter of the host time of a step.)"""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
