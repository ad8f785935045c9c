"""ctypes binding of libt2mi.so (include/t2mi.h).  No fallback: a missing library is an error."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("T2MI_LIB") or os.path.join(_HERE, "libt2mi.so")     # T2MI_LIB: diagnostic builds (tools/)

T2_OK, T2_EINVAL, T2_ENOMEM, T2_EHIP, T2_ESTATE, T2_ERANGE = 0, -1, -2, -3, -4, -5
T2_MAX_BATCH = 64
T2_MAX_PRENET_LAYERS = 4


class T2Dims(C.Structure):
    _fields_ = [
        ("num_symbols", C.c_int32), ("embedding_dim", C.c_int32), ("enc_conv_layers", C.c_int32),
        ("enc_kernel", C.c_int32), ("enc_units", C.c_int32), ("num_prenet_layers", C.c_int32),
        ("prenet_units", C.c_int32 * T2_MAX_PRENET_LAYERS),
        ("attention_lstm_units", C.c_int32), ("decoder_lstm_units", C.c_int32), ("attention_dim", C.c_int32),
        ("attention_filters", C.c_int32), ("attention_kernel", C.c_int32), ("cumulative_weights", C.c_int32),
        ("num_mels", C.c_int32), ("outputs_per_step", C.c_int32), ("n_feed_frame", C.c_int32),
        ("postnet_layers", C.c_int32), ("postnet_kernel", C.c_int32), ("postnet_channels", C.c_int32),
        ("zoneout_cell", C.c_float), ("zoneout_output", C.c_float), ("bn_eps", C.c_float), ("forget_bias", C.c_float),
    ]


class T2V1Dims(C.Structure):
    _fields_ = [
        ("num_symbols", C.c_int32), ("embedding_dim", C.c_int32), ("enc_prenet_units", C.c_int32 * 2),
        ("cbhg_out_units", C.c_int32), ("conv_channels", C.c_int32), ("max_filter_width", C.c_int32),
        ("proj1_channels", C.c_int32), ("proj2_channels", C.c_int32), ("num_highway", C.c_int32),
        ("dec_prenet_units", C.c_int32 * 2), ("attention_out_units", C.c_int32), ("decoder_out_units", C.c_int32),
        ("num_mels", C.c_int32), ("outputs_per_step", C.c_int32), ("n_feed_frame", C.c_int32),
        ("post_cbhg_out_units", C.c_int32), ("post_conv_channels", C.c_int32), ("post_max_filter_width", C.c_int32),
        ("post_proj1_channels", C.c_int32), ("post_proj2_channels", C.c_int32), ("post_num_highway", C.c_int32),
        ("num_freq", C.c_int32), ("bn_eps", C.c_float),
    ]


class T2Error(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libt2mi error {code}: {msg}")
        self.code = code
        self.msg = msg


_p = C.c_void_p
_i64p = C.POINTER(C.c_int64)
_SIGS = {
    "t2_version": (C.c_int, []),
    "t2_last_error": (C.c_char_p, []),
    "t2_ctx_create": (C.c_int, [C.c_int, _p, C.POINTER(_p)]),
    "t2_ctx_destroy": (C.c_int, [_p]),
    "t2_ctx_synchronize": (C.c_int, [_p]),
    "t2_model_create": (C.c_int, [_p, C.POINTER(T2Dims), C.POINTER(_p)]),
    "t2_model_set_weight": (C.c_int, [_p, C.c_char_p, _p, _i64p, C.c_int]),
    "t2_model_finalize": (C.c_int, [_p]),
    "t2_model_destroy": (C.c_int, [_p]),
    "t2_model_set_option": (C.c_int, [_p, C.c_char_p, C.c_int]),
    "t2_model_get_stat": (C.c_int, [_p, C.c_char_p, C.POINTER(C.c_double)]),
    "t2_encode": (C.c_int, [_p, _p, _p, C.c_int, C.c_int, _p]),
    "t2_encode_embedded": (C.c_int, [_p, _p, _p, C.c_int, C.c_int, _p]),
    "t2_postnet": (C.c_int, [_p, _p, C.c_int, C.c_int, _p]),
    "t2_state_create": (C.c_int, [_p, _p, _p, C.c_int, C.c_int, C.POINTER(_p)]),
    "t2_state_reset": (C.c_int, [_p]),
    "t2_state_destroy": (C.c_int, [_p]),
    "t2_state_get": (C.c_int, [_p, C.c_char_p, _p]),
    "t2_decode_step": (C.c_int, [_p, _p, _p, _p, _p]),
    "t2_decode_step_train": (C.c_int, [_p, _p, _p, C.c_float, C.c_int, C.c_uint32, _p, _p]),
    "t2_decode_teacher": (C.c_int, [_p, _p, _p, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p, _p, _p]),
    "t2_decode_free": (C.c_int, [_p, _p, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p, _p, _p, C.POINTER(C.c_int)]),
    "t2_decode_free_launch": (C.c_int, [_p, _p, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p, _p, _p]),
    "t2_decode_free_finish": (C.c_int, [_p, C.POINTER(C.c_int)]),
    "t2_decode_final_state": (C.c_int, [_p, C.c_char_p, _p]),
    "t2v1_model_create": (C.c_int, [_p, C.POINTER(T2V1Dims), C.POINTER(_p)]),
    "t2v1_model_set_weight": (C.c_int, [_p, C.c_char_p, _p, _i64p, C.c_int]),
    "t2v1_model_finalize": (C.c_int, [_p]),
    "t2v1_model_destroy": (C.c_int, [_p]),
    "t2v1_model_set_option": (C.c_int, [_p, C.c_char_p, C.c_int]),
    "t2v1_model_get_stat": (C.c_int, [_p, C.c_char_p, C.POINTER(C.c_double)]),
    "t2v1_encode": (C.c_int, [_p, _p, _p, C.c_int, C.c_int, _p]),
    "t2v1_encode_embedded": (C.c_int, [_p, _p, _p, C.c_int, C.c_int, _p]),
    "t2v1_cbhg": (C.c_int, [_p, C.c_int, _p, _p, C.c_int, C.c_int, _p]),
    "t2v1_postnet": (C.c_int, [_p, _p, C.c_int, C.c_int, _p]),
    "t2v1_decode_teacher": (C.c_int, [_p, _p, _p, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p, _p, _p]),
    "t2v1_decode_free": (C.c_int, [_p, _p, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p, _p, _p, C.POINTER(C.c_int)]),
    "t2v1_postnet_trainer_create": (C.c_int, [_p, C.c_int, C.c_int, C.POINTER(_p)]),
    "t2v1_postnet_trainer_destroy": (C.c_int, [_p]),
    "t2v1_postnet_trainer_bind": (C.c_int, [_p, C.c_char_p, C.c_int64]),
    "t2v1_postnet_train_forward": (C.c_int, [_p, _p, _p, _p]),
    "t2v1_cbhg_train_forward": (C.c_int, [_p, _p, _p, _p, _p]),
    "t2v1_postnet_train_backward": (C.c_int, [_p, _p, _p, _p, _p, _p]),
    "t2v1_postnet_trainer_update_moving_stats": (C.c_int, [_p, _p, C.c_float]),
    "t2v1_postnet_trainer_export": (C.c_int, [_p, _p]),
    "t2v1_decoder_trainer_create": (C.c_int, [_p, C.c_int, C.c_int, C.c_int, C.POINTER(_p)]),
    "t2v1_decoder_trainer_destroy": (C.c_int, [_p]),
    "t2v1_decoder_trainer_bind": (C.c_int, [_p, C.c_char_p, C.c_int64]),
    "t2v1_decoder_trainer_set_loss_counts": (C.c_int, [_p, _p]),
    "t2v1_train_decoder_forward": (C.c_int, [_p, _p, _p, _p, _p, C.c_float, C.c_uint32, _p, _p, _p]),
    "t2v1_train_decoder_backward": (C.c_int, [_p, _p, _p, _p, _p, _p, _p]),
    "t2v1_encoder_trainer_create": (C.c_int, [_p, C.c_int, C.c_int, C.POINTER(_p)]),
    "t2v1_encoder_train_forward": (C.c_int, [_p, _p, _p, _p, C.c_float, C.c_uint32, _p]),
    "t2v1_encoder_train_forward_embedded": (C.c_int, [_p, _p, _p, _p, C.c_float, C.c_uint32, _p]),
    "t2v1_encoder_train_backward": (C.c_int, [_p, _p, _p, _p]),
    "t2_trainer_create": (C.c_int, [_p, C.c_int, C.c_int, C.c_int, C.POINTER(_p)]),
    "t2_trainer_destroy": (C.c_int, [_p]),
    "t2_train_decoder_forward": (C.c_int, [_p, _p, _p, _p, C.c_float, C.c_uint32, _p, _p, _p]),
    "t2_train_encoder_forward": (C.c_int, [_p, _p, _p, C.c_float, C.c_uint32, _p, _p]),
    "t2_train_encoder_forward_embedded": (C.c_int, [_p, _p, _p, C.c_float, C.c_uint32, _p, _p]),
    "t2_train_postnet_forward": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_float, C.c_uint32, _p, _p]),
    "t2_train_encoder_backward": (C.c_int, [_p, _p, _p]),
    "t2_dense": (C.c_int, [_p, _p, _p, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p]),
    "t2_dropout": (C.c_int, [_p, _p, C.c_longlong, C.c_int, C.c_float, C.c_uint32, C.c_int, C.c_int, _p]),
    "t2_zoneout_lstm_cell_train": (C.c_int, [_p, _p, _p, _p, _p, _p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_int,
                                             C.c_int, _p, _p, _p, _p]),
    "t2_zoneout_lstm_cell": (C.c_int, [_p, _p, _p, _p, _p, _p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, _p, _p, _p, _p]),
    "t2_trainer_load_params": (C.c_int, [_p, _p]),
    "t2_trainer_set_loss_counts": (C.c_int, [_p, _p]),
    "t2_trainer_update_moving_stats": (C.c_int, [_p, _p, C.c_float]),
    "t2_trainer_bind_gradient": (C.c_int, [_p, C.c_char_p, C.c_int64]),
    "t2_train_decoder_backward": (C.c_int, [_p, _p, _p, _p, _p, _p, _p]),
    "t2_spec_loss": (C.c_int, [_p, _p, _p, _p, C.c_int, C.c_int, C.c_int, _p]),
    "t2_spec_loss_noalloc": (C.c_int, [_p, _p, _p, _p, C.c_int, C.c_int, C.c_int, _p]),
    "t2_binary_loss": (C.c_int, [_p, _p, _p, _p, C.c_int, C.c_int, _p]),
    "t2_optim_create": (C.c_int, [_p, C.c_int64, C.c_float, C.c_float, C.c_float, C.POINTER(_p)]),
    "t2_optim_destroy": (C.c_int, [_p]),
    "t2_optim_global_norm": (C.c_int, [_p, _p, C.c_float, _p]),
    "t2_optim_step": (C.c_int, [_p, _p, _p, C.c_float, C.c_float, C.c_float]),
    "t2_optim_get_state": (C.c_int, [_p, _p, _p, _i64p]),
    "t2_optim_set_state": (C.c_int, [_p, _p, _p, C.c_int64]),
    "t2_griffin_lim": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, _p]),
}

_lib = None


def exported_symbols():
    return sorted(_SIGS)


def load():
    """Load libt2mi.so (built in-tree by tacotron2_amd/csrc/build.sh).  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: build it with tacotron2_amd/csrc/build.sh "
                          "(there is no CPU fallback for the HIP path)")
    # torch ships its own libamdhip64; import it first so that this library binds to the same HIP
    # runtime instance (two runtimes in one process do not share devices, streams or pointers).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code):
    if code != T2_OK:
        raise T2Error(code, load().t2_last_error().decode("utf-8", "replace"))
