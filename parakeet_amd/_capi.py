"""ctypes binding of include/pk_synth.h (libpk_synth.so).

There is no fallback: if the shared library is missing or a call fails, an
exception is raised.  Status codes map back to the exception classes the
reference raises at the same places.
"""
import ctypes as C
import os

# PyTorch-ROCm bundles its own libamdhip64; two HIP runtimes in one process cannot both own
# the device.  Importing torch first makes libpk_synth.so bind to the runtime torch loaded.
import torch  # noqa: F401  (load order matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
# PK_PROFILE_LIB=1 (read HERE, in Python -- the product library itself reads no environment variable): load the profile
# build (parakeet_amd.build.build(profile=True)), the one that carries the measurement / ablation switches tools/ uses.
PROFILE_LIB = os.environ.get("PK_PROFILE_LIB", "0") not in ("", "0")
LIB_PATH = os.path.join(_HERE, "libpk_synth_prof.so" if PROFILE_LIB else "libpk_synth.so")

PK_OK = 0
PK_HOST_IO = 1
PK_PWG_C_HAS_CONTEXT = 2
PK_APPLY_NORMALIZER = 4
PK_TTS_KEEP_ATT = 8
PK_PWG_MATH_F32, PK_PWG_MATH_BF16X3, PK_PWG_MATH_F16X3 = 0, 1, 2
_EXC = {
    -1: ValueError,
    -2: AssertionError,
    -3: NotImplementedError,
    -4: RuntimeError,
    -5: MemoryError,
    -6: RuntimeError,
}


class PwgCfg(C.Structure):
    _fields_ = [
        ("in_channels", C.c_int32), ("out_channels", C.c_int32), ("kernel_size", C.c_int32),
        ("layers", C.c_int32), ("stacks", C.c_int32), ("residual_channels", C.c_int32),
        ("gate_channels", C.c_int32), ("skip_channels", C.c_int32), ("aux_channels", C.c_int32),
        ("aux_context_window", C.c_int32), ("n_upsample", C.c_int32),
        ("upsample_scales", C.c_int32 * 8), ("use_causal_conv", C.c_int32),
    ]


class Fs2Cfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "idim", "odim", "adim", "aheads", "elayers", "eunits", "dlayers", "dunits",
        "positionwise_conv_kernel_size", "positionwise_layer_type",
        "duration_predictor_layers", "duration_predictor_chans", "duration_predictor_kernel_size",
        "pitch_predictor_layers", "pitch_predictor_chans", "pitch_predictor_kernel_size",
        "energy_predictor_layers", "energy_predictor_chans", "energy_predictor_kernel_size",
        "pitch_embed_kernel_size", "energy_embed_kernel_size",
        "postnet_layers", "postnet_chans", "postnet_filts",
        "use_batch_norm", "use_scaled_pos_enc", "encoder_normalize_before", "decoder_normalize_before",
        "reduction_factor", "num_speakers", "spk_embed_dim", "spk_embed_integration_type", "num_tones", "tone_embed_dim",
        "tone_embed_integration_type", "encoder_concat_after", "decoder_concat_after")]


class WfCfg(C.Structure):
    _fields_ = [("n_upsample", C.c_int32), ("upsample_factors", C.c_int32 * 4), ("n_flows", C.c_int32),
                ("n_layers", C.c_int32), ("n_group", C.c_int32), ("channels", C.c_int32), ("n_mels", C.c_int32),
                ("kernel_h", C.c_int32), ("kernel_w", C.c_int32)]


class SsCfg(C.Structure):
    _fields_ = [("vocab_size", C.c_int32), ("tone_size", C.c_int32), ("encoder_hidden_size", C.c_int32),
                ("encoder_kernel_size", C.c_int32), ("n_encoder_dilations", C.c_int32),
                ("encoder_dilations", C.c_int32 * 32), ("duration_predictor_hidden_size", C.c_int32),
                ("decoder_hidden_size", C.c_int32), ("decoder_output_size", C.c_int32),
                ("decoder_kernel_size", C.c_int32), ("n_decoder_dilations", C.c_int32),
                ("decoder_dilations", C.c_int32 * 32), ("same_padding_resets_dilation", C.c_int32)]


class TtsCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "idim", "odim", "embed_dim", "eprenet_conv_layers", "eprenet_conv_chans", "eprenet_conv_filts",
        "dprenet_layers", "dprenet_units", "adim", "aheads", "elayers", "eunits", "dlayers", "dunits",
        "postnet_layers", "postnet_chans", "postnet_filts", "positionwise_layer_type",
        "positionwise_conv_kernel_size", "use_scaled_pos_enc", "use_batch_norm", "encoder_normalize_before",
        "decoder_normalize_before", "encoder_concat_after", "decoder_concat_after", "reduction_factor",
        "spk_embed_dim", "use_gst", "spk_embed_integration_type", "gst_tokens", "gst_heads", "gst_conv_layers",
        "gst_conv_kernel_size", "gst_conv_stride", "gst_gru_layers", "gst_gru_units")] + [("gst_conv_chans", C.c_int32 * 8)]


class TacoCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "vocab_size", "n_tones", "d_mels", "reduction_factor", "d_encoder", "encoder_conv_layers",
        "encoder_kernel_size", "d_prenet", "d_attention_rnn", "d_decoder_rnn", "d_attention", "attention_filters",
        "attention_kernel_size", "d_postnet", "postnet_kernel_size", "postnet_conv_layers", "d_global_condition",
        "use_stop_token")] + [("p_prenet_dropout", C.c_float)]


class MelCfg(C.Structure):
    _fields_ = [("n_fft", C.c_int32), ("hop_length", C.c_int32), ("center", C.c_int32), ("power", C.c_int32),
                ("n_mels", C.c_int32), ("log_base", C.c_int32), ("log_floor", C.c_float)]


_lib = None


def _declare(lib):
    vp, i32, i64, f32p, i32p, i64p, cstr = (C.c_void_p, C.c_int32, C.c_int64, C.c_void_p,
                                            C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.c_char_p)
    sig = {
        "pk_last_error": (cstr, []),
        "pk_version": (cstr, []),
        "pk_ctx_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
        "pk_ctx_set_stream": (C.c_int, [vp, vp]),
        "pk_sync": (C.c_int, [vp]),
        "pk_ctx_destroy": (None, [vp]),
        "pk_prof_enable": (C.c_int, [vp, C.c_int]),
        "pk_prof_reset": (C.c_int, [vp]),
        "pk_prof_read": (C.c_int, [vp, cstr, C.POINTER(i64), C.POINTER(C.c_double)]),
        "pk_prof_dump": (C.c_int, [vp, C.c_char_p, i64]),
        "pk_pwg_create": (C.c_int, [vp, C.POINTER(PwgCfg), C.POINTER(vp)]),
        "pk_pwg_set_param": (C.c_int, [vp, cstr, f32p, i64p, i32]),
        "pk_pwg_set_normalizer": (C.c_int, [vp, f32p, f32p, i32]),
        "pk_randn": (C.c_int, [vp, f32p, i64, C.c_uint64, C.c_uint64, i32]),
        "pk_pwg_set_seed": (C.c_int, [vp, C.c_uint64]),
        "pk_wf_set_seed": (C.c_int, [vp, C.c_uint64]),
        "pk_pwg_set_math": (C.c_int, [vp, i32]),
        "pk_pwg_set_chunk_samples": (C.c_int, [vp, i64]),
        "pk_pwg_set_option": (C.c_int, [vp, cstr, i64]),
        "pk_pwg_scale_overshoot": (C.c_int, [vp, f32p, i32, i32p]),
        "pk_pwg_finalize": (C.c_int, [vp]),
        "pk_pwg_infer": (C.c_int, [vp, f32p, i32p, i32, f32p, f32p, i32]),
        "pk_pwg_debug_read": (C.c_int, [vp, i32, i32, f32p, i64]),
        "pk_pwg_destroy": (None, [vp]),
        "pk_fs2_create": (C.c_int, [vp, C.POINTER(Fs2Cfg), C.POINTER(vp)]),
        "pk_fs2_set_param": (C.c_int, [vp, cstr, f32p, i64p, i32]),
        "pk_fs2_set_normalizer": (C.c_int, [vp, f32p, f32p, i32]),
        "pk_fs2_set_math": (C.c_int, [vp, i32]),
        "pk_fs2_set_option": (C.c_int, [vp, cstr, i64]),
        "pk_fs2_set_speakers": (C.c_int, [vp, i64p, f32p, i32]),
        "pk_fs2_set_tones": (C.c_int, [vp, i64p, i64]),
        "pk_fs2_finalize": (C.c_int, [vp]),
        "pk_fs2_encode": (C.c_int, [vp, i64p, i32p, i32, C.c_float, i32p]),
        "pk_fs2_decode": (C.c_int, [vp, f32p, i32]),
        "pk_fs2_set_debug": (C.c_int, [vp, i32]),
        "pk_fs2_debug_read": (C.c_int, [vp, i32, i32, f32p, i64]),
        "pk_fs2_destroy": (None, [vp]),
        "pk_wf_create": (C.c_int, [vp, C.POINTER(WfCfg), C.POINTER(vp)]),
        "pk_wf_set_param": (C.c_int, [vp, cstr, f32p, i64p, i32]),
        "pk_wf_set_math": (C.c_int, [vp, i32]),
        "pk_wf_set_option": (C.c_int, [vp, cstr, i64]),
        "pk_wf_finalize": (C.c_int, [vp]),
        "pk_wf_cond_length": (C.c_int, [vp, i32, i32p, i32p]),
        "pk_wf_infer": (C.c_int, [vp, f32p, i32p, i32, f32p, f32p, i32]),
        "pk_wf_destroy": (None, [vp]),
        "pk_ss_create": (C.c_int, [vp, C.POINTER(SsCfg), C.POINTER(vp)]),
        "pk_ss_set_param": (C.c_int, [vp, cstr, f32p, i64p, i32]),
        "pk_ss_set_normalizer": (C.c_int, [vp, f32p, f32p, i32]),
        "pk_ss_set_math": (C.c_int, [vp, i32]),
        "pk_ss_finalize": (C.c_int, [vp]),
        "pk_ss_encode": (C.c_int, [vp, i64p, i64p, i32p, i32, i32p]),
        "pk_ss_decode": (C.c_int, [vp, f32p, i32]),
        "pk_ss_debug_read": (C.c_int, [vp, i32, i32, f32p, i64]),
        "pk_ss_destroy": (None, [vp]),
        "pk_tts_create": (C.c_int, [vp, C.POINTER(TtsCfg), C.POINTER(vp)]),
        "pk_tts_set_param": (C.c_int, [vp, cstr, f32p, i64p, i32]),
        "pk_tts_set_normalizer": (C.c_int, [vp, f32p, f32p, i32]),
        "pk_tts_set_math": (C.c_int, [vp, i32]),
        "pk_tts_set_option": (C.c_int, [vp, cstr, i64]),
        "pk_tts_set_dropout": (C.c_int, [vp, i32]),
        "pk_tts_set_speakers": (C.c_int, [vp, f32p, i32]),
        "pk_tts_set_style_reference": (C.c_int, [vp, f32p, i32p, i32]),
        "pk_tts_finalize": (C.c_int, [vp]),
        "pk_tts_infer": (C.c_int, [vp, i64p, i32p, i32, C.c_double, C.c_double, C.c_double, C.POINTER(C.c_uint64), i32,
                                   i32p]),
        "pk_tts_read": (C.c_int, [vp, f32p, f32p, f32p, i32]),
        "pk_tts_debug_read": (C.c_int, [vp, i32, i32, f32p, i64]),
        "pk_tts_destroy": (None, [vp]),
        "pk_taco_create": (C.c_int, [vp, C.POINTER(TacoCfg), C.POINTER(vp)]),
        "pk_taco_set_param": (C.c_int, [vp, cstr, f32p, i64p, i32]),
        "pk_taco_set_math": (C.c_int, [vp, i32]),
        "pk_taco_set_dropout": (C.c_int, [vp, i32]),
        "pk_taco_finalize": (C.c_int, [vp]),
        "pk_taco_set_global_condition": (C.c_int, [vp, f32p, i32]),
        "pk_taco_infer": (C.c_int, [vp, i64p, i64p, i32p, i32, i32, C.POINTER(C.c_uint64), i32, i32p]),
        "pk_taco_read": (C.c_int, [vp, f32p, f32p, f32p, f32p, i32]),
        "pk_taco_debug_read": (C.c_int, [vp, i32, i32, f32p, i64]),
        "pk_taco_destroy": (None, [vp]),
        "pk_mel_create": (C.c_int, [vp, C.POINTER(MelCfg), f32p, f32p, C.POINTER(vp)]),
        "pk_mel_num_frames": (C.c_int, [vp, i32, i32p]),
        "pk_mel_run": (C.c_int, [vp, f32p, i32p, i32, f32p, i32, i32]),
        "pk_mel_destroy": (None, [vp]),
        "pk_op_expand": (C.c_int, [vp, f32p, i64p, i32, i32, i32, i32, f32p]),
        "pk_op_sinusoid_position_encoding": (C.c_int, [vp, i32, i32, C.c_float, i32, f32p]),
        "pk_op_scaled_dot_product_attention": (C.c_int, [vp, f32p, f32p, f32p, f32p, i32, i32, i32, i32, i32, i32,
                                                         f32p, f32p]),
        "pk_op_matmul": (C.c_int, [vp, f32p, i32, i32, i32, f32p, f32p, f32p]),
        "pk_op_conv1d_cell_step": (C.c_int, [vp, f32p, f32p, f32p, f32p, i32, i32, i32, i32, i32, f32p]),
        "pk_op_conv1d_batchnorm_nlc": (C.c_int, [vp, f32p, i32, i32, i32, i32, i32, i32, f32p, f32p, f32p, f32p,
                                                 f32p, f32p, C.c_float, f32p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    return sig


def lib():
    """Load libpk_synth.so (built by parakeet_amd.build / __graft_entry__.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP engine is not built "
                "(run `python -m parakeet_amd.build`); there is no CPU fallback")
        # The library must have been built from the sources next to it.  PK_ALLOW_STALE_LIB skips the check; so does an
        # install that ships the library without csrc/ or include/ (nothing to compare with).
        if not os.environ.get("PK_ALLOW_STALE_LIB"):
            from . import build as _build
            try:
                want = _build.source_hash()
            except FileNotFoundError:
                want = None
            have = _build.library_hash(LIB_PATH) if want is not None else None
            if have != want:
                raise RuntimeError(
                    f"{LIB_PATH} was built from other sources (library {str(have)[:12]}, tree {want[:12]}): "
                    "run `python -m parakeet_amd.build` (or __graft_entry__.build()); a stale engine is never used")
        _lib = C.CDLL(LIB_PATH)
        _declare(_lib)
    return _lib


def check(status):
    if status != PK_OK:
        msg = lib().pk_last_error().decode("utf-8", "replace")
        raise _EXC.get(status, RuntimeError)(f"pk_synth[{status}]: {msg}")


def fptr(arr):
    """Pointer of a C-contiguous float32 numpy array."""
    return arr.ctypes.data_as(C.c_void_p)
