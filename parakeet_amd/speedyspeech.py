"""SpeedySpeech on the HIP engine -- same class names and call conventions as
parakeet/models/speedyspeech/speedyspeech.py (SpeedySpeech :142-218, SpeedySpeechInference :221-231).

Extensions (supersets): ``inference_batch`` for ragged batches, and ``same_padding_resets_dilation`` (see
include/pk_synth.h, pk_ss_cfg): True (default) reproduces Paddle's conv kernels, which ignore the dilation
under padding="same"; False computes the dilated convolutions as the source is written."""
import ctypes as C

import numpy as np
import torch

from . import _capi
from .runtime import Context, dptr, set_params, to_numpy_f32, wrap


def _ids(v):
    return np.asarray(v.cpu() if isinstance(v, torch.Tensor) else v).astype(np.int64).reshape(-1)


class SpeedySpeech:
    def __init__(self, vocab_size, encoder_hidden_size, encoder_kernel_size, encoder_dilations,
                 duration_predictor_hidden_size, decoder_hidden_size, decoder_output_size, decoder_kernel_size,
                 decoder_dilations, tone_size=None, same_padding_resets_dilation=True, device=None):
        self._ctx = Context.get(device)
        self.odim = decoder_output_size
        self._hidden = encoder_hidden_size
        self.training = True
        cfg = _capi.SsCfg()
        cfg.vocab_size = vocab_size
        cfg.tone_size = int(tone_size or 0)
        cfg.encoder_hidden_size, cfg.encoder_kernel_size = encoder_hidden_size, encoder_kernel_size
        cfg.duration_predictor_hidden_size = duration_predictor_hidden_size
        cfg.decoder_hidden_size, cfg.decoder_output_size = decoder_hidden_size, decoder_output_size
        cfg.decoder_kernel_size = decoder_kernel_size
        if len(encoder_dilations) > 32 or len(decoder_dilations) > 32:
            raise NotImplementedError("at most 32 residual blocks per stack")
        cfg.n_encoder_dilations, cfg.n_decoder_dilations = len(encoder_dilations), len(decoder_dilations)
        for i, d in enumerate(encoder_dilations):
            cfg.encoder_dilations[i] = int(d)
        for i, d in enumerate(decoder_dilations):
            cfg.decoder_dilations[i] = int(d)
        cfg.same_padding_resets_dilation = 1 if same_padding_resets_dilation else 0
        h = C.c_void_p()
        _capi.check(self._ctx.lib.pk_ss_create(self._ctx.handle, C.byref(cfg), C.byref(h)))
        self._h = h
        self._finalized = False
        self._last_tok, self._last_frames = [], []

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self._ctx.lib.pk_ss_destroy(h)
            except Exception:
                pass

    def set_state_dict(self, state_dict):
        set_params(self._ctx.lib.pk_ss_set_param, self._h, state_dict)
        self._finalized = False

    def eval(self):
        self.training = False
        return self

    def set_normalizer(self, normalizer):
        """Register ZScore statistics; applied only by calls passing ``denormalize=True`` (see FastSpeech2)."""
        self._norm_owner = None
        if normalizer is None:
            _capi.check(self._ctx.lib.pk_ss_set_normalizer(self._h, None, None, 0))
        else:
            mu, sigma = to_numpy_f32(normalizer.mu).reshape(-1), to_numpy_f32(normalizer.sigma).reshape(-1)
            _capi.check(self._ctx.lib.pk_ss_set_normalizer(self._h, _capi.fptr(mu), _capi.fptr(sigma), mu.size))
        self._finalized = False

    def set_math(self, mode):
        """'f16x3' (default: split-fp16 MFMA GEMMs, fp32-equivalent error) or 'f32' (exact fp32 MFMA)."""
        _capi.check(self._ctx.lib.pk_ss_set_math(self._h, {"f32": 0, "f16x3": 1}[mode]))

    def _finalize(self):
        if not self._finalized:
            _capi.check(self._ctx.lib.pk_ss_finalize(self._h))
            self._finalized = True

    def encode_batch(self, texts, tones=None):
        ctx = Context.get(self._ctx.device)
        self._finalize()
        ids = [_ids(t) for t in texts]
        lens = np.array([len(i) for i in ids], dtype=np.int32)
        flat = np.ascontiguousarray(np.concatenate(ids))
        tflat = None
        if tones is not None:
            tn = [_ids(t) for t in tones]
            assert [len(t) for t in tn] == [len(i) for i in ids], "one tone per phone"
            tflat = np.ascontiguousarray(np.concatenate(tn))
        frames = np.zeros(len(ids), dtype=np.int32)
        i64p = C.POINTER(C.c_int64)
        _capi.check(ctx.lib.pk_ss_encode(self._h, flat.ctypes.data_as(i64p),
                                         None if tflat is None else tflat.ctypes.data_as(i64p),
                                         lens.ctypes.data_as(C.POINTER(C.c_int32)), len(ids),
                                         frames.ctypes.data_as(C.POINTER(C.c_int32))))
        self._last_tok, self._last_frames = [int(v) for v in lens], [int(v) for v in frames]
        return frames

    def decode_packed(self, denormalize=False):
        ctx = Context.get(self._ctx.device)
        total = int(sum(self._last_frames))
        mel = ctx.empty((total, self.odim))
        if total:
            _capi.check(ctx.lib.pk_ss_decode(self._h, dptr(mel), _capi.PK_APPLY_NORMALIZER if denormalize else 0))
        return mel

    def inference_batch(self, texts, tones=None, denormalize=False):
        """Lists of (T_b,) phone / tone ids -> list of (L_b, output_size) device tensors."""
        frames = self.encode_batch(texts, tones)
        mel = self.decode_packed(denormalize)
        outs, o = [], 0
        for f in frames:
            outs.append(wrap(mel[o:o + int(f)]))
            o += int(f)
        return outs

    def inference(self, text, tones=None, denormalize=False):
        """(T,) int -> (L, output_size); speedyspeech.py:178-218."""
        return self.inference_batch([text], None if tones is None else [tones], denormalize)[0]

    def debug_tap(self, what, b):
        T = self._last_tok[b]
        out = np.empty((T, self._hidden) if what == 0 else (T,), dtype=np.float32)
        _capi.check(self._ctx.lib.pk_ss_debug_read(self._h, what, b, _capi.fptr(out), out.size))
        return out


class SpeedySpeechInference:
    """SpeedySpeechInference (speedyspeech.py:221-231): inference then normalizer.inverse."""

    def __init__(self, normalizer, speedyspeech_model):
        self.normalizer = normalizer
        self.acoustic_model = speedyspeech_model
        self.bind()

    def bind(self):
        m = self.acoustic_model
        if getattr(m, "_norm_owner", None) is not self:
            m.set_normalizer(self.normalizer)
            m._norm_owner = self
        return m

    def forward(self, phones, tones=None):
        return self.bind().inference(phones, tones, denormalize=True)

    __call__ = forward

    def eval(self):
        return self
