"""Tacotron2 acoustic model behind the reference's Python API.

Mirrors parakeet/models/tacotron2.py: ``Tacotron2`` (constructor kwargs :626-649, ``set_state_dict``, ``eval``,
``infer`` :781-840 -> dict of mel_output / mel_outputs_postnet / alignments [/ stop_logits]).  All arithmetic runs in
libpk_synth.so (csrc/taco2.hip).  Training (``forward`` / loss) is out of scope; reduction_factor > 1 is refused (the
reference's ``infer`` cannot run it either: the postnet gets the (B, T, d_mels * r) decoder output, :822-826).

The decoder prenet keeps dropout on at inference (:76-79, training=True); the mask comes from the engine's
counter-based dropout stream (include/pk_synth.h), selected by ``seed=``.

Extension (superset): ``infer_batch`` decodes a ragged batch in lockstep, every utterance with its own stop.
"""
import ctypes as C

import numpy as np
import torch

from . import _capi
from .runtime import Context, dptr, set_params, to_numpy_f32, wrap


def _ids(v):
    if hasattr(v, "numpy") and not isinstance(v, (np.ndarray, torch.Tensor)):
        v = v.numpy()
    return np.asarray(v.cpu() if isinstance(v, torch.Tensor) else v).astype(np.int64)


class Tacotron2:
    def __init__(self, vocab_size, n_tones=None, d_mels=80, d_encoder=512, encoder_conv_layers=3, encoder_kernel_size=5,
                 d_prenet=256, d_attention_rnn=1024, d_decoder_rnn=1024, attention_filters=32, attention_kernel_size=31,
                 d_attention=128, d_postnet=512, postnet_kernel_size=5, postnet_conv_layers=5, reduction_factor=1,
                 p_encoder_dropout=0.5, p_prenet_dropout=0.5, p_attention_dropout=0.1, p_decoder_dropout=0.1,
                 p_postnet_dropout=0.5, d_global_condition=None, use_stop_token=False, device=None):
        self.toned = n_tones is not None
        self.d_mels, self.d_encoder = d_mels, d_encoder
        self.d_global_condition = int(d_global_condition or 0)
        self.use_stop_token = bool(use_stop_token)
        self.training = True
        self._ctx = Context.get(device)
        cfg = _capi.TacoCfg()
        cfg.vocab_size, cfg.n_tones = vocab_size, int(n_tones or 0)
        cfg.d_mels, cfg.reduction_factor = d_mels, reduction_factor
        cfg.d_encoder, cfg.encoder_conv_layers, cfg.encoder_kernel_size = d_encoder, encoder_conv_layers, encoder_kernel_size
        cfg.d_prenet, cfg.d_attention_rnn, cfg.d_decoder_rnn = d_prenet, d_attention_rnn, d_decoder_rnn
        cfg.d_attention, cfg.attention_filters = d_attention, attention_filters
        cfg.attention_kernel_size = attention_kernel_size
        cfg.d_postnet, cfg.postnet_kernel_size, cfg.postnet_conv_layers = d_postnet, postnet_kernel_size, postnet_conv_layers
        cfg.d_global_condition = int(d_global_condition or 0)
        cfg.use_stop_token = 1 if use_stop_token else 0
        cfg.p_prenet_dropout = float(p_prenet_dropout)
        h = C.c_void_p()
        _capi.check(self._ctx.lib.pk_taco_create(self._ctx.handle, C.byref(cfg), C.byref(h)))
        self._h = h
        self._finalized = False
        self._last_tok, self._last_frames = [], []

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self._ctx.lib.pk_taco_destroy(h)
            except Exception:
                pass

    def set_state_dict(self, state_dict):
        set_params(self._ctx.lib.pk_taco_set_param, self._h, state_dict)
        self._finalized = False

    def eval(self):
        self.training = False
        return self

    def set_math(self, mode):
        """'f16x3' (default: split-fp16 MFMA GEMMs, fp32-equivalent error) or 'f32' (exact fp32 MFMA)."""
        _capi.check(self._ctx.lib.pk_taco_set_math(self._h, {"f32": 0, "f16x3": 1}[mode]))

    def set_dropout(self, on):
        """False switches the decoder prenet's dropout off (deterministic; not what the reference computes)."""
        _capi.check(self._ctx.lib.pk_taco_set_dropout(self._h, 1 if on else 0))

    def _finalize(self):
        if not self._finalized:
            _capi.check(self._ctx.lib.pk_taco_finalize(self._h))
            self._finalized = True

    def infer_batch(self, texts, max_decoder_steps=1000, tones=None, seeds=None, global_condition=None):
        """Lists of (T_b,) ids (and tone ids) -> list of dicts like ``infer`` returns, without the batch axis.
        ``global_condition``: (B, d_global_condition), one row per utterance (:816-821)."""
        ctx = Context.get(self._ctx.device)
        self._finalize()
        if global_condition is not None:
            g = to_numpy_f32(global_condition).reshape(len(texts), -1)
            if g.shape[1] != self.d_global_condition:
                raise ValueError(f"global_condition has {g.shape[1]} columns, the model was built with "
                                 f"d_global_condition={self.d_global_condition or None}")
            _capi.check(ctx.lib.pk_taco_set_global_condition(self._h, _capi.fptr(g), g.shape[0]))
        ids = [_ids(t).reshape(-1) for t in texts]
        B = len(ids)
        lens = np.array([len(i) for i in ids], dtype=np.int32)
        flat = np.ascontiguousarray(np.concatenate(ids))
        tflat = None
        if tones is not None:
            tn = [_ids(t).reshape(-1) for t in tones]
            assert [len(t) for t in tn] == [len(i) for i in ids], "one tone per token"
            tflat = np.ascontiguousarray(np.concatenate(tn))
        sd = None
        if seeds is not None:
            sd = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint64).reshape(-1))
            assert sd.size == B, "one dropout seed per utterance"
        frames = np.zeros(B, dtype=np.int32)
        i64p = C.POINTER(C.c_int64)
        _capi.check(ctx.lib.pk_taco_infer(self._h, flat.ctypes.data_as(i64p),
                                          None if tflat is None else tflat.ctypes.data_as(i64p),
                                          lens.ctypes.data_as(C.POINTER(C.c_int32)), B, int(max_decoder_steps),
                                          None if sd is None else sd.ctypes.data_as(C.POINTER(C.c_uint64)), 0,
                                          frames.ctypes.data_as(C.POINTER(C.c_int32))))
        self._last_tok, self._last_frames = [int(v) for v in lens], [int(v) for v in frames]
        total = int(frames.sum())
        mel, post = ctx.empty((total, self.d_mels)), ctx.empty((total, self.d_mels))
        align = ctx.empty((int(sum(L * T for L, T in zip(self._last_frames, self._last_tok))),))
        stop = ctx.empty((total,)) if self.use_stop_token else None
        _capi.check(ctx.lib.pk_taco_read(self._h, dptr(mel), dptr(post), dptr(align),
                                         None if stop is None else dptr(stop), 0))
        outs, o, oa = [], 0, 0
        for L, T in zip(self._last_frames, self._last_tok):
            d = {"mel_output": wrap(mel[o:o + L]), "mel_outputs_postnet": wrap(post[o:o + L]),
                 "alignments": wrap(align[oa:oa + L * T].view(L, T))}
            if stop is not None:
                d["stop_logits"] = wrap(stop[o:o + L])
            outs.append(d)
            o += L
            oa += L * T
        return outs

    def infer(self, text_inputs, max_decoder_steps=1000, tones=None, global_condition=None, seed=0):
        """text_inputs (1, T) [or (T,)] int64 -> {"mel_output": (1, L, C), "mel_outputs_postnet": (1, L, C),
        "alignments": (1, L, T), "stop_logits": (1, L) with a stop token}; tacotron2.py:781-840."""
        x = _ids(text_inputs)
        if x.ndim == 2 and x.shape[0] != 1:
            raise ValueError("infer() takes one utterance (the reference's stop test needs batch size 1, "
                             "tacotron2.py:515-521); use infer_batch for several")
        t = None if tones is None else [_ids(tones).reshape(-1)]
        o = self.infer_batch([x.reshape(-1)], max_decoder_steps, t, [seed], global_condition)[0]
        return {k: wrap(v.unsqueeze(0)) for k, v in o.items()}

    @classmethod
    def from_pretrained(cls, config, checkpoint_path):
        """Tacotron2.from_pretrained (tacotron2.py:843-883): config with ``model`` / ``data`` sections, checkpoint path
        without the ``.pdparams`` suffix."""
        from .checkpoint import load_tacotron2
        return load_tacotron2(config, checkpoint_path)

    def debug_tap(self, what, b):
        """0: encoder outputs (T_b, d_encoder)."""
        out = np.empty((self._last_tok[b], self.d_encoder), dtype=np.float32)
        _capi.check(self._ctx.lib.pk_taco_debug_read(self._h, what, b, _capi.fptr(out), out.size))
        return out
