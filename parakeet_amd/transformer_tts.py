"""TransformerTTS acoustic model behind the reference's Python API.

Mirrors parakeet/models/transformer_tts/transformer_tts.py: ``TransformerTTS`` (constructor kwargs :172-250,
``set_state_dict``, ``eval``, ``inference`` :511-647 -> (outs, probs, att_ws)) and ``TransformerTTSInference``
(:757-767).  All arithmetic runs in libpk_synth.so (csrc/tts.hip on the shared transformer machinery of csrc/fs2.hip).
Training (``forward`` / loss) and teacher forcing are out of scope.

The reference's decoder prenet keeps dropout on at inference (modules/tacotron2/decoder.py:78-81), so its output
depends on Paddle's random generator.  Here the mask comes from the engine's counter-based dropout stream
(include/pk_synth.h): ``seed=`` selects it, the same seed gives the same spectrogram on any batch composition.

Extensions (supersets): ``inference_batch`` decodes a ragged batch in lockstep; ``seed`` / ``dropout``.
"""
import ctypes as C

import numpy as np
import torch

from . import _capi
from .runtime import Context, dptr, set_params, to_numpy_f32, wrap


def _ids(v):
    if hasattr(v, "numpy") and not isinstance(v, (np.ndarray, torch.Tensor)):
        v = v.numpy()
    return np.asarray(v.cpu() if isinstance(v, torch.Tensor) else v).astype(np.int64).reshape(-1)


class TransformerTTS:
    def __init__(self, idim, odim, embed_dim=512, eprenet_conv_layers=3, eprenet_conv_chans=256, eprenet_conv_filts=5,
                 dprenet_layers=2, dprenet_units=256, elayers=6, eunits=1024, adim=512, aheads=4, dlayers=6,
                 dunits=1024, postnet_layers=5, postnet_chans=256, postnet_filts=5, positionwise_layer_type="conv1d",
                 positionwise_conv_kernel_size=1, use_scaled_pos_enc=True, use_batch_norm=True,
                 encoder_normalize_before=True, decoder_normalize_before=True, encoder_concat_after=False,
                 decoder_concat_after=False, reduction_factor=1, spk_embed_dim=None, spk_embed_integration_type="add",
                 use_gst=False, gst_tokens=10, gst_heads=4, gst_conv_layers=6,
                 gst_conv_chans_list=(32, 32, 64, 64, 128, 128), gst_conv_kernel_size=3, gst_conv_stride=2,
                 gst_gru_layers=1, gst_gru_units=128, transformer_enc_dropout_rate=0.1,
                 transformer_enc_positional_dropout_rate=0.1, transformer_enc_attn_dropout_rate=0.1,
                 transformer_dec_dropout_rate=0.1, transformer_dec_positional_dropout_rate=0.1,
                 transformer_dec_attn_dropout_rate=0.1, transformer_enc_dec_attn_dropout_rate=0.1,
                 eprenet_dropout_rate=0.5, dprenet_dropout_rate=0.5, postnet_dropout_rate=0.5,
                 init_type="xavier_uniform", init_enc_alpha=1.0, init_dec_alpha=1.0, use_guided_attn_loss=True,
                 num_heads_applied_guided_attn=2, num_layers_applied_guided_attn=2, device=None):
        if positionwise_layer_type not in ("conv1d", "linear", "conv1d-linear"):
            raise NotImplementedError("Support only linear or conv1d.")   # encoder.py:169
        self.idim, self.odim = idim, odim
        self.eos = idim - 1
        self.reduction_factor = reduction_factor
        self.padding_idx = 0
        self.training = True
        self._adim, self._aheads, self._dlayers = adim, aheads, dlayers
        self.spk_embed_dim = spk_embed_dim
        if spk_embed_dim is not None and spk_embed_integration_type not in ("add", "concat"):
            raise NotImplementedError("support only add or concat.")   # transformer_tts.py:753
        self._ctx = Context.get(device)
        cfg = _capi.TtsCfg()
        cfg.idim, cfg.odim = idim, odim
        cfg.embed_dim, cfg.eprenet_conv_layers = embed_dim, eprenet_conv_layers
        cfg.eprenet_conv_chans, cfg.eprenet_conv_filts = eprenet_conv_chans, eprenet_conv_filts
        cfg.dprenet_layers, cfg.dprenet_units = dprenet_layers, dprenet_units
        cfg.adim, cfg.aheads = adim, aheads
        cfg.elayers, cfg.eunits, cfg.dlayers, cfg.dunits = elayers, eunits, dlayers, dunits
        cfg.postnet_layers, cfg.postnet_chans, cfg.postnet_filts = postnet_layers, postnet_chans, postnet_filts
        cfg.positionwise_layer_type = {"conv1d": 0, "linear": 1, "conv1d-linear": 2}[positionwise_layer_type]
        cfg.positionwise_conv_kernel_size = positionwise_conv_kernel_size
        cfg.use_scaled_pos_enc = 1 if use_scaled_pos_enc else 0
        cfg.use_batch_norm = 1 if use_batch_norm else 0
        cfg.encoder_normalize_before = 1 if encoder_normalize_before else 0
        cfg.decoder_normalize_before = 1 if decoder_normalize_before else 0
        cfg.encoder_concat_after = 1 if encoder_concat_after else 0
        cfg.decoder_concat_after = 1 if decoder_concat_after else 0
        cfg.reduction_factor = reduction_factor
        cfg.spk_embed_dim = 0 if spk_embed_dim is None else int(spk_embed_dim)
        cfg.use_gst = 1 if use_gst else 0
        self.use_gst = bool(use_gst)
        if use_gst:
            chans = [int(v) for v in gst_conv_chans_list]
            if len(chans) != gst_conv_layers:   # style_encoder.py:153-155
                raise ValueError("the number of conv layers and length of channels list must be the same.")
            if len(chans) > 8:
                raise NotImplementedError("at most 8 reference-encoder conv layers")
            cfg.gst_tokens, cfg.gst_heads = gst_tokens, gst_heads
            cfg.gst_conv_layers, cfg.gst_conv_kernel_size, cfg.gst_conv_stride = gst_conv_layers, gst_conv_kernel_size, gst_conv_stride
            cfg.gst_gru_layers, cfg.gst_gru_units = gst_gru_layers, gst_gru_units
            for i, v in enumerate(chans):
                cfg.gst_conv_chans[i] = v
        cfg.spk_embed_integration_type = 1 if spk_embed_integration_type == "concat" else 0
        h = C.c_void_p()
        _capi.check(self._ctx.lib.pk_tts_create(self._ctx.handle, C.byref(cfg), C.byref(h)))
        self._h = h
        self._finalized = False
        self._last_tok, self._last_frames = [], []

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self._ctx.lib.pk_tts_destroy(h)
            except Exception:
                pass

    def set_state_dict(self, state_dict):
        set_params(self._ctx.lib.pk_tts_set_param, self._h, state_dict)
        self._finalized = False

    def eval(self):
        self.training = False
        return self

    def set_normalizer(self, normalizer):
        """Register ZScore statistics; applied only by calls passing ``denormalize=True`` (TransformerTTSInference)."""
        self._norm_owner = None
        if normalizer is None:
            _capi.check(self._ctx.lib.pk_tts_set_normalizer(self._h, None, None, 0))
        else:
            mu, sigma = to_numpy_f32(normalizer.mu).reshape(-1), to_numpy_f32(normalizer.sigma).reshape(-1)
            _capi.check(self._ctx.lib.pk_tts_set_normalizer(self._h, _capi.fptr(mu), _capi.fptr(sigma), mu.size))
        self._finalized = False

    def set_math(self, mode):
        """'f16x3' (default: split-fp16 MFMA GEMMs, fp32-equivalent error) or 'f32' (exact fp32 MFMA)."""
        _capi.check(self._ctx.lib.pk_tts_set_math(self._h, {"f32": 0, "f16x3": 1}[mode]))

    def set_option(self, key, value):
        """Named integer options of the engine handle (include/pk_synth.h, pk_tts_set_option): 'kv_prefix' and the FFT-stack
        options of FastSpeech2.set_option."""
        _capi.check(self._ctx.lib.pk_tts_set_option(self._h, key.encode(), int(value)))

    def set_dropout(self, on):
        """False switches the decoder prenet's dropout off (deterministic; not what the reference computes)."""
        _capi.check(self._ctx.lib.pk_tts_set_dropout(self._h, 1 if on else 0))

    def _finalize(self):
        if not self._finalized:
            _capi.check(self._ctx.lib.pk_tts_finalize(self._h))
            self._finalized = True

    def inference_batch(self, texts, threshold=0.5, minlenratio=0.0, maxlenratio=10.0, seeds=None,
                        return_att=True, denormalize=False, spembs=None, speech=None):
        """Lists of (T_b,) token ids (without <eos>) -> list of (outs (L_b, odim), probs (L_b,),
        att_ws (dlayers, aheads, L_b / reduction_factor, T_b + 1) or None) device tensors.  ``spembs``: (B, spk_embed_dim), one speaker
        embedding per utterance, for a model built with ``spk_embed_dim``.  ``speech``: list of (L_b, odim) reference
        spectrograms, one per utterance, for a ``use_gst`` model."""
        ctx = Context.get(self._ctx.device)
        self._finalize()
        if speech is not None and self.use_gst:
            refs = [to_numpy_f32(y).reshape(-1, self.odim) for y in speech]
            if len(refs) != len(texts):
                raise ValueError("one reference spectrogram per utterance")
            rl = np.array([r.shape[0] for r in refs], dtype=np.int32)
            flat_ref = np.ascontiguousarray(np.concatenate(refs, axis=0))
            _capi.check(ctx.lib.pk_tts_set_style_reference(self._h, _capi.fptr(flat_ref), rl.ctypes.data_as(C.POINTER(C.c_int32)),
                                                           len(refs)))
        if spembs is not None:
            e = to_numpy_f32(spembs).reshape(len(texts), -1)
            if e.shape[1] != (self.spk_embed_dim or 0):
                raise ValueError(f"spembs has {e.shape[1]} columns, the model was built with spk_embed_dim={self.spk_embed_dim}")
            _capi.check(ctx.lib.pk_tts_set_speakers(self._h, _capi.fptr(e), e.shape[0]))
        ids = [_ids(t) for t in texts]
        B = len(ids)
        lens = np.array([len(i) for i in ids], dtype=np.int32)
        flat = np.ascontiguousarray(np.concatenate(ids)) if lens.sum() else np.zeros(1, np.int64)
        frames = np.zeros(B, dtype=np.int32)
        sd = None
        if seeds is not None:
            sd = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint64).reshape(-1))
            assert sd.size == B, "one dropout seed per utterance"
        flags = _capi.PK_TTS_KEEP_ATT if return_att else 0
        _capi.check(ctx.lib.pk_tts_infer(self._h, flat.ctypes.data_as(C.POINTER(C.c_int64)),
                                         lens.ctypes.data_as(C.POINTER(C.c_int32)), B, float(threshold),
                                         float(minlenratio), float(maxlenratio),
                                         None if sd is None else sd.ctypes.data_as(C.POINTER(C.c_uint64)), flags,
                                         frames.ctypes.data_as(C.POINTER(C.c_int32))))
        self._last_tok, self._last_frames = [int(v) + 1 for v in lens], [int(v) for v in frames]
        steps = [L // self.reduction_factor for L in self._last_frames]     # one attention row per decoder step
        total = int(frames.sum())
        mel = ctx.empty((total, self.odim))
        probs = ctx.empty((total,))
        att = None
        if return_att:
            n_att = sum(self._dlayers * self._aheads * S * T for S, T in zip(steps, self._last_tok))
            att = ctx.empty((n_att,))
        _capi.check(ctx.lib.pk_tts_read(self._h, dptr(mel), dptr(probs), None if att is None else dptr(att),
                                        _capi.PK_APPLY_NORMALIZER if denormalize else 0))
        outs, o, oa = [], 0, 0
        for L, S, T in zip(self._last_frames, steps, self._last_tok):
            a = None
            if att is not None:
                n = self._dlayers * self._aheads * S * T
                a = wrap(att[oa:oa + n].view(self._dlayers, self._aheads, S, T))
                oa += n
            outs.append((wrap(mel[o:o + L]), wrap(probs[o:o + L]), a))
            o += L
        return outs

    def inference(self, text, speech=None, spembs=None, threshold=0.5, minlenratio=0.0, maxlenratio=10.0,
                  use_teacher_forcing=False, seed=0, denormalize=False):
        """(T,) int64 -> (outs (L, odim), probs (L,), att_ws (#layers, #heads, L, T + 1)); transformer_tts.py:511-647."""
        if use_teacher_forcing:
            raise NotImplementedError("teacher forcing needs the training graph (transformer_tts.py:568-582)")
        # ``speech`` feeds teacher forcing (refused above) and the style encoder (:552-588); ignored otherwise
        return self.inference_batch([text], threshold, minlenratio, maxlenratio, [seed], True, denormalize,
                                    None if spembs is None else to_numpy_f32(spembs).reshape(1, -1),
                                    None if (speech is None or not self.use_gst) else [speech])[0]

    def debug_tap(self, what, b):
        """0: encoder output (T_b + 1, adim); 1: outs before the postnet (L_b, odim); 2: last decoder layer (L_b, adim)."""
        rows = (self._last_tok[b] if what == 0 else self._last_frames[b] if what == 1
                else self._last_frames[b] // self.reduction_factor)
        out = np.empty((rows, self.odim if what == 1 else self._adim), dtype=np.float32)
        _capi.check(self._ctx.lib.pk_tts_debug_read(self._h, what, b, _capi.fptr(out), out.size))
        return out


class TransformerTTSInference:
    """TransformerTTSInference (transformer_tts.py:757-767): inference()[0] then normalizer.inverse."""

    def __init__(self, normalizer, model):
        self.normalizer = normalizer
        self.acoustic_model = model
        self.bind()

    def bind(self):
        m = self.acoustic_model
        if getattr(m, "_norm_owner", None) is not self:
            m.set_normalizer(self.normalizer)
            m._norm_owner = self
        return m

    def forward(self, text, spk_id=None, seed=0):
        return self.bind().inference(text, seed=seed, denormalize=True)[0]

    __call__ = forward

    def eval(self):
        return self
