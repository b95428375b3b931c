"""FastSpeech2 acoustic model behind the reference's Python API.

Mirrors parakeet/models/fastspeech2/fastspeech2.py: ``FastSpeech2`` (constructor
kwargs :52-118, ``set_state_dict``, ``eval``, ``inference`` :468-558) and
``FastSpeech2Inference`` (:662-671).  All arithmetic runs in libpk_synth.so
(csrc/fs2.hip, csrc/gemm.hip).  Training (``forward`` / loss) is out of scope.

Extension over the reference: ``inference_batch`` runs a ragged batch in one
engine call (the reference's ``inference`` is one utterance per call).
"""
import ctypes as C

import numpy as np
import torch

from . import _capi
from .runtime import Context, dptr, set_params, to_numpy_f32, wrap


class FastSpeech2:
    def __init__(self, idim, odim, adim=384, aheads=4, elayers=6, eunits=1536, dlayers=6, dunits=1536,
                 postnet_layers=5, postnet_chans=512, postnet_filts=5, positionwise_layer_type="conv1d",
                 positionwise_conv_kernel_size=1, use_scaled_pos_enc=True, use_batch_norm=True,
                 encoder_normalize_before=True, decoder_normalize_before=True, encoder_concat_after=False,
                 decoder_concat_after=False, reduction_factor=1, encoder_type="transformer",
                 decoder_type="transformer", duration_predictor_layers=2, duration_predictor_chans=384,
                 duration_predictor_kernel_size=3, energy_predictor_layers=2, energy_predictor_chans=384,
                 energy_predictor_kernel_size=3, energy_predictor_dropout=0.5, energy_embed_kernel_size=9,
                 energy_embed_dropout=0.5, stop_gradient_from_energy_predictor=False,
                 pitch_predictor_layers=2, pitch_predictor_chans=384, pitch_predictor_kernel_size=3,
                 pitch_predictor_dropout=0.5, pitch_embed_kernel_size=9, pitch_embed_dropout=0.5,
                 stop_gradient_from_pitch_predictor=False, num_speakers=None, spk_embed_dim=None,
                 spk_embed_integration_type="add", num_tones=None, tone_embed_dim=None,
                 tone_embed_integration_type="add", transformer_enc_dropout_rate=0.1,
                 transformer_enc_positional_dropout_rate=0.1, transformer_enc_attn_dropout_rate=0.1,
                 transformer_dec_dropout_rate=0.1, transformer_dec_positional_dropout_rate=0.1,
                 transformer_dec_attn_dropout_rate=0.1, duration_predictor_dropout_rate=0.1,
                 postnet_dropout_rate=0.5, init_type="xavier_uniform", init_enc_alpha=1.0,
                 init_dec_alpha=1.0, use_masking=False, use_weighted_masking=False, device=None):
        if encoder_type != "transformer":
            raise ValueError(f"{encoder_type} is not supported.")   # fastspeech2.py:187
        if decoder_type != "transformer":
            raise ValueError(f"{decoder_type} is not supported.")   # fastspeech2.py:268
        if positionwise_layer_type not in ("conv1d", "linear", "conv1d-linear"):
            raise NotImplementedError("Support only linear or conv1d.")   # encoder.py:169
        self.idim, self.odim = idim, odim
        self._adim = adim
        self.eos = idim - 1
        self.reduction_factor = reduction_factor
        self.padding_idx = 0
        self.training = True
        self._ctx = Context.get(device)
        cfg = _capi.Fs2Cfg()
        cfg.idim, cfg.odim, cfg.adim, cfg.aheads = idim, odim, adim, aheads
        cfg.elayers, cfg.eunits, cfg.dlayers, cfg.dunits = elayers, eunits, dlayers, dunits
        cfg.positionwise_conv_kernel_size = positionwise_conv_kernel_size
        cfg.positionwise_layer_type = {"conv1d": 0, "linear": 1, "conv1d-linear": 2}[positionwise_layer_type]
        cfg.duration_predictor_layers = duration_predictor_layers
        cfg.duration_predictor_chans = duration_predictor_chans
        cfg.duration_predictor_kernel_size = duration_predictor_kernel_size
        cfg.pitch_predictor_layers = pitch_predictor_layers
        cfg.pitch_predictor_chans = pitch_predictor_chans
        cfg.pitch_predictor_kernel_size = pitch_predictor_kernel_size
        cfg.energy_predictor_layers = energy_predictor_layers
        cfg.energy_predictor_chans = energy_predictor_chans
        cfg.energy_predictor_kernel_size = energy_predictor_kernel_size
        cfg.pitch_embed_kernel_size = pitch_embed_kernel_size
        cfg.energy_embed_kernel_size = energy_embed_kernel_size
        cfg.postnet_layers, cfg.postnet_chans, cfg.postnet_filts = postnet_layers, postnet_chans, postnet_filts
        cfg.use_batch_norm = 1 if use_batch_norm else 0
        cfg.use_scaled_pos_enc = 1 if use_scaled_pos_enc else 0
        cfg.encoder_normalize_before = 1 if encoder_normalize_before else 0
        cfg.decoder_normalize_before = 1 if decoder_normalize_before else 0
        cfg.encoder_concat_after = 1 if encoder_concat_after else 0
        cfg.decoder_concat_after = 1 if decoder_concat_after else 0
        cfg.reduction_factor = reduction_factor
        if spk_embed_dim is not None and spk_embed_integration_type not in ("add", "concat"):
            raise NotImplementedError("support only add or concat.")   # fastspeech2.py:584
        cfg.num_speakers = 0 if (num_speakers is None or spk_embed_dim is None) else int(num_speakers)
        cfg.spk_embed_dim = 0 if spk_embed_dim is None else int(spk_embed_dim)
        cfg.spk_embed_integration_type = 1 if spk_embed_integration_type == "concat" else 0
        if tone_embed_dim is not None and tone_embed_integration_type != "add":
            raise NotImplementedError("tone_embed_integration_type='concat': the reference's branch cannot "
                                      "broadcast the 1-D tone ids of inference (fastspeech2.py:606-610)")
        cfg.num_tones = 0 if (num_tones is None or tone_embed_dim is None) else int(num_tones)
        cfg.tone_embed_dim = 0 if tone_embed_dim is None else int(tone_embed_dim)
        cfg.tone_embed_integration_type = 0
        self.tone_embed_dim = tone_embed_dim
        self.spk_embed_dim = spk_embed_dim
        h = C.c_void_p()
        _capi.check(self._ctx.lib.pk_fs2_create(self._ctx.handle, C.byref(cfg), C.byref(h)))
        self._h = h
        self._finalized = False
        self._last_tok, self._last_frames = [], []

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self._ctx.lib.pk_fs2_destroy(h)
            except Exception:
                pass

    def set_state_dict(self, state_dict):
        set_params(self._ctx.lib.pk_fs2_set_param, self._h, state_dict)
        self._finalized = False

    def eval(self):
        self.training = False
        return self

    def set_normalizer(self, normalizer):
        """Register ZScore statistics on the engine handle.  Registering changes nothing by itself: they are
        applied only by calls that ask for it (``denormalize=True``, what ``FastSpeech2Inference`` passes), so
        ``inference()`` itself stays in the normalised domain like the reference's (fastspeech2.py:468-558)."""
        self._norm_owner = None
        if normalizer is None:
            _capi.check(self._ctx.lib.pk_fs2_set_normalizer(self._h, None, None, 0))
        else:
            mu, sigma = to_numpy_f32(normalizer.mu).reshape(-1), to_numpy_f32(normalizer.sigma).reshape(-1)
            _capi.check(self._ctx.lib.pk_fs2_set_normalizer(self._h, _capi.fptr(mu), _capi.fptr(sigma), mu.size))
        self._finalized = False

    def _finalize(self):
        if not self._finalized:
            _capi.check(self._ctx.lib.pk_fs2_finalize(self._h))
            self._finalized = True

    def set_math(self, mode):
        """'f16x3' (default: 3-term split-fp16 MFMA GEMMs, fp32-equivalent error) or 'f32' (exact fp32 MFMA)."""
        _capi.check(self._ctx.lib.pk_fs2_set_math(self._h, {"f32": 0, "f16x3": 1}[mode]))

    def set_option(self, key, value):
        """Named integer options of the engine handle (include/pk_synth.h, pk_fs2_set_option): 'ffn_planes',
        'ffn_planes_min_blocks', 'ffnp_variant', 'attn_waves'.  The library reads no environment variable."""
        _capi.check(self._ctx.lib.pk_fs2_set_option(self._h, key.encode(), int(value)))

    def set_debug(self, on=True):
        _capi.check(self._ctx.lib.pk_fs2_set_debug(self._h, 1 if on else 0))

    # -- synthesis -----------------------------------------------------------
    def encode_batch(self, texts, alpha=1.0, spk_ids=None, spembs=None, tone_ids=None):
        """Phase 1: returns the per-utterance frame counts (host ints).  ``spk_ids`` (B,) ints or
        ``spembs`` (B, spk_embed_dim): speaker conditioning of a multi-speaker model (:396-402)."""
        ctx = Context.get(self._ctx.device)
        self._finalize()
        ids = [np.asarray(t.cpu() if isinstance(t, torch.Tensor) else t).astype(np.int64).reshape(-1)
               for t in texts]
        if self.tone_embed_dim is not None and tone_ids is not None:
            tn = [np.asarray(t.cpu() if isinstance(t, torch.Tensor) else t).astype(np.int64).reshape(-1)
                  for t in tone_ids]
            assert [len(t) for t in tn] == [len(i) for i in ids], "one tone id per token"
            tflat = np.ascontiguousarray(np.concatenate(tn))
            _capi.check(ctx.lib.pk_fs2_set_tones(self._h, tflat.ctypes.data_as(C.POINTER(C.c_int64)), tflat.size))
        if self.spk_embed_dim is not None and (spk_ids is not None or spembs is not None):
            if spembs is not None:
                e = np.ascontiguousarray(to_numpy_f32(spembs).reshape(len(ids), self.spk_embed_dim))
                _capi.check(ctx.lib.pk_fs2_set_speakers(self._h, None, _capi.fptr(e), len(ids)))
            else:
                sp = np.ascontiguousarray(np.asarray(
                    spk_ids.cpu() if isinstance(spk_ids, torch.Tensor) else spk_ids).astype(np.int64).reshape(-1))
                assert sp.size == len(ids), "one speaker id per utterance"
                _capi.check(ctx.lib.pk_fs2_set_speakers(self._h, sp.ctypes.data_as(C.POINTER(C.c_int64)), None,
                                                        len(ids)))
        lens = np.array([len(i) for i in ids], dtype=np.int32)
        flat = np.ascontiguousarray(np.concatenate(ids))
        frames = np.zeros(len(ids), dtype=np.int32)
        _capi.check(ctx.lib.pk_fs2_encode(self._h, flat.ctypes.data_as(C.POINTER(C.c_int64)),
                                          lens.ctypes.data_as(C.POINTER(C.c_int32)), len(ids),
                                          C.c_float(alpha), frames.ctypes.data_as(C.POINTER(C.c_int32))))
        self._last_tok, self._last_frames = [int(v) for v in lens], [int(v) for v in frames]
        return frames

    def decode_packed(self, denormalize=False):
        """Phase 2: packed (sum(frames), odim) device tensor of the last encode.  ``denormalize``: apply the
        registered ZScore.inverse in the output epilogue (FastSpeech2Inference.forward :668-671)."""
        ctx = Context.get(self._ctx.device)
        total = int(sum(self._last_frames))
        mel = ctx.empty((total, self.odim))
        if total:
            _capi.check(ctx.lib.pk_fs2_decode(self._h, dptr(mel), _capi.PK_APPLY_NORMALIZER if denormalize else 0))
        return mel

    def inference_batch(self, texts, alpha=1.0, spk_ids=None, spembs=None, tone_ids=None, denormalize=False):
        frames = self.encode_batch(texts, alpha, spk_ids, spembs, tone_ids)
        mel = self.decode_packed(denormalize)
        outs, o = [], 0
        for f in frames:
            outs.append(wrap(mel[o:o + int(f)]))
            o += int(f)
        return outs

    def inference(self, text, speech=None, durations=None, pitch=None, energy=None, alpha=1.0,
                  use_teacher_forcing=False, spembs=None, spk_id=None, tone_id=None, denormalize=False):
        """(T,) int64 -> (L, odim); fastspeech2.py:468-558 (is_inference=True branch)."""
        if use_teacher_forcing:
            raise NotImplementedError("teacher forcing is a training-time path")
        tones = None if tone_id is None else [tone_id]   # (T,) ids, forwarded un-batched by the reference (:546,556)
        if spembs is not None:      # (spk_embed_dim,), unsqueezed by the reference (:541-542)
            return self.inference_batch([text], alpha, spembs=to_numpy_f32(spembs).reshape(1, -1), tone_ids=tones,
                                        denormalize=denormalize)[0]
        if spk_id is not None:
            sid = np.asarray(spk_id.cpu() if isinstance(spk_id, torch.Tensor) else spk_id).reshape(-1)[:1]
            return self.inference_batch([text], alpha, spk_ids=sid, tone_ids=tones, denormalize=denormalize)[0]
        return self.inference_batch([text], alpha, tone_ids=tones, denormalize=denormalize)[0]

    def debug_tap(self, what, b):
        n_rows = self._last_tok[b] if what <= 3 else self._last_frames[b]
        width = {0: -1, 1: 1, 2: 1, 3: 1, 4: -1, 5: -1, 6: self.odim}[what]
        if width == -1:
            width = self._adim
        out = np.empty((n_rows, width), dtype=np.float32)
        _capi.check(self._ctx.lib.pk_fs2_debug_read(self._h, what, b, _capi.fptr(out), out.size))
        return out[:, 0] if width == 1 else out


class FastSpeech2Inference:
    """FastSpeech2Inference (fastspeech2.py:662-671): inference then normalizer.inverse."""

    def __init__(self, normalizer, model):
        self.normalizer = normalizer
        self.acoustic_model = model
        self.bind()

    def bind(self):
        """Make this wrapper's statistics the ones registered on the model's engine handle (a no-op unless
        another wrapper around the same model registered different ones since).  The model's own
        ``inference()`` is unaffected either way."""
        m = self.acoustic_model
        if getattr(m, "_norm_owner", None) is not self:
            m.set_normalizer(self.normalizer)
            m._norm_owner = self
        return m

    def forward(self, text, spk_id=None, alpha=1.0):
        return self.bind().inference(text, spk_id=spk_id, alpha=alpha, denormalize=True)

    __call__ = forward

    def eval(self):
        return self
