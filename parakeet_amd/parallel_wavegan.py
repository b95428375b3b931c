"""Parallel WaveGAN generator behind the reference's Python API.

Mirrors parakeet/models/parallel_wavegan/parallel_wavegan.py:
``PWGGenerator`` (constructor kwargs :369-388, ``set_state_dict``, ``eval``,
``remove_weight_norm`` :485-496, ``inference`` :498-520) and ``PWGInference``
(:766-775).  All arithmetic runs in libpk_synth.so (csrc/pwg.hip).

Extensions over the reference (superset, not a break): ``inference`` takes an
optional ``noise=`` (the reference draws ``paddle.randn`` inside the call, which
cannot be reproduced), and ``inference_batch`` synthesises a ragged batch in
one engine call.
"""
import ctypes as C

import numpy as np
import torch

from . import _capi
from .normalizer import ZScore
from .runtime import Context, dptr, set_params, to_numpy_f32, wrap


class PWGGenerator:
    def __init__(self, in_channels=1, out_channels=1, kernel_size=3, layers=30, stacks=3,
                 residual_channels=64, gate_channels=128, skip_channels=64, aux_channels=80,
                 aux_context_window=2, dropout=0., bias=True, use_weight_norm=True,
                 use_causal_conv=False, upsample_scales=(4, 4, 4, 4), nonlinear_activation=None,
                 nonlinear_activation_params=None, interpolate_mode="nearest",
                 freq_axis_kernel_size=1, device=None):
        assert layers % stacks == 0  # parallel_wavegan.py:398
        if not bias:
            raise NotImplementedError("PWGGenerator(bias=False) is not implemented")
        if nonlinear_activation is not None:
            raise NotImplementedError("upsample nonlinear_activation is not implemented")
        if interpolate_mode != "nearest" or freq_axis_kernel_size != 1:
            raise NotImplementedError("only nearest stretch with freq_axis_kernel_size=1 is implemented")
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.aux_channels = aux_channels
        self.aux_context_window = aux_context_window
        self.layers = layers
        self.stacks = stacks
        self.kernel_size = kernel_size
        self.upsample_factor = int(np.prod(upsample_scales))
        self.training = True
        self._ctx = Context.get(device)
        cfg = _capi.PwgCfg()
        cfg.in_channels, cfg.out_channels, cfg.kernel_size = in_channels, out_channels, kernel_size
        cfg.layers, cfg.stacks = layers, stacks
        cfg.residual_channels, cfg.gate_channels = residual_channels, gate_channels
        cfg.skip_channels, cfg.aux_channels = skip_channels, aux_channels
        cfg.aux_context_window = aux_context_window
        cfg.n_upsample = len(upsample_scales)
        for i, s in enumerate(upsample_scales):
            cfg.upsample_scales[i] = int(s)
        cfg.use_causal_conv = 1 if use_causal_conv else 0
        h = C.c_void_p()
        _capi.check(self._ctx.lib.pk_pwg_create(self._ctx.handle, C.byref(cfg), C.byref(h)))
        self._h = h
        self._finalized = False

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self._ctx.lib.pk_pwg_destroy(h)
            except Exception:
                pass

    # -- nn.Layer look-alikes ------------------------------------------------
    def set_state_dict(self, state_dict):
        set_params(self._ctx.lib.pk_pwg_set_param, self._h, state_dict)
        self._finalized = False

    def eval(self):
        self.training = False
        return self

    def remove_weight_norm(self):
        """Numerically neutral here: weight_g / weight_v pairs are folded at finalize."""
        return None

    def set_math(self, mode):
        """'f16x3' (default: 3-term split-fp16 MFMA, fp32-equivalent error), 'f32' (exact fp32 MFMA) or
        'bf16x3' (split-bf16)."""
        m = {"f32": _capi.PK_PWG_MATH_F32, "bf16x3": _capi.PK_PWG_MATH_BF16X3, "f16x3": _capi.PK_PWG_MATH_F16X3}[mode]
        _capi.check(self._ctx.lib.pk_pwg_set_math(self._h, m))

    def set_option(self, key, value):
        """Named integer options of the engine handle (include/pk_synth.h, pk_pwg_set_option): 'planes', 'scale_guard'."""
        _capi.check(self._ctx.lib.pk_pwg_set_option(self._h, key.encode(), int(value)))

    def scale_overshoot(self):
        """(log2(a-priori bound / measured max|x|) per layer input [layers + 1], fell_back) of the last guarded inference
        (pk_pwg_scale_overshoot)."""
        import numpy as np
        out = np.zeros(self.layers + 1, np.float32)
        fb = C.c_int32(0)
        _capi.check(self._ctx.lib.pk_pwg_scale_overshoot(self._h, _capi.fptr(out), out.size, C.byref(fb)))
        self.fell_back_code = int(fb.value)      # 1: inside a guarded call; 2: by the deferred verdict of a sampled later call
        return out, bool(fb.value)

    def set_chunk_samples(self, samples):
        """Scheduling only (results are unchanged): samples per cache-resident chunk of the residual stack."""
        _capi.check(self._ctx.lib.pk_pwg_set_chunk_samples(self._h, int(samples)))

    def set_seed(self, seed):
        """Seed of the engine's own noise stream (Philox4x32-10 + Box-Muller, ``pk_randn``), used when
        neither ``noise`` nor a torch ``generator`` is given -- the ``paddle.randn`` of :515-516."""
        _capi.check(self._ctx.lib.pk_pwg_set_seed(self._h, int(seed) & (2 ** 64 - 1)))

    def set_normalizer(self, normalizer):
        """Register ZScore statistics; applied only by calls passing ``normalize=True`` (what PWGInference
        does) -- ``inference()`` / ``forward()`` themselves take already-normalised features like the reference's."""
        self._norm_owner = None
        if normalizer is None:
            _capi.check(self._ctx.lib.pk_pwg_set_normalizer(self._h, None, None, 0))
        else:
            mu, sigma = to_numpy_f32(normalizer.mu).reshape(-1), to_numpy_f32(normalizer.sigma).reshape(-1)
            _capi.check(self._ctx.lib.pk_pwg_set_normalizer(self._h, _capi.fptr(mu), _capi.fptr(sigma), mu.size))

    def _finalize(self):
        if not self._finalized:
            _capi.check(self._ctx.lib.pk_pwg_finalize(self._h))
            self._finalized = True

    # -- synthesis -------------------------------------------------------------
    def inference_batch(self, mels, noises=None, generator=None, normalize=False):
        """mels: list of (T'_b, aux) arrays.  Returns a list of (T'_b*hop, out) device tensors."""
        ctx = Context.get(self._ctx.device)
        self._finalize()
        frames = np.array([int(m.shape[0]) for m in mels], dtype=np.int32)
        self._last_frames = [int(f) for f in frames]
        hop = self.upsample_factor
        mel = torch.cat([ctx.to_device(m).reshape(-1, self.aux_channels) for m in mels], dim=0)
        total = int(frames.sum()) * hop
        if noises is None:
            # no noise given: the engine draws it (NULL) unless a torch generator is supplied
            noise = None if generator is None else torch.randn(total, device=ctx.device, dtype=torch.float32,
                                                               generator=generator)
        else:
            noise = torch.cat([ctx.to_device(n).reshape(-1) for n in noises], dim=0)
        assert noise is None or noise.numel() == total, "noise length must be frames * hop"
        wav = ctx.empty((total,))
        _capi.check(ctx.lib.pk_pwg_infer(self._h, dptr(mel), frames.ctypes.data_as(C.POINTER(C.c_int32)),
                                         len(mels), None if noise is None else dptr(noise), dptr(wav),
                                         _capi.PK_APPLY_NORMALIZER if normalize else 0))
        outs, o = [], 0
        for f in frames:
            n = int(f) * hop
            outs.append(wrap(wav[o:o + n].reshape(n, self.out_channels)))
            o += n
        return outs

    def infer_packed(self, mel, frames, noise=None, generator=None, normalize=False):
        """mel: packed (sum(frames), aux) DEVICE tensor (e.g. FastSpeech2.decode_packed());
        returns the packed (sum(frames)*hop,) device waveform -- no host round trip."""
        ctx = Context.get(self._ctx.device)
        self._finalize()
        frames = np.ascontiguousarray(np.asarray(frames, dtype=np.int32))
        self._last_frames = [int(f) for f in frames]
        total = int(frames.sum()) * self.upsample_factor
        mel = ctx.to_device(mel).reshape(-1, self.aux_channels)
        assert mel.shape[0] == int(frames.sum()), "mel rows must equal sum(frames)"
        if noise is None:
            noise = None if generator is None else torch.randn(total, device=ctx.device, dtype=torch.float32,
                                                               generator=generator)
        else:
            noise = ctx.to_device(noise).reshape(-1)
        assert noise is None or noise.numel() == total, "noise length must be sum(frames) * hop"
        wav = ctx.empty((total,))
        _capi.check(ctx.lib.pk_pwg_infer(self._h, dptr(mel), frames.ctypes.data_as(C.POINTER(C.c_int32)),
                                         len(frames), None if noise is None else dptr(noise), dptr(wav),
                                         _capi.PK_APPLY_NORMALIZER if normalize else 0))
        return wav

    def forward(self, x, c):
        """(N, C_in, T) noise, (N, C_aux, T' + 2*ctx) conditioning -> (N, C_out, T);
        parallel_wavegan.py:445-472 (the batch form tests/unit/test_pwg.py exercises)."""
        ctx = Context.get(self._ctx.device)
        self._finalize()
        c = ctx.to_device(c)
        x = ctx.to_device(x)
        N, _, Tc = c.shape
        frames_each = Tc - 2 * self.aux_context_window
        assert x.shape[-1] == frames_each * self.upsample_factor  # assert c.shape[-1] == x.shape[-1] (:462)
        frames = np.full(N, frames_each, dtype=np.int32)
        self._last_frames = [int(f) for f in frames]
        mel = c.transpose(1, 2).contiguous().reshape(-1, self.aux_channels)
        noise = x.reshape(-1).contiguous()
        wav = ctx.empty((noise.numel(),))
        _capi.check(ctx.lib.pk_pwg_infer(self._h, dptr(mel), frames.ctypes.data_as(C.POINTER(C.c_int32)), N,
                                         dptr(noise), dptr(wav), _capi.PK_PWG_C_HAS_CONTEXT))
        return wrap(wav.reshape(N, self.out_channels, -1))

    __call__ = forward

    def inference(self, c=None, noise=None, normalize=False):
        """(T', C_aux) -> (T, C_out); parallel_wavegan.py:498-520."""
        return self.inference_batch([c], None if noise is None else [noise], normalize=normalize)[0]

    def debug_tap(self, what, b):
        # frames of utterance b are known to the engine; size is validated there
        n = self._last_frames[b] * self.upsample_factor
        if what == 3:       # max|x| per 32-sample block of the final residual stream (block-scaled split path)
            out = np.empty(((n + 31) // 32,), dtype=np.float32)
        else:
            rows = {0: 128, 1: 64, 2: 64}[what]
            out = np.empty((rows, n), dtype=np.float32)
        _capi.check(self._ctx.lib.pk_pwg_debug_read(self._h, what, b, _capi.fptr(out), out.size))
        return out


class PWGInference:
    """PWGInference (parallel_wavegan.py:766-775): normalizer(logmel) -> generator.inference."""

    def __init__(self, normalizer, pwg_generator):
        self.normalizer = normalizer
        self.pwg_generator = pwg_generator
        self.bind()

    def bind(self):
        g = self.pwg_generator
        if getattr(g, "_norm_owner", None) is not self:
            g.set_normalizer(self.normalizer)
            g._norm_owner = self
        return g

    def forward(self, logmel, noise=None):
        return self.bind().inference(logmel, noise=noise, normalize=True)

    __call__ = forward

    def eval(self):
        return self
