"""WaveFlow vocoder behind the reference's Python API.

Mirrors parakeet/models/waveflow.py ``ConditionalWaveFlow`` (constructor :741-757, ``infer``
:785-805, ``predict`` :808-825); all arithmetic runs in libpk_synth.so (csrc/waveflow.hip).
Extension: ``infer`` takes an optional ``z=`` (the reference draws ``paddle.randn`` inside).
Synthesis only -- ``forward`` / ``WaveFlowLoss`` are training-time and out of scope.
"""
import ctypes as C

import numpy as np
import torch

from . import _capi, amp
from .runtime import Context, dptr, set_params, wrap

_MATH = {"f32": 0, "f16x3": 1, "f16": 2}


class ConditionalWaveFlow:
    def __init__(self, upsample_factors, n_flows, n_layers, n_group, channels, n_mels, kernel_size, device=None):
        if isinstance(kernel_size, int):
            kernel_size = [kernel_size, kernel_size]
        self.n_group, self.n_mels = n_group, n_mels
        self.training = True
        self._ctx = Context.get(device)
        cfg = _capi.WfCfg()
        cfg.n_upsample = len(upsample_factors)
        for i, f in enumerate(upsample_factors):
            cfg.upsample_factors[i] = int(f)
        cfg.n_flows, cfg.n_layers, cfg.n_group = n_flows, n_layers, n_group
        cfg.channels, cfg.n_mels = channels, n_mels
        cfg.kernel_h, cfg.kernel_w = int(kernel_size[0]), int(kernel_size[1])
        h = C.c_void_p()
        _capi.check(self._ctx.lib.pk_wf_create(self._ctx.handle, C.byref(cfg), C.byref(h)))
        self._h = h
        self._finalized = False
        self._math = "f16x3"

    @classmethod
    def from_pretrained(cls, config, checkpoint_path):
        """waveflow.py:827-852: ``config`` with ``model`` / ``data`` sections (examples/waveflow/config.py; a yacs node, a
        mapping or a yaml path), ``checkpoint_path`` without the ``.pdparams`` suffix.  Like the reference it returns the
        model in training mode with the checkpoint's weight-norm pairs loaded (folded when the engine packs them): the
        recipe goes on with ``layer_tools.recursively_remove_weight_norm(model)`` and ``model.eval()``."""
        from . import checkpoint
        return checkpoint.load_waveflow(config, checkpoint_path, cls=cls, eval_mode=False)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self._ctx.lib.pk_wf_destroy(h)
            except Exception:
                pass

    def set_state_dict(self, state_dict):
        set_params(self._ctx.lib.pk_wf_set_param, self._h, state_dict)
        self._finalized = False

    def eval(self):
        self.training = False
        return self

    def set_math(self, mode):
        """'f16x3' (default: split-fp16 MFMA GEMMs, fp32-equivalent error) or 'f32' (exact fp32 MFMA)."""
        _capi.check(self._ctx.lib.pk_wf_set_math(self._h, _MATH[mode]))
        self._math = mode

    def set_option(self, key, value):
        """Named integer options of the engine handle (include/pk_synth.h, pk_wf_set_option): 'layer_waves'."""
        _capi.check(self._ctx.lib.pk_wf_set_option(self._h, key.encode(), int(value)))

    def set_seed(self, seed):
        """Seed of the engine's own latent stream (``pk_randn``), used when neither ``z`` nor a torch
        ``generator`` is given -- the ``paddle.randn`` of waveflow.py:801."""
        _capi.check(self._ctx.lib.pk_wf_set_seed(self._h, int(seed) & (2 ** 64 - 1)))

    def lengths(self, t_mel):
        a, b = C.c_int32(), C.c_int32()
        _capi.check(self._ctx.lib.pk_wf_cond_length(self._h, int(t_mel), C.byref(a), C.byref(b)))
        return a.value, b.value

    def infer_batch(self, mels, zs=None, generator=None):
        """mels: list of (C_mel, T_b) arrays (ragged).  Returns a list of (T_b',) device tensors."""
        ctx = Context.get(self._ctx.device)
        if not self._finalized:
            _capi.check(ctx.lib.pk_wf_finalize(self._h))
            self._finalized = True
        frames = np.array([int(m.shape[-1]) for m in mels], dtype=np.int32)
        mel = torch.cat([ctx.to_device(m).reshape(self.n_mels, -1).transpose(0, 1) for m in mels], 0).contiguous()
        lens = [self.lengths(int(f)) for f in frames]
        total_z = sum(a for a, _ in lens)
        if zs is None:
            z = None if generator is None else torch.randn(total_z, device=ctx.device, generator=generator)
        else:
            z = torch.cat([ctx.to_device(v).reshape(-1) for v in zs])
        assert z is None or z.numel() == total_z, "z must have cond_len samples per utterance"
        wav = ctx.empty((sum(b for _, b in lens),))
        # inside `with parakeet_amd.amp.auto_cast():` (examples/waveflow/synthesize.py:40) the call runs with fp16 operands
        cast = amp.enabled() and self._math != "f16"
        if cast:
            _capi.check(ctx.lib.pk_wf_set_math(self._h, _MATH["f16"]))
        try:
            _capi.check(ctx.lib.pk_wf_infer(self._h, dptr(mel), frames.ctypes.data_as(C.POINTER(C.c_int32)), len(mels),
                                            None if z is None else dptr(z), dptr(wav), 0))
        finally:
            if cast:
                _capi.check(ctx.lib.pk_wf_set_math(self._h, _MATH[self._math]))
        outs, o = [], 0
        for _, n in lens:
            outs.append(wrap(wav[o:o + n]))
            o += n
        return outs

    def infer(self, mel, z=None):
        """(B, C_mel, T_mel) -> (B, T); waveflow.py:785-805."""
        ctx = Context.get(self._ctx.device)
        mel = ctx.to_device(mel)
        zs = None if z is None else [ctx.to_device(z)[b] for b in range(mel.shape[0])]
        outs = self.infer_batch([mel[b] for b in range(mel.shape[0])], zs)
        return wrap(torch.stack([o.as_subclass(torch.Tensor) for o in outs], 0))

    def predict(self, mel, z=None):
        """np (C_mel, T_mel) -> np (T,); waveflow.py:808-825."""
        z = None if z is None else np.asarray(z)[None]
        return self.infer(np.asarray(mel)[None], z)[0].numpy()
