"""Device plumbing for the Python shim: contexts, tensors in/out of the C ABI.

PyTorch-ROCm is used here only as the device-memory container and stream
provider; no torch operator takes part in the synthesis math.
"""
import ctypes as C

import numpy as np
import torch

from . import _capi


class PKTensor(torch.Tensor):
    """torch.Tensor whose ``.numpy()`` works from device memory, so recipe code
    written for paddle tensors (``wav.numpy()``,
    examples/fastspeech2/ljspeech/synthesize_e2e.py:104-107) runs unchanged."""

    def numpy(self):  # noqa: D401
        return self.detach().cpu().as_subclass(torch.Tensor).numpy()


def wrap(t):
    return t.as_subclass(PKTensor)


class Context:
    """One pk_ctx per HIP device, launched on torch's current stream."""

    _instances = {}

    def __init__(self, device):
        if not torch.cuda.is_available():
            raise RuntimeError("parakeet_amd needs a HIP device (torch.cuda.is_available() is False); "
                               "there is no CPU fallback")
        self.device = torch.device("cuda", device)
        self.lib = _capi.lib()
        h = C.c_void_p()
        _capi.check(self.lib.pk_ctx_create(int(device), C.byref(h)))
        self.handle = h
        self.bind_stream()

    @classmethod
    def get(cls, device=None):
        if device is None:
            device = torch.cuda.current_device() if torch.cuda.is_available() else 0
        if isinstance(device, torch.device):
            device = device.index or 0
        if device not in cls._instances:
            cls._instances[device] = Context(device)
        ctx = cls._instances[device]
        ctx.bind_stream()
        return ctx

    def bind_stream(self):
        s = torch.cuda.current_stream(self.device).cuda_stream
        _capi.check(self.lib.pk_ctx_set_stream(self.handle, C.c_void_p(s)))

    def sync(self):
        _capi.check(self.lib.pk_sync(self.handle))

    # -- profiler ---------------------------------------------------------
    def prof_enable(self, on=True):
        _capi.check(self.lib.pk_prof_enable(self.handle, 1 if on else 0))

    def prof_reset(self):
        _capi.check(self.lib.pk_prof_reset(self.handle))

    def prof_dump(self):
        buf = C.create_string_buffer(1 << 16)
        _capi.check(self.lib.pk_prof_dump(self.handle, buf, len(buf)))
        out = {}
        for line in buf.value.decode().splitlines():
            name, n, ms = line.split()
            out[name] = (int(n), float(ms))
        return out

    # -- tensors ------------------------------------------------------------
    def to_device(self, x, dtype=torch.float32):
        """numpy / torch (any device) -> contiguous tensor on this device."""
        if isinstance(x, torch.Tensor):
            t = x.as_subclass(torch.Tensor)
        elif hasattr(x, "numpy") and not isinstance(x, np.ndarray):
            t = torch.from_numpy(np.asarray(x.numpy()))
        else:
            t = torch.from_numpy(np.ascontiguousarray(np.asarray(x)))
        return t.to(device=self.device, dtype=dtype).contiguous()

    def empty(self, shape, dtype=torch.float32):
        return torch.empty(shape, dtype=dtype, device=self.device)


def to_numpy_f32(v):
    if isinstance(v, torch.Tensor):
        v = v.detach().cpu().numpy()
    elif hasattr(v, "numpy") and not isinstance(v, np.ndarray):
        v = v.numpy()
    return np.ascontiguousarray(np.asarray(v, dtype=np.float32))


def set_params(setter, handle, state):
    """Feed a ``{state_dict_key: array}`` mapping through a pk_*_set_param call."""
    for name, v in state.items():
        a = to_numpy_f32(v)
        shape = (C.c_int64 * max(a.ndim, 1))(*a.shape) if a.ndim else (C.c_int64 * 1)(1)
        ndim = a.ndim if a.ndim else 1
        _capi.check(setter(handle, name.encode(), _capi.fptr(a), shape, ndim))


def dptr(t):
    return C.c_void_p(t.data_ptr())


def randn(n, seed=0, offset=0, device=None):
    """n standard normals from the engine's counter-based generator (``pk_randn``: Philox4x32-10 +
    Box-Muller), a pure function of (seed, offset + i); returns a device tensor."""
    ctx = Context.get(device)
    out = ctx.empty((int(n),))
    _capi.check(ctx.lib.pk_randn(ctx.handle, dptr(out), int(n), int(seed) & (2 ** 64 - 1), int(offset), 0))
    return out
