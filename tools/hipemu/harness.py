"""Point the Python shim at the host-emulated engine (DEVELOPMENT / tests only; see hip/hip_runtime.h).

``install()`` builds tools/hipemu/_build/libpk_synth_emu.so, loads it in place of libpk_synth.so and registers a
context whose "device" memory is host memory, so that the wrappers of parakeet_amd (and through them the C ABI and every
kernel) run on CPU tensors.  Nothing in parakeet_amd refers to this module; a process that has not called install()
cannot reach the emulator.
"""
import ctypes as C
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if HERE not in sys.path:
    sys.path.insert(0, HERE)


def install(verbose=False):
    import build as emu_build                         # tools/hipemu/build.py
    from parakeet_amd import _capi, runtime

    if getattr(runtime.Context, "_hipemu", False):
        return runtime.Context._instances[0]
    lib = C.CDLL(emu_build.build(verbose=verbose))
    _capi._declare(lib)
    _capi._lib = lib

    class EmuContext(runtime.Context):
        def __init__(self):                            # noqa: D401 -- no HIP device, no stream
            self.device = torch.device("cpu")
            self.lib = lib
            h = C.c_void_p()
            _capi.check(lib.pk_ctx_create(0, C.byref(h)))
            self.handle = h

        def bind_stream(self):
            pass

    runtime.Context._instances = {0: EmuContext()}
    runtime.Context._hipemu = True
    return runtime.Context._instances[0]


def reset_error():
    from parakeet_amd import _capi
    _capi._lib.hipemu_reset_error()
