"""Import the reference's synthesis-path modules from /root/reference WITHOUT running the
package __init__ files (they pull in librosa, yacs, visualdl, ... which are not installed).

Two backends for ``import paddle``:

* default (``PARAKEET_REAL_PADDLE`` unset or 0): ``paddle`` resolves to oracle/paddle_shim, the torch-backed
  stand-in -- the only thing that can run in the build container (Paddle is not installable offline);
* ``PARAKEET_REAL_PADDLE=1``: the shim stays OFF ``sys.path`` and the installed PaddlePaddle (>= 2.1.2, README.md:50
  of the reference) executes the reference's source.  This is what ``tools/verify_with_paddle.py`` uses on a machine
  that has Paddle, to retire the "Paddle kernel semantics are documentation-derived" caveat of oracle/__init__.py.

The generators (tools/make_golden*.py) talk to either backend through the three helpers below (``fixed_randn``,
``dropout_hook``, ``same_padding_modes``) and write into ``golden_dir()``: tests/golden for the stand-in,
tests/golden_paddle for real Paddle (``PARAKEET_GOLDEN_DIR`` overrides both).
Only usable where the reference checkout exists (``PARAKEET_REFERENCE``, default /root/reference)."""
import contextlib
import importlib
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("PARAKEET_REFERENCE", "/root/reference")
REAL = os.environ.get("PARAKEET_REAL_PADDLE", "0") not in ("", "0")
SHIM = os.path.join(ROOT, "oracle", "paddle_shim")


def backend():
    return "paddle" if REAL else "shim"


def golden_dir():
    d = os.environ.get("PARAKEET_GOLDEN_DIR")
    if not d:
        d = os.path.join(ROOT, "tests", "golden_paddle" if REAL else "golden")
    os.makedirs(d, exist_ok=True)
    return d


def _stub_typeguard():
    # the reference asserts check_argument_types() in its constructors (fastspeech2.py:120); typeguard >= 3 dropped
    # that function and the package is often absent: the assertion is not part of the arithmetic
    try:
        import typeguard
        if hasattr(typeguard, "check_argument_types"):
            return
    except ImportError:
        pass
    m = types.ModuleType("typeguard")
    m.check_argument_types = lambda *a, **k: True
    sys.modules["typeguard"] = m


def setup():
    if REAL:
        if SHIM in sys.path:
            sys.path.remove(SHIM)
        mod = sys.modules.get("paddle")
        if mod is not None and os.path.abspath(getattr(mod, "__file__", "") or "").startswith(SHIM):
            raise RuntimeError("PARAKEET_REAL_PADDLE=1 but the stand-in paddle is already imported in this process")
        import paddle   # noqa: F401  -- the real one; ImportError here is the message the user needs
        if os.path.abspath(paddle.__file__).startswith(SHIM):
            raise RuntimeError("PARAKEET_REAL_PADDLE=1 resolved `paddle` to oracle/paddle_shim; fix PYTHONPATH")
        paddle.set_device("cpu")     # "the Paddle CPU reference" of north_star; fp32, deterministic reductions
        _stub_typeguard()
    elif SHIM not in sys.path:
        sys.path.insert(0, SHIM)
    if ROOT not in sys.path:
        sys.path.insert(1, ROOT)
    if "parakeet" in sys.modules:
        return

    def ns(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
        return m
    ns("parakeet", os.path.join(REF, "parakeet"))
    for sub in ("models", "modules", "utils", "models.fastspeech2", "models.parallel_wavegan", "models.speedyspeech",
                "models.transformer_tts", "modules.fastspeech2_transformer", "modules.tacotron2"):
        ns("parakeet." + sub, os.path.join(REF, "parakeet", *sub.split(".")))
    # parakeet.utils.checkpoint is imported by waveflow.py at module level only for from_pretrained
    ck = types.ModuleType("parakeet.utils.checkpoint")
    ck.load_parameters = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("not available"))
    sys.modules["parakeet.utils.checkpoint"] = ck


def load(name):
    setup()
    return importlib.import_module(name)


# ---- what the generators need from the backend beyond the public Paddle API ----------------------------------------

@contextlib.contextmanager
def fixed_randn(values):
    """``paddle.randn`` inside the reference's ``inference`` (parallel_wavegan.py:515, waveflow.py:801) returns
    ``values`` reshaped to the requested shape, so the run is reproducible and the noise can be handed to the engine."""
    import numpy as np
    import paddle
    orig = paddle.randn
    flat = np.asarray(values, np.float32)
    paddle.randn = lambda shape, dtype=None, name=None: paddle.to_tensor(flat.reshape([int(s) for s in shape]))
    try:
        yield
    finally:
        paddle.randn = orig


@contextlib.contextmanager
def dropout_hook(keep_fn):
    """Every ACTIVE ``F.dropout`` call (training=True, p > 0: the always-on prenet dropout of Tacotron2-style decoders,
    modules/tacotron2/decoder.py:78-81) keeps the elements ``keep_fn(shape, p) -> bool ndarray`` selects, scaled by
    1 / (1 - p) (Paddle's default ``upscale_in_train``).  The stand-in has a slot for it; real Paddle gets
    ``paddle.nn.functional.dropout`` replaced for the duration (the reference calls it as ``F.dropout``, an attribute
    looked up at call time)."""
    import numpy as np
    import paddle
    import paddle.nn.functional as PF
    if not REAL:
        import torch

        def hook(x, p):
            keep = torch.as_tensor(np.asarray(keep_fn(tuple(int(s) for s in x.shape), p), bool))
            return torch.where(keep, x / (1.0 - p), torch.zeros_like(x))
        PF.DROPOUT_HOOK = hook
        try:
            yield
        finally:
            PF.DROPOUT_HOOK = None
        return
    import inspect
    orig = PF.dropout
    sig = inspect.signature(orig)     # (x, p=0.5, axis=None, training=True, mode="upscale_in_train", name=None) in Paddle 2.1

    def dropout(*args, **kwargs):
        bound = sig.bind(*args, **kwargs)
        bound.apply_defaults()
        arg = dict(bound.arguments)
        arg.update(arg.pop("k", {}) if isinstance(arg.get("k"), dict) else {})   # (a stand-in's **k catch-all)
        x, p = arg["x"], arg.get("p", 0.5)
        if not arg.get("training", True) or p <= 0:
            return orig(*args, **kwargs)
        assert arg.get("mode", "upscale_in_train") == "upscale_in_train" and arg.get("axis") is None, arg
        keep = paddle.to_tensor(np.asarray(keep_fn(tuple(int(s) for s in x.shape), p), bool))
        return paddle.where(keep, x / (1.0 - p), paddle.zeros_like(x))
    PF.dropout = dropout
    try:
        yield
    finally:
        PF.dropout = orig


def same_padding_modes():
    """SpeedySpeech's residual blocks use ``padding="same"`` with a dilation (speedyspeech.py:33-43); what Paddle 2.1
    does with that pair is one of the documentation-derived semantics (oracle/speedyspeech_ref.py).  The stand-in runs
    both readings -> tags "rd" (dilation reset to 1) and "dil"; real Paddle has exactly one behaviour -> tag "real"
    (tools/verify_with_paddle.py reports which reading it equals).  Yields ``(tag, activate)``."""
    if REAL:
        return [("real", lambda: None)]
    import paddle.nn.functional as PF

    def setter(v):
        def f():
            PF.SAME_PADDING_RESETS_DILATION = v
        return f
    return [("rd", setter(True)), ("dil", setter(False))]
