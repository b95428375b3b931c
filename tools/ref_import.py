"""Import the reference's synthesis-path modules from /root/reference WITHOUT running the
package __init__ files (they pull in librosa, yacs, visualdl, ... which are not installed), with
``paddle`` resolved to oracle/paddle_shim.  Only usable in the build container (the reference
does not travel to the GPU box)."""
import importlib
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("PARAKEET_REFERENCE", "/root/reference")


def setup():
    shim = os.path.join(ROOT, "oracle", "paddle_shim")
    if shim not in sys.path:
        sys.path.insert(0, shim)
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
