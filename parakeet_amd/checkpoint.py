"""Checkpoint ingestion without Paddle.

The reference stores checkpoints with ``paddle.save``:

* ``snapshot_iter_*.pdz`` -- ``UpdaterBase.save`` (parakeet/training/updater.py:77-80) pickles the nested
  archive built by ``StandardUpdater.state_dict`` (training/updaters/standard_updater.py:183-190):
  ``{"epoch", "iteration", "<name>_params": layer.state_dict(), "<name>_optimizer": ...}``;
  the recipes read ``["main_params"]`` (FastSpeech2, synthesize_e2e.py:56-57) and
  ``["generator_params"]`` (Parallel WaveGAN, :60-61);
* ``step-N.pdparams`` -- ``utils/checkpoint.py:61-108`` saves a bare ``layer.state_dict()`` (WaveFlow,
  ``ConditionalWaveFlow.from_pretrained`` waveflow.py:827-852);
* ``*_stats.npy`` -- ``np.stack([mean_, scale_])`` float32 (2, n_mels) (utils/compute_statistics.py:101-107);
* ``phone_id_map.txt`` -- one ``<phone> <id>`` pair per line (synthesize_e2e.py:45-50).

``paddle.save`` (python/paddle/framework/io.py of Paddle 2.1.x, the version the reference pins: README.md:50)
writes a pickle stream, default protocol 2, through one of two paths:

* ``save`` -> ``_is_state_dict(obj)`` true (a bare ``layer.state_dict()``: every value a tensor) ->
  ``_legacy_save``: ``_build_saved_state_dict`` turns every tensor into its ``ndarray`` and adds the table
  ``"StructuredToParameterName@@": {structured key: parameter name}``; ``_unpack_saved_dict`` (protocols 2 and 3)
  cuts arrays of more than (2^30 - 1) / itemsize elements into flat slices ``"<key>@@.<i>"`` described by
  ``"UnpackBigParamInfor@@": {key: {"OriginShape": shape, "slices": [names]}}``; then ``pickle.dump``.
* anything else (the updaters' nested archive: ``epoch`` / ``iteration`` are ints) -> ``_pickle_save``: a
  ``pickle.Pickler`` whose ``dispatch_table`` reduces ``VarBase`` / ``ParamBase`` to
  ``(tuple, ((tensor.name, tensor.numpy()),))`` and ``LoDTensor`` to ``(eval, ('data', {'data': ndarray}))``.

So a tensor arrives as the ``ndarray`` itself, as the pair ``(tensor_name, ndarray)``, or -- for the rare
``LoDTensor`` leaves of an optimizer state -- through ``eval('data', {'data': ndarray})``.  ``load_archive`` undoes
all of it (``_pack_loaded_dict`` re-merges the big-parameter slices).  [paddle-format: Paddle is not installable
here, so no file written by Paddle itself has been read; ``tools/make_paddle_fixture.py`` re-implements the two
save paths above around a stand-in tensor class, with the real ``pickle`` / ``numpy`` machinery, and the committed
fixtures under ``tests/golden/paddle21_*`` are its output.]  Because a pickle can execute code, the reader is a
*restricted* unpickler: it reconstructs numpy arrays, numpy scalars / dtypes and built-in containers, maps ``eval``
to a stub that accepts exactly the call above, and refuses every other global.
"""
import io
import os
import pickle
from collections import OrderedDict

import numpy as np

_NAME_TABLE_KEYS = ("StructuredToParameterName@@", "UnpackBigParamInfor@@")

_ALLOWED_GLOBALS = {
    ("collections", "OrderedDict"): OrderedDict,
    ("builtins", "dict"): dict, ("builtins", "list"): list, ("builtins", "tuple"): tuple,
    ("builtins", "set"): set, ("builtins", "frozenset"): frozenset, ("builtins", "int"): int,
    ("builtins", "float"): float, ("builtins", "complex"): complex, ("builtins", "str"): str,
    ("builtins", "bytes"): bytes, ("builtins", "bytearray"): bytearray, ("builtins", "bool"): bool,
    ("__builtin__", "dict"): dict, ("__builtin__", "list"): list, ("__builtin__", "tuple"): tuple,
    ("__builtin__", "int"): int, ("__builtin__", "long"): int, ("__builtin__", "float"): float,
    ("__builtin__", "str"): str, ("__builtin__", "bool"): bool, ("__builtin__", "set"): set,
    ("_codecs", "encode"): __import__("_codecs").encode,   # protocol-2 pickles of numpy byte strings
    ("numpy", "ndarray"): np.ndarray, ("numpy", "dtype"): np.dtype,
}


def _lodtensor_eval(expr, env=None, *rest):
    """Stand-in for the ``eval`` that Paddle's LoDTensor reducer pickles: only ``eval('data', {'data': x})``."""
    if expr != "data" or rest or not isinstance(env, dict) or set(env) != {"data"} or not isinstance(env["data"], np.ndarray):
        raise pickle.UnpicklingError("checkpoint calls eval() with something other than Paddle's LoDTensor reducer")
    return env["data"]


_ALLOWED_GLOBALS[("builtins", "eval")] = _lodtensor_eval
_ALLOWED_GLOBALS[("__builtin__", "eval")] = _lodtensor_eval


def _numpy_global(module, name):
    # numpy moved numpy.core -> numpy._core in 2.x; archives written by either spell the same objects
    import importlib
    if module in ("numpy.core.multiarray", "numpy._core.multiarray") and name in ("_reconstruct", "scalar"):
        sub = "multiarray"
    elif module in ("numpy.core.numeric", "numpy._core.numeric") and name == "_frombuffer":
        sub = "numeric"
    else:
        return None
    for pkg in ("numpy._core.", "numpy.core."):     # numpy 2.x (this image), then numpy 1.x
        try:
            return getattr(importlib.import_module(pkg + sub), name)
        except (ImportError, AttributeError):
            continue
    return None


class _RestrictedUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        obj = _ALLOWED_GLOBALS.get((module, name))
        if obj is None:
            obj = _numpy_global(module, name)
        if obj is None:
            raise pickle.UnpicklingError(
                f"checkpoint refers to {module}.{name}, which a tensor archive has no business containing; "
                "refusing to load it")
        return obj


def _pack_loaded_dict(obj):
    """Undo ``_unpack_saved_dict``: ``"<key>@@.<i>"`` slices + ``"UnpackBigParamInfor@@"`` -> the original array."""
    info = obj.get("UnpackBigParamInfor@@")
    if not isinstance(info, dict):
        return obj
    out = OrderedDict(obj)
    for key, desc in info.items():
        try:
            parts = [np.asarray(_plain(out[name])).reshape(-1) for name in desc["slices"]]
            out[key] = np.concatenate(parts).reshape(tuple(desc["OriginShape"]))
        except (KeyError, TypeError, ValueError) as e:
            raise ValueError(f"archive: cannot re-assemble the sliced parameter {key!r}: {e}") from None
        for name in desc["slices"]:
            out.pop(name, None)
    return out


def _plain(obj):
    """Tensor leaves -> ndarray: accepts ndarray, (name, ndarray) pairs; re-merges sliced big parameters and drops
    Paddle's bookkeeping tables."""
    if isinstance(obj, np.ndarray):
        return obj
    if isinstance(obj, (tuple, list)) and len(obj) == 2 and isinstance(obj[0], str) and isinstance(obj[1], np.ndarray):
        return obj[1]
    if isinstance(obj, dict):
        obj = _pack_loaded_dict(obj)
        return OrderedDict((k, _plain(v)) for k, v in obj.items() if k not in _NAME_TABLE_KEYS)
    if isinstance(obj, (list, tuple)):
        return type(obj)(_plain(v) for v in obj)
    return obj


def load_archive(path_or_file):
    """``paddle.load`` for tensor archives: nested dicts with numpy arrays at the leaves."""
    if hasattr(path_or_file, "read"):
        return _plain(_RestrictedUnpickler(path_or_file).load())
    with open(path_or_file, "rb") as f:
        return _plain(_RestrictedUnpickler(io.BufferedReader(f)).load())


def load_params(path, key=None):
    """State dict ``{reference key: float32 ndarray}`` from a ``.pdz`` (give ``key``, e.g. "main_params",
    "generator_params") or a ``.pdparams`` file (``key=None``).  weight_g / weight_v pairs are kept:
    the engine folds them at finalize (``remove_weight_norm`` semantics)."""
    arch = load_archive(path)
    if key is not None:
        if not isinstance(arch, dict) or key not in arch:
            have = sorted(arch) if isinstance(arch, dict) else type(arch).__name__
            raise KeyError(f"{path}: no entry {key!r} in the archive (has {have})")
        arch = arch[key]
    if not isinstance(arch, dict):
        raise ValueError(f"{path}: expected a state dict, found {type(arch).__name__}")
    state = OrderedDict()
    for name, value in arch.items():
        if not isinstance(value, np.ndarray):
            raise ValueError(f"{path}: entry {name!r} is {type(value).__name__}, not a tensor")
        if value.dtype.kind == "f":
            value = np.ascontiguousarray(value, dtype=np.float32)
        state[name] = value
    return state


def load_stats(path):
    """``(mu, sigma)`` float32 vectors from a ``*_stats.npy`` file of shape (2, n_bins)."""
    stat = np.load(path, allow_pickle=False)
    if stat.ndim != 2 or stat.shape[0] != 2:
        raise ValueError(f"{path}: expected an array of shape (2, n_bins), got {stat.shape}")
    return np.ascontiguousarray(stat[0], np.float32), np.ascontiguousarray(stat[1], np.float32)


def load_phone_id_map(path):
    """``{phone: id}`` and the vocabulary size (= number of lines, synthesize_e2e.py:45-50)."""
    table = OrderedDict()
    with open(path, "rt", encoding="utf-8") as f:
        for line in f:
            parts = line.strip().split()
            if not parts:
                continue
            if len(parts) != 2:
                raise ValueError(f"{path}: malformed line {line!r}")
            table[parts[0]] = int(parts[1])
    return table, len(table)


def _config(cfg):
    """A recipe's ``default.yaml`` (path) or an already parsed mapping."""
    if isinstance(cfg, (str, os.PathLike)):
        import yaml
        with open(cfg, "rt") as f:
            return yaml.safe_load(f)
    return cfg


def load_fastspeech2(config, checkpoint, stats, phones_dict=None, idim=None, speaker_dict=None):
    """The FastSpeech2 half of examples/fastspeech2/ljspeech/synthesize_e2e.py:45-83: returns
    ``(FastSpeech2Inference, phone_id_map)``.  ``config``: the recipe's yaml (path or dict with ``n_mels``
    and ``model``); ``idim`` overrides the vocabulary size read from ``phones_dict``; ``speaker_dict``: the speaker id map
    of the multi-speaker recipes, whose line count is ``num_speakers`` (examples/fastspeech2/aishell3/synthesize_e2e.py:47-57)."""
    from .fastspeech2 import FastSpeech2, FastSpeech2Inference
    from .normalizer import ZScore
    cfg = _config(config)
    phone_id_map = None
    if phones_dict is not None:
        phone_id_map, vocab = load_phone_id_map(phones_dict)
        idim = vocab if idim is None else idim
    if idim is None:
        raise ValueError("load_fastspeech2: give phones_dict or idim")
    kw = {}
    if speaker_dict is not None:
        with open(speaker_dict, "rt") as f:
            kw["num_speakers"] = sum(1 for line in f if line.strip())
    model = FastSpeech2(idim=idim, odim=cfg["n_mels"], **kw, **cfg["model"])
    model.set_state_dict(load_params(checkpoint, "main_params"))
    model.eval()
    mu, sigma = load_stats(stats)
    return FastSpeech2Inference(ZScore(mu, sigma), model), phone_id_map


def load_pwg(config, checkpoint, stats):
    """The vocoder half of synthesize_e2e.py:59-83: returns ``PWGInference``."""
    from .normalizer import ZScore
    from .parallel_wavegan import PWGGenerator, PWGInference
    cfg = _config(config)
    vocoder = PWGGenerator(**cfg["generator_params"])
    vocoder.set_state_dict(load_params(checkpoint, "generator_params"))
    vocoder.remove_weight_norm()
    vocoder.eval()
    mu, sigma = load_stats(stats)
    return PWGInference(ZScore(mu, sigma), vocoder)


def load_speedyspeech(config, checkpoint, stats, phones_dict, tones_dict, same_padding_resets_dilation=True):
    """The acoustic-model half of examples/speedyspeech/baker/synthesize_e2e.py:46-83: returns
    ``(SpeedySpeechInference, phone_id_map, tone_id_map)``."""
    from .normalizer import ZScore
    from .speedyspeech import SpeedySpeech, SpeedySpeechInference
    cfg = _config(config)
    phone_map, vocab = load_phone_id_map(phones_dict)
    tone_map, tone_size = load_phone_id_map(tones_dict)
    model = SpeedySpeech(vocab_size=vocab, tone_size=tone_size,
                         same_padding_resets_dilation=same_padding_resets_dilation, **cfg["model"])
    model.set_state_dict(load_params(checkpoint, "main_params"))
    model.eval()
    mu, sigma = load_stats(stats)
    return SpeedySpeechInference(ZScore(mu, sigma), model), phone_map, tone_map


def load_waveflow(config, checkpoint_path, cls=None, eval_mode=True):
    """``ConditionalWaveFlow.from_pretrained`` (waveflow.py:827-852): ``checkpoint_path`` without the
    ``.pdparams`` suffix, ``config`` with a ``model`` section (examples/waveflow/config.py:32-41).  The classmethod of the
    same name on ``ConditionalWaveFlow`` calls this with ``eval_mode=False`` (the reference returns a model in training mode)."""
    if cls is None:
        from .waveflow import ConditionalWaveFlow as cls
    cfg = _config(config)
    m = cfg["model"]
    model = cls(upsample_factors=list(m["upsample_factors"]), n_flows=m["n_flows"],
                n_layers=m["n_layers"], n_group=m["n_group"], channels=m["channels"],
                n_mels=cfg["data"]["n_mels"] if "data" in cfg else m.get("n_mels", 80),
                kernel_size=m["kernel_size"])
    path = str(checkpoint_path)
    model.set_state_dict(load_params(path if path.endswith(".pdparams") else path + ".pdparams"))
    if eval_mode:
        model.eval()
    return model


def load_transformer_tts(config, checkpoint, stats, phones_dict=None, idim=None):
    """The acoustic-model half of examples/transformer_tts/synthesize.py:45-75: returns
    ``(TransformerTTSInference, phone_id_map)``.  ``config``: the recipe's yaml (path or dict with ``n_mels`` and
    ``model``, examples/transformer_tts/ljspeech/conf/default.yaml)."""
    from .normalizer import ZScore
    from .transformer_tts import TransformerTTS, TransformerTTSInference
    cfg = _config(config)
    phone_id_map = None
    if phones_dict is not None:
        phone_id_map, vocab = load_phone_id_map(phones_dict)
        idim = vocab if idim is None else idim
    if idim is None:
        raise ValueError("load_transformer_tts: give phones_dict or idim")
    model = TransformerTTS(idim=idim, odim=cfg["n_mels"], **cfg["model"])
    model.set_state_dict(load_params(checkpoint, "main_params"))
    model.eval()
    mu, sigma = load_stats(stats)
    return TransformerTTSInference(ZScore(mu, sigma), model), phone_id_map


def load_tacotron2(config, checkpoint_path):
    """``Tacotron2.from_pretrained`` (models/tacotron2.py:843-883): ``config`` with ``model`` and ``data`` sections
    (examples/tacotron2/config.py), ``checkpoint_path`` without the ``.pdparams`` suffix."""
    from .tacotron2 import Tacotron2
    cfg = _config(config)
    m = cfg["model"]
    model = Tacotron2(vocab_size=m["vocab_size"], n_tones=m.get("n_tones"), d_mels=cfg["data"]["n_mels"],
                      d_encoder=m["d_encoder"], encoder_conv_layers=m["encoder_conv_layers"],
                      encoder_kernel_size=m["encoder_kernel_size"], d_prenet=m["d_prenet"],
                      d_attention_rnn=m["d_attention_rnn"], d_decoder_rnn=m["d_decoder_rnn"],
                      attention_filters=m["attention_filters"], attention_kernel_size=m["attention_kernel_size"],
                      d_attention=m["d_attention"], d_postnet=m["d_postnet"],
                      postnet_kernel_size=m["postnet_kernel_size"], postnet_conv_layers=m["postnet_conv_layers"],
                      reduction_factor=m["reduction_factor"], p_encoder_dropout=m["p_encoder_dropout"],
                      p_prenet_dropout=m["p_prenet_dropout"], p_attention_dropout=m["p_attention_dropout"],
                      p_decoder_dropout=m["p_decoder_dropout"], p_postnet_dropout=m["p_postnet_dropout"],
                      d_global_condition=m.get("d_global_condition"), use_stop_token=m["use_stop_token"])
    path = str(checkpoint_path)
    model.set_state_dict(load_params(path if path.endswith(".pdparams") else path + ".pdparams"))
    model.eval()
    return model
