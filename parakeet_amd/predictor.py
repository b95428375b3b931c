"""A ``paddle.inference``-shaped front for the engine's inference objects.

The reference's deployment example (examples/speedyspeech/baker/inference.py:53-130) drives two exported
static graphs through ``create_predictor`` / ``get_input_handle`` / ``copy_from_cpu`` / ``run`` /
``get_output_handle`` / ``copy_to_cpu``.  ``Predictor`` offers exactly those calls around any callable of this
package (``SpeedySpeechInference``, ``FastSpeech2Inference``, ``PWGInference`` ...), so that loop keeps its
shape; nothing is traced or exported -- ``run`` calls the engine.

    am = Predictor(speedyspeech_inference, input_names=["phones", "tones"])
    voc = Predictor(pwg_inference, input_names=["logmel"])
    h = am.get_input_handle(am.get_input_names()[0]); h.reshape(phones.shape); h.copy_from_cpu(phones) ...
    am.run(); mel = am.get_output_handle(am.get_output_names()[0]).copy_to_cpu()

The serving entry proper is ``Config`` + ``create_predictor(config)``: the reference's script names two exported static
graphs per model (``<dir>/speedyspeech.pdmodel`` + ``.pdiparams``, :53-66).  This engine does not interpret Paddle program
descriptions; ``Config`` takes the SAME two paths and resolves, in the same directory, the recipe artefacts the graphs were
exported from -- the model's yaml config, its ``.pdz`` checkpoint, its ``*_stats.npy`` and the phone / tone id maps (the
files ``synthesize_e2e.py`` of each recipe takes as arguments) -- and ``create_predictor`` builds the engine model from them
(``parakeet_amd.checkpoint``), names the inputs as the exported graphs do (``InputSpec`` order of the recipe's
``jit.to_static`` call: speedyspeech.py phones, tones; fastspeech2 text; pwg logmel) and returns a ``Predictor``:

    cfg = Config(f"{d}/speedyspeech.pdmodel", f"{d}/speedyspeech.pdiparams")   # or Config(model="speedyspeech", model_dir=d)
    cfg.enable_use_gpu(100, 0); cfg.enable_memory_optim()
    predictor = create_predictor(cfg)
"""
import glob
import re
import os

import numpy as np
import torch


class _Handle:
    def __init__(self, name):
        self.name = name
        self._value = None
        self._shape = None

    def reshape(self, shape):
        self._shape = tuple(int(s) for s in shape)

    def copy_from_cpu(self, array):
        a = np.asarray(array)
        if self._shape is not None and tuple(a.shape) != self._shape:
            a = a.reshape(self._shape)
        self._value = a

    def share_external_data(self, tensor):
        """Device-side hand-over (no host hop): any tensor the wrapped callable accepts."""
        self._value = tensor

    def copy_to_cpu(self):
        v = self._value
        if v is None:
            raise RuntimeError(f"output {self.name!r} is empty: call run() first")
        if hasattr(v, "numpy") and not isinstance(v, np.ndarray):
            v = v.cpu().numpy() if isinstance(v, torch.Tensor) else v.numpy()
        return np.asarray(v)

    def shape(self):
        v = self._value
        return list(v.shape) if v is not None else (list(self._shape) if self._shape else [])


class Predictor:
    def __init__(self, model, input_names, output_names=("out",)):
        self._model = model
        self._inputs = {n: _Handle(n) for n in input_names}
        self._input_names = list(input_names)
        self._outputs = {n: _Handle(n) for n in output_names}
        self._output_names = list(output_names)

    def get_input_names(self):
        return list(self._input_names)

    def get_output_names(self):
        return list(self._output_names)

    def get_input_handle(self, name):
        return self._inputs[name]

    def get_output_handle(self, name):
        return self._outputs[name]

    def run(self):
        args = []
        for n in self._input_names:
            v = self._inputs[n]._value
            if v is None:
                raise RuntimeError(f"input {n!r} was not set")
            args.append(v)
        out = self._model(*args)
        outs = out if isinstance(out, (tuple, list)) else (out,)
        if len(outs) != len(self._output_names):
            raise RuntimeError(f"model returned {len(outs)} outputs, predictor declares {len(self._output_names)}")
        for n, o in zip(self._output_names, outs):
            self._outputs[n]._value = o
        return True


# model kind -> (input names in the exported graph's InputSpec order, output name)
_KINDS = {
    "speedyspeech": (["phones", "tones"], "logmel"),     # examples/speedyspeech/baker/synthesize_e2e.py:77-83
    "fastspeech2": (["text"], "logmel"),                 # examples/fastspeech2/baker/synthesize_e2e.py (to_static InputSpec)
    "pwg": (["logmel"], "wav"),                          # :85-89
}


def _iteration(path):
    """Sort key of a snapshot: its LAST integer run (``snapshot_iter_10000.pdz`` > ``snapshot_iter_9999.pdz``; a plain
    string sort puts them the other way round), then the name."""
    runs = re.findall(r"\d+", os.path.basename(path))
    return (int(runs[-1]) if runs else -1, os.path.basename(path))


def _first(model_dir, patterns, what, kind, arg=None):
    """``patterns``: most specific first.  In a directory that holds several model kinds (the reference's inference directories
    do: speedyspeech.pdmodel next to pwg.pdmodel) a careless fall-back would silently load the OTHER model's weights or
    statistics, so there: files that carry another kind's name never match, the catch-all patterns (``*.pdz``, ``*stats.npy``)
    are not used, and the vocoder -- whose recipe prefixes every file with its name -- takes nothing but patterns that name it
    (``snapshot_iter_*.pdz`` / ``speech_stats.npy`` / ``default.yaml`` are the ACOUSTIC recipes' names).  ``arg``: the Config
    argument that names the file explicitly (None: the lookup is not kind-sensitive, e.g. the phone id map)."""
    present = [k for k in _KINDS if glob.glob(os.path.join(model_dir, k + "*"))]
    shared = arg is not None and len(set(present) | {kind}) > 1
    others = tuple(k for k in _KINDS if k != kind)
    for pat in patterns:
        if shared and kind not in pat and (pat.startswith("*") or kind == "pwg"):
            continue
        hits = glob.glob(os.path.join(model_dir, pat))
        if shared:
            hits = [h for h in hits if not os.path.basename(h).startswith(others)]
        if hits:
            return sorted(hits, key=_iteration)[-1]   # several snapshots: the highest iteration
    hint = f"; {model_dir} holds several model kinds ({', '.join(present)}): pass {arg}=" if shared else ""
    raise FileNotFoundError(f"{kind}: no {what} in {model_dir} (looked for {', '.join(patterns)}){hint}")


class Config:
    """``paddle.inference.Config(prog_file, params_file)`` for this engine (see the module docstring).  The model kind is
    the stem of ``prog_file`` (speedyspeech | fastspeech2 | pwg) or ``model=``; artefacts are looked up in the directory of
    ``prog_file`` / ``model_dir=`` by the recipes' file names, or given explicitly:
      config      <kind>.yaml | <kind>_default.yaml | default.yaml
      checkpoint  <kind>*.pdz | snapshot_iter_*.pdz | *.pdz      (.pdparams too)
      stat        <kind>_stats.npy | speech_stats.npy | *stats.npy
      phones_dict phone_id_map.txt | phones.txt ;  tones_dict tone_id_map.txt | tones.txt   (acoustic models)"""

    def __init__(self, prog_file=None, params_file=None, model=None, model_dir=None, config=None, checkpoint=None, stat=None,
                 phones_dict=None, tones_dict=None):
        if model is None:
            if prog_file is None:
                raise ValueError("Config needs prog_file (\"<dir>/<kind>.pdmodel\") or model=")
            model = os.path.splitext(os.path.basename(str(prog_file)))[0]
        if model not in _KINDS:
            raise ValueError(f"unknown model kind {model!r}: {sorted(_KINDS)}")
        self.model = model
        self.model_dir = str(model_dir) if model_dir is not None else (os.path.dirname(str(prog_file)) if prog_file else ".")
        self.prog_file, self.params_file = prog_file, params_file
        self.artefacts = dict(config=config, checkpoint=checkpoint, stat=stat, phones_dict=phones_dict, tones_dict=tones_dict)
        self.device_id = 0
        self.use_gpu = True
        self.memory_optim = False

    # -- the calls the reference's script makes on a Config (:56-57, :64-65)
    def enable_use_gpu(self, memory_pool_init_size_mb=100, device_id=0):
        """The engine's workspaces are grow-only device buffers per handle: the pool size has no counterpart."""
        self.use_gpu, self.device_id = True, int(device_id)

    def disable_gpu(self):
        raise RuntimeError("parakeet_amd has no CPU execution path")

    def enable_memory_optim(self):
        self.memory_optim = True

    def resolve(self):
        k, d, a = self.model, self.model_dir, dict(self.artefacts)
        if a["config"] is None:
            a["config"] = _first(d, [f"{k}.yaml", f"{k}_default.yaml", "default.yaml"], "yaml config", k, "config")
        if a["checkpoint"] is None:
            a["checkpoint"] = _first(d, [f"{k}*.pdz", f"{k}*.pdparams", "snapshot_iter_*.pdz", "*.pdz"], "checkpoint", k, "checkpoint")
        if a["stat"] is None:
            a["stat"] = _first(d, [f"{k}_stats.npy", "speech_stats.npy", "*stats.npy"], "statistics file", k, "stat")
        if k != "pwg":
            if a["phones_dict"] is None:
                a["phones_dict"] = _first(d, ["phone_id_map.txt", "phones.txt"], "phone id map", k)
            if k == "speedyspeech" and a["tones_dict"] is None:
                a["tones_dict"] = _first(d, ["tone_id_map.txt", "tones.txt"], "tone id map", k)
        return a


def create_predictor(config, input_names=None, output_names=("out",)):
    """``create_predictor(Config)``: build the engine model the Config describes and wrap it (module docstring);
    ``create_predictor(callable, input_names)``: wrap an inference object that already exists."""
    if not isinstance(config, Config):
        return Predictor(config, input_names, output_names)
    from . import checkpoint as ck
    from .runtime import Context
    a = config.resolve()
    Context.get(config.device_id)
    ins, out = _KINDS[config.model]
    if config.model == "speedyspeech":
        inf, _, _ = ck.load_speedyspeech(a["config"], a["checkpoint"], a["stat"], a["phones_dict"], a["tones_dict"])
        model = lambda phones, tones: inf(np.asarray(phones), np.asarray(tones))
    elif config.model == "fastspeech2":
        inf, _ = ck.load_fastspeech2(a["config"], a["checkpoint"], a["stat"], a["phones_dict"])
        model = lambda text: inf(np.asarray(text))
    else:
        inf = ck.load_pwg(a["config"], a["checkpoint"], a["stat"])
        model = lambda logmel: inf(logmel)[:, 0]       # the exported graph returns wav (T,) for sf.write (:128-131)
    p = Predictor(model, ins, (out,))
    p.inference, p.config = inf, config
    return p
