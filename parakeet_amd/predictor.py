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
"""
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


def create_predictor(model, input_names, output_names=("out",)):
    return Predictor(model, input_names, output_names)
