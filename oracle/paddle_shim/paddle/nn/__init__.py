"""paddle.nn stand-ins on torch.nn.Module.  Parameter names and array layouts follow Paddle:
Linear.weight [in, out]; ConvND.weight [Cout, Cin/groups, *k]; BatchNorm buffers ``_mean`` /
``_variance``; LayerNorm / BatchNorm epsilon 1e-5  [paddle-semantics, from Paddle's API docs]."""
import math

import numpy as np
import torch
import torch.nn.functional as TF

from .. import Tensor, _dt, _wrap
from . import functional  # noqa: F401
from . import initializer  # noqa: F401


def _param(*shape):
    p = torch.nn.Parameter(torch.empty(*shape), requires_grad=False)
    with torch.no_grad():
        if p.dim() >= 2:
            fan_in = int(np.prod(shape[1:]))
            p.uniform_(-1 / math.sqrt(fan_in), 1 / math.sqrt(fan_in))
        else:
            p.zero_()
    return p


class Layer(torch.nn.Module):
    def __init__(self, *a, **k):
        super().__init__()

    def sublayers(self, include_self=False):
        mods = list(self.modules())
        return mods if include_self else mods[1:]

    def register_buffer(self, name, tensor, persistable=True):
        # buffers registered under an auto-generated tensor name (WaveFlow's perms) stay out of state_dict
        persistent = persistable and not str(name).startswith("generated_tensor")
        torch.nn.Module.register_buffer(self, name, tensor, persistent=persistent)

    def named_sublayers(self, prefix="", include_self=False):
        for n, m in self.named_modules(prefix=prefix):
            if m is self and not include_self:
                continue
            yield n, m

    def add_sublayer(self, name, layer):
        self.add_module(str(name), layer)
        return layer

    def set_state_dict(self, state):
        own = torch.nn.Module.state_dict(self)   # persistent entries only; tensors share storage
        missing = [k for k in own if k not in state]
        extra = [k for k in state if k not in own]
        assert not missing and not extra, f"state dict mismatch: missing {missing[:5]}, unexpected {extra[:5]}"
        with torch.no_grad():
            for k, v in state.items():
                t = torch.as_tensor(np.asarray(v))
                assert tuple(t.shape) == tuple(own[k].shape) or t.numel() == own[k].numel(), (k, t.shape, own[k].shape)
                own[k].copy_(t.reshape(own[k].shape).to(own[k].dtype))

    def __call__(self, *a, **k):
        return super().__call__(*a, **k)


class Linear(Layer):
    def __init__(self, in_features, out_features, weight_attr=None, bias_attr=None, name=None):
        super().__init__()
        self.weight = _param(in_features, out_features)
        self.bias = None if bias_attr is False else _param(out_features)

    def forward(self, x):
        y = torch.matmul(x, self.weight)
        return _wrap(y if self.bias is None else y + self.bias)


class _ConvNd(Layer):
    def _setup(self, cin, cout, k, stride, padding, dilation, groups, bias_attr, transposed=False):
        self.stride, self.padding, self.dilation, self.groups = stride, padding, dilation, groups
        shape = (cin, cout // groups, *k) if transposed else (cout, cin // groups, *k)
        self.weight = _param(*shape)
        self.bias = None if bias_attr is False else _param(cout)


class Conv1D(_ConvNd):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 padding_mode="zeros", weight_attr=None, bias_attr=None, data_format="NCL"):
        super().__init__()
        self._setup(in_channels, out_channels, (kernel_size,), stride, padding, dilation, groups, bias_attr)
        self._data_format = data_format

    def forward(self, x):
        return functional.conv1d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups,
                                 data_format=self._data_format)


def _pair(v):
    return tuple(v) if isinstance(v, (list, tuple)) else (v, v)


class Conv2D(_ConvNd):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 padding_mode="zeros", weight_attr=None, bias_attr=None, data_format="NCHW"):
        super().__init__()
        self._setup(in_channels, out_channels, _pair(kernel_size), _pair(stride), padding, _pair(dilation),
                    groups, bias_attr)

    def forward(self, x):
        return functional.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)


class Conv2DTranspose(_ConvNd):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, output_padding=0,
                 groups=1, dilation=1, weight_attr=None, bias_attr=None, data_format="NCHW"):
        super().__init__()
        self._setup(in_channels, out_channels, _pair(kernel_size), _pair(stride), padding, _pair(dilation),
                    groups, bias_attr, transposed=True)
        self.output_padding = output_padding
        self._kernel_size = list(_pair(kernel_size))
        self._stride = list(_pair(stride))

    def forward(self, x):
        return functional.conv2d_transpose(x, self.weight, self.bias, self.stride, self.padding,
                                           self.output_padding, self.dilation, self.groups)


class LayerNorm(Layer):
    def __init__(self, normalized_shape, epsilon=1e-05, weight_attr=None, bias_attr=None, name=None):
        super().__init__()
        if isinstance(normalized_shape, int):
            normalized_shape = (normalized_shape,)
        self._shape = tuple(normalized_shape)
        self._eps = epsilon
        self.weight = torch.nn.Parameter(torch.ones(self._shape), requires_grad=False)
        self.bias = torch.nn.Parameter(torch.zeros(self._shape), requires_grad=False)

    def forward(self, x):
        return _wrap(TF.layer_norm(x, self._shape, self.weight, self.bias, self._eps))


class BatchNorm1D(Layer):
    def __init__(self, num_features, momentum=0.9, epsilon=1e-05, weight_attr=None, bias_attr=None,
                 data_format="NCL", use_global_stats=None, name=None):
        super().__init__()
        self._eps = epsilon
        self.weight = torch.nn.Parameter(torch.ones(num_features), requires_grad=False)
        self.bias = torch.nn.Parameter(torch.zeros(num_features), requires_grad=False)
        self.register_buffer("_mean", torch.zeros(num_features))
        self.register_buffer("_variance", torch.ones(num_features))
        self._data_format = data_format

    def forward(self, x):
        assert not self.training, "only eval-mode batch norm is restated"
        if self._data_format == "NLC" and x.dim() == 3:
            y = TF.batch_norm(x.transpose(1, 2), self._mean, self._variance, self.weight, self.bias, False, 0.0,
                              self._eps)
            return _wrap(y.transpose(1, 2))
        return _wrap(TF.batch_norm(x, self._mean, self._variance, self.weight, self.bias, False, 0.0, self._eps))


class BatchNorm2D(BatchNorm1D):
    """NCHW eval-mode batch norm (same arithmetic as BatchNorm1D on the channel axis)."""


class Embedding(Layer):
    def __init__(self, num_embeddings, embedding_dim, padding_idx=None, sparse=False, weight_attr=None, name=None):
        super().__init__()
        self.weight = _param(num_embeddings, embedding_dim)
        self._padding_idx = padding_idx if (padding_idx is None or padding_idx >= 0) else num_embeddings + padding_idx

    def forward(self, ids):
        out = TF.embedding(ids.to(torch.int64), self.weight)
        if self._padding_idx is not None:  # output rows at padding_idx are zeros [paddle-semantics]
            out = torch.where((ids == self._padding_idx).unsqueeze(-1), torch.zeros_like(out), out)
        return _wrap(out)


class LSTMCell(Layer):
    """paddle.nn.LSTMCell: weight_ih [4H, in], weight_hh [4H, H], bias_ih / bias_hh [4H]; gates in the order
    i, f, g (cell candidate), o along the 4H axis; c' = f * c + i * tanh(g), h' = o * tanh(c')
    [paddle-semantics, from Paddle's API documentation of LSTMCell].  forward -> (h', (h', c'))."""

    def __init__(self, input_size, hidden_size, weight_ih_attr=None, weight_hh_attr=None, bias_ih_attr=None,
                 bias_hh_attr=None, name=None):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        k = 1.0 / math.sqrt(hidden_size)
        for nm, shape in (("weight_ih", (4 * hidden_size, input_size)), ("weight_hh", (4 * hidden_size, hidden_size)),
                          ("bias_ih", (4 * hidden_size,)), ("bias_hh", (4 * hidden_size,))):
            p = torch.nn.Parameter(torch.empty(*shape).uniform_(-k, k), requires_grad=False)
            self.register_parameter(nm, p)

    def forward(self, inputs, states=None):
        if states is None:
            z = torch.zeros(inputs.shape[0], self.hidden_size, dtype=inputs.dtype)
            states = (z, z)
        h, c = states
        gates = torch.matmul(inputs, self.weight_ih.t()) + self.bias_ih + torch.matmul(h, self.weight_hh.t()) + self.bias_hh
        i, f, g, o = torch.chunk(gates, 4, dim=-1)
        c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        h2 = torch.sigmoid(o) * torch.tanh(c2)
        return _wrap(h2), (_wrap(h2), _wrap(c2))


class GRUCell(Layer):
    """paddle.nn.GRUCell: weight_ih [3H, in], weight_hh [3H, H], bias_ih / bias_hh [3H]; gates in the order r, z, c along
    the 3H axis; r = sigmoid(W_ir x + b_ir + W_hr h + b_hr), z likewise, c = tanh(W_ic x + b_ic + r * (W_hc h + b_hc)),
    h' = z * h + (1 - z) * c [paddle-semantics, from Paddle's API documentation of GRUCell].  forward -> (h', h')."""

    def __init__(self, input_size, hidden_size, weight_ih_attr=None, weight_hh_attr=None, bias_ih_attr=None,
                 bias_hh_attr=None, name=None):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        k = 1.0 / math.sqrt(hidden_size)
        for nm, shape in (("weight_ih", (3 * hidden_size, input_size)), ("weight_hh", (3 * hidden_size, hidden_size)),
                          ("bias_ih", (3 * hidden_size,)), ("bias_hh", (3 * hidden_size,))):
            self.register_parameter(nm, torch.nn.Parameter(torch.empty(*shape).uniform_(-k, k), requires_grad=False))

    def forward(self, inputs, states=None):
        h = torch.zeros(inputs.shape[0], self.hidden_size, dtype=inputs.dtype) if states is None else states
        gx = torch.matmul(inputs, self.weight_ih.t()) + self.bias_ih
        gh = torch.matmul(h, self.weight_hh.t()) + self.bias_hh
        xr, xz, xc = torch.chunk(gx, 3, dim=-1)
        hr, hz, hc = torch.chunk(gh, 3, dim=-1)
        r, z = torch.sigmoid(xr + hr), torch.sigmoid(xz + hz)
        c = torch.tanh(xc + r * hc)
        h2 = z * h + (1.0 - z) * c
        return _wrap(h2), _wrap(h2)


class GRU(torch.nn.ModuleList, Layer):
    """paddle.nn.GRU, forward direction: sublayer "{layer}" is RNN(cell); every cell parameter is also registered under
    the cuDNN-style names weight_ih_l{k}, ... (as LSTM below).  Returns (outputs, final_states (num_layers, B, H))."""

    def __init__(self, input_size, hidden_size, num_layers=1, direction="forward", time_major=False, dropout=0.0,
                 weight_ih_attr=None, weight_hh_attr=None, bias_ih_attr=None, bias_hh_attr=None, name=None):
        torch.nn.ModuleList.__init__(self)
        assert direction == "forward", direction
        self.time_major, self.hidden_size, self.num_layers = time_major, hidden_size, num_layers
        for layer in range(num_layers):
            self.append(_RNN(GRUCell(input_size if layer == 0 else hidden_size, hidden_size)))
            for nm in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                self.register_parameter(f"{nm}_l{layer}", getattr(self[layer].cell, nm))

    def flatten_parameters(self):
        return None

    def forward(self, inputs, initial_states=None, sequence_length=None):
        assert initial_states is None and sequence_length is None, "only the inference call pattern is restated"
        x = inputs if self.time_major else inputs.transpose(0, 1)             # (T, B, C)
        x = x.as_subclass(torch.Tensor)
        finals = []
        for layer in range(self.num_layers):
            cell, state, seq = self[layer].cell, None, []
            for t in range(x.shape[0]):
                h, state = cell(x[t], state)
                seq.append(h)
            x = torch.stack(seq, dim=0)
            finals.append(state)
        y = x if self.time_major else x.transpose(0, 1)
        return _wrap(y), _wrap(torch.stack(finals, 0))


class _BiRNN(Layer):
    def __init__(self, cell_fw, cell_bw):
        super().__init__()
        self.cell_fw, self.cell_bw = cell_fw, cell_bw


class _RNN(Layer):
    def __init__(self, cell):
        super().__init__()
        self.cell = cell


class LSTM(torch.nn.ModuleList, Layer):
    """paddle.nn.LSTM (RNNBase is a LayerList): sublayer "{layer}" is RNN(cell) or BiRNN(cell_fw, cell_bw), and every
    cell parameter is ALSO registered on the LSTM itself under the cuDNN-style names weight_ih_l{k}[_reverse], ...
    (RNNBase.__init__ setattr's them), so a state dict carries both "lstm.0.cell_fw.weight_ih" and
    "lstm.weight_ih_l0" for the same array [paddle-semantics, from the 2.1 source of python/paddle/nn/layer/rnn.py as
    recalled; unverified].  batch-first unless time_major; outputs = concat(forward, backward) on the last axis;
    zero initial states; sequence_length=None processes every step."""

    def __init__(self, input_size, hidden_size, num_layers=1, direction="forward", time_major=False, dropout=0.0,
                 weight_ih_attr=None, weight_hh_attr=None, bias_ih_attr=None, bias_hh_attr=None, name=None):
        torch.nn.ModuleList.__init__(self)
        self.bidirectional = direction in ("bidirectional", "bidirect")
        assert self.bidirectional or direction == "forward", direction
        self.time_major, self.hidden_size, self.num_layers = time_major, hidden_size, num_layers
        nd = 2 if self.bidirectional else 1
        for layer in range(num_layers):
            isz = input_size if layer == 0 else nd * hidden_size
            if self.bidirectional:
                self.append(_BiRNN(LSTMCell(isz, hidden_size), LSTMCell(isz, hidden_size)))
            else:
                self.append(_RNN(LSTMCell(isz, hidden_size)))
            cells = [self[layer].cell_fw, self[layer].cell_bw] if self.bidirectional else [self[layer].cell]
            for d, cell in enumerate(cells):
                suffix = "_reverse" if d == 1 else ""
                for nm in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                    self.register_parameter(f"{nm}_l{layer}{suffix}", getattr(cell, nm))

    def flatten_parameters(self):
        return None

    def forward(self, inputs, initial_states=None, sequence_length=None):
        assert initial_states is None and sequence_length is None, "only the inference call pattern is restated"
        x = inputs.transpose(0, 1) if not self.time_major else inputs          # (T, B, C)
        x = x.as_subclass(torch.Tensor) if isinstance(x, torch.Tensor) else x
        hs, cs = [], []
        for layer in range(self.num_layers):
            cells = [self[layer].cell_fw, self[layer].cell_bw] if self.bidirectional else [self[layer].cell]
            outs = []
            for d, cell in enumerate(cells):
                steps = range(x.shape[0] - 1, -1, -1) if d == 1 else range(x.shape[0])
                state, seq = None, [None] * x.shape[0]
                for t in steps:
                    h, state = cell(x[t], state)
                    seq[t] = h
                outs.append(torch.stack(seq, dim=0))
                hs.append(state[0])
                cs.append(state[1])
            x = torch.cat(outs, dim=-1)
        y = x if self.time_major else x.transpose(0, 1)
        return _wrap(y), (_wrap(torch.stack(hs, 0)), _wrap(torch.stack(cs, 0)))


class Dropout(Layer):
    def __init__(self, p=0.5, axis=None, mode="upscale_in_train", name=None):
        super().__init__()
        self.p = p

    def forward(self, x):
        return functional.dropout(x, self.p, self.training)


class ReLU(Layer):
    def forward(self, x):
        return _wrap(torch.relu(x))


class Tanh(Layer):
    def forward(self, x):
        return _wrap(torch.tanh(x))


class Sigmoid(Layer):
    def forward(self, x):
        return _wrap(torch.sigmoid(x))


class LeakyReLU(Layer):
    def __init__(self, negative_slope=0.01):
        super().__init__()
        self.s = negative_slope

    def forward(self, x):
        return _wrap(TF.leaky_relu(x, self.s))


class Softmax(Layer):
    def __init__(self, axis=-1):
        super().__init__()
        self.axis = axis

    def forward(self, x):
        return _wrap(torch.softmax(x, dim=self.axis))


class Pad1D(Layer):
    def __init__(self, padding, mode="constant", value=0.0, data_format="NCL"):
        super().__init__()
        self.padding = (padding, padding) if isinstance(padding, int) else tuple(padding)
        self.mode, self.value = mode, value

    def forward(self, x):
        if self.mode == "constant":
            return _wrap(TF.pad(x, self.padding, value=self.value))
        return _wrap(TF.pad(x, self.padding, mode=self.mode))


class MSELoss(Layer):
    pass


class L1Loss(Layer):
    pass


class Sequential(torch.nn.Sequential, Layer):
    def __init__(self, *layers):
        torch.nn.Sequential.__init__(self, *layers)


class LayerList(torch.nn.ModuleList, Layer):
    def __init__(self, sublayers=None):
        torch.nn.ModuleList.__init__(self, sublayers)


class _Utils:
    """nn.utils.weight_norm(layer, name='weight', dim=0): weight = g * v / ||v|| with the norm over
    all axes but `dim`; parameters weight_g (1-D, tests/unit/test_pwg.py:131-132) and weight_v."""

    @staticmethod
    def weight_norm(layer, name="weight", dim=0):
        w = getattr(layer, name)
        del layer._parameters[name]
        v = torch.nn.Parameter(w.detach().clone(), requires_grad=False)
        g = torch.nn.Parameter(w.detach().reshape(w.shape[0], -1).norm(dim=1), requires_grad=False)
        layer.register_parameter(name + "_g", g)
        layer.register_parameter(name + "_v", v)

        def hook(mod, inputs):
            vv = getattr(mod, name + "_v")
            gg = getattr(mod, name + "_g")
            nrm = vv.reshape(vv.shape[0], -1).norm(dim=1)
            object.__setattr__(mod, name, vv * (gg / nrm).reshape((-1,) + (1,) * (vv.dim() - 1)))

        layer._wn_hook = layer.register_forward_pre_hook(hook)
        layer._wn_name = name
        hook(layer, None)
        return layer

    @staticmethod
    def remove_weight_norm(layer, name="weight"):
        if not hasattr(layer, "_wn_hook"):
            raise ValueError("weight_norm of '{}' not found in {}".format(name, layer))
        vv, gg = getattr(layer, name + "_v"), getattr(layer, name + "_g")
        nrm = vv.reshape(vv.shape[0], -1).norm(dim=1)
        w = vv * (gg / nrm).reshape((-1,) + (1,) * (vv.dim() - 1))
        layer._wn_hook.remove()
        del layer._wn_hook
        del layer._parameters[name + "_g"], layer._parameters[name + "_v"]
        if name in layer.__dict__:
            del layer.__dict__[name]
        layer.register_parameter(name, torch.nn.Parameter(w.detach(), requires_grad=False))
        return layer


utils = _Utils()
