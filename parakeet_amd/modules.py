"""Generic primitives of ``parakeet.modules`` on the engine's kernels.

``scaled_dot_product_attention`` (parakeet/modules/attention.py:22-58),
``sinusoid_position_encoding`` (parakeet/modules/positional_encoding.py:20-39) and
``Conv1dBatchNorm`` (parakeet/modules/conv.py:186-260, eval mode) with the reference's signatures.
"""
import ctypes as C

import numpy as np
import torch

from . import _capi
from .runtime import Context, dptr, to_numpy_f32, wrap


def sinusoid_position_encoding(num_positions, feature_size, omega=1.0, start_pos=0, dtype=None):
    ctx = Context.get()
    out = ctx.empty((num_positions, feature_size))
    _capi.check(ctx.lib.pk_op_sinusoid_position_encoding(ctx.handle, int(num_positions), int(feature_size),
                                                         C.c_float(omega), int(start_pos), dptr(out)))
    return wrap(out)


def expand(encodings, durations):
    """parakeet/modules/expansion.py:19-37: (B, T, C), (B, T) integer durations -> (B, t_dec, C), token t repeated
    durations[b][t] times, shorter sequences zero-padded (the reference builds a dense 0/1 matrix and matmuls)."""
    ctx = Context.get()
    enc = ctx.to_device(encodings).contiguous()
    d = np.ascontiguousarray(np.asarray(durations.cpu() if isinstance(durations, torch.Tensor) else durations)
                             .astype(np.int64))
    B, T, Cc = enc.shape
    assert d.shape == (B, T), "durations must be (B, T)"
    t_dec = max(int(np.maximum(d, 0).sum(-1).max()), 0) if d.size else 0
    out = ctx.empty((B, t_dec, Cc))
    _capi.check(ctx.lib.pk_op_expand(ctx.handle, dptr(enc), d.ctypes.data_as(C.POINTER(C.c_int64)), B, T, Cc, t_dec,
                                     dptr(out) if t_dec else None))
    return wrap(out)


def scaled_dot_product_attention(q, k, v, mask=None, dropout=0.0, training=True):
    if dropout and training:
        raise NotImplementedError("attention dropout is a training-time path")
    ctx = Context.get()
    q, k, v = ctx.to_device(q), ctx.to_device(k), ctx.to_device(v)
    lead = q.shape[:-2]
    B = int(np.prod(lead)) if lead else 1
    Tq, d = q.shape[-2], q.shape[-1]
    Tk, dv = k.shape[-2], v.shape[-1]
    q2, k2, v2 = q.reshape(B, Tq, d), k.reshape(B, Tk, d), v.reshape(B, Tk, dv)
    mptr, mode = None, 0
    if mask is not None:
        m = ctx.to_device(mask)
        while m.dim() < q.dim():
            m = m.unsqueeze(0)
        if m.shape[-2] == 1:
            m2, mode = m.expand(*lead, 1, Tk).reshape(B, 1, Tk).contiguous(), 0
        elif int(np.prod(m.shape[:-2])) == 1 and B > 1:
            m2, mode = m.reshape(1, Tq, Tk).contiguous(), 2
        else:
            m2, mode = m.expand(*lead, Tq, Tk).reshape(B, Tq, Tk).contiguous(), 1
        mptr = dptr(m2)
    out = ctx.empty((B, Tq, dv))
    w = ctx.empty((B, Tq, Tk))
    _capi.check(ctx.lib.pk_op_scaled_dot_product_attention(ctx.handle, dptr(q2), dptr(k2), dptr(v2), mptr, mode,
                                                           B, Tq, Tk, d, dv, dptr(out), dptr(w)))
    return wrap(out.reshape(*lead, Tq, dv)), wrap(w.reshape(*lead, Tq, Tk))


class Conv1dBatchNorm:
    """Conv1D followed by BatchNorm1D (eval mode); state-dict keys conv.weight [Cout,Cin,k], conv.bias,
    bn.weight, bn.bias, bn._mean, bn._variance."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, weight_attr=None,
                 bias_attr=None, data_format="NCL", momentum=0.9, epsilon=1e-05):
        if stride != 1:
            raise NotImplementedError("Conv1dBatchNorm: stride 1 only")
        if not isinstance(padding, int):
            raise NotImplementedError("Conv1dBatchNorm: int padding only")
        self.cin, self.cout, self.k, self.pad = in_channels, out_channels, kernel_size, padding
        self.data_format, self.eps = data_format, epsilon
        self._state = None
        self.training = True

    def set_state_dict(self, state):
        self._state = {k: to_numpy_f32(v) for k, v in state.items()}

    def eval(self):
        self.training = False
        return self

    def forward(self, x):
        if self._state is None:
            raise RuntimeError("Conv1dBatchNorm: parameters were never set")
        ctx = Context.get()
        x = ctx.to_device(x)
        if self.data_format == "NCL":
            x = x.transpose(1, 2).contiguous()
        B, T, _ = x.shape
        tout = T + 2 * self.pad - self.k + 1
        y = ctx.empty((B, tout, self.cout))
        s = self._state
        f = _capi.fptr
        _capi.check(ctx.lib.pk_op_conv1d_batchnorm_nlc(
            ctx.handle, dptr(x), B, T, self.cin, self.cout, self.k, self.pad, f(s["conv.weight"]),
            f(s["conv.bias"]) if "conv.bias" in s else None, f(s["bn.weight"]), f(s["bn.bias"]), f(s["bn._mean"]),
            f(s["bn._variance"]), C.c_float(self.eps), dptr(y)))
        if self.data_format == "NCL":
            y = y.transpose(1, 2).contiguous()
        return wrap(y)

    __call__ = forward


class Conv1dCell:
    """Conv1dCell (modules/conv.py:22-183): a causal dilated Conv1D used like an RNN cell -- ``start_sequence()`` then one
    ``add_input(x_t (B, Cin)) -> y_t (B, Cout)`` per step, the last ``receptive_field`` inputs kept in a device buffer.
    State-dict keys ``weight`` [Cout, Cin, k] and ``bias``.  (No model of the reference uses it since WaveNet was removed.)"""

    def __init__(self, in_channels, out_channels, kernel_size, dilation=1, weight_attr=None, bias_attr=None):
        k = kernel_size[0] if isinstance(kernel_size, (tuple, list)) else kernel_size
        d = dilation[0] if isinstance(dilation, (tuple, list)) else dilation
        self.cin, self.cout, self.k, self.dilation = in_channels, out_channels, int(k), int(d)
        self._r = 1 + (self.k - 1) * self.dilation
        self._has_bias = bias_attr is not False
        self._w = self._b = self._buffer = None
        self._started = False
        self.training = True

    @property
    def receptive_field(self):
        return self._r

    def set_state_dict(self, state):
        ctx = Context.get()
        self._w = ctx.to_device(to_numpy_f32(state["weight"]).reshape(self.cout, self.cin, self.k))
        self._b = ctx.to_device(to_numpy_f32(state["bias"]).reshape(self.cout)) if "bias" in state else None

    def eval(self):
        self.training = False
        return self

    def start_sequence(self):
        if self.training:
            raise Exception("only use start_sequence in evaluation")   # conv.py:108-109
        self._buffer, self._started = None, True

    def add_input(self, x_t):
        if self._w is None:
            raise RuntimeError("Conv1dCell: parameters were never set")
        if not self._started:
            raise RuntimeError("Conv1dCell: call start_sequence() first")
        ctx = Context.get()
        x = ctx.to_device(x_t)
        B = x.shape[0]
        if x.dim() != 2 or x.shape[1] != self.cin:
            raise ValueError(f"Conv1dCell.add_input: expected (B, {self.cin}), got {tuple(x.shape)}")
        if self._r > 1 and self._buffer is None:                           # initialize_buffer (:129-139)
            self._buffer = torch.zeros((B, self.cin, self._r), dtype=torch.float32, device=ctx.device)
        y = ctx.empty((B, self.cout))
        _capi.check(ctx.lib.pk_op_conv1d_cell_step(ctx.handle, None if self._buffer is None else dptr(self._buffer), dptr(x),
                                                   dptr(self._w), None if self._b is None else dptr(self._b), B, self.cin,
                                                   self.cout, self.k, self.dilation, dptr(y)))
        return wrap(y)


class Linear:
    """nn.Linear with Paddle's [in, out] weight, on the engine's GEMM (Conv1D k = 1 without batch norm)."""

    def __init__(self, in_features, out_features):
        self.cin, self.cout = in_features, out_features
        self.weight = self.bias = None

    def set(self, weight, bias):
        w = to_numpy_f32(weight)
        assert w.shape == (self.cin, self.cout), f"Linear weight must be [in, out] = {(self.cin, self.cout)}"
        self.weight = np.ascontiguousarray(w.T.reshape(self.cout, self.cin, 1))
        self.bias = None if bias is None else to_numpy_f32(bias).reshape(self.cout)

    def __call__(self, x):
        ctx = Context.get()
        x = ctx.to_device(x)
        lead = x.shape[:-1]
        x2 = x.reshape(1, -1, self.cin).contiguous()
        y = ctx.empty((1, x2.shape[1], self.cout))
        f = _capi.fptr
        _capi.check(ctx.lib.pk_op_conv1d_batchnorm_nlc(
            ctx.handle, dptr(x2), 1, x2.shape[1], self.cin, self.cout, 1, 0, f(self.weight),
            None if self.bias is None else f(self.bias), None, None, None, None, C.c_float(1e-5), dptr(y)))
        return y.reshape(*lead, self.cout)


class MultiheadAttention:
    """parakeet/modules/attention.py:178-255 (eval mode): affine_q/k/v -> split heads ->
    scaled_dot_product_attention (float mask, additive -1e9) -> concat heads -> affine_o.
    State-dict keys affine_{q,k,v,o}.{weight [in, out], bias}.  Returns (out, attention_weights (B, h, Tq, Tk)).
    The head split / concat are layout moves done with tensor views; the math runs on the engine."""

    def __init__(self, model_dim, num_heads, dropout=0.0, k_dim=None, v_dim=None):
        if model_dim % num_heads != 0:
            raise ValueError("model_dim must be divisible by num_heads")   # attention.py:213-214
        depth = model_dim // num_heads
        k_dim, v_dim = k_dim or depth, v_dim or depth
        self.affine_q = Linear(model_dim, num_heads * k_dim)
        self.affine_k = Linear(model_dim, num_heads * k_dim)
        self.affine_v = Linear(model_dim, num_heads * v_dim)
        self.affine_o = Linear(num_heads * v_dim, model_dim)
        self.num_heads, self.model_dim, self.dropout = num_heads, model_dim, dropout
        self.training = True

    def set_state_dict(self, state):
        for nm in ("q", "k", "v", "o"):
            getattr(self, "affine_" + nm).set(state[f"affine_{nm}.weight"], state.get(f"affine_{nm}.bias"))

    def eval(self):
        self.training = False
        return self

    def _split(self, x):
        B, T, _ = x.shape
        return x.reshape(B, T, self.num_heads, -1).permute(0, 2, 1, 3).contiguous()   # (B, h, T, C)

    def forward(self, q, k, v, mask):
        if self.dropout and self.training:
            raise NotImplementedError("attention dropout is a training-time path")
        ctx = Context.get()
        q, k, v = self._split(self.affine_q(q)), self._split(self.affine_k(k)), self._split(self.affine_v(v))
        m = None if mask is None else ctx.to_device(mask).unsqueeze(1)                  # the h dim (:242)
        ctxv, w = scaled_dot_product_attention(q, k, v, m, 0.0, False)
        B, h, T, C_ = ctxv.shape
        merged = ctxv.as_subclass(torch.Tensor).permute(0, 2, 1, 3).reshape(B, T, h * C_).contiguous()
        return wrap(self.affine_o(merged)), w

    __call__ = forward
