"""Paddle-semantics primitives on torch-CPU tensors (oracle; test infrastructure).

Each helper states which Paddle op it stands for and which semantic choice it
encodes.  Items marked [paddle-semantics] come from Paddle's API
documentation, not from an executed Paddle build (see oracle/__init__.py).

Weights are passed as a flat ``{state_dict_key: tensor}`` mapping with the key
names and array layouts of the reference's ``state_dict()``.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def as_tensor(a, dtype):
    if isinstance(a, torch.Tensor):
        return a.to(dtype)
    return torch.as_tensor(np.asarray(a)).to(dtype)


class Weights:
    """Dict view that converts to a fixed float dtype on access."""

    def __init__(self, state, dtype=torch.float32, prefix=""):
        self.state = state
        self.dtype = dtype
        self.prefix = prefix

    def sub(self, prefix):
        return Weights(self.state, self.dtype, self.prefix + prefix)

    def has(self, key):
        return (self.prefix + key) in self.state

    def __getitem__(self, key):
        return as_tensor(self.state[self.prefix + key], self.dtype)


def linear(x, w, b=None):
    """paddle.nn.Linear: y = x @ W + b with W stored [in, out] [paddle-semantics]."""
    y = torch.matmul(x, w)
    return y if b is None else y + b


def conv1d(x, w, b=None, padding=0, dilation=1):
    """paddle.nn.Conv1D on NCL input, weight [Cout, Cin, k], zero padding."""
    return F.conv1d(x, w, b, stride=1, padding=padding, dilation=dilation)


def layer_norm(x, g, b, eps=1e-5):
    """paddle.nn.LayerNorm over the last axis, epsilon 1e-5 [paddle-semantics]."""
    return F.layer_norm(x, (x.shape[-1],), g, b, eps)


def batch_norm_eval(x, g, b, mean, var, eps=1e-5):
    """paddle.nn.BatchNorm1D in eval mode on NCL input: running stats
    ``_mean`` / ``_variance``, epsilon 1e-5 [paddle-semantics]."""
    shp = (1, -1, 1)
    return (x - mean.view(shp)) / torch.sqrt(var.view(shp) + eps) * g.view(shp) + b.view(shp)


def round_half_away(x):
    """paddle.round: ties round away from zero [paddle-semantics]
    (torch.round / numpy.round tie to even)."""
    return torch.sign(x) * torch.floor(torch.abs(x) + 0.5)


def masked_fill(x, mask, value):
    """parakeet/modules/masked_fill.py:28-37 -- where(mask, value, x)."""
    return torch.where(mask, torch.full_like(x, value), x)


def make_pad_mask(lengths):
    """parakeet/modules/nets_utils.py:54-93 -- True on padded positions."""
    lengths = [int(v) for v in lengths]
    maxlen = max(lengths)
    rng = torch.arange(maxlen).unsqueeze(0)
    return rng >= torch.tensor(lengths).unsqueeze(-1)


def make_non_pad_mask(lengths):
    """parakeet/modules/nets_utils.py:96-125."""
    return ~make_pad_mask(lengths)


def sinusoid_table(length, d_model, dtype):
    """PositionalEncoding.extend_pe, fastspeech2_transformer/embedding.py:46-62.
    The table is built in float32 (as the reference does) and then cast."""
    pe = torch.zeros(length, d_model, dtype=torch.float32)
    position = torch.arange(0, length, dtype=torch.float32).unsqueeze(1)
    div_term = torch.exp(
        torch.arange(0, d_model, 2, dtype=torch.float32) * -(math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.to(dtype)


def fold_weight_norm(state):
    """nn.utils.remove_weight_norm: w = g * v / ||v|| with the norm taken over
    every axis but 0 (weight_norm(dim=0)), ``weight_g`` stored 1-D
    (tests/unit/test_pwg.py:131-132).  Returns a new dict with plain
    ``weight`` keys; entries without a _g/_v pair are passed through."""
    out = {}
    for k, v in state.items():
        if k.endswith("weight_v"):
            base = k[: -len("weight_v")]
            g = np.asarray(state[base + "weight_g"], dtype=np.float64).reshape(-1)
            vv = np.asarray(v, dtype=np.float64)
            nrm = np.sqrt((vv.reshape(vv.shape[0], -1) ** 2).sum(axis=1))
            w = vv * (g / nrm).reshape((-1,) + (1,) * (vv.ndim - 1))
            out[base + "weight"] = w.astype(np.asarray(v).dtype)
        elif k.endswith("weight_g"):
            continue
        else:
            out[k] = v
    return out
