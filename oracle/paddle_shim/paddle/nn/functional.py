"""paddle.nn.functional stand-ins (paddle argument conventions -> torch)."""
import torch
import torch.nn.functional as TF

from .. import _wrap


def _pad2(padding):
    """paddle conv2d padding: int | [h, w] | [top, bottom, left, right] | 'same'/'valid' (str)."""
    if isinstance(padding, int):
        return (padding, padding), None
    padding = list(padding)
    if len(padding) == 2:
        return (padding[0], padding[1]), None
    if len(padding) == 4:
        top, bottom, left, right = padding
        if top == bottom and left == right:
            return (top, left), None
        return (0, 0), (left, right, top, bottom)   # explicit pre-pad (torch F.pad order: last dim first)
    raise ValueError(padding)


# Paddle's conv kernels call UpdatePaddingAndDilation (paddle/fluid/operators/conv_op.h, release 2.1): with
# padding_algorithm "SAME" the pads become pad_sum = max((out - 1) * stride + k - in, 0), before = pad_sum / 2,
# after = pad_sum - before -- computed from the UNDILATED kernel size -- and the dilation is RESET TO 1.
# [paddle-semantics, unverified here: Paddle cannot be installed in this image.]  True reproduces that; False
# gives the mathematically dilated convolution with symmetric "same" padding d * (k - 1).
SAME_PADDING_RESETS_DILATION = True


def conv1d(x, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, data_format="NCL"):
    if data_format == "NLC":
        y = conv1d(x.transpose(1, 2), weight, bias, stride, padding, dilation, groups, "NCL")
        return _wrap(y.transpose(1, 2))
    if isinstance(padding, str):
        assert padding.lower() == "same" and stride in (1, (1,), [1]), padding
        k = weight.shape[-1]
        d = dilation[0] if isinstance(dilation, (list, tuple)) else dilation
        if SAME_PADDING_RESETS_DILATION:
            d = 1
        pad_sum = d * (k - 1)
        before = pad_sum // 2
        x = TF.pad(x, (before, pad_sum - before))
        return _wrap(TF.conv1d(x, weight, bias, stride=1, padding=0, dilation=d, groups=groups))
    if isinstance(padding, (list, tuple)):
        if len(padding) == 1:
            padding = padding[0]
        elif len(padding) == 2 and padding[0] != padding[1]:
            x = TF.pad(x, (padding[0], padding[1]))
            padding = 0
        else:
            padding = padding[0]
    return _wrap(TF.conv1d(x, weight, bias, stride=stride, padding=padding, dilation=dilation, groups=groups))


def conv2d(x, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, data_format="NCHW"):
    pad, pre = _pad2(padding)
    if pre is not None:
        x = TF.pad(x, pre)
    return _wrap(TF.conv2d(x, weight, bias, stride=stride, padding=pad, dilation=dilation, groups=groups))


def conv2d_transpose(x, weight, bias=None, stride=1, padding=0, output_padding=0, dilation=1, groups=1):
    pad, pre = _pad2(padding)
    assert pre is None
    return _wrap(TF.conv_transpose2d(x, weight, bias, stride=stride, padding=pad, output_padding=output_padding,
                                     groups=groups, dilation=dilation))


def interpolate(x, size=None, scale_factor=None, mode="nearest", align_corners=False, data_format="NCHW"):
    assert mode == "nearest"
    # nearest with an integer scale factor: out[i] = in[i // s]  [paddle-semantics]
    sf = scale_factor if isinstance(scale_factor, (list, tuple)) else (scale_factor,) * (x.dim() - 2)
    out = x
    for d, s in enumerate(sf):
        s = int(s)
        if s != 1:
            out = torch.repeat_interleave(out, s, dim=2 + d)
    return _wrap(out)


def pad(x, pad, mode="constant", value=0.0, data_format="NCHW"):
    """paddle.nn.functional.pad: for 3-D/4-D/5-D input and len(pad) == 2*(ndim-2), pad is
    (left, right[, top, bottom[, front, back]]) -- the same order torch uses [paddle-semantics]."""
    pad = [int(p) for p in pad]
    if len(pad) == 2 * x.dim():
        # "pad every dimension, starting from the first" form
        pairs = [(pad[2 * i], pad[2 * i + 1]) for i in range(x.dim())]
        flat = []
        for a, b in reversed(pairs):
            flat += [a, b]
        return _wrap(TF.pad(x, flat, mode=mode, value=value))
    if mode == "constant":
        return _wrap(TF.pad(x, pad, mode="constant", value=value))
    return _wrap(TF.pad(x, pad, mode=mode))


# Tacotron2-style prenets keep dropout ON at inference (modules/tacotron2/decoder.py:78-81, models/tacotron2.py:61-80),
# so a reproducible run needs the mask injected: when DROPOUT_HOOK is set, every *active* dropout call
# (training=True, p > 0) returns DROPOUT_HOOK(x, p) instead of drawing from torch's generator.  The hook owns the
# upscale_in_train convention (x * keep / (1 - p)) [paddle-semantics: the default mode of paddle.nn.functional.dropout].
DROPOUT_HOOK = None


def dropout(x, p=0.5, training=True, **k):
    if DROPOUT_HOOK is not None and training and p > 0:
        return _wrap(DROPOUT_HOOK(x, p))
    return _wrap(TF.dropout(x, p, training))


def sigmoid(x):
    return _wrap(torch.sigmoid(x))


def relu(x):
    return _wrap(torch.relu(x))


def leaky_relu(x, negative_slope=0.01):
    return _wrap(TF.leaky_relu(x, negative_slope))


def softmax(x, axis=-1):
    return _wrap(torch.softmax(x, dim=axis))


def normalize(x, p=2, axis=1, epsilon=1e-12):
    return _wrap(TF.normalize(x, p=p, dim=axis, eps=epsilon))


def tanh(x):
    return _wrap(torch.tanh(x))
