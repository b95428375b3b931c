"""Oracle: Parallel WaveGAN generator forward / inference (test infrastructure).

Follows parakeet/models/parallel_wavegan/parallel_wavegan.py op for op:
  Stretch2D.forward        :48-63
  UpsampleNet.forward      :119-138
  ConvInUpsampleNet.forward:201-216
  ResidualBlock.forward    :284-315
  PWGGenerator.forward     :445-472
  PWGGenerator.inference   :498-520   (noise is an explicit argument here)
  PWGInference.forward     :772-775

State-dict keys are the reference's: first_conv.*,
upsample_net.conv_in.weight, upsample_net.upsample.up_layers.{1,3,5,7}.weight,
conv_layers.{i}.{conv,conv1x1_aux,conv1x1_out,conv1x1_skip}.*,
last_conv_layers.{1,3}.*; weight-norm pairs (weight_g / weight_v) are folded.
Only the non-causal, no-activation upsample configuration of the LJSpeech
recipe is restated (use_causal_conv=False, nonlinear_activation=None).
"""
import math

import torch
import torch.nn.functional as F

from .nn_ref import Weights, conv1d, fold_weight_norm

DEFAULT_CFG = dict(
    in_channels=1, out_channels=1, kernel_size=3, layers=30, stacks=3,
    residual_channels=64, gate_channels=128, skip_channels=64, aux_channels=80,
    aux_context_window=2, upsample_scales=[4, 4, 4, 4])


def upsample_net(W, c, scales):
    """UpsampleNet.forward :119-138.  c: (N, F, T) -> (N, F, T*prod(scales))."""
    c = c.unsqueeze(1)  # (N,1,F,T)
    for i, s in enumerate(scales):
        # Stretch2D :61-62 -- F.interpolate(nearest): out[t] = in[t // s]
        c = torch.repeat_interleave(c, s, dim=3)
        w = W[f"upsample.up_layers.{2 * i + 1}.weight"]  # (1,1,1,2s+1), no bias :103-104
        c = F.conv2d(c, w, None, padding=(0, s))
    return c.squeeze(1)


def conv_in_upsample(W, c, cfg):
    """ConvInUpsampleNet.forward :201-216 (non-causal): Conv1D(k=2*ctx+1, no
    padding, no bias) then UpsampleNet."""
    c_ = conv1d(c, W["conv_in.weight"])
    return upsample_net(W, c_, cfg["upsample_scales"])


def residual_block(W, x, c, dilation, kernel_size):
    """ResidualBlock.forward :284-315."""
    x_in = x
    pad = (kernel_size - 1) // 2 * dilation
    x = conv1d(x, W["conv.weight"], W["conv.bias"], padding=pad, dilation=dilation)
    x = x + conv1d(c, W["conv1x1_aux.weight"])
    a, b = torch.chunk(x, 2, dim=1)
    x = torch.tanh(a) * torch.sigmoid(b)
    skip = conv1d(x, W["conv1x1_skip.weight"], W["conv1x1_skip.bias"])
    res = (conv1d(x, W["conv1x1_out.weight"], W["conv1x1_out.bias"]) + x_in) * math.sqrt(0.5)
    return res, skip


def generator_forward(state, x, c, cfg=None, dtype=torch.float32, return_parts=False):
    """PWGGenerator.forward :445-472.  x: (N,1,T) noise, c: (N,80,T'+2*ctx)."""
    cfg = dict(DEFAULT_CFG, **(cfg or {}))
    W = Weights(fold_weight_norm(state), dtype)
    x = x.to(dtype)
    c = c.to(dtype)
    c_up = conv_in_upsample(W.sub("upsample_net."), c, cfg)
    assert c_up.shape[-1] == x.shape[-1]
    x = conv1d(x, W["first_conv.weight"], W["first_conv.bias"])
    layers_per_stack = cfg["layers"] // cfg["stacks"]
    skips = 0
    for i in range(cfg["layers"]):
        d = 2 ** (i % layers_per_stack)
        x, s = residual_block(W.sub(f"conv_layers.{i}."), x, c_up, d, cfg["kernel_size"])
        skips = skips + s
    skips = skips * math.sqrt(1.0 / cfg["layers"])
    h = torch.relu(skips)
    h = conv1d(h, W["last_conv_layers.1.weight"], W["last_conv_layers.1.bias"])
    h = torch.relu(h)
    out = conv1d(h, W["last_conv_layers.3.weight"], W["last_conv_layers.3.bias"])
    if return_parts:
        return out, dict(c_up=c_up, x_last=x, skips=skips)
    return out


def generator_inference(state, c, noise, cfg=None, dtype=torch.float32):
    """PWGGenerator.inference :498-520 with the noise passed in.
    c: (T', 80) normalised log-mel; noise: (T'*hop,) or (1,1,T'*hop).
    Returns (T'*hop, 1)."""
    cfg = dict(DEFAULT_CFG, **(cfg or {}))
    c = c.to(dtype)
    hop = 1
    for s in cfg["upsample_scales"]:
        hop *= s
    x = noise.to(dtype).reshape(1, cfg["in_channels"], c.shape[0] * hop)
    cc = c.transpose(0, 1).unsqueeze(0)
    cc = F.pad(cc, (cfg["aux_context_window"],) * 2, mode="replicate")  # nn.Pad1D :518
    out = generator_forward(state, x, cc, cfg, dtype)
    return out.squeeze(0).transpose(0, 1)


def pwg_inference(state, mu, sigma, logmel, noise, cfg=None, dtype=torch.float32):
    """PWGInference.forward :772-775 -- ZScore.forward then inference."""
    mu = torch.as_tensor(mu).to(dtype)
    sigma = torch.as_tensor(sigma).to(dtype)
    normalized = (logmel.to(dtype) - mu) / sigma  # normalizer.py:25-28
    return generator_inference(state, normalized, noise, cfg, dtype)
