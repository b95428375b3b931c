"""Oracle: ConditionalWaveFlow.infer (test infrastructure).

Follows parakeet/models/waveflow.py:
  fold                              :32-51
  UpsampleNet.forward(trim=True)    :103-132
  ResidualBlock.add_input           :248-294   (incremental causal conv with a row buffer)
  ResidualNet.add_input             :368-392
  Flow._predict_row_parameters      :496-501, _inverse_transform_row :503-505,
  Flow.inverse                      :515-556
  WaveFlow._create_perm             :602-615, _trim :617-625, inverse :674-711
  ConditionalWaveFlow.infer         :785-805   (z passed in instead of paddle.randn)
and parakeet/modules/geometry.py shuffle_dim :18-50 (gather along an axis).

State-dict keys (weight-norm pairs are folded):
  encoder.{i}.{weight [1,1,3,2f], bias [1]}                       Conv2DTranspose
  decoder.{f}.input_proj.{weight [C,1,1,1], bias [C]}
  decoder.{f}.resnet.{l}.conv.{weight [2C,C,kh,kw], bias [2C]}
  decoder.{f}.resnet.{l}.condition_proj.{weight [2C,n_mels,1,1], bias [2C]}
  decoder.{f}.resnet.{l}.out_proj.{weight [2C,C,1,1], bias [2C]}
  decoder.{f}.output_proj.{weight [2,C,1,1], bias [2]}
"""
import torch
import torch.nn.functional as F

from .nn_ref import Weights, fold_weight_norm

DEFAULT_CFG = dict(upsample_factors=[16, 16], n_flows=8, n_layers=8, n_group=16, channels=128, n_mels=80,
                   kernel_size=[3, 3])

DILATIONS_H = {8: [1] * 8, 16: [1] * 8, 32: [1, 2, 4, 1, 2, 4, 1, 2], 64: [1, 2, 4, 8, 16, 1, 2, 4],
               128: [1, 2, 4, 8, 16, 32, 64, 1]}   # Flow.dilations_dict :419-425


def create_perms(n_group, n_flows):
    """WaveFlow._create_perm :602-615."""
    idx = list(range(n_group))
    half = n_group // 2
    perms = []
    for i in range(n_flows):
        if i < n_flows // 2:
            perms.append(idx[::-1])
        else:
            perms.append(list(reversed(idx[:half])) + list(reversed(idx[half:])))
    return perms


def upsample(W, mel, factors, trim=True):
    """UpsampleNet.forward :103-132: per layer Conv2DTranspose(1,1,(3,2f),stride (1,f),padding (1,f//2)),
    trim the last (2f - f) columns, leaky_relu(0.4)."""
    x = mel.unsqueeze(1)
    for i, f in enumerate(factors):
        x = F.conv_transpose2d(x, W[f"{i}.weight"], W[f"{i}.bias"], stride=(1, f), padding=(1, f // 2))
        if trim:
            x = x[:, :, :, :-(2 * f - f)]
        x = F.leaky_relu(x, 0.4)
    return x.squeeze(1)


def flow_inverse(W, z, cond, n_layers, dil_h):
    """Flow.inverse :515-556 with ResidualNet/ResidualBlock.add_input :248-294,368-392.
    z (B,1,H,Wd), cond (B,Cm,H,Wd) -> x (B,1,H,Wd)."""
    B, _, H, Wd = z.shape
    x = torch.zeros_like(z)
    x[:, :, :1, :] = z[:, :, :1, :]
    bufs = [None] * n_layers
    for i in range(1, H):
        x_row = x[:, :, i - 1:i, :]
        z_row = z[:, :, i:i + 1, :]
        c_row = cond[:, :, i:i + 1, :]
        h = F.conv2d(x_row, W["input_proj.weight"], W["input_proj.bias"])
        skips = []
        for l in range(n_layers):
            wl = W.sub(f"resnet.{l}.")
            cw = wl["conv.weight"]
            kh, kw = cw.shape[2], cw.shape[3]
            dh, dw = dil_h[l], 2 ** l
            rh, rw = 1 + (kh - 1) * dh, 1 + (kw - 1) * dw
            if bufs[l] is None:                      # _init_buffer :287-290
                bufs[l] = torch.zeros(B, h.shape[1], rh, Wd, dtype=h.dtype)
            bufs[l] = torch.cat([bufs[l][:, :, 1:, :], h], dim=2)   # _update_buffer :292-294
            xin = F.pad(bufs[l], (rw // 2, (rw - 1) // 2, 0, 0))    # padding=[0,0,rw//2,(rw-1)//2] :271-276
            y = F.conv2d(xin, cw, wl["conv.bias"], dilation=(dh, dw))
            y = y + F.conv2d(c_row, wl["condition_proj.weight"], wl["condition_proj.bias"])
            content, gate = torch.chunk(y, 2, dim=1)
            y = torch.tanh(content) * torch.sigmoid(gate)
            y = F.conv2d(y, wl["out_proj.weight"], wl["out_proj.bias"])
            res, skip = torch.chunk(y, 2, dim=1)
            h = h + res
            skips.append(skip)
        out = torch.stack(skips, 0).sum(0)           # ResidualNet.add_input :390-391
        params = F.conv2d(out, W["output_proj.weight"], W["output_proj.bias"])
        logs, b = torch.chunk(params, 2, dim=1)
        x[:, :, i:i + 1, :] = (z_row - b) * torch.exp(-logs)   # _inverse_transform_row :503-505
    return x


def waveflow_inverse(W, z, cond, cfg):
    """WaveFlow.inverse :674-711.  z (B,T), cond (B,Cm,T) -> x (B,T')."""
    ng = cfg["n_group"]
    pruned = z.shape[-1] // ng * ng                  # _trim :617-625
    z = z[:, :pruned]
    cond = cond[:, :, :pruned]
    B = z.shape[0]
    z = z.reshape(B, pruned // ng, ng).transpose(1, 2).unsqueeze(1)             # (B,1,H,Wd)
    cond = cond.reshape(B, cond.shape[1], pruned // ng, ng).transpose(2, 3)     # (B,Cm,H,Wd)
    perms = create_perms(ng, cfg["n_flows"])
    dil_h = DILATIONS_H[ng]
    for i in reversed(range(cfg["n_flows"])):
        p = torch.tensor(perms[i])
        z = torch.index_select(z, 2, p)              # geo.shuffle_dim(z, 2, perm) :704
        cond = torch.index_select(cond, 2, p)        # :705
        z = flow_inverse(W.sub(f"{i}."), z, cond, cfg["n_layers"], dil_h)
    x = z.squeeze(1)
    return x.transpose(1, 2).reshape(B, -1)


def infer(state, mel, z, cfg=None, dtype=torch.float32):
    """ConditionalWaveFlow.infer :785-805 with z given.  mel (B,Cm,T_mel); z (B, T_cond) where
    T_cond is the trimmed upsampled length.  Returns (B, T)."""
    cfg = dict(DEFAULT_CFG, **(cfg or {}))
    W = Weights(fold_weight_norm(state), dtype)
    cond = upsample(W.sub("encoder."), mel.to(dtype), cfg["upsample_factors"], trim=True)
    assert z.shape[-1] == cond.shape[-1], (z.shape, cond.shape)
    return waveflow_inverse(W.sub("decoder."), z.to(dtype), cond, cfg)


def cond_length(t_mel, factors):
    """Length of the trimmed upsampled condition for t_mel frames: each layer maps T -> f*T - f... precisely
    (T-1)*f - 2*(f//2) + 2f - f = f*T - f for even f."""
    t = t_mel
    for f in factors:
        t = (t - 1) * f - 2 * (f // 2) + 2 * f - (2 * f - f)
    return t
