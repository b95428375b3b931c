"""Oracle: FastSpeech2 single-utterance inference (test infrastructure).

Restates, op for op, the inference branch of
parakeet/models/fastspeech2/fastspeech2.py:
  FastSpeech2.inference          :468-558  (is_inference=True branch)
  FastSpeech2._forward           :377-466
  FastSpeech2Inference.forward   :668-671
and the modules it calls:
  Encoder.forward                fastspeech2_transformer/encoder.py:171-192
  EncoderLayer.forward           fastspeech2_transformer/encoder_layer.py:64-115
  MultiHeadedAttention           fastspeech2_transformer/attention.py:51-156
  ScaledPositionalEncoding       fastspeech2_transformer/embedding.py:46-62,111-126
  MultiLayeredConv1d.forward     fastspeech2_transformer/multi_layer_conv.py:62-77
  DurationPredictor._forward     fastspeech2_predictor/duration_predictor.py:85-103
  VariancePredictor.forward      fastspeech2_predictor/variance_predictor.py:77-104
  LayerNorm(dim=1)               modules/layer_norm.py:34-63
  LengthRegulator.forward/expand fastspeech2_predictor/length_regulator.py:46-89
  Postnet.forward                modules/tacotron2/decoder.py:127-198
  masked_fill                    modules/masked_fill.py:28-37
  make_pad_mask/non_pad_mask     modules/nets_utils.py:54-125

Configuration = examples/fastspeech2/ljspeech/conf/default.yaml:33-75
(transformer encoder/decoder, conv1d position-wise layers, pre-norm, scaled
positional encoding, no speaker / tone embedding).
"""
import math

import numpy as np
import torch

from .nn_ref import (Weights, batch_norm_eval, conv1d, layer_norm, linear,
                     make_non_pad_mask, make_pad_mask, masked_fill,
                     round_half_away, sinusoid_table)

DEFAULT_CFG = dict(
    adim=384, aheads=2, elayers=4, eunits=1536, dlayers=4, dunits=1536,
    positionwise_conv_kernel_size=3,
    duration_predictor_layers=2, duration_predictor_chans=256, duration_predictor_kernel_size=3,
    pitch_predictor_layers=5, pitch_predictor_chans=256, pitch_predictor_kernel_size=5,
    energy_predictor_layers=2, energy_predictor_chans=256, energy_predictor_kernel_size=3,
    pitch_embed_kernel_size=1, energy_embed_kernel_size=1,
    postnet_layers=5, postnet_chans=256, postnet_filts=5,
    spk_embed_dim=None, spk_embed_integration_type="add", tone_embed_dim=None)


def scaled_posenc(W, x):
    """ScaledPositionalEncoding.forward embedding.py:111-126: x + alpha * pe
    (no sqrt(d) scaling, unlike PositionalEncoding.forward :78)."""
    pe = sinusoid_table(x.shape[1], x.shape[2], x.dtype)
    return x + W["alpha"] * pe.unsqueeze(0)


def attention(W, x, mask, n_head):
    """MultiHeadedAttention.forward attention.py:133-156 with query=key=value=x.
    mask: (B,1,T) bool non-pad mask or None."""
    B, T, D = x.shape
    dk = D // n_head
    q = linear(x, W["linear_q.weight"], W["linear_q.bias"]).reshape(B, T, n_head, dk).transpose(1, 2)
    k = linear(x, W["linear_k.weight"], W["linear_k.bias"]).reshape(B, T, n_head, dk).transpose(1, 2)
    v = linear(x, W["linear_v.weight"], W["linear_v.bias"]).reshape(B, T, n_head, dk).transpose(1, 2)
    scores = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(dk)
    if mask is not None:
        m = ~mask.unsqueeze(1)  # :110-111
        min_value = float(np.finfo(np.float32).min)  # :112-114 (scores are float32 in the reference)
        scores = masked_fill(scores, m, min_value)
        attn = torch.softmax(scores, dim=-1)
        attn = masked_fill(attn, m, 0.0)
    else:
        attn = torch.softmax(scores, dim=-1)
    ctx = torch.matmul(attn, v).transpose(1, 2).reshape(B, T, D)
    return linear(ctx, W["linear_out.weight"], W["linear_out.bias"])


def conv_ffn(W, x):
    """The position-wise layer of an FFT block, by weight rank (Linear weights are [in, out], Conv1D weights
    [Cout, Cin, k]): MultiLayeredConv1d multi_layer_conv.py:62-77 ("conv1d"), Conv1dLinear :112-127
    ("conv1d-linear"), PositionwiseFeedForward positionwise_feed_forward.py:41-44 ("linear")."""
    if W["w_1.weight"].dim() == 3:
        k = W["w_1.weight"].shape[-1]
        h = torch.relu(conv1d(x.transpose(1, 2), W["w_1.weight"], W["w_1.bias"], padding=(k - 1) // 2)).transpose(1, 2)
    else:
        h = torch.relu(linear(x, W["w_1.weight"], W["w_1.bias"]))
    if W["w_2.weight"].dim() == 3:
        k2 = W["w_2.weight"].shape[-1]
        return conv1d(h.transpose(1, 2), W["w_2.weight"], W["w_2.bias"], padding=(k2 - 1) // 2).transpose(1, 2)
    return linear(h, W["w_2.weight"], W["w_2.bias"])


def encoder_layer(W, x, mask, n_head, normalize_before=True, concat_after=False):
    """EncoderLayer.forward encoder_layer.py:64-115 (no cache, dropout off)."""
    residual = x
    if normalize_before:
        x = layer_norm(x, W["norm1.weight"], W["norm1.bias"])
    att = attention(W.sub("self_attn."), x, mask, n_head)
    if concat_after:                                                          # :103-106
        x = residual + linear(torch.cat([x, att], dim=-1), W["concat_linear.weight"], W["concat_linear.bias"])
    else:
        x = residual + att
    if not normalize_before:
        x = layer_norm(x, W["norm1.weight"], W["norm1.bias"])
    residual = x
    if normalize_before:
        x = layer_norm(x, W["norm2.weight"], W["norm2.bias"])
    x = residual + conv_ffn(W.sub("feed_forward."), x)
    if not normalize_before:
        x = layer_norm(x, W["norm2.weight"], W["norm2.bias"])
    return x


def encoder(W, xs, mask, n_layers, n_head, embed_ids, normalize_before=True, concat_after=False):
    """Encoder.forward encoder.py:171-192.  embed_ids=True: input_layer is
    nn.Embedding(padding_idx=0) followed by ScaledPositionalEncoding (embed.0 /
    embed.1); False: positional encoding only (embed.0) -- fastspeech2.py:250-266."""
    if embed_ids:
        table = W["embed.0.weight"].clone()
        table[0] = 0.0  # padding_idx=0 -> zero row [paddle-semantics]
        x = table[xs]
        x = scaled_posenc(W.sub("embed.1."), x)
    else:
        x = scaled_posenc(W.sub("embed.0."), xs)
    for i in range(n_layers):
        x = encoder_layer(W.sub(f"encoders.{i}."), x, mask, n_head, normalize_before, concat_after)
    if not normalize_before:                                                  # encoder.py:190-191
        return x
    return layer_norm(x, W["after_norm.weight"], W["after_norm.bias"])


def conv_relu_ln_stack(W, xs, n_layers):
    """The conv stacks shared by DurationPredictor (:64-80) and
    VariancePredictor (:61-75): [Conv1D -> ReLU -> LayerNorm(dim=1) -> Dropout]*n
    on (B, C, T)."""
    for j in range(n_layers):
        w = W[f"conv.{j}.0.weight"]
        xs = torch.relu(conv1d(xs, w, W[f"conv.{j}.0.bias"], padding=(w.shape[-1] - 1) // 2))
        # LayerNorm(dim=1): transpose -> LN over channels -> transpose (layer_norm.py:50-63)
        xs = layer_norm(xs.transpose(1, 2), W[f"conv.{j}.2.weight"], W[f"conv.{j}.2.bias"]).transpose(1, 2)
    return xs


def variance_predictor(W, hs, pad_mask, n_layers):
    """VariancePredictor.forward variance_predictor.py:77-104 -> (B,T,1)."""
    xs = conv_relu_ln_stack(W, hs.transpose(1, 2), n_layers)
    xs = linear(xs.transpose(1, 2), W["linear.weight"], W["linear.bias"])
    return masked_fill(xs, pad_mask.unsqueeze(-1), 0.0)


def duration_inference(W, hs, pad_mask, n_layers, offset=1.0):
    """DurationPredictor.inference duration_predictor.py:122-137 -> (B,T)
    integer-valued float: clip(round(exp(x) - offset), min=0) :98."""
    xs = conv_relu_ln_stack(W, hs.transpose(1, 2), n_layers)
    xs = linear(xs.transpose(1, 2), W["linear.weight"], W["linear.bias"]).squeeze(-1)
    xs = torch.clamp(round_half_away(torch.exp(xs) - offset), min=0)
    return masked_fill(xs, pad_mask, 0.0)


def length_regulate(hs, ds, alpha=1.0):
    """LengthRegulator.forward/expand length_regulator.py:46-89.  hs (B,T,C),
    ds (B,T) integer-valued.  The reference builds a dense 0/1 matrix in
    float64 numpy and multiplies; the result is a pure row repeat, built here
    the same way (matmul) so rows beyond an utterance's length are zero."""
    if alpha != 1.0:
        assert alpha > 0
        ds = round_half_away(ds.to(torch.float32) * alpha)
    ds = ds.to(torch.int64).numpy()
    B, T = ds.shape
    slens = ds.sum(-1)
    t_dec = int(slens.max())
    M = np.zeros([B, t_dec, T])
    for i in range(B):
        k = 0
        for j in range(T):
            d = int(ds[i, j])
            if d >= 1:
                M[i, k:k + d, j] = 1
            k += d
    return torch.matmul(torch.as_tensor(M).to(hs.dtype), hs)


def postnet(W, xs, n_layers):
    """Postnet.forward tacotron2/decoder.py:182-198 on (B, odim, T): Conv1D(no
    bias) -> BatchNorm1D(eval) -> Tanh for all but the last layer, the last
    one without Tanh (:127-169)."""
    for j in range(n_layers):
        w = W[f"postnet.{j}.0.weight"]
        xs = conv1d(xs, w, None, padding=(w.shape[-1] - 1) // 2)
        xs = batch_norm_eval(xs, W[f"postnet.{j}.1.weight"], W[f"postnet.{j}.1.bias"],
                             W[f"postnet.{j}.1._mean"], W[f"postnet.{j}.1._variance"])
        if j != n_layers - 1:
            xs = torch.tanh(xs)
    return xs


def integrate_spk_embed(W, hs, spembs, integration_type):
    """FastSpeech2._integrate_with_spk_embed fastspeech2.py:560-586.  hs (B,T,adim), spembs (B,D).
    F.normalize: x / max(||x||_2, 1e-12) along axis 1."""
    n = spembs / spembs.norm(dim=1, keepdim=True).clamp_min(1e-12)
    if integration_type == "add":
        return hs + linear(n, W["spk_projection.weight"], W["spk_projection.bias"]).unsqueeze(1)
    if integration_type == "concat":
        n = n.unsqueeze(1).expand(-1, hs.shape[1], -1)
        return linear(torch.cat([hs, n], dim=-1), W["spk_projection.weight"], W["spk_projection.bias"])
    raise NotImplementedError("support only add or concat.")   # :584


def inference(state, ids, cfg=None, alpha=1.0, dtype=torch.float32, return_parts=False, spk_id=None,
              spembs=None, tone_id=None):
    """FastSpeech2.inference fastspeech2.py:468-558 for one utterance.
    ids: (T,) int64 -> normalised mel (L, odim).  spk_id (int) / spembs (D,): speaker conditioning of
    the multi-speaker recipes (:396-402; spembs wins when both are given)."""
    cfg = dict(DEFAULT_CFG, **(cfg or {}))
    W = Weights(state, dtype)
    x = torch.as_tensor(np.asarray(ids)).to(torch.int64)
    ilens = [int(x.shape[0])]                       # :519-521
    xs = x.unsqueeze(0)                             # :522
    x_masks = make_non_pad_mask(ilens).unsqueeze(-2)  # _source_mask :618-641
    hs = encoder(W.sub("encoder."), xs, x_masks, cfg["elayers"], cfg["aheads"], True,
                 cfg.get("encoder_normalize_before", True), cfg.get("encoder_concat_after", False))  # :393
    if cfg.get("spk_embed_dim") is not None:        # :396-402
        emb = None
        if spembs is not None:
            emb = torch.as_tensor(np.asarray(spembs)).to(dtype).reshape(1, -1)
        elif spk_id is not None:
            emb = W["spk_embedding_table.weight"][int(spk_id)].reshape(1, -1)
            if int(spk_id) == 0:                    # nn.Embedding(padding_idx=0) returns zeros [paddle-semantics]
                emb = torch.zeros_like(emb)
        if emb is not None:
            hs = integrate_spk_embed(W, hs, emb, cfg.get("spk_embed_integration_type", "add"))
    if cfg.get("tone_embed_dim") is not None and tone_id is not None:   # :404-408
        # inference forwards the (T,) ids un-batched (:546,556): tone_embs is (T, Dt), F.normalize's default
        # axis 1 is then the feature axis, and hs (1,T,adim) + (T,adim) broadcasts ("add", :598-601)
        tid = torch.as_tensor(np.asarray(tone_id)).to(torch.int64)
        te = W["tone_embedding_table.weight"][tid]
        te = torch.where((tid == 0).unsqueeze(-1), torch.zeros_like(te), te)   # padding_idx 0 [paddle-semantics]
        te = te / te.norm(dim=1, keepdim=True).clamp_min(1e-12)
        hs = hs + linear(te, W["tone_projection.weight"], W["tone_projection.bias"])
    d_masks = make_pad_mask(ilens)                  # :410
    p_outs = variance_predictor(W.sub("pitch_predictor."), hs, d_masks, cfg["pitch_predictor_layers"])
    e_outs = variance_predictor(W.sub("energy_predictor."), hs, d_masks, cfg["energy_predictor_layers"])
    d_outs = duration_inference(W.sub("duration_predictor."), hs, d_masks,
                                cfg["duration_predictor_layers"])        # :423
    kp = W["pitch_embed.0.weight"].shape[-1]
    ke = W["energy_embed.0.weight"].shape[-1]
    p_embs = conv1d(p_outs.transpose(1, 2), W["pitch_embed.0.weight"], W["pitch_embed.0.bias"],
                    padding=(kp - 1) // 2).transpose(1, 2)               # :426-427
    e_embs = conv1d(e_outs.transpose(1, 2), W["energy_embed.0.weight"], W["energy_embed.0.bias"],
                    padding=(ke - 1) // 2).transpose(1, 2)               # :428-429
    hs2 = hs + e_embs + p_embs                      # :430
    hs_up = length_regulate(hs2, d_outs, alpha)     # :432
    zs = encoder(W.sub("decoder."), hs_up, None, cfg["dlayers"], cfg["aheads"], False,
                 cfg.get("decoder_normalize_before", True), cfg.get("decoder_concat_after", False))  # :455 (h_masks=None)
    before = linear(zs, W["feat_out.weight"], W["feat_out.bias"])        # :457
    odim = before.shape[-1] // cfg.get("reduction_factor", 1)
    before = before.reshape(before.shape[0], -1, odim)                   # (B, L * r, odim)
    after = before                                                       # :460-461 (postnet is None)
    if cfg["postnet_layers"] > 0:
        after = before + postnet(W.sub("postnet."), before.transpose(1, 2),
                                 cfg["postnet_layers"]).transpose(1, 2)  # :463-464
    if return_parts:
        return after[0], dict(hs=hs[0], p=p_outs[0, :, 0], e=e_outs[0, :, 0], d=d_outs[0],
                              hs_up=hs_up[0], zs=zs[0], before=before[0])
    return after[0]


def fastspeech2_inference(state, mu, sigma, ids, cfg=None, alpha=1.0, dtype=torch.float32):
    """FastSpeech2Inference.forward fastspeech2.py:668-671: inference then
    ZScore.inverse (normalizer.py:30-33): x * sigma + mu."""
    mel = inference(state, ids, cfg, alpha, dtype)
    return mel * torch.as_tensor(sigma).to(dtype) + torch.as_tensor(mu).to(dtype)
