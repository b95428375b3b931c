"""Oracle: TransformerTTS single-utterance inference (test infrastructure).

Restates, op for op, parakeet/models/transformer_tts/transformer_tts.py
  TransformerTTS.inference                   :511-647  (no teacher forcing, no GST)
  TransformerTTS._integrate_with_spk_embed   :725-755
  StyleEncoder / ReferenceEncoder / StyleTokenLayer  parakeet/modules/style_encoder.py:24-308 (use_gst)
and the modules it calls:
  Encoder.forward                            fastspeech2_transformer/encoder.py:171-192
  EncoderPrenet (tacotron2 Encoder, elayers=0) modules/tacotron2/encoder.py:150-176
  Decoder.forward_one_step                   fastspeech2_transformer/decoder.py:190-227
  DecoderLayer.forward (cache branch)        fastspeech2_transformer/decoder_layer.py:74-158
  MultiHeadedAttention                       fastspeech2_transformer/attention.py:51-156
  PositionwiseFeedForward                    fastspeech2_transformer/positionwise_feed_forward.py:41-44
  DecoderPrenet (tacotron2 Prenet)           modules/tacotron2/decoder.py:62-81
  Postnet                                    modules/tacotron2/decoder.py:127-198
  subsequent_mask                            fastspeech2_transformer/mask.py:18-35

Two properties of the reference loop that this restatement keeps, because they decide the numbers:

* ``Decoder.forward_one_step`` applies ``self.embed`` to the WHOLE prefix ``ys`` at every step, and the decoder
  prenet inside it calls ``F.dropout(x)`` with the defaults p = 0.5, training = True -- dropout stays on at
  inference and ignores ``dprenet_dropout_rate`` (modules/tacotron2/decoder.py:78-81).  So at step s the
  first decoder layer sees s freshly re-dropped rows; only the layer OUTPUTS are cached (decoder.py:213-218), and
  layers >= 1 read the cached rows of the layer below.
* with a cache, a layer computes the last query row only and its mask row is all ones (decoder_layer.py:110-120):
  the self-attention of step s is unmasked over the s rows.

A reproducible run needs the dropout mask injected.  ``drop`` is a callable (step, layer, rows, units) -> keep
array of shape (rows, units); the default is the engine's counter-based dropout stream (oracle/philox_ref.py
``dropout_keep``; include/pk_synth.h), element index ((step*(step-1)/2 + pos) * n_layers + layer) * units + unit
with step counted from 1.  ``drop=None`` switches dropout off (x unchanged: the deterministic variant).
"""
import math

import numpy as np
import torch

from . import philox_ref
from .fastspeech2_ref import conv_ffn, encoder_layer, postnet, scaled_posenc
from .nn_ref import Weights, batch_norm_eval, conv1d, layer_norm, linear, sinusoid_table

DEFAULT_CFG = dict(
    embed_dim=0, eprenet_conv_layers=0, eprenet_conv_filts=0, eprenet_conv_chans=0,
    dprenet_layers=2, dprenet_units=256, adim=512, aheads=8, elayers=6, eunits=1024, dlayers=6, dunits=1024,
    postnet_layers=5, postnet_filts=5, postnet_chans=256, reduction_factor=1, use_scaled_pos_enc=True,
    spk_embed_dim=None, spk_embed_integration_type="add", use_gst=False, gst_tokens=10, gst_heads=4, gst_conv_layers=6,
    gst_conv_chans_list=(32, 32, 64, 64, 128, 128), gst_conv_kernel_size=3, gst_conv_stride=2, gst_gru_layers=1,
    gst_gru_units=128)

PRENET_DROPOUT_P = 0.5   # F.dropout's default; Prenet.forward passes no rate (modules/tacotron2/decoder.py:80)


def stream_dropout(seed, n_layers, units, p=PRENET_DROPOUT_P):
    def drop(step, layer, rows, n_units):
        assert n_units == units
        tri = step * (step - 1) // 2
        pos = np.arange(rows, dtype=np.uint64)[:, None]
        u = np.arange(units, dtype=np.uint64)[None, :]
        idx = ((np.uint64(tri) + pos) * np.uint64(n_layers) + np.uint64(layer)) * np.uint64(units) + u
        return philox_ref.dropout_keep(idx, p, seed)
    return drop


def mha(W, q_in, kv_in, n_head):
    """MultiHeadedAttention.forward attention.py:133-156, mask None (or all ones).  q_in (B,Tq,D), kv_in (B,Tk,D).
    Returns (output, attention weights (B, H, Tq, Tk))."""
    B, Tq, D = q_in.shape
    Tk = kv_in.shape[1]
    dk = D // n_head
    q = linear(q_in, W["linear_q.weight"], W["linear_q.bias"]).reshape(B, Tq, n_head, dk).transpose(1, 2)
    k = linear(kv_in, W["linear_k.weight"], W["linear_k.bias"]).reshape(B, Tk, n_head, dk).transpose(1, 2)
    v = linear(kv_in, W["linear_v.weight"], W["linear_v.bias"]).reshape(B, Tk, n_head, dk).transpose(1, 2)
    attn = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(dk), dim=-1)
    ctx = torch.matmul(attn, v).transpose(1, 2).reshape(B, Tq, D)
    return linear(ctx, W["linear_out.weight"], W["linear_out.bias"]), attn


def posenc(W, x, cfg):
    """ScaledPositionalEncoding.forward embedding.py:111-126 (x + alpha * pe) or, with use_scaled_pos_enc=False,
    PositionalEncoding.forward :64-80 (x * sqrt(d_model) + pe)."""
    if cfg.get("use_scaled_pos_enc", True):
        return scaled_posenc(W, x)
    pe = sinusoid_table(x.shape[1], x.shape[2], x.dtype)
    return x * math.sqrt(x.shape[2]) + pe.unsqueeze(0)


def encoder_input(W, ids, cfg):
    """The encoder's ``embed`` Sequential (transformer_tts.py:258-277, encoder.py:112-121): Embedding(padding_idx=0)
    [+ conv prenet + Linear] followed by ScaledPositionalEncoding."""
    if cfg["eprenet_conv_layers"] > 0:
        P = W.sub("embed.0.0.")
        table = P["embed.weight"].clone()
        table[0] = 0.0                                   # padding_idx=0 [paddle-semantics]
        x = table[ids].transpose(1, 2)                   # tacotron2/encoder.py:150
        for i in range(cfg["eprenet_conv_layers"]):      # Conv1D(no bias) -> BatchNorm1D -> ReLU -> Dropout(eval: off)
            w = P[f"convs.{i}.0.weight"]
            x = conv1d(x, w, None, padding=(w.shape[-1] - 1) // 2)
            x = torch.relu(batch_norm_eval(x, P[f"convs.{i}.1.weight"], P[f"convs.{i}.1.bias"],
                                           P[f"convs.{i}.1._mean"], P[f"convs.{i}.1._variance"]))
        x = linear(x.transpose(1, 2), W["embed.0.1.weight"], W["embed.0.1.bias"])
    else:
        table = W["embed.0.weight"].clone()
        table[0] = 0.0
        x = table[ids]
    return posenc(W.sub("embed.1."), x, cfg)


def encode(W, ids, cfg):
    """Encoder.forward encoder.py:171-192 with masks=None (transformer_tts.py:585)."""
    x = encoder_input(W, ids, cfg)
    for i in range(cfg["elayers"]):
        x = encoder_layer(W.sub(f"encoders.{i}."), x, None, cfg["aheads"], cfg.get("encoder_normalize_before", True),
                          cfg.get("encoder_concat_after", False))
    if not cfg.get("encoder_normalize_before", True):                            # encoder.py:190-191
        return x
    return layer_norm(x, W["after_norm.weight"], W["after_norm.bias"])


def decoder_embed(W, ys, step, cfg, drop):
    """decoder.embed = Sequential(Sequential(DecoderPrenet, Linear), ScaledPositionalEncoding) on the whole prefix
    ys (1, step, odim)."""
    x = ys
    if cfg["dprenet_layers"] == 0:
        # input_layer "linear" (decoder.py:112-118): Linear -> LayerNorm -> Dropout(eval: off) -> ReLU -> pos_enc
        x = linear(x, W["embed.0.weight"], W["embed.0.bias"])
        x = torch.relu(layer_norm(x, W["embed.1.weight"], W["embed.1.bias"]))
        return posenc(W.sub("embed.4."), x, cfg)
    for j in range(cfg["dprenet_layers"]):
        x = torch.relu(linear(x, W[f"embed.0.0.prenet.{j}.0.weight"], W[f"embed.0.0.prenet.{j}.0.bias"]))
        if drop is not None:
            keep = torch.as_tensor(drop(step, j, x.shape[1], x.shape[2]))
            x = torch.where(keep.unsqueeze(0), x / (1.0 - PRENET_DROPOUT_P), torch.zeros_like(x))
    x = linear(x, W["embed.0.1.weight"], W["embed.0.1.bias"])
    return posenc(W.sub("embed.1."), x, cfg)


def integrate_with_spk_embed(W, hs, spembs, kind):
    """TransformerTTS._integrate_with_spk_embed :725-755.  hs (1, T, adim), spembs (1, D)."""
    e = spembs / torch.clamp(torch.linalg.vector_norm(spembs, dim=1, keepdim=True), min=1e-12)   # F.normalize
    if kind == "add":
        return hs + linear(e, W["projection.weight"], W["projection.bias"]).unsqueeze(1)
    if kind == "concat":
        e = e.unsqueeze(1).expand(-1, hs.shape[1], -1)
        return linear(torch.cat([hs, e], dim=-1), W["projection.weight"], W["projection.bias"])
    raise NotImplementedError("support only add or concat.")


def gru_cell(W, x, h):
    """paddle.nn.GRUCell.forward [paddle-semantics, from Paddle's API documentation]: gate order r, z, c."""
    gx = linear(x, W["weight_ih"].t(), W["bias_ih"])
    gh = linear(h, W["weight_hh"].t(), W["bias_hh"])
    xr, xz, xc = torch.chunk(gx, 3, dim=-1)
    hr, hz, hc = torch.chunk(gh, 3, dim=-1)
    r, z = torch.sigmoid(xr + hr), torch.sigmoid(xz + hz)
    c = torch.tanh(xc + r * hc)
    return z * h + (1.0 - z) * c


def style_encoder(W, speech, cfg):
    """StyleEncoder.forward style_encoder.py:93-106 for one reference spectrogram speech (L, odim) -> (1, adim).
    ReferenceEncoder.forward :187-215: Conv2D(no bias) -> BatchNorm2D(eval) -> ReLU stack, transpose / reshape, GRU, last
    hidden state; StyleTokenLayer.forward :266-288: multi-head attention of that one query over tanh(gst_embs), with
    the q / k / v input widths of style_encoder.MultiHeadedAttention :291-308."""
    R = W.sub("ref_enc.")
    k, stride = cfg["gst_conv_kernel_size"], cfg["gst_conv_stride"]
    x = speech.reshape(1, 1, speech.shape[0], speech.shape[1])
    for i in range(cfg["gst_conv_layers"]):
        x = torch.nn.functional.conv2d(x, R[f"convs.{3 * i}.weight"], None, stride=stride, padding=(k - 1) // 2)
        shp = (1, -1, 1, 1)
        bn = R.sub(f"convs.{3 * i + 1}.")
        x = (x - bn["_mean"].view(shp)) / torch.sqrt(bn["_variance"].view(shp) + 1e-5) * bn["weight"].view(shp) + bn["bias"].view(shp)
        x = torch.relu(x)
    hs = x.transpose(1, 2).reshape(1, x.shape[2], -1)                     # (1, L', C * F')
    for l in range(cfg["gst_gru_layers"]):
        C = R.sub(f"gru.{l}.cell.")
        h = torch.zeros(1, C["weight_hh"].shape[1], dtype=hs.dtype)
        seq = []
        for t in range(hs.shape[1]):
            h = gru_cell(C, hs[:, t], h)
            seq.append(h)
        hs = torch.stack(seq, dim=1)
    ref = h                                                                # (1, gru_units): ref_embs[-1]
    S = W.sub("stl.")
    M = S.sub("mha.")
    n_head = cfg["gst_heads"]
    gst = torch.tanh(S["gst_embs"]).unsqueeze(0)                           # (1, tokens, adim / heads)
    q = linear(ref.unsqueeze(1), M["linear_q.weight"], M["linear_q.bias"])
    kk = linear(gst, M["linear_k.weight"], M["linear_k.bias"])
    vv = linear(gst, M["linear_v.weight"], M["linear_v.bias"])
    A = q.shape[-1]
    dk = A // n_head
    qh = q.reshape(1, 1, n_head, dk).transpose(1, 2)
    kh = kk.reshape(1, -1, n_head, dk).transpose(1, 2)
    vh = vv.reshape(1, -1, n_head, dk).transpose(1, 2)
    attn = torch.softmax(torch.matmul(qh, kh.transpose(-1, -2)) / math.sqrt(dk), dim=-1)
    ctx = torch.matmul(attn, vh).transpose(1, 2).reshape(1, 1, A)
    return linear(ctx, M["linear_out.weight"], M["linear_out.bias"]).squeeze(1)


def decoder_layer_step(W, tgt, memory, cache, n_head, normalize_before=True, concat_after=False):
    """DecoderLayer.forward decoder_layer.py:74-158.
    tgt (1, s, D); cache (1, s-1, D) or None.  Returns (x (1, s, D), src attention weights (H, T) of the last row)."""
    residual = tgt
    t = layer_norm(tgt, W["norm1.weight"], W["norm1.bias"]) if normalize_before else tgt
    if cache is None:
        tq = t
    else:
        tq = t[:, -1:, :]
        residual = residual[:, -1:, :]
    a = mha(W.sub("self_attn."), tq, t, n_head)[0]
    if concat_after:                                                               # :126-129
        x = residual + linear(torch.cat([tq, a], dim=-1), W["concat_linear1.weight"], W["concat_linear1.bias"])
    else:
        x = residual + a
    if not normalize_before:
        x = layer_norm(x, W["norm1.weight"], W["norm1.bias"])
    residual = x
    h = layer_norm(x, W["norm2.weight"], W["norm2.bias"]) if normalize_before else x
    a, attn = mha(W.sub("src_attn."), h, memory, n_head)
    if concat_after:                                                               # :139-142
        x = residual + linear(torch.cat([h, a], dim=-1), W["concat_linear2.weight"], W["concat_linear2.bias"])
    else:
        x = residual + a
    if not normalize_before:
        x = layer_norm(x, W["norm2.weight"], W["norm2.bias"])
    residual = x
    h = layer_norm(x, W["norm3.weight"], W["norm3.bias"]) if normalize_before else x
    x = residual + conv_ffn(W.sub("feed_forward."), h)
    if not normalize_before:
        x = layer_norm(x, W["norm3.weight"], W["norm3.bias"])
    if cache is not None:
        x = torch.cat([cache, x], dim=1)
    return x, attn[0, :, -1]


def inference(state, ids, cfg=None, threshold=0.5, minlenratio=0.0, maxlenratio=10.0, seed=0, drop="stream",
              dtype=torch.float32, return_parts=False, spembs=None, speech=None):
    """TransformerTTS.inference transformer_tts.py:511-647.  ids (T,) int64 without <eos>.
    Returns (outs (L, odim), probs (L,), att_ws (dlayers, aheads, L, T+1))."""
    cfg = dict(DEFAULT_CFG, **(cfg or {}))
    r = cfg.get("reduction_factor", 1)
    W = Weights(state, dtype)
    idim = (state["encoder.embed.0.weight"] if "encoder.embed.0.weight" in state
            else state["encoder.embed.0.0.embed.weight"]).shape[0]
    odim = state["feat_out.weight"].shape[1] // r
    x = np.pad(np.asarray(ids), (0, 1), "constant", constant_values=idim - 1)      # :563-565 add <eos>
    xs = torch.as_tensor(x).to(torch.int64).unsqueeze(0)
    hs = encode(W.sub("encoder."), xs, cfg)                                        # :584-585
    enc_out = hs
    if cfg.get("use_gst"):                                                         # :586-588
        style = style_encoder(W.sub("gst."), torch.as_tensor(np.asarray(speech)).to(dtype), cfg)
        hs = hs + style.unsqueeze(1)
    if cfg.get("spk_embed_dim"):                                                   # :591-593
        e = torch.as_tensor(np.asarray(spembs)).to(dtype).reshape(1, -1)
        hs = integrate_with_spk_embed(W, hs, e, cfg["spk_embed_integration_type"])
    maxlen = int(hs.shape[1] * maxlenratio / r)                                    # :597-598
    minlen = int(hs.shape[1] * minlenratio / r)
    if drop == "stream":
        drop = stream_dropout(seed, cfg["dprenet_layers"], cfg["dprenet_units"])
    D = W.sub("decoder.")
    idx = 0
    ys = torch.zeros(1, 1, odim, dtype=dtype)                                      # :601-602
    outs, probs, att_ws = [], [], []
    cache = None
    parts = {}
    while True:
        idx += 1
        xdec = decoder_embed(D, ys, idx, cfg, drop)                                # decoder.py:210
        if cache is None:
            cache = [None] * cfg["dlayers"]
        new_cache, att_step = [], []
        for l in range(cfg["dlayers"]):                                            # decoder.py:214-218
            xdec, a = decoder_layer_step(D.sub(f"decoders.{l}."), xdec, hs, cache[l], cfg["aheads"],
                                         cfg.get("decoder_normalize_before", True), cfg.get("decoder_concat_after", False))
            new_cache.append(xdec)
            att_step.append(a)
        cache = new_cache
        z = xdec[:, -1]
        if cfg.get("decoder_normalize_before", True):
            z = layer_norm(z, D["after_norm.weight"], D["after_norm.bias"])        # decoder.py:220-221
        out = linear(z, W["feat_out.weight"], W["feat_out.bias"]).reshape(r, odim)  # :613-615
        outs.append(out)
        probs.append(torch.sigmoid(linear(z, W["prob_out.weight"], W["prob_out.bias"]))[0])   # :616  (r,)
        ys = torch.cat([ys, out[-1].reshape(1, 1, odim)], dim=1)                   # :619-621: the LAST of the r frames
        att_ws.append(torch.stack(att_step, dim=0))                                # (dlayers, H, T)
        if int((probs[-1] >= threshold).sum()) > 0 or idx >= maxlen:               # :638-639
            if idx < minlen:                                                       # :641-642
                continue
            before = torch.cat(outs, dim=0).unsqueeze(0).transpose(1, 2)           # :644-645
            after = before
            if cfg["postnet_layers"] > 0:
                after = before + postnet(W.sub("postnet."), before, cfg["postnet_layers"])   # :646-648
            mel = after.transpose(1, 2).squeeze(0)
            probs_t = torch.cat(probs, dim=0)
            break
    att = torch.stack(att_ws, dim=2)                                               # (dlayers, H, L, T)
    if return_parts:
        parts.update(hs=hs[0], enc=enc_out[0], before=before[0].transpose(0, 1), zs=cache[-1][0])
        return mel, probs_t, att, parts
    return mel, probs_t, att
