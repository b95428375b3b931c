"""Oracle: Tacotron2 single-utterance inference (test infrastructure).

Restates, op for op, parakeet/models/tacotron2.py
  Tacotron2.infer                 :781-840
  Tacotron2Encoder.forward        :216-241   (Conv1dBatchNorm -> ReLU -> dropout(eval: off), then a bidirectional LSTM)
  Tacotron2Decoder.infer          :474-541   (stop rules :515-528)
  Tacotron2Decoder._decode        :378-417
  Tacotron2Decoder._initialize_decoder_states :352-376
  DecoderPreNet.forward           :61-79     (F.dropout(..., training=True): dropout stays on at inference)
  DecoderPostNet.forward          :147-171
and parakeet/modules/attention.py LocationSensitiveAttention.forward :300-348, parakeet/modules/conv.py
Conv1dBatchNorm :186-260.

LSTM semantics [paddle-semantics, from Paddle's API documentation]: gates = x W_ih^T + b_ih + h W_hh^T + b_hh, split
in the order i, f, g, o; c' = sigmoid(f) c + sigmoid(i) tanh(g); h' = sigmoid(o) tanh(c').  The bidirectional LSTM
concatenates (forward, backward) outputs; zero initial states.

The prenet dropout mask comes from the engine's counter-based stream (oracle/philox_ref.py; include/pk_synth.h):
element index ((step * 2 + layer) * d_prenet + unit), step counted from 0, p = p_prenet_dropout.  ``drop=None``
switches it off.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import philox_ref
from .nn_ref import Weights, batch_norm_eval, conv1d, linear

DEFAULT_CFG = dict(
    vocab_size=37, n_tones=None, d_mels=80, reduction_factor=1, d_encoder=512, encoder_conv_layers=3,
    encoder_kernel_size=5, d_prenet=256, d_attention_rnn=1024, d_decoder_rnn=1024, d_attention=128,
    attention_filters=32, attention_kernel_size=31, d_postnet=512, postnet_kernel_size=5, postnet_conv_layers=5,
    p_prenet_dropout=0.5, d_global_condition=None, use_stop_token=True)


def stream_dropout(seed, units, p):
    def drop(step, layer, n_units):
        assert n_units == units
        idx = np.uint64((step * 2 + layer) * units) + np.arange(units, dtype=np.uint64)
        return philox_ref.dropout_keep(idx, p, seed)
    return drop


def lstm_cell(W, x, h, c):
    """paddle.nn.LSTMCell.forward."""
    gates = linear(x, W["weight_ih"].t(), W["bias_ih"]) + linear(h, W["weight_hh"].t(), W["bias_hh"])
    i, f, g, o = torch.chunk(gates, 4, dim=-1)
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    return torch.sigmoid(o) * torch.tanh(c2), c2


def conv_bn_nlc(W, x):
    """Conv1dBatchNorm.forward conv.py:245-260 with data_format "NLC": x (B, T, C)."""
    w = W["conv.weight"]
    y = conv1d(x.transpose(1, 2), w, W["conv.bias"], padding=int((w.shape[-1] - 1) / 2))
    y = batch_norm_eval(y, W["bn.weight"], W["bn.bias"], W["bn._mean"], W["bn._variance"])
    return y.transpose(1, 2)


def encoder(W, x, n_conv):
    """Tacotron2Encoder.forward :216-241, input_lens=None.  x (1, T, d_encoder)."""
    for i in range(n_conv):
        x = torch.relu(conv_bn_nlc(W.sub(f"conv_batchnorms.{i}."), x))
    T = x.shape[1]
    outs = []
    for cell, order in (("cell_fw", range(T)), ("cell_bw", range(T - 1, -1, -1))):
        C = W.sub(f"lstm.0.{cell}.")
        Hh = C["weight_hh"].shape[1]
        h = torch.zeros(1, Hh, dtype=x.dtype)
        c = torch.zeros(1, Hh, dtype=x.dtype)
        seq = [None] * T
        for t in order:
            h, c = lstm_cell(C, x[:, t], h, c)
            seq[t] = h
        outs.append(torch.stack(seq, dim=1))
    return torch.cat(outs, dim=-1)


def location_sensitive_attention(W, query, processed_key, value, attw_cat):
    """LocationSensitiveAttention.forward attention.py:300-348, mask None.
    query (1, d_query); processed_key (1, T, d_att); value (1, T, d_key); attw_cat (1, T, 2)."""
    pq = linear(query.unsqueeze(1), W["query_layer.weight"])
    w = W["location_conv.weight"]
    loc = conv1d(attw_cat.transpose(1, 2), w, None, padding=int((w.shape[-1] - 1) / 2)).transpose(1, 2)
    ploc = linear(loc, W["location_layer.weight"])
    alignment = linear(torch.tanh(ploc + processed_key + pq), W["value.weight"])      # (1, T, 1)
    weights = torch.softmax(alignment, dim=1)
    context = torch.matmul(weights.transpose(1, 2), value)                            # (1, 1, d_key)
    return context.squeeze(1), weights.squeeze(-1)


def postnet(W, x, n_layers):
    """DecoderPostNet.forward :147-171 (eval: dropout off).  x (1, T, d_mels)."""
    for i in range(n_layers - 1):
        x = torch.tanh(conv_bn_nlc(W.sub(f"conv_batchnorms.{i}."), x))
    return conv_bn_nlc(W.sub(f"conv_batchnorms.{n_layers - 1}."), x)


def infer(state, ids, cfg=None, tones=None, max_decoder_steps=1000, seed=0, drop="stream", dtype=torch.float32,
          return_parts=False, global_condition=None):
    """Tacotron2.infer :781-840 for one utterance.  ids (T,) int64.  Returns a dict with mel_output (L, d_mels),
    mel_outputs_postnet (L, d_mels), alignments (L, T) and, with a stop token, stop_logits (L,)."""
    cfg = dict(DEFAULT_CFG, **(cfg or {}))
    if cfg.get("reduction_factor", 1) != 1:
        raise NotImplementedError("reduction_factor != 1: Tacotron2.infer cannot run it (postnet on (B, T, C * r), :822-826)")
    W = Weights(state, dtype)
    x = torch.as_tensor(np.asarray(ids)).to(torch.int64).reshape(1, -1)
    emb = W["embedding.weight"][x]                                                    # :807-808
    if cfg.get("n_tones"):
        tn = torch.as_tensor(np.asarray(tones)).to(torch.int64).reshape(1, -1)
        te = W["embedding_tones.weight"][tn]
        emb = emb + torch.where((tn == 0).unsqueeze(-1), torch.zeros_like(te), te)    # padding_idx=0 [paddle-semantics]
    key = encoder(W.sub("encoder."), emb, cfg["encoder_conv_layers"])                 # :811
    enc_out = key
    if global_condition is not None:                                                  # :816-821
        g = torch.as_tensor(np.asarray(global_condition)).to(dtype).reshape(1, 1, -1)
        key = torch.cat([key, g.expand(-1, key.shape[1], -1)], dim=-1)
    D = W.sub("decoder.")
    A = D.sub("attention_layer.")
    T = key.shape[1]
    Ha, Hd = D["attention_rnn.weight_hh"].shape[1], D["decoder_rnn.weight_hh"].shape[1]
    M = cfg["d_mels"]
    z = lambda n: torch.zeros(1, n, dtype=dtype)                                      # noqa: E731  (:352-372)
    att_h, att_c, dec_h, dec_c = z(Ha), z(Ha), z(Hd), z(Hd)
    attw, attw_cum, ctx = z(T), z(T), z(key.shape[2])
    pkey = linear(key, A["key_layer.weight"])                                         # :376
    p = float(cfg["p_prenet_dropout"])
    if drop == "stream":
        drop = stream_dropout(seed, cfg["d_prenet"], p)
    query = z(M)                                                                      # :493-497
    first_hit_end = None
    mels, aligns, stops = [], [], []
    for i in range(max_decoder_steps):
        q = query
        for j, nm in enumerate(("linear1", "linear2")):                               # DecoderPreNet :76-79
            q = torch.relu(linear(q, D[f"prenet.{nm}.weight"]))
            if drop is not None and p > 0:
                keep = torch.as_tensor(drop(i, j, q.shape[1]))
                q = torch.where(keep.unsqueeze(0), q / (1.0 - p), torch.zeros_like(q))
        att_h, att_c = lstm_cell(D.sub("attention_rnn."), torch.cat([q, ctx], dim=-1), att_h, att_c)   # :381-385
        ctx, attw = location_sensitive_attention(A, att_h, pkey, key, torch.stack([attw, attw_cum], dim=-1))
        attw_cum = attw_cum + attw                                                    # :397
        dec_h, dec_c = lstm_cell(D.sub("decoder_rnn."), torch.cat([att_h, ctx], dim=-1), dec_h, dec_c)   # :400-403
        hc = torch.cat([dec_h, ctx], dim=-1)
        mel = linear(hc, D["linear_projection.weight"], D["linear_projection.bias"])  # :411-413
        mels.append(mel)
        aligns.append(attw)
        if cfg["use_stop_token"]:
            stop = linear(hc, D["stop_layer.weight"], D["stop_layer.bias"])
            stops.append(stop)
            if float(torch.sigmoid(stop)) > 0.5:                                      # :515-518
                break
        else:
            if int(torch.argmax(attw[0])) == T - 1:                                   # :520-525
                if first_hit_end is None:
                    first_hit_end = i
                elif i > first_hit_end + 20:
                    break
        if len(mels) == max_decoder_steps:                                            # :526-528
            break
        query = mel
    mel_out = torch.stack(mels, dim=1)                                                # (1, L, M)
    post = mel_out + postnet(W.sub("postnet."), mel_out, cfg["postnet_conv_layers"])  # :825-826
    out = dict(mel_output=mel_out[0], mel_outputs_postnet=post[0], alignments=torch.stack(aligns, dim=1)[0])
    if cfg["use_stop_token"]:
        out["stop_logits"] = torch.cat(stops, dim=1)[0]
    if return_parts:
        out["encoder_outputs"] = enc_out[0]
    return out
