"""Autoregressive acoustic models: the oracle against golden vectors produced by the reference's
own Python source over the paddle stand-in with the dropout stream injected (tools/make_golden_ar.py).  CPU only."""
import os
import sys

import numpy as np

from oracle import philox_ref
from oracle import tacotron2_ref as t2
from oracle import transformer_tts_ref as tt
from parakeet_amd import synthetic as syn

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ar_cases import T2_CASES, TTS_CASES  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_dropout_stream_basics():
    idx = np.arange(1 << 16, dtype=np.uint64)
    keep = philox_ref.dropout_keep(idx, 0.5, seed=7)
    assert abs(keep.mean() - 0.5) < 0.01
    assert abs(philox_ref.dropout_keep(idx, 0.1, seed=7).mean() - 0.9) < 0.01
    # a pure function of (seed, index): any slice equals the same slice of the whole
    assert np.array_equal(philox_ref.dropout_keep(idx[1001:1013], 0.5, seed=7), keep[1001:1013])
    assert not np.array_equal(philox_ref.dropout_keep(idx[:4096], 0.5, seed=8), keep[:4096])
    # a stream of its own: word 3 of the counter separates it from the pk_randn stream
    ctr = np.array([[5, 0, 0, 0]], dtype=np.uint64)
    a = philox_ref.philox4x32_10(ctr, (7, 0))
    ctr[0, 3] = philox_ref.DROPOUT_STREAM
    assert not np.array_equal(a, philox_ref.philox4x32_10(ctr, (7, 0)))
    assert philox_ref.dropout_threshold(0.5) == 1 << 31 and philox_ref.dropout_threshold(0.0) == 0


def test_transformer_tts_oracle_matches_reference_source():
    g = np.load(os.path.join(GOLD, "transformer_tts.npz"))
    for name, over, idim, T, seed, skw, kw in TTS_CASES:
        cfg = dict(syn.TRANSFORMER_TTS_LJSPEECH, **over)
        state = syn.transformer_tts_state(idim, 80, cfg, seed=seed, **skw)
        mel, probs, att = tt.inference(state, g[f"{name}_ids"], cfg, seed=seed,
                                       spembs=g[f"{name}_spemb"] if cfg.get("spk_embed_dim") else None,
                                       speech=g[f"{name}_speech"] if cfg.get("use_gst") else None, **kw)
        assert mel.shape == g[f"{name}_mel"].shape, name          # same stop decision
        assert np.abs(mel.numpy() - g[f"{name}_mel"]).max() < 2e-5, name
        assert np.abs(probs.numpy() - g[f"{name}_probs"]).max() < 1e-5, name
        assert att.shape == g[f"{name}_att"].shape
        assert np.abs(att.numpy() - g[f"{name}_att"]).max() < 1e-5, name


def test_transformer_tts_stop_logic():
    g = np.load(os.path.join(GOLD, "transformer_tts.npz"))
    # "stop": the loop ended because the stop probability crossed the threshold, before maxlen
    name, over, idim, T, seed, skw, kw = TTS_CASES[1]
    probs = g["stop_probs"]
    assert probs[-1] >= 0.5 and (probs[:-1] < 0.5).all() and len(probs) < int((T + 1) * kw["maxlenratio"])
    # "minlen": every probability is above the threshold, the length is int((T + 1) * minlenratio)
    name, over, idim, T, seed, skw, kw = TTS_CASES[2]
    assert (g["minlen_probs"] >= 0.5).all() and len(g["minlen_probs"]) == int((T + 1) * kw["minlenratio"])
    # the dropout mask matters: another seed gives another spectrogram (the prenet dropout is live at inference)
    cfg = dict(syn.TRANSFORMER_TTS_LJSPEECH, **TTS_CASES[0][1])
    state = syn.transformer_tts_state(40, 80, cfg, seed=11, stop_bias=-6.0)
    a = tt.inference(state, g["lj_ids"], cfg, seed=11, maxlenratio=0.5)[0].numpy()
    b = tt.inference(state, g["lj_ids"], cfg, seed=12, maxlenratio=0.5)[0].numpy()
    c = tt.inference(state, g["lj_ids"], cfg, drop=None, maxlenratio=0.5)[0].numpy()
    assert np.abs(a - b).max() > 1e-3 and np.abs(a - c).max() > 1e-3


def _engine_algorithm(state, texts, cfg, seeds, threshold=0.5, minlenratio=0.0, maxlenratio=10.0):
    """The decoding schedule of csrc/tts.hip restated on torch-CPU: utterances in lockstep, position-major rows
    (row = pos * B + b), layer 0 recomputed over the whole prefix at every step, K/V rows of layers >= 1 projected
    once (when new) and cached, the stop state per utterance, finished utterances stepped on and ignored."""
    import math
    import torch
    from oracle.nn_ref import Weights, layer_norm, linear, sinusoid_table
    W = Weights(state, torch.float64)
    D = W.sub("decoder.")
    B, A, H = len(texts), cfg["adim"], cfg["aheads"]
    dk, J, U, NL = A // H, cfg["dprenet_layers"], cfg["dprenet_units"], cfg["dlayers"]
    idim = state["encoder.embed.0.weight"].shape[0]
    mems = []
    for t in texts:
        x = torch.as_tensor(np.pad(np.asarray(t), (0, 1), constant_values=idim - 1)).to(torch.int64).unsqueeze(0)
        mems.append(tt.encode(W.sub("encoder."), x, cfg)[0])
    T = [m.shape[0] for m in mems]
    maxlen = [int(t * maxlenratio) for t in T]
    minlen = [int(t * minlenratio) for t in T]
    Lcap = max(max(1, a, b) for a, b in zip(maxlen, minlen))
    pe = sinusoid_table(Lcap, A, torch.float64) * D["embed.1.alpha"]
    Y = torch.zeros((Lcap + 1) * B, 80, dtype=torch.float64)
    XC = [torch.zeros(Lcap * B, A, dtype=torch.float64) for _ in range(NL)]
    QKV = [torch.zeros(Lcap * B, 3 * A, dtype=torch.float64) for _ in range(NL)]
    MKV = []
    for l in range(NL):
        S = D.sub(f"decoders.{l}.src_attn.")
        MKV.append([(linear(m, S["linear_k.weight"], S["linear_k.bias"]), linear(m, S["linear_v.weight"], S["linear_v.bias"]))
                    for m in mems])
    probs = torch.zeros(Lcap * B, dtype=torch.float64)
    length = [0] * B

    def qkv_proj(Wl, x):
        return torch.cat([linear(x, Wl[f"linear_{n}.weight"], Wl[f"linear_{n}.bias"]) for n in "qkv"], dim=-1)

    def attend(q, K, V):   # q (A,), K / V (n, A)
        out = torch.zeros(A, dtype=torch.float64)
        for h in range(H):
            sl = slice(h * dk, (h + 1) * dk)
            p = torch.softmax((K[:, sl] @ q[sl]) / math.sqrt(dk), dim=0)
            out[sl] = p @ V[:, sl]
        return out

    for s in range(1, Lcap + 1):
        R, nr = s * B, (s - 1) * B
        x = Y[:R]
        for j in range(J):
            x = torch.relu(linear(x, D[f"embed.0.0.prenet.{j}.0.weight"], D[f"embed.0.0.prenet.{j}.0.bias"]))
            r = np.arange(R, dtype=np.uint64)
            pos, b = r // np.uint64(B), (r % np.uint64(B)).astype(np.int64)
            tri = np.uint64(s * (s - 1) // 2)
            idx = ((tri + pos)[:, None] * np.uint64(J) + np.uint64(j)) * np.uint64(U) + np.arange(U, dtype=np.uint64)[None, :]
            keep = np.zeros((R, U), dtype=bool)
            for bb in range(B):
                keep[b == bb] = philox_ref.dropout_keep(idx[b == bb], 0.5, seeds[bb])
            x = torch.where(torch.as_tensor(keep), x * 2.0, torch.zeros_like(x))
        X0 = linear(x, D["embed.0.1.weight"], D["embed.0.1.bias"]) + pe[:s].repeat_interleave(B, dim=0)
        L0 = D.sub("decoders.0.")
        QKV[0][:R] = qkv_proj(L0.sub("self_attn."), layer_norm(X0, L0["norm1.weight"], L0["norm1.bias"]))
        for l in range(NL):
            Wl = D.sub(f"decoders.{l}.")
            xin = (X0 if l == 0 else XC[l - 1])[nr:R]
            if l > 0:
                QKV[l][nr:R] = qkv_proj(Wl.sub("self_attn."), layer_norm(xin, Wl["norm1.weight"], Wl["norm1.bias"]))
            ctx = torch.stack([attend(QKV[l][nr + b, :A], QKV[l][b:R:B, A:2 * A], QKV[l][b:R:B, 2 * A:]) for b in range(B)])
            rx = xin + linear(ctx, Wl["self_attn.linear_out.weight"], Wl["self_attn.linear_out.bias"])
            q = linear(layer_norm(rx, Wl["norm2.weight"], Wl["norm2.bias"]), Wl["src_attn.linear_q.weight"],
                       Wl["src_attn.linear_q.bias"])
            ctx = torch.stack([attend(q[b], *MKV[l][b]) for b in range(B)])
            rx = rx + linear(ctx, Wl["src_attn.linear_out.weight"], Wl["src_attn.linear_out.bias"])
            t = layer_norm(rx, Wl["norm3.weight"], Wl["norm3.bias"])
            f = torch.relu(linear(t, Wl["feed_forward.w_1.weight"], Wl["feed_forward.w_1.bias"]))
            XC[l][nr:R] = rx + linear(f, Wl["feed_forward.w_2.weight"], Wl["feed_forward.w_2.bias"])
        z = layer_norm(XC[-1][nr:R], D["after_norm.weight"], D["after_norm.bias"])
        Y[R:R + B] = linear(z, W["feat_out.weight"], W["feat_out.bias"])
        p = torch.sigmoid(linear(z, W["prob_out.weight"], W["prob_out.bias"]))[:, 0]
        probs[nr:R] = p
        for b in range(B):
            if length[b] == 0 and (p[b] >= threshold or s >= maxlen[b]) and s >= minlen[b]:
                length[b] = s
        if all(length):
            break
    return [(Y[B + b:(length[b] + 1) * B:B], probs[b:length[b] * B:B]) for b in range(B)]


def test_engine_decoding_schedule_is_the_reference_loop():
    """Lockstep, position-major, KV-cached decoding (what csrc/tts.hip launches) == the reference's per-utterance
    loop that re-embeds and re-projects the whole prefix at every step."""
    import torch
    cfg = dict(syn.TRANSFORMER_TTS_LJSPEECH, elayers=1, dlayers=2, postnet_layers=0)
    state = syn.transformer_tts_state(40, 80, dict(cfg, postnet_layers=2), seed=12, stop_bias=-0.7, stop_gain=2.0)
    texts = [syn.phoneme_ids(T, idim=40, seed=300 + T) for T in (6, 2, 4)]
    seeds = [12, 13, 15]
    got = _engine_algorithm(state, texts, cfg, seeds, maxlenratio=3.0)
    lens = []
    for t, sd, (outs, probs) in zip(texts, seeds, got):
        ref, rprobs, _ = tt.inference(state, t, cfg, maxlenratio=3.0, seed=sd, dtype=torch.float64)
        assert outs.shape == ref.shape
        assert np.abs(outs.numpy() - ref.numpy()).max() < 1e-10
        assert np.abs(probs.numpy() - rprobs.numpy()).max() < 1e-12
        lens.append(ref.shape[0])
    assert len(set(lens)) > 1


def test_tacotron2_oracle_matches_reference_source():
    g = np.load(os.path.join(GOLD, "tacotron2.npz"))
    for name, over, T, seed, skw, max_steps in T2_CASES:
        cfg = dict(syn.TACOTRON2_LJSPEECH, **over)
        state = syn.tacotron2_state(cfg, seed=seed, **skw)
        o = t2.infer(state, g[f"{name}_ids"], cfg, tones=g[f"{name}_tones"] if cfg["n_tones"] else None,
                     max_decoder_steps=max_steps, seed=seed,
                     global_condition=g[f"{name}_global_condition"] if cfg.get("d_global_condition") else None)
        for k, v in o.items():
            assert v.shape == g[f"{name}_{k}"].shape, (name, k)          # same stop decision
            tol = 2e-3 if (k == "stop_logits" and name == "stop") else 2e-5   # that head has a gain of 500
            assert np.abs(v.numpy() - g[f"{name}_{k}"]).max() < tol, (name, k)
    # the three ways a run ends
    p = 1.0 / (1.0 + np.exp(-g["stop_stop_logits"]))
    assert p[-1] > 0.5 and (p[:-1] <= 0.5).all() and len(p) == 29 and abs(p[-1] - 0.5) > 0.02
    assert "nostop_stop_logits" not in g.files
    a = g["nostop_alignments"].argmax(-1)
    first = int(np.argmax(a == a.shape[0] * 0 + g["nostop_ids"].shape[0] - 1))
    assert len(a) == first + 22                                          # i > first_hit_end + 20 (:523-525)
    assert g["maxsteps_mel_output"].shape[0] == 12


def test_tacotron2_lstm_aliases_and_dropout():
    import torch
    cfg = dict(syn.TACOTRON2_LJSPEECH, **T2_CASES[2][1])
    a = syn.tacotron2_state(cfg, seed=5, stop_bias=-8.0, lstm_aliases=True)
    b = syn.tacotron2_state(cfg, seed=5, stop_bias=-8.0, lstm_aliases=False)
    assert set(a) - set(b) == {f"encoder.lstm.{p}_l0{s}" for p in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")
                               for s in ("", "_reverse")}
    assert np.array_equal(a["encoder.lstm.weight_hh_l0_reverse"], a["encoder.lstm.0.cell_bw.weight_hh"])
    ids = np.arange(1, 8)
    x = t2.infer(b, ids, cfg, max_decoder_steps=5, seed=1)["mel_output"].numpy()
    y = t2.infer(b, ids, cfg, max_decoder_steps=5, seed=2)["mel_output"].numpy()
    z = t2.infer(b, ids, cfg, max_decoder_steps=5, drop=None)["mel_output"].numpy()
    assert x.shape == (5, 80) and np.abs(x - y).max() > 1e-3 and np.abs(x - z).max() > 1e-3
    assert np.array_equal(x[0], y[0])   # step 0: the query is zero, relu(0) = 0, the mask is moot
