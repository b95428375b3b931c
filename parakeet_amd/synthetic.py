"""Synthetic configurations, weights and inputs of LJSpeech shape.

There are no checkpoints in this environment (and ``.pdz`` files need Paddle
to unpickle), so benchmarks and parity tests run on random-initialised weights
laid out exactly like the reference's ``state_dict()``: same key
names, same array shapes.  The generators follow SURVEY.md 8(d):

* FastSpeech2: Xavier-uniform for every >=2-D weight (init_type
  ``xavier_uniform``, examples/fastspeech2/ljspeech/conf/default.yaml:53,
  parakeet/modules/nets_utils.py:144-146), alpha = 1.  With ``perturb=True``
  biases, LayerNorm / BatchNorm parameters and running stats are randomised so
  that a parity test cannot pass with a dropped bias or a swapped gamma/beta.
* duration head: ``fixed_duration=d`` sets ``duration_predictor.linear`` to
  weight 0 / bias ln(d+1) so every token gets exactly d frames through the
  normal inference path (the throughput configuration, L = 5*T); otherwise a
  random head with bias ln 4 gives ragged integer durations.
* Parallel WaveGAN: U(-1/sqrt(Cin*k), 1/sqrt(Cin*k)), weight-norm already
  folded (``weight_norm=True`` emits weight_g / weight_v pairs instead).
"""
import math

import numpy as np

FS2_LJSPEECH = dict(
    adim=384, aheads=2, elayers=4, eunits=1536, dlayers=4, dunits=1536,
    positionwise_layer_type="conv1d", positionwise_conv_kernel_size=3,
    duration_predictor_layers=2, duration_predictor_chans=256, duration_predictor_kernel_size=3,
    postnet_layers=5, postnet_filts=5, postnet_chans=256,
    use_scaled_pos_enc=True, encoder_normalize_before=True, decoder_normalize_before=True,
    reduction_factor=1, init_type="xavier_uniform", init_enc_alpha=1.0, init_dec_alpha=1.0,
    pitch_predictor_layers=5, pitch_predictor_chans=256, pitch_predictor_kernel_size=5,
    pitch_embed_kernel_size=1,
    energy_predictor_layers=2, energy_predictor_chans=256, energy_predictor_kernel_size=3,
    energy_embed_kernel_size=1)

PWG_LJSPEECH = dict(
    in_channels=1, out_channels=1, kernel_size=3, layers=30, stacks=3,
    residual_channels=64, gate_channels=128, skip_channels=64, aux_channels=80,
    aux_context_window=2, dropout=0.0, use_weight_norm=True, upsample_scales=[4, 4, 4, 4])

WAVEFLOW_LJSPEECH = dict(upsample_factors=[16, 16], n_flows=8, n_layers=8, n_group=16, channels=128,
                         n_mels=80, kernel_size=[3, 3])   # examples/waveflow/config.py:32-41 (C=64: paper's small model)

SPEEDYSPEECH_BAKER = dict(   # examples/speedyspeech/baker/conf/default.yaml:23-31
    encoder_hidden_size=128, encoder_kernel_size=3, encoder_dilations=[1, 3, 9, 27, 1, 3, 9, 27, 1, 1],
    duration_predictor_hidden_size=128, decoder_hidden_size=128, decoder_output_size=80, decoder_kernel_size=3,
    decoder_dilations=[1, 3, 9, 27, 1, 3, 9, 27, 1, 3, 9, 27, 1, 3, 9, 27, 1, 1])

SAMPLE_RATE = 22050
HOP = 256


def _xavier(rng, shape):
    """Xavier-uniform with Paddle's fan computation: for a [in, out] Linear
    weight fan_in = shape[0], fan_out = shape[1]; for conv [Cout, Cin, k]
    fan_in = Cin*k, fan_out = Cout*k."""
    if len(shape) == 2:
        fan_in, fan_out = shape
    else:
        rf = int(np.prod(shape[2:]))
        fan_in, fan_out = shape[1] * rf, shape[0] * rf
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def fastspeech2_state(idim=80, odim=80, cfg=None, seed=10086, fixed_duration=None, perturb=True,
                      num_speakers=None, num_tones=None):
    """``cfg`` may carry spk_embed_dim / spk_embed_integration_type (aishell3 / vctk recipes: 256, "concat");
    ``num_speakers`` then sizes ``spk_embedding_table`` (fastspeech2.py:147-151, 190-194)."""
    cfg = dict(FS2_LJSPEECH, **(cfg or {}))
    rng = np.random.default_rng(seed)
    A = cfg["adim"]
    st = {}

    def small(n, scale=0.1):
        return (rng.uniform(-scale, scale, size=(n,)) if perturb else np.zeros(n)).astype(np.float32)

    def gamma(n):
        return (rng.uniform(0.5, 1.5, size=(n,)) if perturb else np.ones(n)).astype(np.float32)

    def ln(prefix, n):
        st[prefix + ".weight"] = gamma(n)
        st[prefix + ".bias"] = small(n)

    def fft_stack(prefix, n_layers, units, embed_idx):
        st[f"{prefix}.embed.{embed_idx}.alpha"] = np.array(
            [cfg["init_enc_alpha" if prefix == "encoder" else "init_dec_alpha"]], dtype=np.float32)
        k = cfg["positionwise_conv_kernel_size"]
        for i in range(n_layers):
            p = f"{prefix}.encoders.{i}"
            for nm in ("q", "k", "v", "out"):
                st[f"{p}.self_attn.linear_{nm}.weight"] = _xavier(rng, (A, A))
                st[f"{p}.self_attn.linear_{nm}.bias"] = small(A)
            kind = cfg.get("positionwise_layer_type", "conv1d")   # encoder.py:145-170
            st[f"{p}.feed_forward.w_1.weight"] = _xavier(rng, (A, units) if kind == "linear" else (units, A, k))
            st[f"{p}.feed_forward.w_1.bias"] = small(units)
            st[f"{p}.feed_forward.w_2.weight"] = _xavier(rng, (A, units, k) if kind == "conv1d" else (units, A))
            st[f"{p}.feed_forward.w_2.bias"] = small(A)
            ln(f"{p}.norm1", A)
            ln(f"{p}.norm2", A)
            if cfg.get(f"{prefix}_concat_after"):       # encoder_layer.py:61-62
                st[f"{p}.concat_linear.weight"] = _xavier(rng, (2 * A, A))
                st[f"{p}.concat_linear.bias"] = small(A)
        if cfg.get(f"{prefix}_normalize_before", True):  # encoder.py:142-143
            ln(f"{prefix}.after_norm", A)

    emb = _xavier(rng, (idim, A))
    emb[0] = 0.0  # padding_idx row
    st["encoder.embed.0.weight"] = emb
    fft_stack("encoder", cfg["elayers"], cfg["eunits"], 1)
    fft_stack("decoder", cfg["dlayers"], cfg["dunits"], 0)

    def predictor(prefix, n_layers, chans, k):
        for j in range(n_layers):
            cin = A if j == 0 else chans
            st[f"{prefix}.conv.{j}.0.weight"] = _xavier(rng, (chans, cin, k))
            st[f"{prefix}.conv.{j}.0.bias"] = small(chans)
            ln(f"{prefix}.conv.{j}.2", chans)
        st[f"{prefix}.linear.weight"] = _xavier(rng, (chans, 1))
        st[f"{prefix}.linear.bias"] = small(1)

    predictor("duration_predictor", cfg["duration_predictor_layers"],
              cfg["duration_predictor_chans"], cfg["duration_predictor_kernel_size"])
    predictor("pitch_predictor", cfg["pitch_predictor_layers"],
              cfg["pitch_predictor_chans"], cfg["pitch_predictor_kernel_size"])
    predictor("energy_predictor", cfg["energy_predictor_layers"],
              cfg["energy_predictor_chans"], cfg["energy_predictor_kernel_size"])
    if fixed_duration is not None:
        st["duration_predictor.linear.weight"][:] = 0.0
        st["duration_predictor.linear.bias"][:] = math.log(fixed_duration + 1.0)
    else:
        c = cfg["duration_predictor_chans"]
        st["duration_predictor.linear.weight"] = rng.uniform(-0.05, 0.05, size=(c, 1)).astype(np.float32)
        st["duration_predictor.linear.bias"] = np.array([math.log(4.0)], dtype=np.float32)
    for nm in ("pitch", "energy"):
        k = cfg[f"{nm}_embed_kernel_size"]
        st[f"{nm}_embed.0.weight"] = _xavier(rng, (A, 1, k))
        st[f"{nm}_embed.0.bias"] = small(A)
    st["feat_out.weight"] = _xavier(rng, (A, odim * cfg["reduction_factor"]))
    st["feat_out.bias"] = small(odim * cfg["reduction_factor"])
    n, ch, kf = cfg["postnet_layers"], cfg["postnet_chans"], cfg["postnet_filts"]
    for j in range(n):
        cin = odim if j == 0 else ch
        cout = odim if j == n - 1 else ch
        st[f"postnet.postnet.{j}.0.weight"] = _xavier(rng, (cout, cin, kf))
        st[f"postnet.postnet.{j}.1.weight"] = rng.uniform(0.5, 1.5, size=(cout,)).astype(np.float32)
        st[f"postnet.postnet.{j}.1.bias"] = small(cout)
        st[f"postnet.postnet.{j}.1._mean"] = small(cout)
        st[f"postnet.postnet.{j}.1._variance"] = (
            rng.uniform(0.5, 1.5, size=(cout,)) if perturb else np.ones(cout)).astype(np.float32)
    if cfg.get("tone_embed_dim") is not None:
        Dt = cfg["tone_embed_dim"]
        tt = rng.normal(size=(num_tones, Dt)).astype(np.float32)
        tt[0] = 0.0
        st["tone_embedding_table.weight"] = tt
        st["tone_projection.weight"] = _xavier(rng, (Dt, A))   # "add" integration (:197-199)
        st["tone_projection.bias"] = small(A)
    if cfg.get("spk_embed_dim") is not None:
        D = cfg["spk_embed_dim"]
        tab = rng.normal(size=(num_speakers, D)).astype(np.float32)
        tab[0] = 0.0  # padding_idx row (nn.Embedding(padding_idx=0), :147-151)
        st["spk_embedding_table.weight"] = tab
        kin = D if cfg.get("spk_embed_integration_type", "add") == "add" else A + D
        st["spk_projection.weight"] = _xavier(rng, (kin, A))
        st["spk_projection.bias"] = small(A)
    return st


def pwg_state(cfg=None, seed=42, weight_norm=False):
    cfg = dict(PWG_LJSPEECH, **(cfg or {}))
    rng = np.random.default_rng(seed)
    st = {}

    def conv(name, cout, cin, k, bias=True):
        lim = 1.0 / math.sqrt(cin * k)
        w = rng.uniform(-lim, lim, size=(cout, cin, k)).astype(np.float32)
        _put(name, w)
        if bias:
            st[name + ".bias"] = rng.uniform(-lim, lim, size=(cout,)).astype(np.float32)

    def _put(name, w):
        if weight_norm:
            # weight_norm(dim=0): g is the per-output-channel norm (1-D, tests/unit/test_pwg.py:131-132)
            g = rng.uniform(0.5, 1.5, size=(w.shape[0],)).astype(np.float32)
            v = w
            st[name + ".weight_g"] = g
            st[name + ".weight_v"] = v
        else:
            st[name + ".weight"] = w

    R, G, S, A = (cfg["residual_channels"], cfg["gate_channels"], cfg["skip_channels"],
                  cfg["aux_channels"])
    conv("first_conv", R, cfg["in_channels"], 1)
    conv("upsample_net.conv_in", A, A, 2 * cfg["aux_context_window"] + 1, bias=False)
    for i, s in enumerate(cfg["upsample_scales"]):
        k = 2 * s + 1
        w = (1.0 / k + rng.uniform(-0.05, 0.05, size=(1, 1, 1, k))).astype(np.float32)
        _put(f"upsample_net.upsample.up_layers.{2 * i + 1}", w)
    for i in range(cfg["layers"]):
        p = f"conv_layers.{i}"
        conv(p + ".conv", G, R, cfg["kernel_size"])
        conv(p + ".conv1x1_aux", G, A, 1, bias=False)
        conv(p + ".conv1x1_out", R, G // 2, 1)
        conv(p + ".conv1x1_skip", S, G // 2, 1)
    conv("last_conv_layers.1", S, S, 1)
    conv("last_conv_layers.3", cfg["out_channels"], S, 1)
    return st


def phoneme_ids(n_tokens, idim=80, seed=10086):
    """ids in [1, idim-2]: never the pad id 0, never <eos> = idim-1
    (parakeet/models/fastspeech2/fastspeech2.py:126,143)."""
    rng = np.random.default_rng(seed)
    return rng.integers(1, idim - 1, size=n_tokens).astype(np.int64)


def mel_stats(odim=80, seed=7, identity=False):
    """(mu, sigma) pairs like *_stats.npy = np.stack([mean_, scale_])
    (utils/compute_statistics.py:101-107)."""
    if identity:
        return np.zeros(odim, np.float32), np.ones(odim, np.float32)
    rng = np.random.default_rng(seed)
    return (rng.normal(-1.0, 0.5, size=odim).astype(np.float32),
            rng.uniform(0.5, 1.5, size=odim).astype(np.float32))


def waveflow_state(cfg=None, seed=2021, weight_norm=False, zero_output_proj=False):
    """ConditionalWaveFlow state dict (parakeet/models/waveflow.py): encoder = 2 Conv2DTranspose,
    decoder = n_flows Flows.  The reference initialises output_proj to zero (:441-446), which makes
    every flow the identity; parity tests need a non-trivial transform, so it is small-random here
    unless ``zero_output_proj``."""
    cfg = dict(WAVEFLOW_LJSPEECH, **(cfg or {}))
    rng = np.random.default_rng(seed)
    C, M = cfg["channels"], cfg["n_mels"]
    kh, kw = cfg["kernel_size"]
    st = {}

    def put(name, w, b):
        if weight_norm:
            st[name + ".weight_g"] = np.sqrt((w.reshape(w.shape[0], -1) ** 2).sum(1)).astype(np.float32) * \
                rng.uniform(0.8, 1.2, size=w.shape[0]).astype(np.float32)
            st[name + ".weight_v"] = w
        else:
            st[name + ".weight"] = w
        st[name + ".bias"] = b

    def u(lim, shape):
        return rng.uniform(-lim, lim, size=shape).astype(np.float32)

    for i, f in enumerate(cfg["upsample_factors"]):
        std = math.sqrt(1 / (3 * 2 * f))
        # positive-mean taps so the two leaky-relu stages keep a usable dynamic range
        put(f"encoder.{i}", (u(std, (1, 1, 3, 2 * f)) + std).astype(np.float32), u(std, (1,)))
    for fl in range(cfg["n_flows"]):
        p = f"decoder.{fl}"
        put(p + ".input_proj", u(1.0, (C, 1, 1, 1)), u(1.0, (C,)))
        for l in range(cfg["n_layers"]):
            q = f"{p}.resnet.{l}"
            std = math.sqrt(1.0 / (C * kh * kw))
            put(q + ".conv", u(std, (2 * C, C, kh, kw)), u(std, (2 * C,)))
            std = math.sqrt(1.0 / M)
            put(q + ".condition_proj", u(std, (2 * C, M, 1, 1)), u(std, (2 * C,)))
            std = math.sqrt(1.0 / C)
            put(q + ".out_proj", u(std, (2 * C, C, 1, 1)), u(std, (2 * C,)))
        if zero_output_proj:
            st[p + ".output_proj.weight"] = np.zeros((2, C, 1, 1), np.float32)
            st[p + ".output_proj.bias"] = np.zeros((2,), np.float32)
        else:
            std = 0.3 * math.sqrt(1.0 / C)
            st[p + ".output_proj.weight"] = u(std, (2, C, 1, 1))
            st[p + ".output_proj.bias"] = u(0.05, (2,))
    return st


def speedyspeech_state(cfg=None, vocab_size=70, tone_size=7, seed=303, mean_duration=3.0):
    """SpeedySpeech state dict (parakeet/models/speedyspeech/speedyspeech.py:21-170).  Keys follow the
    attribute paths: Sequential(Conv1D, ReLU, BatchNorm1D) -> ``.0`` conv, ``.2`` batch norm.  The duration
    head gets a bias of ln(mean_duration) and small weights so that round(exp(.)) spreads over 1..6."""
    cfg = dict(SPEEDYSPEECH_BAKER, **(cfg or {}))
    rng = np.random.default_rng(seed)
    H = cfg["encoder_hidden_size"]
    st = {}

    def small(n, scale=0.1):
        return rng.uniform(-scale, scale, size=(n,)).astype(np.float32)

    def bn(prefix, n):
        # small gains keep the 28-block residual streams O(1..10) with random weights
        st[prefix + ".weight"] = rng.uniform(0.1, 0.4, size=(n,)).astype(np.float32)
        st[prefix + ".bias"] = small(n)
        st[prefix + "._mean"] = small(n)
        st[prefix + "._variance"] = rng.uniform(0.5, 1.5, size=(n,)).astype(np.float32)

    def lin(prefix, cin, cout):
        st[prefix + ".weight"] = _xavier(rng, (cin, cout))
        st[prefix + ".bias"] = small(cout)

    def res_block(prefix, ch, k, n):
        for j in range(n):
            st[f"{prefix}.blocks.{j}.0.weight"] = _xavier(rng, (ch, ch, k))
            st[f"{prefix}.blocks.{j}.0.bias"] = small(ch)
            bn(f"{prefix}.blocks.{j}.2", ch)

    emb = rng.normal(scale=0.5, size=(vocab_size, H)).astype(np.float32)
    emb[0] = 0.0
    st["encoder.embedding.text_embedding.weight"] = emb
    if tone_size:
        t = rng.normal(scale=0.5, size=(tone_size, H)).astype(np.float32)
        t[0] = 0.0
        st["encoder.embedding.tone_embedding.weight"] = t
    lin("encoder.prenet.0", H, H)
    for i, _ in enumerate(cfg["encoder_dilations"]):
        res_block(f"encoder.res_blocks.{i}", H, cfg["encoder_kernel_size"], 2)
    lin("encoder.postnet1.0", H, H)
    bn("encoder.postnet2.1", H)
    lin("encoder.postnet2.2", H, H)
    Hd = cfg["duration_predictor_hidden_size"]
    for i, k in enumerate((4, 3, 1)):
        res_block(f"duration_predictor.layers.{i}", Hd, k, 1)
    st["duration_predictor.layers.3.weight"] = rng.uniform(-0.3, 0.3, size=(Hd, 1)).astype(np.float32)
    st["duration_predictor.layers.3.bias"] = np.array([math.log(mean_duration)], dtype=np.float32)
    D = cfg["decoder_hidden_size"]
    for i, _ in enumerate(cfg["decoder_dilations"]):
        res_block(f"decoder.res_blocks.{i}", D, cfg["decoder_kernel_size"], 2)
    lin("decoder.postnet1.0", D, D)
    res_block("decoder.postnet2.0", D, cfg["decoder_kernel_size"], 2)
    lin("decoder.postnet2.1", D, cfg["decoder_output_size"])
    return st


# examples/transformer_tts/ljspeech/conf/default.yaml:19-45
TRANSFORMER_TTS_LJSPEECH = dict(
    embed_dim=0, eprenet_conv_layers=0, eprenet_conv_filts=0, eprenet_conv_chans=0,
    dprenet_layers=2, dprenet_units=256, adim=512, aheads=8, elayers=6, eunits=1024, dlayers=6, dunits=1024,
    positionwise_layer_type="conv1d", positionwise_conv_kernel_size=1,
    postnet_layers=5, postnet_filts=5, postnet_chans=256, use_scaled_pos_enc=True,
    encoder_normalize_before=True, decoder_normalize_before=True, reduction_factor=1,
    init_type="xavier_uniform", init_enc_alpha=1.0, init_dec_alpha=1.0)


def transformer_tts_state(idim=80, odim=80, cfg=None, seed=4242, stop_bias=0.0, stop_gain=1.0):
    """TransformerTTS state dict (parakeet/models/transformer_tts/transformer_tts.py:172-358): key names and shapes
    are the reference's ``state_dict()``.  Encoder input layer: nn.Embedding(padding_idx=0) when
    ``eprenet_conv_layers == 0`` (the LJSpeech recipe), else Sequential(EncoderPrenet, Linear) (:258-277); decoder
    input layer: Sequential(DecoderPrenet, Linear) (:311-321).  ``stop_bias`` / ``stop_gain`` shift and scale the
    stop-token head ``prob_out`` so that a test can choose where sigmoid(.) >= threshold fires."""
    cfg = dict(TRANSFORMER_TTS_LJSPEECH, **(cfg or {}))
    rng = np.random.default_rng(seed)
    A, H = cfg["adim"], cfg["aheads"]
    assert A % H == 0
    st = {}

    def small(n, scale=0.1):
        return rng.uniform(-scale, scale, size=(n,)).astype(np.float32)

    def ln(prefix, n):
        st[prefix + ".weight"] = rng.uniform(0.5, 1.5, size=(n,)).astype(np.float32)
        st[prefix + ".bias"] = small(n)

    def lin(prefix, cin, cout):
        st[prefix + ".weight"] = _xavier(rng, (cin, cout))
        st[prefix + ".bias"] = small(cout)

    def bn(prefix, n):
        st[prefix + ".weight"] = rng.uniform(0.5, 1.5, size=(n,)).astype(np.float32)
        st[prefix + ".bias"] = small(n)
        st[prefix + "._mean"] = small(n)
        st[prefix + "._variance"] = rng.uniform(0.5, 1.5, size=(n,)).astype(np.float32)

    def mha(prefix):
        for nm in ("q", "k", "v", "out"):
            lin(f"{prefix}.linear_{nm}", A, A)

    if cfg["eprenet_conv_layers"] > 0:
        E, C, kf = cfg["embed_dim"], cfg["eprenet_conv_chans"], cfg["eprenet_conv_filts"]
        emb = _xavier(rng, (idim, E))
        emb[0] = 0.0
        st["encoder.embed.0.0.embed.weight"] = emb
        for i in range(cfg["eprenet_conv_layers"]):
            st[f"encoder.embed.0.0.convs.{i}.0.weight"] = _xavier(rng, (C, E if i == 0 else C, kf))
            bn(f"encoder.embed.0.0.convs.{i}.1", C)
        lin("encoder.embed.0.1", C, A)
    else:
        emb = _xavier(rng, (idim, A))
        emb[0] = 0.0
        st["encoder.embed.0.weight"] = emb
    scaled = cfg.get("use_scaled_pos_enc", True)   # PositionalEncoding has no alpha parameter (embedding.py:44-62)
    if scaled:
        st["encoder.embed.1.alpha"] = np.array(cfg["init_enc_alpha"], dtype=np.float32)
    k = cfg["positionwise_conv_kernel_size"]
    kind = cfg.get("positionwise_layer_type", "conv1d")
    for i in range(cfg["elayers"]):
        p = f"encoder.encoders.{i}"
        mha(p + ".self_attn")
        U = cfg["eunits"]
        st[f"{p}.feed_forward.w_1.weight"] = _xavier(rng, (A, U) if kind == "linear" else (U, A, k))
        st[f"{p}.feed_forward.w_1.bias"] = small(U)
        st[f"{p}.feed_forward.w_2.weight"] = _xavier(rng, (A, U, k) if kind == "conv1d" else (U, A))
        st[f"{p}.feed_forward.w_2.bias"] = small(A)
        ln(p + ".norm1", A)
        ln(p + ".norm2", A)
        if cfg.get("encoder_concat_after"):
            lin(p + ".concat_linear", 2 * A, A)
    if cfg.get("encoder_normalize_before", True):
        ln("encoder.after_norm", A)
    P = cfg["dprenet_units"]
    for j in range(cfg["dprenet_layers"]):
        lin(f"decoder.embed.0.0.prenet.{j}.0", odim if j == 0 else P, P)
    if cfg["dprenet_layers"] > 0:
        lin("decoder.embed.0.1", P, A)
        pos = "decoder.embed.1"
    else:   # input_layer "linear": Sequential(Linear, LayerNorm, Dropout, ReLU, pos_enc) (decoder.py:112-118)
        lin("decoder.embed.0", odim, A)
        ln("decoder.embed.1", A)
        pos = "decoder.embed.4"
    if scaled:
        st[pos + ".alpha"] = np.array(cfg["init_dec_alpha"], dtype=np.float32)
    for i in range(cfg["dlayers"]):
        p = f"decoder.decoders.{i}"
        mha(p + ".self_attn")
        mha(p + ".src_attn")
        lin(p + ".feed_forward.w_1", A, cfg["dunits"])
        lin(p + ".feed_forward.w_2", cfg["dunits"], A)
        for n in (1, 2, 3):
            ln(f"{p}.norm{n}", A)
        if cfg.get("decoder_concat_after"):   # decoder_layer.py:66-68
            lin(p + ".concat_linear1", 2 * A, A)
            lin(p + ".concat_linear2", 2 * A, A)
    if cfg.get("decoder_normalize_before", True):
        ln("decoder.after_norm", A)
    r = cfg["reduction_factor"]
    lin("feat_out", A, odim * r)
    lin("prob_out", A, r)
    st["prob_out.weight"] = (st["prob_out.weight"] * stop_gain).astype(np.float32)
    st["prob_out.bias"] = (st["prob_out.bias"] + stop_bias).astype(np.float32)
    n, ch, kf = cfg["postnet_layers"], cfg["postnet_chans"], cfg["postnet_filts"]
    for j in range(n):
        cin = odim if j == 0 else ch
        cout = odim if j == n - 1 else ch
        st[f"postnet.postnet.{j}.0.weight"] = _xavier(rng, (cout, cin, kf))
        bn(f"postnet.postnet.{j}.1", cout)
    if cfg.get("use_gst"):   # StyleEncoder (modules/style_encoder.py), idim = odim, gst_token_dim = adim (:299-310)
        chans = list(cfg.get("gst_conv_chans_list", (32, 32, 64, 64, 128, 128)))
        kk, stride = cfg.get("gst_conv_kernel_size", 3), cfg.get("gst_conv_stride", 2)
        F = odim
        for i, co in enumerate(chans):
            st[f"gst.ref_enc.convs.{3 * i}.weight"] = _xavier(rng, (co, 1 if i == 0 else chans[i - 1], kk, kk))
            bn(f"gst.ref_enc.convs.{3 * i + 1}", co)
            F = (F - kk + 2 * ((kk - 1) // 2)) // stride + 1
        Hg = cfg.get("gst_gru_units", 128)
        for l in range(cfg.get("gst_gru_layers", 1)):
            isz = F * chans[-1] if l == 0 else Hg
            kq = 1.0 / math.sqrt(Hg)
            for nm, shape in (("weight_ih", (3 * Hg, isz)), ("weight_hh", (3 * Hg, Hg)), ("bias_ih", (3 * Hg,)),
                              ("bias_hh", (3 * Hg,))):
                w = rng.uniform(-kq, kq, size=shape).astype(np.float32)
                st[f"gst.ref_enc.gru.{l}.cell.{nm}"] = w
                st[f"gst.ref_enc.gru.{nm}_l{l}"] = w      # paddle registers every cell parameter twice (as nn.LSTM)
        heads, tokens = cfg.get("gst_heads", 4), cfg.get("gst_tokens", 10)
        st["gst.stl.gst_embs"] = rng.standard_normal((tokens, A // heads)).astype(np.float32)
        lin("gst.stl.mha.linear_q", Hg, A)
        lin("gst.stl.mha.linear_k", A // heads, A)
        lin("gst.stl.mha.linear_v", A // heads, A)
        lin("gst.stl.mha.linear_out", A, A)
    D = cfg.get("spk_embed_dim")
    if D:   # :313-317
        lin("projection", D if cfg.get("spk_embed_integration_type", "add") == "add" else A + D, A)
    return st


# examples/tacotron2/config.py:31-54
TACOTRON2_LJSPEECH = dict(
    vocab_size=37, n_tones=None, d_mels=80, reduction_factor=1, d_encoder=512, encoder_conv_layers=3,
    encoder_kernel_size=5, d_prenet=256, d_attention_rnn=1024, d_decoder_rnn=1024, d_attention=128,
    attention_filters=32, attention_kernel_size=31, d_postnet=512, postnet_kernel_size=5, postnet_conv_layers=5,
    p_encoder_dropout=0.5, p_prenet_dropout=0.5, p_attention_dropout=0.1, p_decoder_dropout=0.1,
    p_postnet_dropout=0.5, d_global_condition=None, use_stop_token=True)


def tacotron2_state(cfg=None, seed=2222, stop_bias=0.0, stop_gain=1.0, lstm_aliases=True):
    """Tacotron2 state dict (parakeet/models/tacotron2.py:626-689).  ``lstm_aliases``: paddle.nn.LSTM registers every
    cell parameter twice -- "encoder.lstm.0.cell_fw.weight_ih" and the cuDNN-style "encoder.lstm.weight_ih_l0"
    (RNNBase.__init__) -- and a state dict carries both names for the same array; False keeps only the cell names."""
    cfg = dict(TACOTRON2_LJSPEECH, **(cfg or {}))
    rng = np.random.default_rng(seed)
    E, M = cfg["d_encoder"], cfg["d_mels"] * cfg["reduction_factor"]
    st = {}

    def u(shape, lim):
        return rng.uniform(-lim, lim, size=shape).astype(np.float32)

    def conv_bn(prefix, cout, cin, k):
        st[prefix + ".conv.weight"] = _xavier(rng, (cout, cin, k))
        st[prefix + ".conv.bias"] = u((cout,), 0.1)
        st[prefix + ".bn.weight"] = rng.uniform(0.5, 1.5, size=(cout,)).astype(np.float32)
        st[prefix + ".bn.bias"] = u((cout,), 0.1)
        st[prefix + ".bn._mean"] = u((cout,), 0.1)
        st[prefix + ".bn._variance"] = rng.uniform(0.5, 1.5, size=(cout,)).astype(np.float32)

    def lstm_cell(prefix, isz, hsz):
        k = 1.0 / math.sqrt(hsz)
        st[prefix + ".weight_ih"] = u((4 * hsz, isz), k)
        st[prefix + ".weight_hh"] = u((4 * hsz, hsz), k)
        st[prefix + ".bias_ih"] = u((4 * hsz,), k)
        st[prefix + ".bias_hh"] = u((4 * hsz,), k)

    st["embedding.weight"] = u((cfg["vocab_size"], E), 0.5)
    if cfg.get("n_tones"):
        t = u((cfg["n_tones"], E), 0.3)
        t[0] = 0.0   # padding_idx=0
        st["embedding_tones.weight"] = t
    for i in range(cfg["encoder_conv_layers"]):
        conv_bn(f"encoder.conv_batchnorms.{i}", E, E, cfg["encoder_kernel_size"])
    Hh = E // 2
    for d, nm in enumerate(("cell_fw", "cell_bw")):
        lstm_cell(f"encoder.lstm.0.{nm}", E, Hh)
        if lstm_aliases:
            suffix = "_reverse" if d else ""
            for p in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                st[f"encoder.lstm.{p}_l0{suffix}"] = st[f"encoder.lstm.0.{nm}.{p}"]
    De = E + (cfg.get("d_global_condition") or 0)
    P, Ha, Hd, Da = cfg["d_prenet"], cfg["d_attention_rnn"], cfg["d_decoder_rnn"], cfg["d_attention"]
    st["decoder.prenet.linear1.weight"] = _xavier(rng, (M, P))
    st["decoder.prenet.linear2.weight"] = _xavier(rng, (P, P))
    lstm_cell("decoder.attention_rnn", P + De, Ha)
    st["decoder.attention_layer.query_layer.weight"] = _xavier(rng, (Ha, Da))
    st["decoder.attention_layer.key_layer.weight"] = _xavier(rng, (De, Da))
    st["decoder.attention_layer.value.weight"] = (_xavier(rng, (Da, 1)) * 4.0).astype(np.float32)   # peaky alignments
    F, K = cfg["attention_filters"], cfg["attention_kernel_size"]
    st["decoder.attention_layer.location_conv.weight"] = _xavier(rng, (F, 2, K))
    st["decoder.attention_layer.location_layer.weight"] = _xavier(rng, (F, Da))
    lstm_cell("decoder.decoder_rnn", Ha + De, Hd)
    st["decoder.linear_projection.weight"] = _xavier(rng, (Hd + De, M))
    st["decoder.linear_projection.bias"] = u((M,), 0.1)
    if cfg["use_stop_token"]:
        st["decoder.stop_layer.weight"] = (_xavier(rng, (Hd + De, 1)) * stop_gain).astype(np.float32)
        st["decoder.stop_layer.bias"] = np.array([stop_bias], dtype=np.float32)
    n, C, kf = cfg["postnet_conv_layers"], cfg["d_postnet"], cfg["postnet_kernel_size"]
    for j in range(n):
        conv_bn(f"postnet.conv_batchnorms.{j}", M if j == n - 1 else C, M if j == 0 else C, kf)
    return st
