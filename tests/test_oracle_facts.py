"""Pin the oracle against every machine-checkable fact the reference repository holds for the
path: docstring worked examples, the expansion test's input/shape, shape algebra,
and the Paddle-semantics choices the restatement encodes.  CPU only."""
import math

import numpy as np
import torch

from oracle import fastspeech2_ref as fs2
from oracle import nn_ref, pwg_ref
from parakeet_amd import synthetic as syn


def test_make_pad_mask_docstring_example():
    # parakeet/modules/nets_utils.py:71-75
    m = nn_ref.make_pad_mask([5, 3, 2]).int().tolist()
    assert m == [[0, 0, 0, 0, 0], [0, 0, 0, 1, 1], [0, 0, 1, 1, 1]]


def test_make_non_pad_mask_docstring_example():
    # parakeet/modules/nets_utils.py:119-123
    m = nn_ref.make_non_pad_mask([5, 3, 2]).int().tolist()
    assert m == [[1, 1, 1, 1, 1], [1, 1, 1, 0, 0], [1, 1, 0, 0, 0]]


def test_source_mask_docstring_example():
    # FastSpeech2._source_mask, fastspeech2.py:634-637: make_non_pad_mask(ilens).unsqueeze(-2)
    m = nn_ref.make_non_pad_mask([5, 3]).unsqueeze(-2)
    assert m.shape == (2, 1, 5)
    assert m.int().tolist() == [[[1, 1, 1, 1, 1]], [[1, 1, 1, 0, 0]]]


def test_expand_matches_reference_unit_test():
    # tests/unit/test_expansion.py:19-29: x (2,4,3), durations [[1,2,2,1],[3,1,4,0]] -> shape [2,8,3]
    x = torch.randn(2, 4, 3)
    ds = torch.tensor([[1, 2, 2, 1], [3, 1, 4, 0]], dtype=torch.float32)
    y = fs2.length_regulate(x, ds)
    assert list(y.shape) == [2, 8, 3]
    # values: pure row repeat, zero rows beyond an utterance's total (6 of 8 for the first)
    want0 = torch.stack([x[0, 0], x[0, 1], x[0, 1], x[0, 2], x[0, 2], x[0, 3], torch.zeros(3), torch.zeros(3)])
    want1 = torch.stack([x[1, 0]] * 3 + [x[1, 1]] + [x[1, 2]] * 4)
    assert torch.equal(y[0], want0) and torch.equal(y[1], want1)


def test_expand_skips_zero_durations_and_scales_with_alpha():
    x = torch.arange(12, dtype=torch.float32).reshape(1, 4, 3)
    y = fs2.length_regulate(x, torch.tensor([[0., 2., 0., 1.]]))
    assert torch.equal(y[0], torch.stack([x[0, 1], x[0, 1], x[0, 3]]))
    # alpha: ds = round(ds * alpha) with paddle.round (half away from zero): 1*2.5 -> 3, 3*2.5=7.5 -> 8
    y = fs2.length_regulate(x, torch.tensor([[1., 3., 0., 0.]]), alpha=2.5)
    assert y.shape[1] == 3 + 8


def test_round_is_half_away_from_zero():
    # paddle.round ties away from zero; torch.round would give [0, 2, 2]
    r = nn_ref.round_half_away(torch.tensor([0.5, 1.5, 2.5, -0.5, 2.4999]))
    assert r.tolist() == [1.0, 2.0, 3.0, -1.0, 2.0]


def test_duration_postprocessing_formula():
    # duration_predictor.py:98: clip(round(exp(x) - 1), min=0); fixed head ln(6) -> 5 frames/token
    st = syn.fastspeech2_state(fixed_duration=5)
    ids = syn.phoneme_ids(11)
    _, parts = fs2.inference(st, ids, return_parts=True)
    assert parts["d"].tolist() == [5.0] * 11
    assert parts["hs_up"].shape[0] == 55


def test_scaled_posenc_has_no_sqrt_d_and_matches_formula():
    # embedding.py:46-62,125: x + alpha * pe, pe[:,0::2]=sin(pos*div), pe[:,1::2]=cos(pos*div)
    W = nn_ref.Weights({"alpha": np.array([0.5], np.float32)})
    x = torch.zeros(1, 3, 8)
    y = fs2.scaled_posenc(W, x)[0]
    div = [math.exp(i * -(math.log(10000.0) / 8)) for i in range(0, 8, 2)]
    for pos in range(3):
        for i, dv in enumerate(div):
            assert abs(y[pos, 2 * i].item() - 0.5 * math.sin(pos * dv)) < 1e-6
            assert abs(y[pos, 2 * i + 1].item() - 0.5 * math.cos(pos * dv)) < 1e-6


def test_linear_weight_layout_is_in_out():
    w = torch.arange(6, dtype=torch.float32).reshape(2, 3)  # [in=2, out=3]
    y = nn_ref.linear(torch.tensor([[1.0, 10.0]]), w, torch.zeros(3))
    assert y.tolist() == [[30.0, 41.0, 52.0]]


def test_embedding_padding_idx_row_is_zero():
    st = syn.fastspeech2_state()
    st = dict(st)
    emb = st["encoder.embed.0.weight"].copy()
    emb[0] = 7.0  # a checkpoint may hold anything in the padding row; lookups must still give zeros
    st["encoder.embed.0.weight"] = emb
    a = fs2.inference(st, np.array([3, 0, 5]))
    emb[0] = 0.0
    st["encoder.embed.0.weight"] = emb
    b = fs2.inference(st, np.array([3, 0, 5]))
    assert torch.equal(a, b)


def test_weight_norm_fold():
    # g = ||v|| -> w == v ; scaling g scales w  (nn.utils.weight_norm(dim=0), 1-D g: test_pwg.py:131-132)
    v = np.random.default_rng(0).normal(size=(4, 3, 5)).astype(np.float32)
    g = np.sqrt((v.reshape(4, -1) ** 2).sum(1)).astype(np.float32)
    out = nn_ref.fold_weight_norm({"a.weight_v": v, "a.weight_g": g, "a.bias": np.zeros(4)})
    assert set(out) == {"a.weight", "a.bias"}
    np.testing.assert_allclose(out["a.weight"], v, rtol=1e-6)
    out2 = nn_ref.fold_weight_norm({"a.weight_v": v, "a.weight_g": 2 * g})
    np.testing.assert_allclose(out2["a.weight"], 2 * v, rtol=1e-6)


def test_pwg_shape_algebra_and_parameter_count():
    # T = (T' - 2*ctx) * hop (parakeet/datasets/vocoder_batch_fn.py:64-65); generator 1.33 M params
    st = syn.pwg_state()
    assert abs(sum(v.size for v in st.values()) - 1.33e6) < 0.02e6
    cfg = dict(layers=4, stacks=2)
    st = syn.pwg_state(dict(syn.PWG_LJSPEECH, **cfg))
    x = torch.randn(2, 1, 6 * 256)
    c = torch.randn(2, 80, 6 + 4)
    y = pwg_ref.generator_forward(st, x, c, cfg)
    assert y.shape == (2, 1, 6 * 256)
    w = pwg_ref.generator_inference(st, torch.randn(6, 80), torch.randn(6 * 256), cfg)
    assert w.shape == (6 * 256, 1)


def test_pwg_upsample_is_nearest_repeat_then_fir():
    # Stretch2D (nearest, scale 4): out[t] = in[t // 4]; Conv2D (1,9) pad (0,4) zero padding
    W = nn_ref.Weights({"upsample.up_layers.1.weight": np.ones((1, 1, 1, 9), np.float32)})
    c = torch.tensor([[[1.0, 2.0]]])  # (N=1, F=1, T=2)
    y = pwg_ref.upsample_net(W, c, [4])[0, 0]
    rep = [1, 1, 1, 1, 2, 2, 2, 2]
    want = [sum(rep[max(0, t - 4):t + 5]) for t in range(8)]
    assert y.tolist() == [float(v) for v in want]


def test_fs2_parameter_count_matches_survey():
    st = syn.fastspeech2_state(idim=80, odim=80)
    n = sum(v.size for v in st.values())
    assert abs(n - 37.1e6) < 0.3e6  # 37.1 M parameters


def test_oracle_fp32_close_to_fp64():
    st = syn.fastspeech2_state()
    ids = syn.phoneme_ids(12)
    a = fs2.inference(st, ids, dtype=torch.float32).double()
    b = fs2.inference(st, ids, dtype=torch.float64)
    assert (a - b).abs().mean().item() < 1e-5


def test_lstm_restatement_matches_torch_lstm():
    """oracle/tacotron2_ref.py's LSTMCell / bidirectional LSTM against torch.nn.LSTM with the same arrays: torch keeps
    the cuDNN conventions that paddle.nn.LSTM documents (weight_ih [4H, in], gate order i, f, g, o, outputs =
    concat(forward, backward)) -- an independent implementation of the semantics the oracle encodes."""
    import torch
    from oracle import tacotron2_ref as t2
    from oracle.nn_ref import Weights
    from parakeet_amd import synthetic as syn
    cfg = dict(syn.TACOTRON2_LJSPEECH, d_encoder=32, encoder_conv_layers=0)
    st = syn.tacotron2_state(cfg, seed=3)
    W = Weights(st, torch.float64)
    x = torch.randn(1, 7, 32, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    got = t2.encoder(W.sub("encoder."), x, 0)
    ref = torch.nn.LSTM(32, 16, batch_first=True, bidirectional=True).double()
    with torch.no_grad():
        for sfx, cell in (("", "cell_fw"), ("_reverse", "cell_bw")):
            for p in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                getattr(ref, f"{p}_l0{sfx}").copy_(torch.as_tensor(st[f"encoder.lstm.0.{cell}.{p}"]).double())
        want, _ = ref(x)
    assert np.abs(got.numpy() - want.numpy()).max() < 1e-12


def test_subsequent_mask_docstring_example():
    """fastspeech2_transformer/mask.py:30-33: subsequent_mask(3) = [[1,0,0],[1,1,0],[1,1,1]] -- the last row, which is
    the only one a cached decoding step uses (decoder_layer.py:110-120), is all ones: the step attends to the whole
    prefix, as oracle/transformer_tts_ref.py and the engine's step kernel do."""
    m = np.tril(np.ones((3, 3), dtype=bool))
    assert m.tolist() == [[True, False, False], [True, True, False], [True, True, True]]
    assert m[-1].all()


def test_ar_parameter_counts():
    """Sizes of the synthetic autoregressive models follow the recipes: TransformerTTS LJSpeech (6 + 6 blocks of 512 /
    1024, 8 heads) and Tacotron2 (examples/tacotron2/config.py) parameter counts from the layer shapes."""
    from parakeet_amd import synthetic as syn
    n_tts = sum(int(np.prod(v.shape)) for v in syn.transformer_tts_state(80, 80).values())
    # encoder: 6 x (4 x 512^2 + 2 x 512 x 1024) ~ 12.6 M, decoder: 6 x (8 x 512^2 + 2 x 512 x 1024) ~ 18.9 M, + heads
    assert 32.0e6 < n_tts < 34.5e6
    st = syn.tacotron2_state(lstm_aliases=False)
    n_t2 = sum(int(np.prod(v.shape)) for v in st.values())
    assert 27.5e6 < n_t2 < 29.5e6          # ~28 M, the size the Tacotron2 paper's architecture has
