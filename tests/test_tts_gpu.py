"""TransformerTTS on the HIP engine (csrc/tts.hip) vs the golden vectors of the reference source (dropout stream
injected, tools/make_golden_ar.py) and vs the fp64 oracle, through the C ABI."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import transformer_tts_ref as tt
from parakeet_amd import synthetic as syn

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ar_cases import TTS_CASES  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _model(cfg, idim, state, math=None):
    from parakeet_amd.transformer_tts import TransformerTTS
    m = TransformerTTS(idim=idim, odim=80, **cfg)
    m.set_state_dict(state)
    m.eval()
    if math:
        m.set_math(math)
    return m


def _close(a, b, l1=1e-4, mx=2e-3):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return a.shape == b.shape and np.abs(a - b).mean() < l1 and np.abs(a - b).max() < mx


@pytest.mark.parametrize("math", ["f32", "f16x3"])
@pytest.mark.parametrize("case", [c[0] for c in TTS_CASES])
def test_engine_matches_reference_source(case, math):
    name, over, idim, T, seed, skw, kw = [c for c in TTS_CASES if c[0] == case][0]
    g = np.load(os.path.join(GOLD, "transformer_tts.npz"))
    cfg = dict(syn.TRANSFORMER_TTS_LJSPEECH, **over)
    m = _model(cfg, idim, syn.transformer_tts_state(idim, 80, cfg, seed=seed, **skw), math)
    mel, probs, att = m.inference(g[f"{name}_ids"], seed=seed,
                                  spembs=g[f"{name}_spemb"] if cfg.get("spk_embed_dim") else None,
                                  speech=g[f"{name}_speech"] if cfg.get("use_gst") else None, **kw)
    assert mel.shape == g[f"{name}_mel"].shape                      # same stop decision
    assert _close(mel.numpy(), g[f"{name}_mel"])                    # mel L1 bar of the north star
    assert np.abs(probs.numpy() - g[f"{name}_probs"]).max() < 1e-4
    assert att.shape == g[f"{name}_att"].shape
    assert np.abs(att.numpy() - g[f"{name}_att"]).max() < 1e-4


@pytest.mark.parametrize("math", ["f32", "f16x3"])
def test_engine_vs_fp64_oracle_with_taps(math):
    cfg = dict(syn.TRANSFORMER_TTS_LJSPEECH, elayers=2, dlayers=2, postnet_layers=3)
    state = syn.transformer_tts_state(50, 80, cfg, seed=77, stop_bias=-6.0)
    ids = syn.phoneme_ids(12, idim=50, seed=78)
    m = _model(cfg, 50, state, math)
    mel, probs, att = m.inference(ids, maxlenratio=2.0, seed=5)
    ref, rprobs, ratt, parts = tt.inference(state, ids, cfg, maxlenratio=2.0, seed=5, dtype=torch.float64,
                                            return_parts=True)
    assert mel.shape == ref.shape == (26, 80)
    assert np.abs(m.debug_tap(0, 0) - parts["hs"].numpy()).max() < 1e-4          # encoder output
    assert np.abs(m.debug_tap(2, 0) - parts["zs"].numpy()).max() < 2e-4          # last decoder layer, every step
    assert np.abs(m.debug_tap(1, 0) - parts["before"].numpy()).max() < 2e-4      # outs before the postnet
    assert _close(mel.numpy(), ref.numpy())
    assert np.abs(probs.numpy() - rprobs.numpy()).max() < 1e-4
    assert np.abs(att.numpy() - ratt.numpy()).max() < 1e-4
    assert np.abs(att.numpy().sum(-1) - 1.0).max() < 1e-5                        # softmax rows


def test_ragged_batch_equals_single_utterances():
    """Lockstep decoding of utterances that stop at different steps, each with its own dropout seed: every
    utterance must reproduce its single-utterance run (rows of finished utterances are ignored)."""
    cfg = dict(syn.TRANSFORMER_TTS_LJSPEECH, elayers=1, dlayers=2, postnet_layers=2)
    state = syn.transformer_tts_state(40, 80, cfg, seed=12, stop_bias=-0.7, stop_gain=2.0)
    m = _model(cfg, 40, state)
    texts = [syn.phoneme_ids(T, idim=40, seed=300 + T) for T in (6, 2, 9, 4)]
    seeds = [12, 13, 14, 15]
    outs = m.inference_batch(texts, maxlenratio=3.0, seeds=seeds)
    lens = [int(o[0].shape[0]) for o in outs]
    assert lens == [18, 9, 30, 4]                                                # stop token, maxlen, maxlen, stop token
    for t, sd, (mel, probs, att) in zip(texts, seeds, outs):
        ref, rprobs, ratt = tt.inference(state, t, cfg, maxlenratio=3.0, seed=sd, dtype=torch.float64)
        assert _close(mel.numpy(), ref.numpy())
        assert np.abs(probs.numpy() - rprobs.numpy()).max() < 1e-4
        assert np.abs(att.numpy() - ratt.numpy()).max() < 1e-4
        one = m.inference(t, maxlenratio=3.0, seed=sd)
        assert one[0].shape == mel.shape and np.abs(one[0].numpy() - mel.numpy()).max() < 1e-5


def test_dropout_switch_normalizer_and_errors():
    from parakeet_amd.normalizer import ZScore
    from parakeet_amd.transformer_tts import TransformerTTS, TransformerTTSInference
    cfg = dict(syn.TRANSFORMER_TTS_LJSPEECH, elayers=1, dlayers=1, postnet_layers=0)
    state = syn.transformer_tts_state(40, 80, cfg, seed=3, stop_bias=-6.0)
    ids = syn.phoneme_ids(5, idim=40, seed=4)
    m = _model(cfg, 40, state)
    a = m.inference(ids, maxlenratio=1.0, seed=1)[0].numpy()
    b = m.inference(ids, maxlenratio=1.0, seed=2)[0].numpy()
    assert np.abs(a - b).max() > 1e-3                                            # the mask is live
    assert np.array_equal(a, m.inference(ids, maxlenratio=1.0, seed=1)[0].numpy())   # and reproducible
    m.set_dropout(False)
    c = m.inference(ids, maxlenratio=1.0)[0].numpy()
    ref = tt.inference(state, ids, cfg, maxlenratio=1.0, drop=None, dtype=torch.float64)[0].numpy()
    assert _close(c, ref)
    m.set_dropout(True)
    mu, sigma = syn.mel_stats(seed=9)
    inf = TransformerTTSInference(ZScore(mu, sigma), m)
    lm = inf(ids, seed=1).numpy()                                                # default maxlenratio 10
    raw = m.inference(ids, seed=1)[0].numpy()
    assert np.abs(lm - (raw * sigma + mu)).max() < 1e-5
    with pytest.raises(NotImplementedError):
        TransformerTTS(idim=40, odim=80, **dict(cfg, reduction_factor=32))
    with pytest.raises(ValueError):
        TransformerTTS(idim=40, odim=80, **dict(cfg, use_gst=True, gst_conv_layers=2))   # 2 layers, 6 channel entries
    with pytest.raises(NotImplementedError):
        TransformerTTS(idim=40, odim=80, **dict(cfg, spk_embed_dim=64, spk_embed_integration_type="mul"))
    with pytest.raises(ValueError):
        m.inference(np.array([1, 2, 40]))                                        # id out of range


def test_speaker_embeddings_ragged_batch():
    """One speaker embedding per utterance of a ragged batch ("concat": a dense layer on the encoder rows plus a
    per-utterance vector); the conditioning is per call -- a model built with spk_embed_dim refuses to run without."""
    cfg = dict(syn.TRANSFORMER_TTS_LJSPEECH, elayers=1, dlayers=1, postnet_layers=2, spk_embed_dim=32,
               spk_embed_integration_type="concat")
    state = syn.transformer_tts_state(40, 80, cfg, seed=41, stop_bias=-6.0)
    m = _model(cfg, 40, state)
    rng = np.random.default_rng(42)
    texts = [syn.phoneme_ids(T, idim=40, seed=400 + T) for T in (5, 2, 7)]
    emb = rng.standard_normal((3, 32)).astype(np.float32)
    outs = m.inference_batch(texts, maxlenratio=1.0, seeds=[1, 2, 3], spembs=emb)
    for b, (t, (mel, probs, att)) in enumerate(zip(texts, outs)):
        ref, rprobs, ratt, parts = tt.inference(state, t, cfg, maxlenratio=1.0, seed=b + 1, dtype=torch.float64,
                                                spembs=emb[b], return_parts=True)
        assert np.abs(m.debug_tap(0, b) - parts["hs"].numpy()).max() < 1e-4      # encoder output after the integration
        assert _close(mel.numpy(), ref.numpy())
        assert np.abs(att.numpy() - ratt.numpy()).max() < 1e-4
    with pytest.raises(ValueError):
        m.inference_batch(texts, maxlenratio=1.0)
    with pytest.raises(ValueError):
        m.inference_batch(texts, maxlenratio=1.0, spembs=emb[:, :16])
    one = m.inference(texts[1], spembs=emb[1], maxlenratio=1.0, seed=2)
    assert np.abs(one[0].numpy() - outs[1][0].numpy()).max() < 1e-5


def test_reduction_factor_ragged_batch():
    """reduction_factor 2: two frames per decoder step, lengths counted in steps, one attention row per step; utterances
    of a ragged batch stop at different steps (stop token through either of the step's two probabilities, maxlen)."""
    cfg = dict(syn.TRANSFORMER_TTS_LJSPEECH, elayers=1, dlayers=2, postnet_layers=2, reduction_factor=2)
    state = syn.transformer_tts_state(40, 80, cfg, seed=51, stop_bias=-3.5, stop_gain=2.0)
    m = _model(cfg, 40, state)
    texts = [syn.phoneme_ids(T, idim=40, seed=500 + T) for T in (6, 3, 8)]
    seeds = [5, 6, 7]
    outs = m.inference_batch(texts, maxlenratio=3.0, seeds=seeds)
    lens = []
    for b, (t, sd, (mel, probs, att)) in enumerate(zip(texts, seeds, outs)):
        ref, rprobs, ratt, parts = tt.inference(state, t, cfg, maxlenratio=3.0, seed=sd, dtype=torch.float64, return_parts=True)
        assert mel.shape == ref.shape and mel.shape[0] % 2 == 0
        assert att.shape == ratt.shape and att.shape[2] == mel.shape[0] // 2
        assert np.abs(rprobs.numpy() - 0.5).min() > 2e-3                         # the stop decisions are not marginal
        assert np.abs(m.debug_tap(1, b) - parts["before"].numpy()).max() < 2e-4  # frames before the postnet
        assert np.abs(m.debug_tap(2, b) - parts["zs"].numpy()).max() < 2e-4      # one decoder row per step
        assert _close(mel.numpy(), ref.numpy())
        assert np.abs(probs.numpy() - rprobs.numpy()).max() < 1e-4
        assert np.abs(att.numpy() - ratt.numpy()).max() < 1e-4
        lens.append(int(mel.shape[0]))
    assert lens == [20, 12, 2]                                                   # maxlen, maxlen, stop token at step 1


def test_style_tokens_ragged_batch():
    """use_gst: reference spectrograms of different lengths (1 .. 3 GRU steps after the stride-2 conv stack, one of them a
    single frame), one per utterance; a two-layer GRU; the conditioning is per call."""
    cfg = dict(syn.TRANSFORMER_TTS_LJSPEECH, elayers=1, dlayers=1, postnet_layers=0, use_gst=True, gst_tokens=7, gst_heads=4,
               gst_conv_layers=4, gst_conv_chans_list=(8, 8, 16, 16), gst_gru_layers=2, gst_gru_units=32)
    state = syn.transformer_tts_state(40, 80, cfg, seed=61, stop_bias=-6.0)
    m = _model(cfg, 40, state)
    rng = np.random.default_rng(62)
    texts = [syn.phoneme_ids(T, idim=40, seed=600 + T) for T in (4, 6, 3)]
    refs = [rng.standard_normal((L, 80)).astype(np.float32) for L in (33, 1, 17)]
    outs = m.inference_batch(texts, maxlenratio=1.0, seeds=[1, 2, 3], speech=refs)
    for b, (t, (mel, probs, att)) in enumerate(zip(texts, outs)):
        ref, rprobs, ratt, parts = tt.inference(state, t, cfg, maxlenratio=1.0, seed=b + 1, dtype=torch.float64,
                                                speech=refs[b], return_parts=True)
        assert np.abs(m.debug_tap(0, b) - parts["hs"].numpy()).max() < 1e-4      # encoder output + style embedding
        assert _close(mel.numpy(), ref.numpy())
    with pytest.raises(ValueError):
        m.inference_batch(texts, maxlenratio=1.0)                                # a use_gst model needs its references
    one = m.inference(texts[2], speech=refs[2], maxlenratio=1.0, seed=3)
    assert np.abs(one[0].numpy() - outs[2][0].numpy()).max() < 1e-5


@pytest.mark.parametrize("variant", ["prenorm", "postnorm_concat", "linear_input"])
def test_prefix_overlap_is_bit_identical(variant):
    """Option "overlap_prefix" (default on): the next step's prefix work -- prenet with fresh dropout, input layer, layer 0's
    q | k | v of the row blocks that already exist -- runs on a side stream, into a second set of buffers, under the current step's
    layer chain.  Every kernel involved works row by row with per-row operand scales, so splitting the prefix into "old blocks
    early, new block now" must give the sequential path's spectrogram, stop probabilities and attention weights bit for bit --
    for pre-norm, post-norm / concat_after blocks and the "linear" decoder input layer, on a ragged batch."""
    over = {"prenorm": {}, "postnorm_concat": dict(decoder_normalize_before=False, decoder_concat_after=True),
            "linear_input": dict(dprenet_layers=0)}[variant]
    cfg = dict(syn.TRANSFORMER_TTS_LJSPEECH, elayers=1, dlayers=2, postnet_layers=2, **over)
    state = syn.transformer_tts_state(40, 80, cfg, seed=81, stop_bias=-0.7, stop_gain=2.0)
    m = _model(cfg, 40, state)
    texts = [syn.phoneme_ids(T, idim=40, seed=800 + T) for T in (6, 2, 5)]
    seeds = [3, 4, 5]
    m.set_option("overlap_prefix", 1)
    a = m.inference_batch(texts, maxlenratio=3.0, seeds=seeds)
    again = m.inference_batch(texts, maxlenratio=3.0, seeds=seeds)     # (buffers of the second set are reused across calls)
    m.set_option("overlap_prefix", 0)
    b = m.inference_batch(texts, maxlenratio=3.0, seeds=seeds)
    for (ma, pa, wa), (mb, pb, wb), (mc, pc, wc) in zip(a, b, again):
        np.testing.assert_array_equal(ma.numpy(), mb.numpy())
        np.testing.assert_array_equal(pa.numpy(), pb.numpy())
        np.testing.assert_array_equal(wa.numpy(), wb.numpy())
        np.testing.assert_array_equal(ma.numpy(), mc.numpy())


def test_kv_only_prefix_projection_experiment():
    """Option "kv_prefix" = 1 (pk_tts_set_option; off by default): layer 0 projects k | v only for the prefix rows and q for the new rows with a
    row GEMM; same result as the fused q | k | v projection up to the two GEMM kernels' rounding."""
    cfg = dict(syn.TRANSFORMER_TTS_LJSPEECH, elayers=1, dlayers=2, postnet_layers=2)
    state = syn.transformer_tts_state(40, 80, cfg, seed=71, stop_bias=-6.0)
    m = _model(cfg, 40, state)
    texts = [syn.phoneme_ids(T, idim=40, seed=700 + T) for T in (5, 3)]
    base = m.inference_batch(texts, maxlenratio=2.0, seeds=[1, 2])
    m.set_option("kv_prefix", 1)
    alt = m.inference_batch(texts, maxlenratio=2.0, seeds=[1, 2])
    for (a, pa, wa), (b, pb, wb) in zip(base, alt):
        assert a.shape == b.shape and np.abs(a.numpy() - b.numpy()).max() < 2e-4
        assert np.abs(wa.numpy() - wb.numpy()).max() < 1e-5
    ref = tt.inference(state, texts[0], cfg, maxlenratio=2.0, seed=1, dtype=torch.float64)[0].numpy()
    assert _close(alt[0][0].numpy(), ref)


def test_query_projection_inside_the_source_attention():
    """Option "fuse_src_q" (default on; 64-wide heads, <= 128 memory rows): norm2 + linear_q of the encoder-decoder attention are
    computed inside the step's attention kernel instead of a row GEMM of their own.  Same spectrogram, stop probabilities and
    attention weights up to summation order, on a ragged batch; the fused path is the one that ran."""
    from parakeet_amd.runtime import Context
    cfg = dict(syn.TRANSFORMER_TTS_LJSPEECH, elayers=1, dlayers=2, postnet_layers=2)
    state = syn.transformer_tts_state(40, 80, cfg, seed=91, stop_bias=-6.0)
    m = _model(cfg, 40, state)
    texts = [syn.phoneme_ids(T, idim=40, seed=900 + T) for T in (7, 3, 5)]
    ctx = Context.get()
    ctx.prof_enable(True)
    ctx.prof_reset()
    fused = m.inference_batch(texts, maxlenratio=2.0, seeds=[1, 2, 3])
    names = {k for k, (n, _) in ctx.prof_dump().items() if n > 0}
    ctx.prof_enable(False)
    assert "tts_attn_src_q" in names and "tts_row_src_q" not in names and "tts_row_feat_out_stop" in names
    assert "tts_prenet_embed" in names and "tts_row_prenet" not in names     # (option "fuse_prenet": one launch for the new rows' prenet + input layer)
    m.set_option("fuse_src_q", 0)
    m.set_option("fuse_prenet", 0)
    plain = m.inference_batch(texts, maxlenratio=2.0, seeds=[1, 2, 3])
    for (a, pa, wa), (b, pb, wb) in zip(fused, plain):
        assert a.shape == b.shape and np.abs(a.numpy() - b.numpy()).max() < 2e-4
        assert np.abs(pa.numpy() - pb.numpy()).max() < 1e-5
        assert np.abs(wa.numpy() - wb.numpy()).max() < 1e-5
    ref = tt.inference(state, texts[0], cfg, maxlenratio=2.0, seed=1, dtype=torch.float64)[0].numpy()
    assert _close(fused[0][0].numpy(), ref)
