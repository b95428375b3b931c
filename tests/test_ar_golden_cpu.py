"""Autoregressive acoustic models: the oracle against golden vectors produced by the reference's
own Python source over the paddle stand-in with the dropout stream injected (tools/make_golden_ar.py).  CPU only."""
import os
import sys

import numpy as np

from oracle import philox_ref
from oracle import transformer_tts_ref as tt
from parakeet_amd import synthetic as syn

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ar_cases import TTS_CASES  # noqa: E402

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
        mel, probs, att = tt.inference(state, g[f"{name}_ids"], cfg, seed=seed, **kw)
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
