"""Tacotron2 on the HIP engine (csrc/taco2.hip) vs the golden vectors of the reference source (dropout stream injected,
tools/make_golden_ar.py) and vs the fp64 oracle, through the C ABI."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import tacotron2_ref as t2
from parakeet_amd import synthetic as syn

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ar_cases import T2_CASES  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KEYS = ("mel_output", "mel_outputs_postnet", "alignments", "stop_logits")


def _model(cfg, state, math=None):
    from parakeet_amd.tacotron2 import Tacotron2
    kw = {k: v for k, v in cfg.items()}
    m = Tacotron2(**kw)
    m.set_state_dict(state)
    m.eval()
    if math:
        m.set_math(math)
    return m


def _check(got, ref, name, stop_tol=1e-3):
    for k in KEYS:
        if k not in ref:
            assert k not in got
            continue
        a, b = np.asarray(got[k], np.float64), np.asarray(ref[k], np.float64)
        assert a.shape == b.shape, (name, k, a.shape, b.shape)               # same stop decision
        if k == "alignments":
            assert np.abs(a - b).max() < 1e-4, (name, k)
        elif k == "stop_logits":
            assert np.abs(a - b).max() < stop_tol, (name, k)
        else:
            assert np.abs(a - b).mean() < 1e-4 and np.abs(a - b).max() < 2e-3, (name, k)   # mel L1 bar


@pytest.mark.parametrize("math", ["f32", "f16x3"])
@pytest.mark.parametrize("case", [c[0] for c in T2_CASES])
def test_engine_matches_reference_source(case, math):
    name, over, T, seed, skw, max_steps = [c for c in T2_CASES if c[0] == case][0]
    g = np.load(os.path.join(GOLD, "tacotron2.npz"))
    cfg = dict(syn.TACOTRON2_LJSPEECH, **over)
    m = _model(cfg, syn.tacotron2_state(cfg, seed=seed, **skw), math)
    o = m.infer(g[f"{name}_ids"][None, :], max_decoder_steps=max_steps,
                tones=g[f"{name}_tones"][None, :] if cfg["n_tones"] else None, seed=seed,
                global_condition=g[f"{name}_global_condition"][None, :] if cfg.get("d_global_condition") else None)
    got = {k: v.numpy()[0] for k, v in o.items()}
    ref = {k: g[f"{name}_{k}"] for k in KEYS if f"{name}_{k}" in g.files}
    _check(got, ref, name, stop_tol=0.05 if name == "stop" else 1e-3)       # that stop head has a gain of 500


@pytest.mark.parametrize("math", ["f32", "f16x3"])
def test_engine_vs_fp64_oracle_ragged_batch(math):
    """Lockstep decoding of utterances of different lengths that end at different steps for different reasons, each
    with its own dropout seed, against one oracle run per utterance; aliased LSTM parameter names only."""
    over = dict(T2_CASES[0][1])
    cfg = dict(syn.TACOTRON2_LJSPEECH, **over)
    state = syn.tacotron2_state(cfg, seed=21, stop_bias=-31.4, stop_gain=500.0)
    aliased = {k: v for k, v in state.items() if ".lstm.0." not in k}        # "encoder.lstm.weight_ih_l0" ... only
    m = _model(cfg, aliased, math)
    texts, seeds = [np.load(os.path.join(GOLD, "tacotron2.npz"))["stop_ids"]], [21]
    for k in (7, 20, 14):                                                    # picked for their distance from the threshold
        rng = np.random.default_rng(100 + k)
        texts.append(rng.integers(1, 37, size=int(rng.integers(2, 20))))
        seeds.append(k)
    outs = m.infer_batch(texts, max_decoder_steps=40, seeds=seeds)
    lens = []
    for b, (t, sd, o) in enumerate(zip(texts, seeds, outs)):
        ref = t2.infer(state, t, cfg, max_decoder_steps=40, seed=sd, dtype=torch.float64, return_parts=True)
        enc = ref.pop("encoder_outputs").numpy()
        assert np.abs(m.debug_tap(0, b) - enc).max() < 1e-4                  # conv stack + bidirectional LSTM
        _check({k: v.numpy() for k, v in o.items()}, {k: v.numpy() for k, v in ref.items()}, f"utt{b}", stop_tol=0.05)
        assert np.abs(o["alignments"].numpy().sum(-1) - 1.0).max() < 1e-5
        lens.append(int(o["mel_output"].shape[0]))
    assert lens == [29, 2, 40, 40]                                          # stop token twice, max_decoder_steps twice


def test_dropout_switch_and_errors():
    from parakeet_amd.tacotron2 import Tacotron2
    over = dict(T2_CASES[2][1])
    cfg = dict(syn.TACOTRON2_LJSPEECH, **over)
    state = syn.tacotron2_state(cfg, seed=5, stop_bias=-8.0)
    m = _model(cfg, state)
    ids = np.arange(1, 8)
    a = m.infer(ids, max_decoder_steps=6, seed=1)["mel_output"].numpy()
    b = m.infer(ids, max_decoder_steps=6, seed=2)["mel_output"].numpy()
    assert a.shape == (1, 6, 80) and np.abs(a - b).max() > 1e-3              # the mask is live
    assert np.array_equal(a, m.infer(ids, max_decoder_steps=6, seed=1)["mel_output"].numpy())
    m.set_dropout(False)
    c = m.infer(ids, max_decoder_steps=6)["mel_outputs_postnet"].numpy()[0]
    ref = t2.infer(state, ids, cfg, max_decoder_steps=6, drop=None, dtype=torch.float64)["mel_outputs_postnet"].numpy()
    assert np.abs(c - ref).mean() < 1e-4 and np.abs(c - ref).max() < 2e-3
    with pytest.raises(NotImplementedError):
        Tacotron2(**dict(cfg, reduction_factor=2))
    with pytest.raises(NotImplementedError):
        Tacotron2(**dict(cfg, d_global_condition=24))                        # multiples of 16 only
    with pytest.raises(ValueError):
        m.infer(np.array([1, 2, 37]))                                        # id out of range
    with pytest.raises(ValueError):
        m.infer(np.ones((2, 5), dtype=np.int64))                             # one utterance per infer() call


def test_global_condition_ragged_batch():
    """d_global_condition: every utterance of a ragged batch gets its own vector (:816-821); the conditioning is per
    call -- a model built with it refuses to run without, and a second call does not reuse the first call's rows."""
    over = dict(T2_CASES[-1][1])
    assert over["d_global_condition"] == 32
    cfg = dict(syn.TACOTRON2_LJSPEECH, **over)
    state = syn.tacotron2_state(cfg, seed=31, stop_bias=-8.0)
    m = _model(cfg, state)
    rng = np.random.default_rng(32)
    texts = [rng.integers(1, 37, size=n) for n in (5, 11, 3)]
    gc = rng.standard_normal((3, 32)).astype(np.float32)
    outs = m.infer_batch(texts, max_decoder_steps=7, seeds=[1, 2, 3], global_condition=gc)
    for b, (t, o) in enumerate(zip(texts, outs)):
        ref = t2.infer(state, t, cfg, max_decoder_steps=7, seed=b + 1, dtype=torch.float64, global_condition=gc[b],
                       return_parts=True)
        enc = ref.pop("encoder_outputs").numpy()
        assert np.abs(m.debug_tap(0, b) - enc).max() < 1e-4                  # the tap stays d_encoder wide
        _check({k: v.numpy() for k, v in o.items()}, {k: v.numpy() for k, v in ref.items()}, f"utt{b}")
    with pytest.raises(ValueError):
        m.infer_batch(texts, max_decoder_steps=7)                            # no condition for a model that needs one
    with pytest.raises(ValueError):
        m.infer_batch(texts, max_decoder_steps=7, global_condition=gc[:, :16])
    plain = _model(dict(syn.TACOTRON2_LJSPEECH, **T2_CASES[2][1]), syn.tacotron2_state(dict(syn.TACOTRON2_LJSPEECH, **T2_CASES[2][1]), seed=5))
    with pytest.raises(ValueError):
        plain.infer(np.arange(1, 6), max_decoder_steps=3, global_condition=gc[:1])
