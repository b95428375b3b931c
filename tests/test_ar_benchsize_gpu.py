"""The autoregressive acoustic models at the sizes bench.py times them at: the LJSpeech recipe
configurations unshrunk (TransformerTTS 6 + 6 layers, adim 512... as in synthetic.TRANSFORMER_TTS_LJSPEECH; Tacotron2 as in
examples/tacotron2/config.py), a ragged batch of ~128-token utterances decoded in lockstep for 200+ steps with the prenet
dropout stream on, first and last utterance against one fp64 oracle run each.  Bars: same number of frames (same stop
step), mel L1 < 1e-4 over the utterance AND on every single frame (the north star's bar, per frame so that a drift late in
the decode cannot hide in the mean).
Reference: parakeet/models/transformer_tts/transformer_tts.py:511-647, parakeet/models/tacotron2.py:474-560, 781-840."""
import numpy as np
import pytest
import torch

from oracle import tacotron2_ref as t2
from oracle import transformer_tts_ref as tt
from parakeet_amd import synthetic as syn

pytestmark = pytest.mark.gpu

TOKENS = (128, 97, 113, 121)


def _frame_l1(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape
    return np.abs(a - b).mean(), np.abs(a - b).mean(axis=-1).max()


def test_transformer_tts_ljspeech_sizes_224_steps():
    from parakeet_amd.transformer_tts import TransformerTTS
    cfg = dict(syn.TRANSFORMER_TTS_LJSPEECH)
    state = syn.transformer_tts_state(80, 80, cfg, seed=61, stop_bias=-8.0)      # stop token held off: ends at maxlen
    m = TransformerTTS(idim=80, odim=80, **cfg)
    m.set_state_dict(state)
    m.eval()
    texts = [syn.phoneme_ids(T, idim=80, seed=600 + T) for T in TOKENS]
    seeds = [7, 8, 9, 10]
    ratio = 224.5 / 129                                                          # int((T + 1) * ratio): 224, 170, 198, 212
    outs = m.inference_batch(texts, maxlenratio=ratio, seeds=seeds)
    assert [int(o[0].shape[0]) for o in outs] == [int((T + 1) * ratio) for T in TOKENS]
    for b in (0, len(texts) - 1):
        ref, rprobs, ratt = tt.inference(state, texts[b], cfg, maxlenratio=ratio, seed=seeds[b], dtype=torch.float64)
        l1, worst = _frame_l1(outs[b][0].cpu().numpy(), ref.numpy())
        assert l1 < 1e-4 and worst < 1e-4, (b, l1, worst)
        assert np.abs(outs[b][1].cpu().numpy() - rprobs.numpy()).max() < 1e-4
        assert np.abs(outs[b][2].cpu().numpy() - ratt.numpy()).max() < 1e-4


def test_tacotron2_ljspeech_sizes_256_steps():
    from parakeet_amd.tacotron2 import Tacotron2
    cfg = dict(syn.TACOTRON2_LJSPEECH)
    state = syn.tacotron2_state(cfg, seed=62, stop_bias=-8.0)
    m = Tacotron2(**cfg)
    m.set_state_dict(state)
    m.eval()
    rng = np.random.default_rng(63)
    texts = [rng.integers(1, 37, size=T) for T in TOKENS]
    seeds = [3, 4, 5, 6]
    outs = m.infer_batch(texts, max_decoder_steps=256, seeds=seeds)
    for b in (0, len(texts) - 1):
        ref = t2.infer(state, texts[b], cfg, max_decoder_steps=256, seed=seeds[b], dtype=torch.float64)
        assert outs[b]["mel_output"].shape[0] == ref["mel_output"].shape[0]       # same stop decision
        for key in ("mel_output", "mel_outputs_postnet"):
            l1, worst = _frame_l1(outs[b][key].cpu().numpy(), ref[key].numpy())
            assert l1 < 1e-4 and worst < 1e-4, (b, key, l1, worst)
        assert np.abs(outs[b]["alignments"].cpu().numpy() - ref["alignments"].numpy()).max() < 1e-4
