"""SpeedySpeech on the HIP engine vs the golden vectors of the reference source and vs the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import speedyspeech_ref as ssr
from parakeet_amd import synthetic as syn

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _model(quirk, seed, **kw):
    from parakeet_amd.speedyspeech import SpeedySpeech
    m = SpeedySpeech(vocab_size=70, tone_size=7, same_padding_resets_dilation=quirk, **syn.SPEEDYSPEECH_BAKER, **kw)
    m.set_state_dict(syn.speedyspeech_state(seed=seed))
    m.eval()
    return m


@pytest.mark.parametrize("tag,quirk", [("rd", True), ("dil", False)])
def test_engine_matches_reference_source(tag, quirk):
    from parakeet_amd.normalizer import ZScore
    from parakeet_amd.speedyspeech import SpeedySpeechInference
    g = np.load(os.path.join(GOLD, "speedyspeech_baker.npz"))
    if f"{tag}_mel0" not in g.files:
        pytest.skip("a file made by real Paddle holds only the reading Paddle implements")
    m = _model(quirk, int(g["seed"]))
    for i in range(3):
        mel = m.inference(g[f"{tag}_text{i}"], g[f"{tag}_tones{i}"]).numpy()
        assert mel.shape == g[f"{tag}_mel{i}"].shape                      # integer durations identical
        assert np.abs(mel - g[f"{tag}_mel{i}"]).mean() < 1e-4             # mel L1 bar of the north star
        assert np.abs(mel - g[f"{tag}_mel{i}"]).max() < 2e-3
    nt = m.inference(g[f"{tag}_text1"]).numpy()                           # tones=None (:183)
    assert nt.shape == g[f"{tag}_notone_mel"].shape and np.abs(nt - g[f"{tag}_notone_mel"]).mean() < 1e-4
    # one ragged batch == the single-utterance references (zero gaps isolate the utterances, incl. d = 27)
    outs = m.inference_batch([g[f"{tag}_text{i}"] for i in range(3)], [g[f"{tag}_tones{i}"] for i in range(3)])
    for i, o in enumerate(outs):
        assert o.shape == g[f"{tag}_mel{i}"].shape
        assert np.abs(o.numpy() - g[f"{tag}_mel{i}"]).mean() < 1e-4
    inf = SpeedySpeechInference(ZScore(g["mu"], g["sigma"]), m)
    assert np.abs(inf(g[f"{tag}_text0"], g[f"{tag}_tones0"]).numpy() - g[f"{tag}_logmel0"]).mean() < 1e-4


@pytest.mark.parametrize("mode", ["f32", "f16x3"])
def test_engine_vs_oracle_taps_and_math_modes(mode):
    m = _model(True, 909)
    m.set_math(mode)
    state = syn.speedyspeech_state(seed=909)
    rng = np.random.default_rng(3)
    texts = [rng.integers(1, 70, size=T) for T in (1, 7, 33)]
    tones = [rng.integers(0, 7, size=len(t)) for t in texts]              # tone 0 = padding row
    frames = m.encode_batch(texts, tones)
    for b, (tx, tn) in enumerate(zip(texts, tones)):
        ref, parts = ssr.inference(state, tx, tn, dtype=torch.float64, return_parts=True)
        enc = m.debug_tap(0, b)
        assert np.abs(enc - parts["enc"].numpy()).max() < 1e-4
        assert np.abs(m.debug_tap(1, b) - parts["pred"].numpy()).max() < 1e-4
        e = np.exp(parts["pred"].numpy())
        if np.abs(e - np.floor(e) - 0.5).min() > 1e-3:                     # away from rounding ties
            assert np.array_equal(m.debug_tap(2, b).astype(np.int64), parts["durs"].numpy())
            assert frames[b] == int(parts["durs"].sum())
    outs = m.inference_batch(texts, tones)
    for b, (tx, tn) in enumerate(zip(texts, tones)):
        ref = ssr.inference(state, tx, tn, dtype=torch.float64).numpy()
        if ref.shape == tuple(outs[b].shape):
            assert np.abs(outs[b].numpy() - ref).mean() < 1e-4


def test_error_mapping_and_shapes():
    from parakeet_amd.speedyspeech import SpeedySpeech
    with pytest.raises(AssertionError):      # widths must agree (PK_ESHAPE)
        SpeedySpeech(70, 128, 3, [1], 64, 128, 80, 3, [1])
    m = _model(True, 1)
    with pytest.raises(ValueError):
        m.inference(np.array([1, 2, 70]))    # id out of range
    with pytest.raises(ValueError):
        m.inference(np.array([1, 2]), np.array([1, 9]))


def test_predictor_style_pipeline_speedyspeech_pwg():
    """The loop body of examples/speedyspeech/baker/inference.py:100-126 over the engine."""
    from parakeet_amd.normalizer import ZScore
    from parakeet_amd.parallel_wavegan import PWGGenerator, PWGInference
    from parakeet_amd.predictor import create_predictor
    from parakeet_amd.speedyspeech import SpeedySpeechInference
    m = _model(True, 5)
    am = create_predictor(SpeedySpeechInference(ZScore(*syn.mel_stats(seed=1)), m), ["phones", "tones"])
    gen = PWGGenerator(**syn.PWG_LJSPEECH)
    gen.set_state_dict(syn.pwg_state())
    gen.eval()
    gen.set_seed(3)
    voc = create_predictor(PWGInference(ZScore(*syn.mel_stats(seed=2)), gen), ["logmel"])
    phones, tones = np.arange(1, 8), np.array([1, 2, 3, 4, 5, 6, 1])
    for name, v in zip(am.get_input_names(), (phones, tones)):
        h = am.get_input_handle(name)
        h.reshape(v.shape)
        h.copy_from_cpu(v)
    am.run()
    mel = am.get_output_handle(am.get_output_names()[0]).copy_to_cpu()
    assert mel.ndim == 2 and mel.shape[1] == 80 and mel.shape[0] > 0
    mh = voc.get_input_handle(voc.get_input_names()[0])
    mh.reshape(mel.shape)
    mh.copy_from_cpu(mel)
    voc.run()
    wav = voc.get_output_handle(voc.get_output_names()[0]).copy_to_cpu()
    assert wav.shape == (mel.shape[0] * 256, 1) and np.isfinite(wav).all()


def test_speedyspeech_to_baker_pwg_end_to_end():
    """examples/speedyspeech/baker/synthesize_e2e.py:113-131 as one ragged batch: SpeedySpeechInference (baker
    configuration) into the baker vocoder, upsample_scales [4, 5, 3, 5] = hop 300, mel staying in HBM -- against
    the two CPU oracles chained the same way."""
    from oracle import pwg_ref
    from parakeet_amd.normalizer import ZScore
    from parakeet_amd.parallel_wavegan import PWGGenerator, PWGInference
    from parakeet_amd.speedyspeech import SpeedySpeechInference
    from parakeet_amd.synthesize import Synthesizer
    m = _model(True, 5)
    mu_a, sg_a = syn.mel_stats(seed=1)
    mu_v, sg_v = syn.mel_stats(seed=2)
    pcfg = dict(syn.PWG_LJSPEECH, upsample_scales=[4, 5, 3, 5])
    pstate = syn.pwg_state(pcfg, seed=9)
    gen = PWGGenerator(**pcfg)
    gen.set_state_dict(pstate)
    gen.remove_weight_norm()
    gen.eval()
    synth = Synthesizer(SpeedySpeechInference(ZScore(mu_a, sg_a), m), PWGInference(ZScore(mu_v, sg_v), gen))
    rng = np.random.default_rng(2)
    lens = [9, 4, 13]
    phones = [rng.integers(1, 70, size=T) for T in lens]
    tones = [rng.integers(1, 7, size=T) for T in lens]
    mels = m.inference_batch(phones, tones, denormalize=True)          # (acoustic-model parity: the tests above)
    frames = [int(x.shape[0]) for x in mels]
    noises = [rng.normal(size=(L * 300,)).astype(np.float32) for L in frames]
    wavs = synth.synthesize_batch(phones, noises=noises, tones=tones)
    ocfg = {k: pcfg[k] for k in ("layers", "stacks", "kernel_size", "aux_context_window", "upsample_scales")}
    for b, L in enumerate(frames):
        assert wavs[b].shape == (L * 300, 1)
        ref = pwg_ref.pwg_inference(pstate, mu_v, sg_v, torch.from_numpy(mels[b].numpy()), torch.from_numpy(noises[b]),
                                    ocfg, torch.float64).numpy()
        got = wavs[b].numpy()
        err = np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30)
        assert err < 1e-4, f"utt {b}: wav rel err {err}"
