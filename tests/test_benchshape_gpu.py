"""Oracle parity at the shapes bench.py times.

The small-shape parity tests (<= 28 frames per batch) never let a wave of the persistent PWG layer kernel
enter its second loop iteration: 256 workgroups x 8 waves = 2048 wave-tiles = 256 frames per sweep, so the
cross-tile software pipeline (ring continuation into the next tile, prefetch_head(next_wt)) that produces
every sample of the benchmark only runs for batches of more than 256 frames.  These tests compare that
steady state -- and the tile variants WaveFlow / SpeedySpeech / FastSpeech2 pick for big grids -- with the
fp64 CPU oracle on full utterances (one 640-frame utterance costs the oracle about 7 s).
"""
import math

import numpy as np
import pytest
import torch

from parakeet_amd import synthetic as syn

pytestmark = pytest.mark.gpu

# Two bars everywhere: the north star's (mel L1 < 1e-4 against the reference) and a REGRESSION bar at about ten times the error
# the engine actually delivers (mel L1 1e-6, waveform 1.3e-6 of the peak, WaveFlow 3e-7: profiles/r03_wf_error.txt, bench.py's
# parity_check), so that a 100x numerical regression cannot stay green.
MEL_L1_NORTH_STAR = 1e-4
MEL_L1_BAR = 1e-5


def _rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.parametrize("mode", ["f16x3", "f32"])
def test_pwg_steady_state_tile_loop_vs_oracle(mode):
    """3 x 640 frames = 15 360 wave-tiles on 2048 wave slots: every wave loops 7-8 times.  The middle
    utterance (tiles handled in sweeps 3..5, by waves that have both a previous and a next tile) and the
    last one (ends in a partial sweep) are compared with the fp64 oracle, internal taps included."""
    from oracle import pwg_ref
    from parakeet_amd.parallel_wavegan import PWGGenerator
    state = syn.pwg_state(seed=21)
    rng = np.random.default_rng(22)
    frames = [640, 640, 640]
    mels = [rng.normal(size=(L, 80)).astype(np.float32) for L in frames]
    noises = [rng.normal(size=(L * 256,)).astype(np.float32) for L in frames]
    gen = PWGGenerator(**syn.PWG_LJSPEECH)
    gen.set_state_dict(state)
    gen.remove_weight_norm()
    gen.eval()
    gen.set_math(mode)
    outs = gen.inference_batch(mels, noises)
    cfg = syn.PWG_LJSPEECH
    ocfg = {k: cfg[k] for k in ("layers", "stacks", "kernel_size", "aux_context_window", "upsample_scales")}
    for b in (1, 2):
        c = torch.from_numpy(mels[b]).transpose(0, 1).unsqueeze(0)
        c = torch.nn.functional.pad(c, (cfg["aux_context_window"],) * 2, mode="replicate")
        x = torch.from_numpy(noises[b]).reshape(1, 1, -1)
        with torch.no_grad():
            ref, parts = pwg_ref.generator_forward(state, x, c, ocfg, torch.float64, return_parts=True)
        assert _rel_err(gen.debug_tap(1, b), parts["x_last"][0].numpy()) < 1e-4
        skips = gen.debug_tap(2, b) * math.sqrt(1.0 / cfg["layers"])
        assert _rel_err(skips, parts["skips"][0].numpy()) < 1e-4
        err = _rel_err(outs[b].numpy()[:, 0], ref[0, 0].numpy())
        assert err < 1e-4, f"{mode} utt {b}: wav rel err {err}"
        if mode == "f16x3":
            assert err < 5e-6, f"split-fp16 is meant to be fp32-equivalent, got {err}"


def test_e2e_batch32_bench_shape_vs_oracle():
    """bench.py's step: 32 utterances x 128 tokens -> 640 frames -> 163 840 samples each, FastSpeech2 ->
    PWG with the mel staying in HBM.  Utterances 0, 17 and 31 vs the fp64 oracle (durations exact, mel L1
    < 1e-4 -- the north star's bar -- wav rel. max < 1e-4)."""
    from oracle import fastspeech2_ref, pwg_ref
    from parakeet_amd.fastspeech2 import FastSpeech2, FastSpeech2Inference
    from parakeet_amd.normalizer import ZScore
    from parakeet_amd.parallel_wavegan import PWGGenerator, PWGInference
    from parakeet_amd.synthesize import Synthesizer
    fs2_state = syn.fastspeech2_state(80, 80, fixed_duration=5)
    pwg_state = syn.pwg_state()
    am = FastSpeech2(80, 80, **syn.FS2_LJSPEECH)
    am.set_state_dict(fs2_state)
    am.eval()
    voc = PWGGenerator(**syn.PWG_LJSPEECH)
    voc.set_state_dict(pwg_state)
    voc.remove_weight_norm()
    voc.eval()
    mu_f, sg_f = syn.mel_stats(seed=7)
    mu_p, sg_p = syn.mel_stats(seed=8)
    synth = Synthesizer(FastSpeech2Inference(ZScore(mu_f, sg_f), am), PWGInference(ZScore(mu_p, sg_p), voc))
    texts = [syn.phoneme_ids(128, seed=10086 + i) for i in range(32)]
    per = 128 * 5 * 256
    noise = torch.randn(32 * per, device="cuda", generator=torch.Generator(device="cuda").manual_seed(42))
    wav, frames = synth.synthesize_packed(texts, noise=noise)
    mel = am.decode_packed(denormalize=True)   # the de-normalised log-mel of the same call (FastSpeech2Inference domain)
    assert frames.tolist() == [640] * 32
    wav = wav.cpu().numpy()
    mel = mel.as_subclass(torch.Tensor).cpu().numpy().reshape(32, 640, 80)
    for b in (0, 17, 31):
        nz = noise[b * per:(b + 1) * per].cpu()
        with torch.no_grad():
            logmel = fastspeech2_ref.fastspeech2_inference(fs2_state, mu_f, sg_f, texts[b], dtype=torch.float64)
            want = pwg_ref.pwg_inference(pwg_state, mu_p, sg_p, logmel, nz, dtype=torch.float64)[:, 0].numpy()
        assert logmel.shape == (640, 80)
        l1 = float(np.abs(mel[b] - logmel.numpy()).mean())
        assert l1 < MEL_L1_NORTH_STAR and l1 < MEL_L1_BAR, f"utt {b}: mel L1 {l1}"
        err = _rel_err(wav[b * per:(b + 1) * per], want)
        assert err < 1e-4 and err < 1e-5, f"utt {b}: wav rel err {err}"      # (measured 1.3e-6)


def test_e2e_global_batch_256_in_one_call():
    """BASELINE config 4's whole global batch on ONE GPU in ONE engine call (the N = 1 point of the strong-scaling curve with
    --minibatch 256): 256 ragged utterances (64-128 tokens, 5 frames per token: 31 M samples, a 10 GB working set).  Sizes this
    large are where 32-bit offsets would wrap: utterances from the start, the middle and the very end of the packed buffers
    against the same utterances synthesised three at a time (an utterance's result does not depend on its batch: mel bit for
    bit), and the last one against the fp64 oracle."""
    from oracle import fastspeech2_ref, pwg_ref
    from parakeet_amd.fastspeech2 import FastSpeech2, FastSpeech2Inference
    from parakeet_amd.normalizer import ZScore
    from parakeet_amd.parallel_wavegan import PWGGenerator, PWGInference
    from parakeet_amd.synthesize import Synthesizer
    fs2_state = syn.fastspeech2_state(80, 80, fixed_duration=5)
    pwg_state = syn.pwg_state()
    am = FastSpeech2(80, 80, **syn.FS2_LJSPEECH)
    am.set_state_dict(fs2_state)
    am.eval()
    voc = PWGGenerator(**syn.PWG_LJSPEECH)
    voc.set_state_dict(pwg_state)
    voc.remove_weight_norm()
    voc.eval()
    mu_f, sg_f = syn.mel_stats(seed=7)
    mu_p, sg_p = syn.mel_stats(seed=8)
    synth = Synthesizer(FastSpeech2Inference(ZScore(mu_f, sg_f), am), PWGInference(ZScore(mu_p, sg_p), voc))
    rng = np.random.default_rng(256)
    tokens = [int(t) for t in rng.integers(64, 129, size=256)]
    tokens[-1] = 128
    texts = [syn.phoneme_ids(t, seed=20000 + i) for i, t in enumerate(tokens)]
    samples = [t * 5 * 256 for t in tokens]
    offs = np.concatenate([[0], np.cumsum(samples)])
    noise = torch.randn(int(offs[-1]), device="cuda", generator=torch.Generator(device="cuda").manual_seed(4))
    wav, frames = synth.synthesize_packed(texts, noise=noise)
    assert frames.tolist() == [5 * t for t in tokens]
    mel = am.decode_packed(denormalize=True).as_subclass(torch.Tensor).cpu().numpy()
    wav = wav.cpu().numpy()
    assert wav.shape == (int(offs[-1]),) and np.isfinite(wav).all()
    foffs = np.concatenate([[0], np.cumsum([5 * t for t in tokens])])
    pick = [0, 127, 255]
    small_noise = torch.cat([noise[offs[b]:offs[b + 1]] for b in pick])
    wav3, frames3 = synth.synthesize_packed([texts[b] for b in pick], noise=small_noise)
    mel3 = am.decode_packed(denormalize=True).as_subclass(torch.Tensor).cpu().numpy()
    wav3 = wav3.cpu().numpy()
    o3 = np.concatenate([[0], np.cumsum([samples[b] for b in pick])])
    f3 = np.concatenate([[0], np.cumsum([5 * tokens[b] for b in pick])])
    for i, b in enumerate(pick):
        assert np.array_equal(mel[foffs[b]:foffs[b + 1]], mel3[f3[i]:f3[i + 1]]), f"utt {b}: mel depends on the batch"
        err = _rel_err(wav[offs[b]:offs[b + 1]], wav3[o3[i]:o3[i + 1]])
        assert err < 2e-6, f"utt {b}: waveform differs from the three-utterance call by {err}"
    b = 255
    with torch.no_grad():
        logmel = fastspeech2_ref.fastspeech2_inference(fs2_state, mu_f, sg_f, texts[b], dtype=torch.float64)
        want = pwg_ref.pwg_inference(pwg_state, mu_p, sg_p, logmel, noise[offs[b]:offs[b + 1]].cpu(), dtype=torch.float64)[:, 0].numpy()
    l1 = float(np.abs(mel[foffs[b]:foffs[b + 1]] - logmel.numpy()).mean())
    assert l1 < MEL_L1_BAR, f"utt {b}: mel L1 {l1}"
    err = _rel_err(wav[offs[b]:offs[b + 1]], want)
    assert err < 1e-5, f"utt {b}: wav rel err {err}"


def test_fs2_batch16_ragged_vs_oracle():
    """BASELINE config 2's parity shape: 16 ragged utterances (T in [37, 128]) with a random duration head;
    every utterance against its own oracle call (the existing bookkeeping test only checks shapes)."""
    from oracle import fastspeech2_ref as ref
    from parakeet_amd.fastspeech2 import FastSpeech2
    st = syn.fastspeech2_state(80, 80, seed=5)
    am = FastSpeech2(80, 80, **syn.FS2_LJSPEECH)
    am.set_state_dict(st)
    am.eval()
    rng = np.random.default_rng(0)
    lens = rng.integers(37, 129, size=16).tolist()
    texts = [syn.phoneme_ids(T, seed=200 + i) for i, T in enumerate(lens)]
    outs = am.inference_batch(texts)
    worst = 0.0
    for b in range(16):
        with torch.no_grad():
            want = ref.inference(st, texts[b], dtype=torch.float64).numpy()
        got = outs[b].numpy()
        assert got.shape == want.shape, f"utt {b}: durations differ"
        worst = max(worst, float(np.abs(got - want).mean()))
        assert np.abs(got - want).max() < 2e-3
    assert worst < MEL_L1_NORTH_STAR and worst < MEL_L1_BAR, worst


def _waveflow_bench_shape(channels, frames, maths, seed=31, check=(0,)):
    """maths: {math mode or None: bar}; the fp64 oracle runs once per checked utterance, every mode is compared with it."""
    from oracle import waveflow_ref as ref
    from parakeet_amd.waveflow import ConditionalWaveFlow
    cfg = dict(syn.WAVEFLOW_LJSPEECH, channels=channels)
    state = syn.waveflow_state(cfg, seed=seed, weight_norm=True)
    model = ConditionalWaveFlow(**cfg)
    model.set_state_dict(state)
    model.eval()
    rng = np.random.default_rng(seed + 1)
    mels = [np.maximum(rng.normal(-4, 2, size=(80, T)), np.log(1e-5)).astype(np.float32) for T in frames]
    zs = [rng.normal(size=(model.lengths(T)[0],)).astype(np.float32) for T in frames]
    want = {}
    for b in check:
        with torch.no_grad():
            want[b] = ref.infer(state, torch.from_numpy(mels[b])[None], torch.from_numpy(zs[b])[None], cfg,
                                torch.float64)[0].numpy()
    for math, tol in maths.items():
        model.set_math(math or "f16x3")
        outs = model.infer_batch(mels, zs)
        for b in check:
            got = outs[b].numpy()
            assert got.shape == want[b].shape
            err = _rel_err(got, want[b])
            assert err < tol, f"{channels} channels, math {math}, utt {b}: rel err {err}"


def test_waveflow_bench_shape_vs_oracle():
    """BASELINE config 5's utterance shape (640 mel frames, C = 64, all 8 flows): 2 x 640 frames -- full rounds of the layer
    kernel's wave tiles, utterance boundaries inside a workgroup -- vs the fp64 oracle, both utterances; the default math at
    its bar (5e-6: ten times the measured error), the fp16-operand mode (the reference's AMP precision) at its own: 2e-3 of the
    peak over 8 flows x 15 rows of feedback."""
    _waveflow_bench_shape(64, [640, 640], {None: 5e-6, "f16": 2e-3}, check=(0, 1))      # measured 3.3e-7 / 1.3e-4


def test_waveflow_c128_bench_shape_vs_oracle():
    """The 128-channel model (the reference repository's default width, examples/waveflow/config.py:32-41) at a 640-frame
    utterance, default math and fp16 operands -- the other two WaveFlow configurations bench.py times."""
    _waveflow_bench_shape(128, [640], {None: 5e-6, "f16": 2e-3}, seed=41)               # measured 2.8e-7 / 9.1e-5


def test_speedyspeech_batch32_vs_oracle():
    """bench.py's SpeedySpeech extra: 32 x 128 phones with tones; utterances 0, 15, 31 vs the fp64 oracle."""
    from oracle import speedyspeech_ref as ssr
    from parakeet_amd.speedyspeech import SpeedySpeech
    state = syn.speedyspeech_state(seed=77)
    m = SpeedySpeech(vocab_size=70, tone_size=7, **syn.SPEEDYSPEECH_BAKER)
    m.set_state_dict(state)
    m.eval()
    rng = np.random.default_rng(0)
    ph = [rng.integers(1, 70, size=128) for _ in range(32)]
    tn = [rng.integers(1, 7, size=128) for _ in range(32)]
    outs = m.inference_batch(ph, tn)
    for b in (0, 15, 31):
        with torch.no_grad():
            want = ssr.inference(state, ph[b], tn[b], dtype=torch.float64).numpy()
        got = outs[b].numpy()
        assert got.shape == want.shape, f"utt {b}: durations differ"
        assert np.abs(got - want).mean() < MEL_L1_BAR
