"""GPU parity: ConditionalWaveFlow.infer (HIP through the C ABI) vs the CPU oracle (fp64)."""
import numpy as np
import pytest
import torch

from parakeet_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _run(cfg_over, frames, seed, tol=2e-4, math=None, expect_kernel=None):
    from oracle import waveflow_ref as ref
    from parakeet_amd.waveflow import ConditionalWaveFlow
    cfg = dict(syn.WAVEFLOW_LJSPEECH, **cfg_over)
    state = syn.waveflow_state(cfg, seed=seed, weight_norm=True)
    model = ConditionalWaveFlow(**cfg)
    model.set_state_dict(state)
    model.eval()
    if math:
        model.set_math(math)
    rng = np.random.default_rng(seed + 1)
    mels = [np.maximum(rng.normal(-4, 2, size=(cfg["n_mels"], T)), np.log(1e-5)).astype(np.float32) for T in frames]
    zs = [rng.normal(size=(model.lengths(T)[0],)).astype(np.float32) for T in frames]
    if expect_kernel:
        from parakeet_amd.runtime import Context
        ctx = Context.get()
        ctx.prof_enable(True)
        ctx.prof_reset()
    outs = model.infer_batch(mels, zs)
    if expect_kernel:
        names = set(ctx.prof_dump().keys())
        ctx.prof_enable(False)
        present, absent = expect_kernel
        assert any(n.startswith(present) for n in names) and not any(n.startswith(absent) for n in names), names
    for b, T in enumerate(frames):
        want = ref.infer(state, torch.from_numpy(mels[b])[None], torch.from_numpy(zs[b])[None], cfg,
                         torch.float64)[0].numpy()
        got = outs[b].numpy()
        assert got.shape == want.shape == (model.lengths(T)[1],)
        err = np.abs(got - want).max() / (np.abs(want).max() + 1e-30)
        assert err < tol, f"utt {b}: rel err {err}"


def test_waveflow_c64_two_flows_ragged():
    _run(dict(channels=64, n_flows=2), [4, 7, 3], seed=1)


def test_waveflow_c64_all_flows():
    # all 8 flows: both permutation kinds and their cumulative effect on the condition
    _run(dict(channels=64), [5, 3], seed=2, tol=1e-3)


def test_waveflow_c128_repo_default_width():
    _run(dict(channels=128, n_flows=2), [4], seed=3)


def test_waveflow_96_mel_channels_runs_unfused():
    """the fused layer kernel needs a free condition channel (channel n_mels of its 96 carries the folded biases).
    A 96-mel model must take the unfused GEMM path -- same result bars -- and refuse the fp16-operand mode, which exists on the
    fused kernel only; an 80-mel model runs the fused kernel."""
    from parakeet_amd.waveflow import ConditionalWaveFlow
    _run(dict(channels=64, n_flows=2, n_mels=96), [4, 3], seed=6, expect_kernel=("wf_gemm_conv_gate", "wf_layer"))
    _run(dict(channels=64, n_flows=2), [4, 3], seed=6, expect_kernel=("wf_layer", "wf_gemm_conv_gate"))
    cfg = dict(syn.WAVEFLOW_LJSPEECH, channels=64, n_flows=2, n_mels=96)
    m = ConditionalWaveFlow(**cfg)
    with pytest.raises(NotImplementedError):
        m.set_math("f16")


def test_waveflow_fp16_operand_mode():
    """set_math("f16"): the reference's own inference precision for this model (examples/waveflow/synthesize.py:40 runs under
    paddle.amp.auto_cast) -- fp16 conv operands, fp32 accumulation, the residual stream kept at 22 bits.  Not the default and
    not fp32-equivalent: the bar is 2e-3 of the waveform's peak against the fp64 oracle (measured: 5e-5 .. 8e-5 with two
    flows), ten times the split-fp16 default's bar."""
    _run(dict(channels=64, n_flows=2), [4, 7, 3], seed=1, tol=2e-3, math="f16")
    _run(dict(channels=64), [5, 3], seed=2, tol=2e-3, math="f16")
    _run(dict(channels=128, n_flows=2), [4], seed=3, tol=2e-3, math="f16")


def test_waveflow_infer_api_and_errors():
    from parakeet_amd.waveflow import ConditionalWaveFlow
    cfg = dict(syn.WAVEFLOW_LJSPEECH, channels=64, n_flows=2)
    with pytest.raises(ValueError):
        ConditionalWaveFlow(**dict(cfg, n_flows=3))     # odd flows (waveflow.py:586-589)
    m = ConditionalWaveFlow(**cfg)
    m.set_state_dict(syn.waveflow_state(cfg, seed=5))
    m.eval()
    mel = np.random.default_rng(0).normal(-4, 1, size=(2, 80, 4)).astype(np.float32)
    y = m.infer(mel)
    assert tuple(y.shape) == (2, m.lengths(4)[1])
    assert np.isfinite(y.numpy()).all()
    w = m.predict(mel[0])
    assert w.shape == (m.lengths(4)[1],)
