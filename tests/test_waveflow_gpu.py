"""GPU parity: ConditionalWaveFlow.infer (HIP through the C ABI) vs the CPU oracle (fp64)."""
import numpy as np
import pytest
import torch

from parakeet_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _run(cfg_over, frames, seed, tol=1e-5, math=None, expect_kernel=None):   # default math: measured 3e-7 (the bar was 2e-4 until round 5)
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
        names = {k for k, (n, _) in ctx.prof_dump().items() if n > 0}   # (names of earlier tests stay registered with 0 launches)
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
    _run(dict(channels=64), [5, 3], seed=2)


def test_waveflow_c128_repo_default_width():
    _run(dict(channels=128, n_flows=2), [4], seed=3)


def test_waveflow_96_mel_channels_runs_unfused():
    """the fused layer kernel needs a free condition channel (channel n_mels of its 96 carries the folded biases).
    A 96-mel model must take the unfused GEMM path -- same result bars -- and refuse the fp16-operand mode, which exists on the
    fused kernel only; an 80-mel model runs the fused kernel."""
    from parakeet_amd.waveflow import ConditionalWaveFlow
    _run(dict(channels=64, n_flows=2, n_mels=96), [4, 3], seed=6, expect_kernel=("wf_gemm_conv_gate", ("wf_layer", "wf_row")))
    _run(dict(channels=64, n_flows=2), [4, 3], seed=6, expect_kernel=(("wf_row", "wf_layer"), "wf_gemm_conv_gate"))
    cfg = dict(syn.WAVEFLOW_LJSPEECH, channels=64, n_flows=2, n_mels=96)
    m = ConditionalWaveFlow(**cfg)
    with pytest.raises(NotImplementedError):
        m.set_math("f16")


@pytest.mark.parametrize("math", [None, "f16"])
def test_waveflow_12_wave_workgroups_bit_identical(math):
    """Option "layer_waves": the 64-channel layer kernel in 12-wave workgroups (three waves per SIMD in 168 registers, one round
    of 11 tiles at the benchmark's shape instead of 8 + 3) does the arithmetic of the 8-wave kernel tile for tile -- ring depth
    and register allocation differ, not a single operation: the waveforms are equal bit for bit, in both math modes, on a ragged
    batch whose tiles straddle utterances and gaps; and the result meets the oracle bar."""
    from oracle import waveflow_ref as ref
    from parakeet_amd.waveflow import ConditionalWaveFlow
    cfg = dict(syn.WAVEFLOW_LJSPEECH, channels=64, n_flows=2)
    state = syn.waveflow_state(cfg, seed=11, weight_norm=True)
    model = ConditionalWaveFlow(**cfg)
    model.set_state_dict(state)
    model.eval()
    if math:
        model.set_math(math)
    rng = np.random.default_rng(12)
    frames = [9, 4, 6]
    mels = [np.maximum(rng.normal(-4, 2, size=(80, T)), np.log(1e-5)).astype(np.float32) for T in frames]
    zs = [rng.normal(size=(model.lengths(T)[0],)).astype(np.float32) for T in frames]
    outs = {}
    for w in (8, 12, 6):   # 6: two independent 6-wave workgroups per CU with 24 KB weight slabs
        model.set_option("layer_waves", w)
        outs[w] = [o.numpy().copy() for o in model.infer_batch(mels, zs)]
    for w in (12, 6):
        for a, b in zip(outs[8], outs[w]):
            np.testing.assert_array_equal(a, b)
    want = ref.infer(state, torch.from_numpy(mels[0])[None], torch.from_numpy(zs[0])[None], cfg, torch.float64)[0].numpy()
    err = np.abs(outs[12][0] - want).max() / np.abs(want).max()
    assert err < (2e-3 if math else 1e-5), err
    with pytest.raises(ValueError):
        model.set_option("layer_waves", 10)
    m128 = ConditionalWaveFlow(**dict(cfg, channels=128))
    with pytest.raises(NotImplementedError):
        m128.set_option("layer_waves", 12)


@pytest.mark.parametrize("channels", [64, 128])
def test_waveflow_row_kernel_variants_bit_identical(channels):
    """the residual stack of a row as ONE launch -- the eight layers behind barriers across the grid
    (csrc/pk_grid.h, a cooperative launch; option "persistent", off by default: measured slower than eight launches on the
    MI355X), the row's affine step and the next row's input projection in the epilogue of the last layer (option "fuse_step",
    on): 960 launches per batch instead of 960 + 120, or 120.  The options "persistent" and "fuse_step" only move launch
    boundaries: every combination gives the same waveform bit for bit (and the first one is checked against the oracle by the
    tests above).  Under the host emulation (PK_EMU) there is no grid barrier: "persistent" is then one launch per layer."""
    from parakeet_amd.waveflow import ConditionalWaveFlow
    cfg = dict(syn.WAVEFLOW_LJSPEECH, channels=channels, n_flows=2)
    model = ConditionalWaveFlow(**cfg)
    model.set_state_dict(syn.waveflow_state(cfg, seed=21, weight_norm=True))
    model.eval()
    rng = np.random.default_rng(22)
    frames = [7, 3, 5]
    mels = [np.maximum(rng.normal(-4, 2, size=(80, T)), np.log(1e-5)).astype(np.float32) for T in frames]
    zs = [rng.normal(size=(model.lengths(T)[0],)).astype(np.float32) for T in frames]
    ref = None
    for persistent, fuse in ((1, 1), (0, 1), (1, 0), (0, 0)):
        model.set_option("persistent", persistent)
        model.set_option("fuse_step", fuse)
        outs = [o.numpy().copy() for o in model.infer_batch(mels, zs)]
        assert all(np.isfinite(o).all() for o in outs)
        if ref is None:
            ref = outs
        else:
            for a, b in zip(ref, outs):
                np.testing.assert_array_equal(a, b)


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
