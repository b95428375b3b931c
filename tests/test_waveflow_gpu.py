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


def test_waveflow_12_wave_workgroups_bit_identical():
    """Option "layer_waves": the 64-channel layer kernel also runs in 12-wave workgroups (three waves per SIMD in 168 registers, one
    round of 11 tiles at the benchmark's shape instead of 8 + 3) and as two 6-wave workgroups per CU -- the arithmetic of the 8-wave
    kernel tile for tile: the waveforms are equal bit for bit on a ragged batch whose tiles straddle utterances and gaps, in both
    maths, and meet the oracle bar.  (Round 5 refused three waves per SIMD in the default math: wrong tiles in 7 - 25 % of the calls
    at the benchmark's shape.  Round 6 found the instruction -- DESIGN 4.3, "the op_sel rule" -- and every configuration is
    back; the calls that failed are in test_waveflow_calls_with_two_working_waves_per_simd_are_deterministic_and_right.)"""
    from oracle import waveflow_ref as ref
    from parakeet_amd.waveflow import ConditionalWaveFlow
    cfg = dict(syn.WAVEFLOW_LJSPEECH, channels=64, n_flows=2)
    state = syn.waveflow_state(cfg, seed=11, weight_norm=True)
    model = ConditionalWaveFlow(**cfg)
    model.set_state_dict(state)
    model.eval()
    rng = np.random.default_rng(12)
    frames = [9, 4, 6]
    mels = [np.maximum(rng.normal(-4, 2, size=(80, T)), np.log(1e-5)).astype(np.float32) for T in frames]
    zs = [rng.normal(size=(model.lengths(T)[0],)).astype(np.float32) for T in frames]
    outs = {}
    for math in ("f16x3", "f16"):
        model.set_math(math)
        for w in (8, 12, 6, 0):
            model.set_option("layer_waves", w)
            outs[(math, w)] = [o.numpy().copy() for o in model.infer_batch(mels, zs)]
        for w in (12, 6, 0):
            for a, b in zip(outs[(math, 8)], outs[(math, w)]):
                np.testing.assert_array_equal(a, b)
    outs = {8: outs[("f16", 8)], 12: outs[("f16", 12)]}
    for a, b in zip(outs[8], outs[12]):
        np.testing.assert_array_equal(a, b)
    want = ref.infer(state, torch.from_numpy(mels[0])[None], torch.from_numpy(zs[0])[None], cfg, torch.float64)[0].numpy()
    err = np.abs(outs[12][0] - want).max() / np.abs(want).max()
    assert err < 2e-3, err
    model.set_math("f16x3")
    model.set_option("layer_waves", 12)
    err = np.abs(model.infer_batch(mels, zs)[0].numpy() - want).max() / np.abs(want).max()
    assert err < 1e-5, err
    with pytest.raises(ValueError):
        model.set_option("layer_waves", 10)
    m128 = ConditionalWaveFlow(**dict(cfg, channels=128))
    with pytest.raises(NotImplementedError):
        m128.set_option("layer_waves", 12)


@pytest.mark.parametrize("channels", [64, 128])
def test_waveflow_row_kernel_variants_bit_identical(channels):
    """The row's affine step and the next row's input projection in the epilogue of the last layer (option "fuse_step", on): 960
    launches per batch instead of 960 + 120.  The option only moves a launch boundary: the same waveform bit for bit.  (SURVEY
    K20, the residual stack of a row as ONE cooperative launch behind grid barriers -- option "persistent" -- was built in round 4,
    measured slower than eight launches and left off; round 5 found it non-deterministic at sizes beyond these tests' and took it
    out of the product: the profile build keeps it.)"""
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
    with pytest.raises(NotImplementedError):      # round 5: not in the product (not deterministic beyond small sizes, HISTORY 9.9)
        model.set_option("persistent", 1)
    for persistent, fuse in ((0, 1), (0, 0)):
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


@pytest.mark.parametrize("channels", [64, 128])
def test_waveflow_large_call_equals_small_calls(channels):
    """Sizes well beyond the benchmark's (8 x 640 frames): 24 ragged utterances of 500 - 2 000 frames in ONE call (about 7.5 M
    samples, 470 k positions per row) against the same utterances two at a time.  An utterance's position in the packed row decides
    where the 32-position scale blocks fall inside it, so the two results differ in the last bits (both fp32-equivalent: 3e-7
    against fp64) -- a wrapped offset or a tile reading across a gap would show as an error of order one."""
    from parakeet_amd.waveflow import ConditionalWaveFlow
    cfg = dict(syn.WAVEFLOW_LJSPEECH, channels=channels)
    model = ConditionalWaveFlow(**cfg)
    model.set_state_dict(syn.waveflow_state(cfg, seed=77, weight_norm=True))
    model.eval()
    rng = np.random.default_rng(78)
    frames = [int(t) for t in rng.integers(500, 2001, size=24)]
    mels = [np.maximum(rng.normal(-4, 2, size=(cfg["n_mels"], T)), np.log(1e-5)).astype(np.float32) for T in frames]
    zs = [rng.normal(size=(model.lengths(T)[0],)).astype(np.float32) for T in frames]
    big = [o.numpy() for o in model.infer_batch(mels, zs)]
    assert all(np.isfinite(o).all() for o in big)
    for pair in ((0, 1), (11, 12), (22, 23)):
        small = model.infer_batch([mels[b] for b in pair], [zs[b] for b in pair])
        for o, b in zip(small, pair):
            assert big[b].shape == (model.lengths(frames[b])[1],)
            err = np.abs(o.numpy() - big[b]).max() / np.abs(big[b]).max()
            assert err < 2e-6, f"utterance {b}: the large call differs from the small one by {err}"


@pytest.mark.parametrize("channels,math,frames", [
    (64, None, [1200, 1200]),        # 5 tiles per workgroup: the smallest call in which two waves of a SIMD both work
    (64, None, [1950, 1950]),        # 8: every wave of the 8-wave kernel works
    (64, None, [2560, 2560]),        # 11 tiles on 12 waves (the benchmark's workgroup shape)
    (64, "f16", [1950, 1950]),
    (128, None, [1200, 1200]),
    (64, None, [640] * 8),           # BASELINE config 5 itself
    (64, "f16", [640] * 8),
])
def test_waveflow_calls_with_two_working_waves_per_simd_are_deterministic_and_right(channels, math, frames):
    """Round 5 (HISTORY 9.9): with more than four tiles per workgroup two waves of a SIMD run matrix instructions side by side, and
    the layer kernel's out projection read an accumulator one issue slot too early now and then -- a few tiles per call off by
    1e-3, different on every run.  None of the earlier tests had such a shape (2 x 640 frames: three tiles per workgroup).  Every
    shape here runs three times (bit-identical) and against the exact-fp32 unfused path (another kernel family altogether)."""
    from parakeet_amd.waveflow import ConditionalWaveFlow
    cfg = dict(syn.WAVEFLOW_LJSPEECH, channels=channels)
    state = syn.waveflow_state(cfg, seed=77, weight_norm=True)

    def make(m):
        model = ConditionalWaveFlow(**cfg)
        model.set_state_dict(state)
        model.eval()
        if m:
            model.set_math(m)
        return model
    model, exact = make(math), make("f32")
    rng = np.random.default_rng(78)
    mels = [np.maximum(rng.normal(-4, 2, size=(cfg["n_mels"], T)), np.log(1e-5)).astype(np.float32) for T in frames]
    zs = [rng.normal(size=(model.lengths(T)[0],)).astype(np.float32) for T in frames]
    want = np.concatenate([o.numpy() for o in exact.infer_batch(mels, zs)])
    runs = [np.concatenate([o.numpy() for o in model.infer_batch(mels, zs)]) for _ in range(3)]
    assert np.array_equal(runs[0], runs[1]) and np.array_equal(runs[0], runs[2]), "two runs of the same call differ"
    err = np.abs(runs[0] - want).max() / np.abs(want).max()
    assert err < (2e-3 if math == "f16" else 2e-6), f"differs from the exact-fp32 path by {err}"
    if frames == [640] * 8:
        # VERDICT r5 "weak" #2: BASELINE config 5 against the ORACLE itself (fp64 restatement of waveflow.py:785-805), not only
        # against the engine's other kernel family: the first and the last utterance of the 8 x 640 call
        from oracle import waveflow_ref as ref
        o = 0
        for b, T in enumerate(frames):
            n = model.lengths(T)[1]
            if b in (0, 7):
                with torch.no_grad():
                    w64 = ref.infer(state, torch.from_numpy(mels[b])[None], torch.from_numpy(zs[b])[None], cfg, torch.float64)[0].numpy()
                e = np.abs(runs[0][o:o + n] - w64).max() / np.abs(w64).max()
                assert e < (2e-3 if math == "f16" else 5e-6), f"utterance {b} of the 8 x 640 call differs from the fp64 oracle by {e}"
            o += n


def test_waveflow_recipe_with_three_imports_swapped(tmp_path):
    """VERDICT r5 "missing" #2 / "next" #6: the body of examples/waveflow/synthesize.py:29-44 with its three imports swapped --
    `ConditionalWaveFlow.from_pretrained(config, checkpoint_path)` on the class (waveflow.py:827-852),
    `layer_tools.recursively_remove_weight_norm(model)` (utils/layer_tools.py:40-46), `model.eval()`, then per mel file
    `with amp.auto_cast(): audio = model.predict(mel)` -- on a weight-normalised checkpoint in the released layout
    (`<path>.pdparams`, yacs-style config with attribute access).  Under auto_cast the call runs with fp16 operands (the
    reference's AMP precision) and returns to the default math afterwards."""
    import pickle
    from oracle import waveflow_ref as ref
    # --- the recipe's three imports, swapped:
    from parakeet_amd.waveflow import ConditionalWaveFlow          # parakeet.models.waveflow
    from parakeet_amd.utils import layer_tools                     # parakeet.utils
    from parakeet_amd import amp                                   # paddle.amp

    class Node(dict):                                              # what yacs' CfgNode offers the recipe: items as attributes
        __getattr__ = dict.__getitem__
    wcfg = dict(syn.WAVEFLOW_LJSPEECH, channels=64, n_flows=4)
    config = Node(data=Node(n_mels=80, sample_rate=22050),
                  model=Node({k: v for k, v in wcfg.items() if k != "n_mels"}))
    state = syn.waveflow_state(wcfg, seed=5, weight_norm=True)
    assert any(k.endswith("weight_g") for k in state)
    with open(tmp_path / "step-2000000.pdparams", "wb") as f:
        pickle.dump(dict(state), f, protocol=2)
    mel_dir, output_dir = tmp_path / "mels", tmp_path / "out"
    mel_dir.mkdir()
    rng = np.random.default_rng(11)
    for i, T in enumerate((9, 14)):
        np.save(mel_dir / f"LJ00{i}.npy", np.maximum(rng.normal(-4, 2, size=(80, T)), np.log(1e-5)).astype(np.float32))

    # --- synthesize.py:29-44
    model = ConditionalWaveFlow.from_pretrained(config, str(tmp_path / "step-2000000"))
    assert model.training
    layer_tools.recursively_remove_weight_norm(model)
    model.eval()
    output_dir.mkdir(parents=True, exist_ok=True)
    model.set_seed(1234)
    for file_path in sorted(mel_dir.glob("*.npy")):
        mel = np.load(str(file_path))
        with amp.auto_cast():
            audio = model.predict(mel)
        assert isinstance(audio, np.ndarray) and audio.shape == (model.lengths(mel.shape[1])[1],) and np.isfinite(audio).all()
        # the same call with the latent given: fp16 operands inside auto_cast, the default math outside
        z = rng.normal(size=(model.lengths(mel.shape[1])[0],)).astype(np.float32)
        with torch.no_grad():
            want = ref.infer(state, torch.from_numpy(mel)[None], torch.from_numpy(z)[None], wcfg, torch.float64)[0].numpy()
        with amp.auto_cast():
            a16 = model.predict(mel, z)
        a32 = model.predict(mel, z)
        e16 = np.abs(a16 - want).max() / np.abs(want).max()
        e32 = np.abs(a32 - want).max() / np.abs(want).max()
        assert e32 < 1e-5 < e16 < 2e-3, (e32, e16)


@pytest.mark.parametrize("channels,frames,runs", [(64, [640] * 8, 40), (64, [2560, 2560], 30), (128, [640] * 8, 30)])
def test_waveflow_repeat_run_gate(channels, frames, runs):
    """ADVICE r5 / VERDICT r5 #2: the gate every rebuild of the layer kernel has to pass -- the shapes in which round 5's defect lived
    (8 x 640: 11 tiles on 12 waves; 2 x 2560; 128 channels with two working waves per SIMD), `runs` calls each, every one compared
    with the exact-fp32 unfused path (another kernel family) and with the first.  Before round 6's fix 7 - 25 % of these calls were
    wrong (1 - 5 % at 128 channels); a three-run test passes such a kernel 60 % of the time, 40 runs 1 % of the time."""
    from parakeet_amd.waveflow import ConditionalWaveFlow
    cfg = dict(syn.WAVEFLOW_LJSPEECH, channels=channels)
    state = syn.waveflow_state(cfg, seed=77, weight_norm=True)

    def make(m):
        model = ConditionalWaveFlow(**cfg)
        model.set_state_dict(state)
        model.eval()
        if m:
            model.set_math(m)
        return model
    model, exact = make(None), make("f32")
    rng = np.random.default_rng(78)
    mels = [np.maximum(rng.normal(-4, 2, size=(80, T)), np.log(1e-5)).astype(np.float32) for T in frames]
    zs = [rng.normal(size=(model.lengths(T)[0],)).astype(np.float32) for T in frames]
    want = np.concatenate([o.numpy() for o in exact.infer_batch(mels, zs)])
    peak = np.abs(want).max()
    first = None
    for r in range(runs):
        got = np.concatenate([o.numpy() for o in model.infer_batch(mels, zs)])
        if first is None:
            first = got
            assert np.abs(got - want).max() / peak < 2e-6
        else:
            assert np.array_equal(got, first), f"run {r} differs from run 0 in {int((got != first).sum())} samples"
