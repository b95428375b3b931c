"""GPU parity: STFT / magnitude / log-mel (HIP through the C ABI) vs the CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_stft_magnitude_matches_oracle():
    from oracle import audio_ref
    from parakeet_amd.audio import STFT
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, size=(3, 46080 // 4)).astype(np.float32)   # tests/unit/test_stft.py uses U(-1,1), n_fft 1024, hop 256
    st = STFT(1024, 256)
    re, im = st(x)
    rr, ri = audio_ref.stft(torch.from_numpy(x), n_fft=1024, hop_length=256, dtype=torch.float64)
    assert tuple(re.shape) == tuple(rr.shape)
    scale = float(rr.abs().max())
    assert np.abs(re.numpy() - rr.numpy()).max() < 2e-5 * scale
    assert np.abs(im.numpy() - ri.numpy()).max() < 2e-5 * scale
    mag = st.magnitude(x).numpy()
    want = audio_ref.magnitude(torch.from_numpy(x), n_fft=1024, hop_length=256, dtype=torch.float64).numpy()
    assert np.abs(mag - want).max() < 2e-5 * scale
    pw = st.power(x).numpy()
    assert np.abs(pw - want ** 2).max() < 4e-5 * scale ** 2


def test_log_mel_ragged_batch_matches_oracle():
    from oracle import audio_ref
    from parakeet_amd.audio import LogMelFBank
    rng = np.random.default_rng(2)
    wavs = [rng.uniform(-0.5, 0.5, size=n).astype(np.float32) for n in (5000, 22050, 1024, 777)]
    fb = LogMelFBank(sr=22050, n_fft=1024, hop_length=256, n_mels=80, fmin=80, fmax=7600)
    outs = fb.get_log_mel_fbank_batch(wavs)
    for w, o in zip(wavs, outs):
        want = audio_ref.log_mel(torch.from_numpy(w)[None], 22050, 1024, 256, 80, 80, 7600, dtype=torch.float64)[0]
        assert tuple(o.shape) == tuple(want.shape) == (1 + len(w) // 256, 80)
        assert np.abs(o.numpy() - want.numpy()).max() < 1e-4     # log10 domain
    single = fb.get_log_mel_fbank(wavs[1]).numpy()
    np.testing.assert_array_equal(single, outs[1].numpy())
    nat = fb.get_log_mel_fbank(wavs[0], base="e").numpy()
    assert np.abs(nat - outs[0].numpy() * np.log(10.0)).max() < 3e-4


def test_mel_l1_metric_on_synthesised_audio():
    # the acceptance metric of the path (mel L1) computed on-device: identical wav -> 0, scaled wav -> > 0
    from parakeet_amd.audio import LogMelFBank
    rng = np.random.default_rng(3)
    w = rng.normal(size=16384).astype(np.float32) * 0.1
    fb = LogMelFBank(sr=22050, n_fft=1024, hop_length=256, n_mels=80, fmin=80, fmax=7600)
    a = fb.get_log_mel_fbank(w).numpy()
    b = fb.get_log_mel_fbank(w * 2.0).numpy()
    assert np.abs(a - fb.get_log_mel_fbank(w).numpy()).mean() == 0.0
    assert abs(np.abs(b - a).mean() - np.log10(2.0)) < 1e-3


def test_stft_function_of_stft_loss():
    """parakeet.modules.stft_loss.stft (:20-67): clipped magnitude, (B, frames, bins)."""
    from parakeet_amd.audio import stft
    rng = np.random.default_rng(4)
    x = rng.normal(scale=0.3, size=(2, 4096)).astype(np.float32)
    x[1, 1000:3000] = 0.0                                            # silence: exercises the 1e-7 floor
    got = stft(x, 512, 128, 512, "hann").cpu().numpy()
    w = np.asarray(__import__("scipy.signal", fromlist=["get_window"]).get_window("hann", 512, fftbins=True))
    xp = np.pad(x.astype(np.float64), ((0, 0), (256, 256)), mode="reflect")
    frames = 1 + (xp.shape[1] - 512) // 128
    ref = np.empty((2, frames, 257))
    for f in range(frames):
        spec = np.fft.rfft(xp[:, f * 128:f * 128 + 512] * w, axis=1)
        ref[:, f] = np.sqrt(np.clip(spec.real ** 2 + spec.imag ** 2, 1e-7, None))
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < 2e-4 * max(1.0, np.abs(ref).max())
    assert got.min() >= np.sqrt(np.float32(1e-7)) * 0.999


def test_audio_processor_and_log_magnitude(tmp_path):
    """AudioProcessor.spectrogram / mel_spectrogram (audio/audio.py:95-102) + LogMagnitude (the WaveFlow feature
    domain, examples/waveflow/preprocess.py:57-87) against the oracle's librosa-algorithm restatement."""
    from oracle import audio_ref
    from parakeet_amd.audio import AudioProcessor, LogMagnitude
    rng = np.random.default_rng(8)
    wav = (0.3 * rng.normal(size=6000)).astype(np.float32)
    p = AudioProcessor(sample_rate=22050, n_fft=1024, win_length=1024, hop_length=256, n_mels=80, fmin=0, fmax=8000)
    S = p.spectrogram(wav)
    M = p.mel_spectrogram(wav)
    re_im = audio_ref.stft(torch.from_numpy(wav)[None], 1024, 256, 1024, "hann")
    ref_S = torch.sqrt(re_im[0] ** 2 + re_im[1] ** 2)[0].numpy()
    assert S.shape == ref_S.shape == (513, 1 + 6000 // 256)
    assert np.abs(S - ref_S).max() < 2e-4 * ref_S.max()
    basis = audio_ref.mel_filterbank(22050, 1024, 80, 0, 8000)
    ref_M = basis @ ref_S
    assert np.abs(M - ref_M).max() < 2e-4 * ref_M.max()
    norm = LogMagnitude(1e-5)
    logm = norm.transform(M)
    assert np.allclose(norm.inverse(logm), np.maximum(M, 1e-5), rtol=1e-6)
    p.write_wav(tmp_path / "x.wav", wav)
    back = AudioProcessor(22050, 1024, 1024, 256, normalize=False).read_wav(tmp_path / "x.wav")
    assert np.abs(back - np.clip(wav, -1, 1)).max() < 2.0 / 32768      # written x 32767, read / 32768, + rounding


def test_melscale_runs_on_the_engine_and_matches_fp64():
    """MelScale.forward (modules/audio.py:226-229) = matmul(basis, spectrogram): through pk_op_matmul, not torch."""
    from parakeet_amd.audio import STFT, MelScale
    rng = np.random.default_rng(6)
    x = rng.uniform(-1, 1, size=(2, 9000)).astype(np.float32)
    st = STFT(1024, 256)
    mag = st.magnitude(x)                                     # (2, 513, frames)
    ms = MelScale(22050, 1024, 80, 80, 7600)
    got = ms(mag).numpy()
    want = np.einsum("mf,bft->bmt", ms.weight.numpy().astype(np.float64), mag.numpy().astype(np.float64))
    assert got.shape == want.shape == (2, 80, mag.shape[-1])
    assert np.abs(got - want).max() < 2e-6 * np.abs(want).max()
    one = ms(mag[0]).numpy()                                  # un-batched (n_freq, frames) input
    np.testing.assert_array_equal(one, got[0])
