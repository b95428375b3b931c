"""CPU checks of the STFT / mel oracle: against numpy's FFT, against mel-scale identities, and the
product-side filterbank builder against the oracle's (two independent writings of librosa's algorithm)."""
import numpy as np
import torch

from oracle import audio_ref


def test_oracle_stft_matches_numpy_rfft():
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, size=(2, 4096)).astype(np.float32)
    re, im = audio_ref.stft(torch.from_numpy(x), n_fft=1024, hop_length=256, dtype=torch.float64)
    assert re.shape == (2, 513, 1 + 4096 // 256)          # frames = 1 + T // hop (audio.py:103-105)
    xp = np.pad(x.astype(np.float64), ((0, 0), (512, 512)), mode="reflect")
    win = audio_ref.window_padded("hann", 1024, 1024)
    for f in (0, 3, 16):
        spec = np.fft.rfft(xp[:, f * 256:f * 256 + 1024] * win, axis=-1)
        np.testing.assert_allclose(re[:, :, f].numpy(), spec.real, atol=1e-9)
        np.testing.assert_allclose(im[:, :, f].numpy(), spec.imag, atol=1e-9)


def test_mel_scale_identities():
    # Slaney scale: linear below 1 kHz (200/3 Hz per mel), log above; 1 kHz <-> 15 mel
    assert abs(audio_ref.hz_to_mel(1000.0) - 15.0) < 1e-12
    assert abs(audio_ref.mel_to_hz(15.0) - 1000.0) < 1e-9
    assert abs(audio_ref.hz_to_mel(500.0) - 7.5) < 1e-12
    f = np.array([80.0, 440.0, 1000.0, 4000.0, 7600.0])
    np.testing.assert_allclose(audio_ref.mel_to_hz(audio_ref.hz_to_mel(f)), f, rtol=1e-12)
    fb = audio_ref.mel_filterbank(22050, 1024, 80, 80, 7600)
    assert fb.shape == (80, 513) and fb.dtype == np.float32
    assert (fb >= 0).all() and (fb.sum(1) > 0).all()
    # slaney norm: each triangle integrates to ~1 over frequency (bin width sr/n_fft)
    area = fb.sum(1) * (22050 / 1024)
    assert np.all(np.abs(area[10:] - 1.0) < 0.1)
    # filters outside [fmin, fmax] are zero
    freqs = np.linspace(0, 22050 / 2, 513)
    assert fb[:, freqs < 80 - 1e-9].sum() == 0 and fb[:, freqs > 7600 + 1e-9].sum() == 0


def test_product_filterbank_equals_oracle():
    from parakeet_amd.audio import mel_filterbank
    for args in [(22050, 1024, 80, 80, 7600), (24000, 2048, 80, 80, 7600), (22050, 1024, 80, 0, 8000),
                 (16000, 512, 40, 0, None)]:
        a = mel_filterbank(*args)
        b = audio_ref.mel_filterbank(*args)
        assert a.shape == b.shape
        np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-9)


def test_write_wav_roundtrip(tmp_path):
    import wave
    from parakeet_amd.audio import write_wav
    rng = np.random.default_rng(0)
    x = np.clip(rng.normal(scale=0.4, size=2205), -1.5, 1.5).astype(np.float32)
    p = tmp_path / "a.wav"
    write_wav(p, x.reshape(-1, 1), 22050)
    with wave.open(str(p), "rb") as w:
        assert (w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()) == (1, 2, 22050, 2205)
        pcm = np.frombuffer(w.readframes(2205), dtype="<i2")
    assert np.array_equal(pcm, np.rint(np.clip(x, -1, 1) * 32767).astype(np.int16))
    write_wav(tmp_path / "f.wav", x, 22050, subtype="FLOAT")
    raw = (tmp_path / "f.wav").read_bytes()
    assert raw[:4] == b"RIFF" and raw[8:12] == b"WAVE" and len(raw) == 44 + 4 * 2205
    assert np.array_equal(np.frombuffer(raw[44:], dtype="<f4"), x)


def test_spec_normalizers_match_reference_source():
    """LogMagnitude / UnitMagnitude (parakeet/audio/spec_normalizer.py:39-75): the reference file itself is numpy-only, so it is
    executed here when /root/reference is present; its formulas are also checked through their defining properties."""
    import importlib.util
    import os
    from parakeet_amd.audio import LogMagnitude, UnitMagnitude
    rng = np.random.default_rng(3)
    x = np.abs(rng.normal(size=(80, 50))) * 10.0 ** rng.uniform(-7, 2, size=(80, 50))
    lm, um = LogMagnitude(), UnitMagnitude()
    assert np.array_equal(lm.transform(x), np.log(np.maximum(x, 1e-5)))
    assert np.allclose(lm.inverse(lm.transform(x)), np.maximum(x, 1e-5))
    u = um.transform(x)
    assert u.min() >= 0 and u.max() <= 1
    assert np.allclose(um.transform(np.array([1e-4, 10.0 ** (-4 + 100 / 20)])), [0.0, 1.0])   # -80 dB - 20 -> 0; +20 dB - 20 -> 1
    inside = (u > 0) & (u < 1)
    assert np.allclose(um.inverse(u)[inside], x[inside], rtol=1e-10)
    path = "/root/reference/parakeet/audio/spec_normalizer.py"
    if os.path.exists(path):
        spec = importlib.util.spec_from_file_location("ref_spec_normalizer", path)
        ref = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref)
        for mine, theirs in ((lm, ref.LogMagnitude()), (um, ref.UnitMagnitude())):
            assert np.array_equal(mine.transform(x), theirs.transform(x))
            assert np.array_equal(mine.inverse(mine.transform(x)), theirs.inverse(theirs.transform(x)))
