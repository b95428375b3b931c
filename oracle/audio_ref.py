"""Oracle: STFT / mel / log features (test infrastructure).

Follows parakeet/modules/audio.py STFT :74-215 (DFT-as-conv1d on the reflect-padded signal,
weight = np.fft.fft(np.eye(n_fft))[:n_bin] * window), MelScale :218-229, and the host feature
extractor parakeet/data/get_feats.py LogMelFBank :20-88 (log10(clip(mel_basis @ |STFT|, 1e-10))).

Third-party dependency not in /root/reference: ``librosa`` (listed unpinned in setup.py:53-81; absent
here).  ``mel_filterbank`` restates librosa.filters.mel's published algorithm (Slaney mel scale,
htk=False, norm='slaney', float32 output) -- parity for it is anchored on the reference's call sites
(audio.py:221, get_feats.py:49-55) and on scale identities checked in tests/test_audio_cpu.py.
"""
import numpy as np
import scipy.signal
import torch
import torch.nn.functional as F


def hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    fmax = sr / 2.0 if fmax is None else fmax
    n_bin = 1 + n_fft // 2
    fftfreqs = np.linspace(0, sr / 2.0, n_bin)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    w = np.zeros((n_mels, n_bin))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return (w * enorm[:, None]).astype(np.float32)


def window_padded(window, win_length, n_fft):
    w = scipy.signal.get_window(window, win_length, fftbins=True)   # audio.py:133
    if n_fft != win_length:                                         # pad_center :136-137
        lpad = (n_fft - win_length) // 2
        w = np.pad(w, (lpad, n_fft - win_length - lpad))
    return w


def stft(x, n_fft, hop_length, win_length=None, window="hann", center=True, dtype=torch.float32):
    """STFT.forward :161-200 -> (real, imag), each (B, n_bin, frames)."""
    win_length = win_length or n_fft
    n_bin = 1 + n_fft // 2
    w = window_padded(window, win_length, n_fft)
    basis = np.fft.fft(np.eye(n_fft))[:n_bin]
    weight = np.concatenate([basis.real, basis.imag], 0) * w          # :146-152
    weight = torch.from_numpy(weight[:, None, :]).to(dtype)
    x = x.to(dtype).unsqueeze(1)
    if center:
        x = F.pad(x, (n_fft // 2, n_fft // 2), mode="reflect")
    out = F.conv1d(x, weight, stride=hop_length)
    return torch.chunk(out, 2, dim=1)


def magnitude(x, **kw):
    re, im = stft(x, **kw)
    return torch.sqrt(re ** 2 + im ** 2)                              # :198-215


def log_mel(x, sr, n_fft, hop_length, n_mels, fmin, fmax, win_length=None, window="hann", base="10",
            dtype=torch.float32):
    """LogMelFBank.get_log_mel_fbank get_feats.py:80-88 -> (B, frames, n_mels)."""
    mag = magnitude(x, n_fft=n_fft, hop_length=hop_length, win_length=win_length, window=window, dtype=dtype)
    basis = torch.from_numpy(mel_filterbank(sr, n_fft, n_mels, fmin, fmax)).to(dtype)
    mel = torch.matmul(basis, mag)                                    # audio.py:228 / get_feats.py:75
    mel = torch.clamp(mel, min=1e-10)
    mel = torch.log10(mel) if base == "10" else torch.log(mel)
    return mel.transpose(1, 2)
