"""STFT / mel features behind the reference's Python API.

Mirrors parakeet/modules/audio.py ``STFT`` (:74-215: forward / power / magnitude) and ``MelScale``
(:218-229), and the host feature extractor ``LogMelFBank`` of parakeet/data/get_feats.py (:20-88);
the arithmetic (reflect padding, DFT-as-GEMM, magnitude, mel GEMM, log) runs in libpk_synth.so
(csrc/mel.hip).  The window and the mel filterbank are host-side constants handed to the engine;
``mel_filterbank`` follows the algorithm of librosa.filters.mel (Slaney scale and normalisation),
which the reference calls at audio.py:221 / get_feats.py:49-55.
"""
import ctypes as C
import math

import numpy as np
import scipy.signal
import torch

from . import _capi
from .runtime import Context, dptr, wrap


def _slaney_hz(m):
    f_sp, brk_mel, step = 200.0 / 3.0, 15.0, math.log(6.4) / 27.0
    m = np.asarray(m, dtype=np.float64)
    lin = m * f_sp
    return np.where(m >= brk_mel, 1000.0 * np.exp(step * (m - brk_mel)), lin)


def _slaney_mel(f):
    f_sp, brk_mel, step = 200.0 / 3.0, 15.0, math.log(6.4) / 27.0
    return f / f_sp if f < 1000.0 else brk_mel + math.log(f / 1000.0) / step


def mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    """Triangular Slaney-normalised mel filters, (n_mels, 1 + n_fft//2) float32."""
    fmax = sr / 2.0 if fmax is None else float(fmax)
    freqs = np.arange(1 + n_fft // 2, dtype=np.float64) * (sr / 2.0) / (n_fft // 2)
    edges = _slaney_hz(np.linspace(_slaney_mel(float(fmin)), _slaney_mel(fmax), n_mels + 2))
    lo, mid, hi = edges[:-2, None], edges[1:-1, None], edges[2:, None]
    up = (freqs[None, :] - lo) / (mid - lo)
    down = (hi - freqs[None, :]) / (hi - mid)
    tri = np.clip(np.minimum(up, down), 0.0, None)
    return (tri * (2.0 / (hi - lo))).astype(np.float32)


def _window(window, win_length, n_fft):
    name = "hann" if window == "hanning" else window
    w = scipy.signal.get_window(name, win_length, fftbins=True)
    if n_fft != win_length:
        left = (n_fft - win_length) // 2
        w = np.pad(w, (left, n_fft - win_length - left))
    return np.ascontiguousarray(w, dtype=np.float32)


class _Engine:
    def __init__(self, n_fft, hop_length, win_length, window, center, power, mel_basis, log_base, device=None):
        self.ctx = Context.get(device)
        cfg = _capi.MelCfg(n_fft, hop_length, 1 if center else 0, 1 if power else 0,
                           0 if mel_basis is None else mel_basis.shape[0], log_base, 1e-10)
        self.n_bin = 1 + n_fft // 2
        self.n_mels = cfg.n_mels
        win = _window(window, win_length or n_fft, n_fft)
        basis = None if mel_basis is None else np.ascontiguousarray(mel_basis, np.float32)
        h = C.c_void_p()
        _capi.check(self.ctx.lib.pk_mel_create(self.ctx.handle, C.byref(cfg), _capi.fptr(win),
                                               None if basis is None else _capi.fptr(basis), C.byref(h)))
        self.h = h

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            try:
                self.ctx.lib.pk_mel_destroy(h)
            except Exception:
                pass

    def frames(self, n):
        f = C.c_int32()
        _capi.check(self.ctx.lib.pk_mel_num_frames(self.h, int(n), C.byref(f)))
        return f.value

    def run(self, wavs, what):
        """wavs: list of 1-D arrays -> list of (frames, cols) device tensors."""
        ctx = Context.get(self.ctx.device)
        lens = np.array([int(np.prod(w.shape)) for w in wavs], dtype=np.int32)
        x = torch.cat([ctx.to_device(w).reshape(-1) for w in wavs])
        cols = {0: 2 * self.n_bin, 1: self.n_bin, 2: self.n_mels}[what]
        nf = [self.frames(n) for n in lens]
        out = ctx.empty((sum(nf), cols))
        _capi.check(ctx.lib.pk_mel_run(self.h, dptr(x), lens.ctypes.data_as(C.POINTER(C.c_int32)), len(wavs),
                                       dptr(out), what, 0))
        res, o = [], 0
        for f in nf:
            res.append(out[o:o + f])
            o += f
        return res


class STFT:
    """(B, T) -> real, imag (B, n_bin, frames); audio.py:74-215."""

    def __init__(self, n_fft, hop_length=None, win_length=None, window="hanning", center=True, pad_mode="reflect"):
        if pad_mode != "reflect":
            raise NotImplementedError("only pad_mode='reflect' is implemented")
        win_length = win_length or n_fft
        hop_length = hop_length or int(win_length // 4)
        self.n_fft, self.hop_length, self.n_bin, self.center = n_fft, hop_length, 1 + n_fft // 2, center
        self._mag = _Engine(n_fft, hop_length, win_length, window, center, False, None, 0)
        self._pow = None
        self._args = (n_fft, hop_length, win_length, window, center)

    def _batch(self, x, eng, what):
        x = x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))
        outs = eng.run([x[b] for b in range(x.shape[0])], what)
        return torch.stack([o.transpose(0, 1) for o in outs], 0)

    def forward(self, x):
        y = self._batch(x, self._mag, 0)
        return wrap(y[:, :self.n_bin]), wrap(y[:, self.n_bin:])

    __call__ = forward

    def power(self, x):
        if self._pow is None:
            n_fft, hop, win, window, center = self._args
            self._pow = _Engine(n_fft, hop, win, window, center, True, None, 0)
        return wrap(self._batch(x, self._pow, 1))

    def magnitude(self, x):
        return wrap(self._batch(x, self._mag, 1))


class MelScale:
    """(B, n_freq, frames) -> (B, n_mels, frames); audio.py:218-229 (a plain matmul with the basis)."""

    def __init__(self, sr, n_fft, n_mels, fmin, fmax):
        self.weight = torch.from_numpy(mel_filterbank(sr, n_fft, n_mels, fmin, fmax))
        self._wkn = np.ascontiguousarray(self.weight.numpy().T)          # [n_freq][n_mels]

    def forward(self, spec):
        """``paddle.matmul(self.weight, spectrogram)`` as one engine GEMM over rows = (batch, frame)
        (``pk_op_matmul``); torch only moves data (the NCL <-> rows transposes)."""
        ctx = Context.get()
        spec = ctx.to_device(spec)
        squeeze = spec.dim() == 2
        if squeeze:
            spec = spec[None]
        B, F, T = spec.shape
        assert F == self._wkn.shape[0], "MelScale: spectrogram has the wrong number of frequency bins"
        n_mels = self._wkn.shape[1]
        x = spec.transpose(1, 2).contiguous().reshape(B * T, F)
        y = ctx.empty((B * T, n_mels))
        _capi.check(ctx.lib.pk_op_matmul(ctx.handle, dptr(x), B * T, F, n_mels, _capi.fptr(self._wkn), None, dptr(y)))
        out = y.reshape(B, T, n_mels).transpose(1, 2).contiguous()
        return wrap(out[0] if squeeze else out)

    __call__ = forward


class LogMelFBank:
    """get_feats.py:20-88 on the device: wav -> (num_frames, n_mels) log-mel."""

    def __init__(self, sr=24000, n_fft=2048, hop_length=300, win_length=None, window="hann", n_mels=80,
                 fmin=80, fmax=7600, eps=1e-10):
        self.sr, self.n_fft, self.hop_length = sr, n_fft, hop_length
        self.fmin = 0 if fmin is None else fmin
        self.fmax = sr / 2 if fmax is None else fmax
        self.mel_filter = mel_filterbank(sr, n_fft, n_mels, self.fmin, self.fmax)
        self._args = (n_fft, hop_length, win_length or n_fft, window)
        self._eng = {}

    def _engine(self, base):
        if base not in self._eng:
            n_fft, hop, win, window = self._args
            self._eng[base] = _Engine(n_fft, hop, win, window, True, False, self.mel_filter,
                                      10 if base == "10" else 2)
        return self._eng[base]

    def get_log_mel_fbank(self, wav, base="10"):
        return wrap(self._engine(base).run([wav], 2)[0])

    def get_log_mel_fbank_batch(self, wavs, base="10"):
        return [wrap(o) for o in self._engine(base).run(list(wavs), 2)]


def wav_bytes(wav, samplerate, subtype="PCM_16"):
    """The bytes ``write_wav`` puts into the file (a serving process sends them instead of writing a file)."""
    import io
    buf = io.BytesIO()
    write_wav(buf, wav, samplerate, subtype)
    return buf.getvalue()


def write_wav(path, wav, samplerate, subtype="PCM_16"):
    """Minimal stand-in for the ``soundfile.write(path, wav.numpy(), samplerate=fs)`` at the end of the
    synthesis recipes (examples/fastspeech2/ljspeech/synthesize_e2e.py:104-107): RIFF/WAVE, mono or
    (T, C) float input in [-1, 1].  "PCM_16" (libsndfile's default for .wav: clip, scale by 32767, round to
    nearest) or "FLOAT" (32-bit IEEE samples, lossless)."""
    import struct
    x = np.asarray(wav.cpu() if isinstance(wav, torch.Tensor) else wav, dtype=np.float32)
    if x.ndim == 1:
        x = x[:, None]
    if x.ndim != 2:
        raise ValueError("write_wav: expected (T,) or (T, channels)")
    n, ch = x.shape
    if subtype == "PCM_16":
        data = np.rint(np.clip(x, -1.0, 1.0) * 32767.0).astype("<i2").tobytes()
        fmt, bits = 1, 16
    elif subtype == "FLOAT":
        data = x.astype("<f4").tobytes()
        fmt, bits = 3, 32
    else:
        raise ValueError(f"write_wav: unsupported subtype {subtype!r}")
    block = ch * bits // 8
    header = b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE" + b"fmt " + struct.pack(
        "<IHHIIHH", 16, fmt, ch, int(samplerate), int(samplerate) * block, block, bits) + b"data" + struct.pack(
        "<I", len(data))
    if hasattr(path, "write"):       # a binary file object (wav_bytes)
        path.write(header)
        path.write(data)
        return
    with open(path, "wb") as f:
        f.write(header)
        f.write(data)


def stft(x, fft_size, hop_length=None, win_length=None, window="hann", center=True, pad_mode="reflect"):
    """``parakeet.modules.stft_loss.stft`` (:20-67): (B, T) -> magnitude spectrogram (B, frames, fft_size//2+1),
    ``sqrt(clip(re^2 + im^2, min=1e-7))``.  The transform runs on the engine (STFT-as-GEMM + magnitude kernel);
    the floor is one elementwise maximum with sqrt(1e-7) on the result."""
    win_length = win_length or fft_size
    t = STFT(fft_size, hop_length, win_length, window, center, pad_mode)
    mag = t.magnitude(x).as_subclass(torch.Tensor)                    # (B, bins, frames)
    return wrap(torch.clamp_min(mag, float(np.sqrt(np.float32(1e-7)))).transpose(1, 2).contiguous())


class LogMagnitude:
    """parakeet/audio/spec_normalizer.py:37-53 (the WaveFlow / Tacotron2 feature domain): log(max(x, min))."""

    def __init__(self, min=1e-5):   # noqa: A002  (the reference's argument name)
        self.min = min

    def transform(self, x):
        return np.log(np.maximum(np.asarray(x), self.min))

    def inverse(self, x):
        return np.exp(np.asarray(x))


class UnitMagnitude:
    """parakeet/audio/spec_normalizer.py:56-75: dB scale mapped to [0, 1] (20 log10(max(x, min)) - 20, then (. + 100) / 100,
    clipped) and back.  Host-side numpy, like the reference."""

    def __init__(self, min=1e-5):   # noqa: A002
        self.min = min

    def transform(self, x):
        db = 20.0 * np.log10(np.maximum(self.min, np.asarray(x))) - 20.0
        return np.clip((db + 100.0) / 100.0, 0, 1)

    def inverse(self, x):
        db = np.clip(np.asarray(x), 0, 1) * 100.0 - 100.0
        return np.exp((db + 20.0) / 20.0 * np.log(10))


class AudioProcessor:
    """parakeet/audio/audio.py:20-102 with the transforms on the engine: ``spectrogram`` = |STFT|,
    ``mel_spectrogram`` = mel_filter . |STFT| (the Slaney filterbank of ``librosa.filters.mel``), both returned
    as (bins, frames) numpy arrays like the reference.  ``read_wav`` reads PCM / float RIFF files at the
    processor's sample rate (the reference resamples with librosa, which this image does not have)."""

    def __init__(self, sample_rate, n_fft, win_length, hop_length, n_mels=80, fmin=0, fmax=None, window="hann",
                 center=True, pad_mode="reflect", normalize=True):
        if pad_mode != "reflect":
            raise NotImplementedError("only pad_mode='reflect' is implemented")
        self.sample_rate, self.normalize = sample_rate, normalize
        self.n_fft, self.win_length, self.hop_length = n_fft, win_length, hop_length
        self.window, self.center, self.pad_mode = window, center, pad_mode
        self.n_mels, self.fmin, self.fmax = n_mels, fmin, fmax
        self.mel_filter = mel_filterbank(sample_rate, n_fft, n_mels, fmin or 0, fmax or sample_rate / 2)
        self.inv_mel_filter = np.linalg.pinv(self.mel_filter)
        self._spec = _Engine(n_fft, hop_length, win_length, window, center, False, None, 0)
        self._mel = _Engine(n_fft, hop_length, win_length, window, center, False, self.mel_filter, 0)

    def read_wav(self, filename):
        import wave
        with wave.open(str(filename), "rb") as w:
            if w.getframerate() != self.sample_rate:
                raise NotImplementedError(f"{filename}: {w.getframerate()} Hz, resampling to {self.sample_rate} Hz "
                                          "needs librosa")
            raw = w.readframes(w.getnframes())
            ch, width = w.getnchannels(), w.getsampwidth()
        if width != 2:
            raise NotImplementedError("read_wav: 16-bit PCM only")
        wav = np.frombuffer(raw, dtype="<i2").astype(np.float32).reshape(-1, ch).mean(axis=1) / 32768.0
        if self.normalize:
            wav = wav / np.max(np.abs(wav)) * 0.999
        return wav

    def write_wav(self, path, wav):
        write_wav(path, wav, self.sample_rate)

    def spectrogram(self, wav):
        return self._spec.run([np.asarray(wav, dtype=np.float32)], 1)[0].cpu().numpy().T

    def mel_spectrogram(self, wav):
        return self._mel.run([np.asarray(wav, dtype=np.float32)], 2)[0].cpu().numpy().T
