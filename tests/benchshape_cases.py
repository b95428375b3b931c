"""Inputs of tests/golden/benchshape.npz (tools/make_golden_benchshape.py): the recorded random inputs are numpy streams
from stored seeds; their sha256 is in the file and is checked here -- a numpy whose stream differs fails loudly instead of
comparing against outputs of other inputs."""
import hashlib
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FRAMES, HOP = 640, 256


def load(gold_dir=GOLD):
    return np.load(os.path.join(gold_dir, "benchshape.npz"))


def _stream(seed, n):
    return np.random.default_rng(int(seed)).standard_normal(n, dtype=np.float32)


def _checked(a, digest, what):
    got = np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)
    assert np.array_equal(got, digest), f"numpy's default_rng stream differs from the one the golden's {what} was drawn from"
    return a


def pwg_inputs(g):
    mel = _checked(_stream(g["pwg_mel_seed"], FRAMES * 80).reshape(FRAMES, 80), g["pwg_mel_sha256"], "PWG mel")
    noise = _checked(_stream(g["pwg_noise_seed"], FRAMES * HOP), g["pwg_noise_sha256"], "PWG noise")
    return mel, noise


def waveflow_inputs(g):
    mel = np.maximum(_stream(g["wf_mel_seed"], 80 * FRAMES).reshape(1, 80, FRAMES) * 2 - 4, np.log(1e-5)).astype(np.float32)
    _checked(mel, g["wf_mel_sha256"], "WaveFlow mel")
    z = _checked(_stream(g["wf_z_seed"], g["wf_wav"].shape[0]).reshape(1, -1), g["wf_z_sha256"], "WaveFlow latent")
    return mel, z
