"""ZScore normaliser with the reference's interface (parakeet/modules/normalizer.py:18-33).

Inside the engine the two affines are fused into the kernels that produce /
consume the mel (FastSpeech2Inference output, PWGInference input); this class
carries the statistics and offers forward/inverse on host or device arrays.
"""
import numpy as np
import torch

from .runtime import to_numpy_f32


class ZScore:
    # feature last
    def __init__(self, mu, sigma):
        self.mu = to_numpy_f32(mu)
        self.sigma = to_numpy_f32(sigma)

    def _stats(self, x):
        if isinstance(x, torch.Tensor):
            return (torch.as_tensor(self.mu, device=x.device), torch.as_tensor(self.sigma, device=x.device))
        return self.mu, self.sigma

    def forward(self, x):
        mu, sigma = self._stats(x)
        return (x - mu) / sigma

    __call__ = forward

    def inverse(self, x):
        mu, sigma = self._stats(x)
        return x * sigma + mu
