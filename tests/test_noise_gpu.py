"""pk_randn and the noise == NULL paths of the vocoders (internal Philox(seed))."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import philox_ref
from parakeet_amd import _capi
from parakeet_amd import synthetic as syn
from parakeet_amd.runtime import Context, randn

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,seed,offset", [(1, 0, 0), (4099, 7, 0), (70001, 2 ** 40 + 3, 4 * 123457), (5, 9, 2 ** 34)])
def test_randn_matches_oracle(n, seed, offset):
    got = randn(n, seed=seed, offset=offset).cpu().numpy()
    want = philox_ref.randn(n, seed=seed, offset=offset)
    # same integers, then logf / sincosf in fp32 vs fp64: |z| <= 6.7
    np.testing.assert_allclose(got, want, rtol=0, atol=4e-6)


def test_randn_stream_property_and_host_io():
    a = randn(1000, seed=3).cpu().numpy()
    b = randn(1000 - 16, seed=3, offset=16).cpu().numpy()
    assert np.array_equal(a[16:], b)
    ctx = Context.get()
    host = np.empty(1000, np.float32)
    _capi.check(ctx.lib.pk_randn(ctx.handle, host.ctypes.data_as(C.c_void_p), 1000, 3, 0, _capi.PK_HOST_IO))
    assert np.array_equal(host, a)
    with pytest.raises(ValueError):
        randn(8, seed=1, offset=2)   # offset must be a multiple of 4


def _small_pwg():
    from parakeet_amd.parallel_wavegan import PWGGenerator
    gen = PWGGenerator(**syn.PWG_LJSPEECH)
    gen.set_state_dict(syn.pwg_state())
    gen.eval()
    return gen


def test_pwg_internal_noise_equals_explicit_stream():
    gen = _small_pwg()
    rng = np.random.default_rng(0)
    mels = [rng.normal(size=(L, 80)).astype(np.float32) for L in (5, 9)]
    gen.set_seed(1234)
    a = [o.numpy() for o in gen.inference_batch(mels)]          # noise drawn by the engine
    a2 = [o.numpy() for o in gen.inference_batch(mels)]         # next range of the stream: different
    total = (5 + 9) * 256
    noise = randn(2 * total, seed=1234).cpu().numpy()
    b = [o.numpy() for o in gen.inference_batch(mels, [noise[:5 * 256], noise[5 * 256:total]])]
    b2 = [o.numpy() for o in gen.inference_batch(mels, [noise[total:total + 5 * 256], noise[total + 5 * 256:]])]
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    for x, y in zip(a2, b2):
        assert np.array_equal(x, y)
    assert not np.array_equal(a[0], a2[0])
    gen.set_seed(1234)                                           # re-seeding rewinds the stream
    c = [o.numpy() for o in gen.inference_batch(mels)]
    assert np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1])


def test_waveflow_internal_latent_equals_explicit_stream():
    from parakeet_amd.waveflow import ConditionalWaveFlow
    cfg = dict(syn.WAVEFLOW_LJSPEECH)
    cfg.update(channels=64)
    m = ConditionalWaveFlow(**cfg)
    m.set_state_dict(syn.waveflow_state(cfg))
    m.eval()
    rng = np.random.default_rng(1)
    mel = rng.normal(-4, 2, size=(80, 6)).astype(np.float32)
    m.set_seed(99)
    a = m.infer_batch([mel])[0].numpy()
    zlen, _ = m.lengths(6)
    z = randn(zlen, seed=99).cpu().numpy()
    b = m.infer_batch([mel], [z])[0].numpy()
    assert np.array_equal(a, b)
