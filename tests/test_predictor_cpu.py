"""The paddle.inference-shaped wrapper (host logic only; the GPU path is exercised in test_speedyspeech_gpu)."""
import numpy as np
import pytest

from parakeet_amd.predictor import create_predictor


def test_predictor_protocol():
    calls = []

    def model(phones, tones):
        calls.append((phones.copy(), tones.copy()))
        return (phones + tones).astype(np.float32)[:, None] * np.ones((1, 3), np.float32)

    p = create_predictor(model, ["phones", "tones"])
    assert p.get_input_names() == ["phones", "tones"] and p.get_output_names() == ["out"]
    with pytest.raises(RuntimeError):
        p.run()
    a, b = np.arange(6, dtype=np.int64), np.ones(6, dtype=np.int64)
    for name, v in zip(p.get_input_names(), (a, b)):
        h = p.get_input_handle(name)
        h.reshape(v.shape)
        h.copy_from_cpu(v)
    with pytest.raises(RuntimeError):
        p.get_output_handle("out").copy_to_cpu()
    assert p.run()
    out = p.get_output_handle(p.get_output_names()[0]).copy_to_cpu()
    assert out.shape == (6, 3) and np.array_equal(out[:, 0], (a + b).astype(np.float32))
    assert len(calls) == 1


def test_nets_utils_worked_examples():
    """The docstring examples of parakeet/modules/nets_utils.py:36-42,71-75,119-123 and fastspeech2.py:634-637."""
    from parakeet_amd import nets_utils as nu
    x = [np.ones(4), np.ones(2), np.ones(1)]
    assert np.array_equal(nu.pad_list(x, 0), [[1, 1, 1, 1], [1, 1, 0, 0], [1, 0, 0, 0]])
    assert np.array_equal(nu.make_pad_mask([5, 3, 2]).astype(int), [[0, 0, 0, 0, 0], [0, 0, 0, 1, 1], [0, 0, 1, 1, 1]])
    assert np.array_equal(nu.make_non_pad_mask(np.array([5, 3, 2])).astype(int),
                          [[1, 1, 1, 1, 1], [1, 1, 1, 0, 0], [1, 1, 0, 0, 0]])
    assert nu.source_mask([5, 3]).shape == (2, 1, 5)
    with pytest.raises(ValueError):
        nu.make_pad_mask([3], length_dim=0)
