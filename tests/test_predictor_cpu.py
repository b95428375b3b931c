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
