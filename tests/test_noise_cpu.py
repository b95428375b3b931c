"""Oracle of the engine's noise stream against published known-answer vectors (no GPU)."""
import numpy as np

from oracle import philox_ref


def test_philox4x32_10_known_answer_vectors():
    # Random123 (D. E. Shaw Research) kat_vectors, philox4x32 with 10 rounds
    kat = [
        ((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
        ((0xFFFFFFFF,) * 4, (0xFFFFFFFF, 0xFFFFFFFF), (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
        ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
         (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
    ]
    for ctr, key, want in kat:
        got = philox_ref.philox4x32_10(np.array(ctr, dtype=np.uint32), key)
        assert [int(v) for v in got] == list(want)


def test_stream_is_a_function_of_seed_and_index():
    a = philox_ref.randn(1003, seed=5)
    b = philox_ref.randn(1003 - 8, seed=5, offset=8)
    assert np.array_equal(a[8:], b)
    assert not np.array_equal(a, philox_ref.randn(1003, seed=6))


def test_moments():
    z = philox_ref.randn(400_000, seed=2021)
    assert abs(z.mean()) < 5e-3 and abs(z.var() - 1) < 1e-2
    assert abs(((z - z.mean()) ** 4).mean() / z.var() ** 2 - 3) < 5e-2
    assert np.isfinite(z).all() and np.abs(z).max() < 6.7   # sqrt(-2 ln 2^-32) = 6.66
