"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- numpy restatement of the engine's noise stream.

Philox4x32-10 as published (J. K. Salmon, M. A. Moraes, R. O. Dror, D. E. Shaw, "Parallel random numbers:
as easy as 1, 2, 3", SC'11; constants M0 = 0xD2511F53, M1 = 0xCD9E8D57, W0 = 0x9E3779B9, W1 = 0xBB67AE85),
pinned by the known-answer vectors of the Random123 distribution (tests/test_golden_cpu.py), followed by
Box-Muller exactly as include/pk_synth.h documents for pk_randn.  The reference itself has no counterpart:
it calls paddle.randn (parallel_wavegan.py:515-516, waveflow.py:801).
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter, key):
    """counter: (..., 4) uint32, key: (2,) ints -> (..., 4) uint32."""
    c = np.asarray(counter, dtype=np.uint64).copy()
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c[..., 0]
        p1 = M1 * c[..., 2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
        n0 = hi1 ^ c[..., 1] ^ np.uint64(k0)
        n2 = hi0 ^ c[..., 3] ^ np.uint64(k1)
        c = np.stack([n0, lo1, n2, lo0], axis=-1)
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return c.astype(np.uint32)


def randn(n, seed=0, offset=0, dtype=np.float64):
    """The pk_randn stream: element i depends on (seed, offset + i) only."""
    assert offset % 4 == 0
    nblk = (n + 3) // 4
    ctr = np.uint64(offset // 4) + np.arange(nblk, dtype=np.uint64)
    counter = np.stack([ctr & MASK, ctr >> np.uint64(32), np.zeros_like(ctr), np.zeros_like(ctr)], axis=-1)
    r = philox4x32_10(counter, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)).astype(np.float64)
    out = np.empty((nblk, 4), dtype=np.float64)
    for h in range(2):
        # the engine forms u1, u2 in fp32: (float)c rounds the 32-bit integer to 24 bits first
        u1 = (np.float32(r[:, 2 * h]).astype(np.float32) + np.float32(1.0)).astype(np.float64) * 2.0 ** -32
        u2 = np.float32(r[:, 2 * h + 1]).astype(np.float64) * 2.0 ** -32
        rad = np.sqrt(-2.0 * np.log(u1))
        out[:, 2 * h] = rad * np.cos(2.0 * np.pi * u2)
        out[:, 2 * h + 1] = rad * np.sin(2.0 * np.pi * u2)
    return out.reshape(-1)[:n].astype(dtype)


DROPOUT_STREAM = 0x44524F50   # "DROP": counter word 3 of the dropout stream (the noise stream uses 0)


def dropout_threshold(p):
    """keep <=> random 32-bit word >= threshold; P(keep) = 1 - threshold / 2^32 (exactly 1 - p for dyadic p)."""
    return int(min(max(np.floor(float(p) * 4294967296.0), 0.0), 4294967295.0))


def dropout_keep(index, p, seed=0):
    """The engine's dropout stream (include/pk_synth.h, "dropout stream"): element ``index`` (any integer array,
    < 2^64) -> keep flag.  Word index & 3 of the Philox block with counter (lo(index >> 2), hi(index >> 2), 0, "DROP")
    and key (lo(seed), hi(seed))."""
    idx = np.asarray(index, dtype=np.uint64)
    blk = idx >> np.uint64(2)
    counter = np.stack([blk & MASK, blk >> np.uint64(32), np.zeros_like(blk),
                        np.full_like(blk, DROPOUT_STREAM)], axis=-1)
    r = philox4x32_10(counter, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    word = np.take_along_axis(r, (idx & np.uint64(3)).astype(np.int64)[..., None], axis=-1)[..., 0]
    return word >= np.uint32(dropout_threshold(p))
