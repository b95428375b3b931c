// pk_philox.h -- Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11) and the engine's dropout stream
// (include/pk_synth.h, "dropout stream"; restated in oracle/philox_ref.py).
#pragma once
#include <hip/hip_runtime.h>

// Philox4x32-10 block: counter (c0..c3), key (k0, k1) -> 4 x uint32
__device__ __forceinline__ void philox4x32_10(unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3, unsigned k0,
                                              unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned lo0 = 0xD2511F53u * c0, hi0 = __umulhi(0xD2511F53u, c0);
        const unsigned lo1 = 0xCD9E8D57u * c2, hi1 = __umulhi(0xCD9E8D57u, c2);
        const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

constexpr unsigned PK_DROPOUT_STREAM = 0x44524F50u;   // "DROP": counter word 3 (the noise stream uses 0)

// The four 32-bit words of dropout-stream elements e .. e + 3 (e a multiple of 4); keep <=> word >= threshold
__device__ __forceinline__ void pk_dropout_words(unsigned long long e, unsigned long long seed, unsigned (&w)[4]) {
    const unsigned long long blk = e >> 2;
    unsigned c0 = (unsigned)blk, c1 = (unsigned)(blk >> 32), c2 = 0u, c3 = PK_DROPOUT_STREAM;
    philox4x32_10(c0, c1, c2, c3, (unsigned)seed, (unsigned)(seed >> 32));
    w[0] = c0; w[1] = c1; w[2] = c2; w[3] = c3;
}

// host: keep <=> word >= threshold, P(keep) = 1 - threshold / 2^32
static inline unsigned pk_dropout_threshold(double p) {
    double t = p * 4294967296.0;
    t = t < 0.0 ? 0.0 : (t > 4294967295.0 ? 4294967295.0 : t);
    return (unsigned)(long long)t;   // floor
}
