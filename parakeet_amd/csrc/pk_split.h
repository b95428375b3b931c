// pk_split.h -- block scaling shared by every split-fp16 kernel (pwg.hip, gemm.hip, fs2.hip attention).
//
// A product a*b of fp32 values is evaluated as a_hi*b_hi + a_lo*b_hi + a_hi*b_lo on v_mfma_f32_32x32x16_f16 with
// x_hi = fp16(x), x_lo = fp16(x - x_hi).  fp16 has 5 exponent bits: x_hi is a full 11-bit part only for
// |x| >= 2^-14 and x_lo only for |x| >= 2^-3 (below that the low part is a subnormal and the pair keeps an
// ABSOLUTE error of 2^-25).  The kernels therefore never split raw values: every MFMA operand is first multiplied
// by a power of two that brings the largest magnitude of its block to [2^13, 2^14), and the accumulator is
// brought back by the exact inverse in the epilogue.  Multiplying by a power of two is exact, so the result is the
// unscaled algorithm's with the subnormal floor moved from 2^-25 (absolute) to 2^-39 relative to the block
// maximum: the error no longer depends on the scale of weights or activations.
//   weights      one exponent per tensor (PWG) / per 128-column block (GEMM), folded into the stored fragments;
//   activations  one exponent per block the producer can name: PWG x per wave tile (64 ch x 32 samples and the
//                blocks its taps touch), GEMM A per output row (the rows its taps read), attention V per
//                (utterance, head); gate outputs |z| < 1 and softmax weights p <= 1 use the fixed 2^14.
#pragma once
#include <cmath>
#include <cstddef>

constexpr int PK_BLK_TOP = 13;            // block maximum goes to [2^13, 2^14)
constexpr int PK_EXP_MIN = 87;            // blocks below 2^-40 (all-zero gaps) are scaled as if they were 2^-40
constexpr int PK_EXP_MAX = 200;
constexpr float PK_UNIT_SCALE = 16384.f;  // 2^14 for operands bounded by 1
constexpr int PK_UNIT_EXP = 14;

// exponent k of the scale 2^k from the fp32 bits of the block maximum (sign bit clear)
__host__ __device__ __forceinline__ int blk_scale_exp(unsigned amax_bits) {
    int e = (int)(amax_bits >> 23);
    e = e < PK_EXP_MIN ? PK_EXP_MIN : (e > PK_EXP_MAX ? PK_EXP_MAX : e);
    return PK_BLK_TOP + 127 - e;
}
__device__ __forceinline__ float pow2f(int k) { return __uint_as_float((unsigned)(k + 127) << 23); }   // |k| <= 126

// host: exponent k with max|w| * 2^k in [2^13, 2^14) over n values with the given stride
static inline int pk_weight_scale_exp(const float* w, size_t n, size_t stride = 1) {
    float m = 0.f;
    for (size_t i = 0; i < n; ++i) {
        const float v = std::fabs(w[i * stride]);
        if (std::isfinite(v) && v > m) m = v;
    }
    if (m == 0.f) return 0;
    int e;
    (void)std::frexp(m, &e);   // m = f * 2^e, f in [0.5, 1)
    const int k = 14 - e;
    return k < -40 ? -40 : (k > 40 ? 40 : k);
}
