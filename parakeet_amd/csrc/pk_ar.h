// pk_ar.h -- small kernels shared by the autoregressive acoustic models (tts.hip: TransformerTTS, taco2.hip: Tacotron2).
// Rows of per-step tensors are POSITION-MAJOR: row = pos * B + b (all utterances of a batch are decoded in lockstep).
#pragma once
#include "pk_common.h"
#include "pk_philox.h"

// Prenet dropout (always on at inference: modules/tacotron2/decoder.py:78-81, models/tacotron2.py:76-79), in place,
// one thread per 4 units of a row.  Row r belongs to utterance r % B at prefix position r / B; element index of the
// dropout stream (include/pk_synth.h): ((base + r / B) * J + j) * U + u, seed per utterance.
// amax (optional, U == 256 only: a row is exactly one wave): max|.| of every row after the dropout -- the operand scale the
// split-fp16 GEMM that reads these rows would otherwise compute with a pass of its own (k_row_amax).
static __global__ __launch_bounds__(256) void k_ar_dropout(float* __restrict__ x, int ld, int rows, int U, int B,
                                                           unsigned long long base, int J, int j,
                                                           const unsigned long long* __restrict__ seeds,
                                                           unsigned thr, float scale, float* __restrict__ amax) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    const int per_row = U >> 2;
    if (q >= (long)rows * per_row) return;   // (U == 256: whole waves leave)
    const int r = (int)(q / per_row), u4 = (int)(q - (long)r * per_row) * 4;
    const int pos = r / B, b = r - pos * B;
    const unsigned long long e = ((base + (unsigned long long)pos) * (unsigned long long)J + (unsigned long long)j) *
                                     (unsigned long long)U + (unsigned long long)u4;
    unsigned w[4];
    pk_dropout_words(e, seeds ? seeds[b] : 0ull, w);
    float4* p = reinterpret_cast<float4*>(x + (long)r * ld + u4);
    float4 v = *p;
    v.x = w[0] >= thr ? v.x * scale : 0.f;
    v.y = w[1] >= thr ? v.y * scale : 0.f;
    v.z = w[2] >= thr ? v.z * scale : 0.f;
    v.w = w[3] >= thr ? v.w * scale : 0.f;
    *p = v;
    if (amax) {
        float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if ((threadIdx.x & 63) == 0) amax[r] = m;
    }
}

// Position-major rows -> a row timeline (or packed rows through rowmap): timeline row r of utterance u at position p
// takes src row (p + off) * B + u; gap rows are zeroed when rowmap == NULL.  Optional per-column affine.
static __global__ __launch_bounds__(128) void k_ar_gather(const float* __restrict__ src, int C, int B, int off,
                                                          const int* __restrict__ row_utt,
                                                          const int* __restrict__ row_pos,
                                                          const int* __restrict__ rowmap,
                                                          const float* __restrict__ cscale,
                                                          const float* __restrict__ cshift, float* __restrict__ dst) {
    const long r = blockIdx.x;
    const int u = row_utt[r];
    const long o = rowmap ? rowmap[r] : r;
    if (o < 0) return;
    if (u < 0) {
        if (!rowmap)
            for (int c = threadIdx.x; c < C; c += blockDim.x) dst[o * C + c] = 0.f;
        return;
    }
    const float* s = src + ((long)(row_pos[r] + off) * B + u) * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float v = s[c];
        if (cscale) v = v * cscale[c] + cshift[c];
        dst[o * C + c] = v;
    }
}
