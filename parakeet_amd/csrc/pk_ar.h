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

// The NEW row block of a decoding step through the decoder prenet (two Linear + ReLU + dropout layers, always-on dropout) and
// the input layer (Linear + positional encoding) in ONE launch, a workgroup per utterance: three dependent row GEMMs of 6 us
// each (+ 3 us between two launches) become three phases behind __syncthreads.  A phase: thread (n4 = tid % (N / 4), part =
// tid / (N / 4)) owns 4 consecutive outputs and a slice of K, all of its float4 weight loads ([K][N] row-major, L2-resident: every
// workgroup reads the same 850 KB) in flight in batches of 32, the input row broadcast from LDS; the parts are summed through
// LDS in order (deterministic).  LJSpeech recipe shapes: odim 80 -> 256 -> 256 -> adim 512.  we == NULL (Tacotron2: the prenet
// alone): the second layer's output is the result (x0 [B][ldx0], U columns).
struct pk_prenet_embed {
    const float* y; int ldy;                 // [B][ldy]: the previous step's last frame (K0 = O values)
    int O, U, A, B;
    const float *w1, *b1, *w2, *b2, *we, *be;   // [O][U], [U][U], [U][A] row-major (+ biases, NULL = none)
    const float* peb; int ldpe;              // positional encoding (scaled) of the new rows [B][ldpe]
    float* x0; int ldx0;                     // out: [B][ldx0]
    int dropout; unsigned long long base; int J; const unsigned long long* seeds; unsigned thr; float scale;
};
static __device__ __forceinline__ void pk_ar_dense_phase(const float* in, int K, const float* __restrict__ W, int N, float* red, int tid) {
    const int ng = N >> 2, n4 = tid % ng, part = tid / ng, nparts = 512 / ng;
    const int kper = (K + nparts - 1) / nparts, kbeg = part * kper, kend = min(K, kbeg + kper);
    const float4* W4 = reinterpret_cast<const float4*>(W);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k0 = kbeg; k0 < kend; k0 += 32) {
        float4 w[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) w[i] = W4[(long)min(k0 + i, kend - 1) * ng + n4];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const float x = k0 + i < kend ? in[k0 + i] : 0.f;
            acc.x = fmaf(x, w[i].x, acc.x);
            acc.y = fmaf(x, w[i].y, acc.y);
            acc.z = fmaf(x, w[i].z, acc.z);
            acc.w = fmaf(x, w[i].w, acc.w);
        }
    }
    if (part < nparts) *reinterpret_cast<float4*>(red + part * N + 4 * n4) = acc;
}
static __global__ __launch_bounds__(512) void k_ar_prenet_embed(pk_prenet_embed a) {
    __shared__ __attribute__((aligned(16))) float hin[512];
    __shared__ __attribute__((aligned(16))) float red[8 * 256];   // (N / 4 column groups x 512 / (N / 4) parts = 2048 floats for every N)
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid < a.O) hin[tid] = a.y[(long)b * a.ldy + tid];
    __syncthreads();
    for (int j = 0; j < 2; ++j) {
        const float* W = j == 0 ? a.w1 : a.w2;
        const float* bias = j == 0 ? a.b1 : a.b2;
        const int K = j == 0 ? a.O : a.U, N = a.U;
        pk_ar_dense_phase(hin, K, W, N, red, tid);
        __syncthreads();
        const int ng = N >> 2, nparts = 512 / ng;
        if (tid < ng) {
            float4 s = bias ? *reinterpret_cast<const float4*>(bias + 4 * tid) : make_float4(0.f, 0.f, 0.f, 0.f);
            for (int p = 0; p < nparts; ++p) {
                const float4 r = *reinterpret_cast<const float4*>(red + p * N + 4 * tid);
                s.x += r.x; s.y += r.y; s.z += r.z; s.w += r.w;
            }
            s.x = fmaxf(s.x, 0.f); s.y = fmaxf(s.y, 0.f); s.z = fmaxf(s.z, 0.f); s.w = fmaxf(s.w, 0.f);
            if (a.dropout) {
                const unsigned long long e = (a.base * (unsigned long long)a.J + (unsigned long long)j) * (unsigned long long)N + 4ull * tid;
                unsigned w4[4];
                pk_dropout_words(e, a.seeds ? a.seeds[b] : 0ull, w4);
                s.x = w4[0] >= a.thr ? s.x * a.scale : 0.f;
                s.y = w4[1] >= a.thr ? s.y * a.scale : 0.f;
                s.z = w4[2] >= a.thr ? s.z * a.scale : 0.f;
                s.w = w4[3] >= a.thr ? s.w * a.scale : 0.f;
            }
            *reinterpret_cast<float4*>(hin + 4 * tid) = s;
        }
        __syncthreads();
    }
    if (!a.we) {   // (uniform)
        if (tid < (a.U >> 2)) *reinterpret_cast<float4*>(a.x0 + (long)b * a.ldx0 + 4 * tid) = *reinterpret_cast<const float4*>(hin + 4 * tid);
        return;
    }
    pk_ar_dense_phase(hin, a.U, a.we, a.A, red, tid);
    __syncthreads();
    const int ng = a.A >> 2, nparts = 512 / ng;
    if (tid < ng) {
        float4 s = a.be ? *reinterpret_cast<const float4*>(a.be + 4 * tid) : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int p = 0; p < nparts; ++p) {
            const float4 r = *reinterpret_cast<const float4*>(red + p * a.A + 4 * tid);
            s.x += r.x; s.y += r.y; s.z += r.z; s.w += r.w;
        }
        const float4 pe = *reinterpret_cast<const float4*>(a.peb + (long)b * a.ldpe + 4 * tid);
        s.x += pe.x; s.y += pe.y; s.z += pe.z; s.w += pe.w;
        *reinterpret_cast<float4*>(a.x0 + (long)b * a.ldx0 + 4 * tid) = s;
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
