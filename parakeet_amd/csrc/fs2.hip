// fs2.hip -- FastSpeech2 inference on gfx950: kernels + pk_fs2_* entry points.
//
// Reference: parakeet/models/fastspeech2/fastspeech2.py FastSpeech2.inference :468-558
// (_forward(is_inference=True) :377-466) and the modules listed in SURVEY.md 8a
// (encoder.py, encoder_layer.py, attention.py, embedding.py, multi_layer_conv.py,
// duration_predictor.py, variance_predictor.py, length_regulator.py, layer_norm.py,
// tacotron2/decoder.py Postnet, normalizer.py).
//
// Data layout ("row timeline").  Activations are channels-last [rows][C] fp32.  All
// utterances of a batch share one row axis, separated and framed by GAPR zero rows:
//
//      |GAPR| utt 0 (T_0 rows) |GAPR| utt 1 (T_1 rows) |GAPR| ...
//
// GAPR >= the largest (k-1)/2 of any Conv1D on the path, and every tensor that
// feeds a k>1 convolution has its gap rows forced to zero by the kernel that
// produces it -- so a batched conv sees exactly the zero padding the reference's
// one-utterance-per-call inference applies, and ragged batches are exact.
// row_utt[r] = utterance id or -1 (gap), row_pos[r] = position inside the utterance.
// There are two timelines per call: token rate (encoder, variance adaptor) and
// frame rate (decoder, postnet); the length regulator maps one onto the other.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "pk_fft.h"
#include "pk_ffn_planes.h"
#include "pk_gemm.h"
#include "pk_split.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int FS2_MAX_HEADS = PK_FFT_MAX_HEADS;   // (q|k|v, head) magnitude-bound constants are passed to a kernel by value

namespace {

__device__ __forceinline__ int mfma_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// pe[pos][2i] = sin(pos * div[i]), pe[pos][2i+1] = cos(pos * div[i])   (embedding.py:46-62)
__global__ void k_build_pe(float* pe, const float* div, int maxlen, int d) {
    const int pos = blockIdx.x;
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
        const float ang = (float)pos * div[c >> 1];
        pe[(long)pos * d + c] = (c & 1) ? cosf(ang) : sinf(ang);
    }
}

// x[r] = Emb[tok[r]] * xscale + alpha * PE[pos[r]]   (fastspeech2.py:165-168, embedding.py:111-126;
// xscale = 1 for ScaledPositionalEncoding).  Gap rows are zeroed.  padding_idx row 0 of the table
// is zero (set at finalize).
__global__ void k_embed(const int* __restrict__ tok, const int* __restrict__ row_utt,
                        const int* __restrict__ row_pos, const float* __restrict__ table,
                        const float* __restrict__ pe, float alpha, float xscale, int d,
                        float* __restrict__ x) {
    const int r = blockIdx.x;
    const bool valid = row_utt[r] >= 0;
    const float* e = table + (long)(valid ? tok[r] : 0) * d;
    const float* p = pe + (long)(valid ? row_pos[r] : 0) * d;
    for (int c = threadIdx.x; c < d; c += blockDim.x)
        x[(long)r * d + c] = valid ? (e[c] * xscale + alpha * p[c]) : 0.f;
}

// LayerNorm over the channel axis, one wave per row (nn.LayerNorm, eps 1e-5; also
// LayerNorm(dim=1) of the predictors, which is the same thing in channels-last).
// Gap rows -> 0.
constexpr int LN_MAXPER = PK_FFT_LN_MAXPER;
__global__ __launch_bounds__(256) void k_layernorm(const float* __restrict__ x, const float* __restrict__ g,
                                                   const float* __restrict__ b, const int* __restrict__ row_utt,
                                                   int rows, int C, float eps, float* __restrict__ y,
                                                   float* __restrict__ amax) {
    // amax (optional): max|y[r, :]| per row for the block scaling of the split-fp16 GEMM that consumes y (pk_split.h)
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    const int nper = C >> 6;
    float* yo = y + (long)r * C;
    if (row_utt[r] < 0) {
#pragma unroll
        for (int e = 0; e < LN_MAXPER; ++e)
            if (e < nper) yo[lane + 64 * e] = 0.f;
        if (amax && lane == 0) amax[r] = 0.f;
        return;
    }
    const float* xi = x + (long)r * C;
    float v[LN_MAXPER];
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < LN_MAXPER; ++e)
        if (e < nper) {
            v[e] = xi[lane + 64 * e];
            s += v[e];
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < LN_MAXPER; ++e)
        if (e < nper) {
            const float dlt = v[e] - mean;
            q += dlt * dlt;
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float inv = 1.0f / sqrtf(q / (float)C + eps);
    float am = 0.f;
#pragma unroll
    for (int e = 0; e < LN_MAXPER; ++e)
        if (e < nper) {
            const int c = lane + 64 * e;
            const float yv = (v[e] - mean) * inv * g[c] + b[c];
            yo[c] = yv;
            am = fmaxf(am, fabsf(yv));
        }
    if (amax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o));
        if (lane == 0) amax[r] = am;
    }
}

// Multi-head self-attention for one (utterance, head, 32-query tile) per wave
// (attention.py:133-156 + forward_attention :88-131).  qkv rows are [q | k | v], each
// heads*DK wide.  Flash-style: scores never leave registers.
//   S^T[key][q]  = K . Q^T      (A = K rows, B = Q rows; the d axis is split so that a
//                                lane reads contiguous floats: d = hi*DK/2 + ks)
//   online softmax over keys (keys live in accumulator registers, queries in lanes)
//   O[q][dv]    += P[q][key] . V[key][dv]   (A = P, taken straight from the S registers
//                                thanks to a K-order permutation; B = V rows, coalesced)
// Keys >= len get -inf (== masked_fill(min) -> softmax -> masked_fill(0) of the
// reference); the decoder passes no mask (fastspeech2.py:452-455) and attends to its
// whole utterance.
struct AttnArgs {
    const float* qkv;
    int ld;
    float* out;
    int ldo;
    const int* seg_start;
    const int* seg_len;
    int D;      // heads * DK
    float scale;
    const unsigned* amax;   // split-fp16 kernels: fp32 bits of max|q|, max|k|, max|v| per (utterance, head) [B][H][3]
};

// Block maxima for the split-fp16 attention kernels (pk_split.h): one (utterance, head) = one block of Q, of K
// and of V.  grid (ceil(maxlen / 32), heads, B), 4 waves x 8 rows; atomicMax on fp32 bits of non-negative values
// into zeroed memory.
__global__ __launch_bounds__(256) void k_qkv_amax(const float* __restrict__ qkv, int ld, const int* __restrict__ seg_start,
                                                  const int* __restrict__ seg_len, int D, int dk,
                                                  unsigned* __restrict__ amax) {
    const int b = blockIdx.z, h = blockIdx.y;
    const int len = seg_len[b], start = seg_start[b];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r0 = blockIdx.x * 32 + wave * 8;
    if (r0 >= len) return;
    const int r1 = min(r0 + 8, len);
    for (int part = 0; part < 3; ++part) {
        const float* p = qkv + (long)start * ld + part * D + h * dk;
        float m = 0.f;
        for (int r = r0; r < r1; ++r)
            for (int c = lane; c < dk; c += 64) m = fmaxf(m, fabsf(p[(long)r * ld + c]));
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0) atomicMax(amax + ((long)b * gridDim.y + h) * 3 + part, __float_as_uint(m));
    }
}

template <int DK>
__global__ __launch_bounds__(256, 1) void k_attention(AttnArgs a) {
    constexpr int KH = DK / 2;   // k-steps of the QK^T product
    constexpr int DT = DK / 32;  // 32-wide tiles of the value dimension
    const int b = blockIdx.z, h = blockIdx.y;
    const int len = a.seg_len[b], start = a.seg_start[b];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q0 = (blockIdx.x * 4 + wave) * 32;
    if (q0 >= len) return;
    const int j = lane & 31, hi = lane >> 5;
    const long ld = a.ld;
    const float* base = a.qkv + (long)start * ld + h * DK;

    float qf[KH];
    {
        const int qr = min(q0 + j, len - 1);
        const float* qp = base + (long)qr * ld + hi * KH;
#pragma unroll
        for (int c = 0; c < KH / 4; ++c) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(qp + 4 * c);
            qf[4 * c] = v[0];
            qf[4 * c + 1] = v[1];
            qf[4 * c + 2] = v[2];
            qf[4 * c + 3] = v[3];
        }
    }
    f32x16 O[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    for (int k0 = 0; k0 < len; k0 += 32) {
        f32x16 S;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.f;
        {
            const int kr = min(k0 + j, len - 1);
            const float* kp = base + a.D + (long)kr * ld + hi * KH;
#pragma unroll
            for (int c = 0; c < KH / 4; ++c) {
                const f32x4 kv = *reinterpret_cast<const f32x4*>(kp + 4 * c);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    S = __builtin_amdgcn_mfma_f32_32x32x2f32(kv[e], qf[4 * c + e], S, 0, 0, 0);
            }
        }
        float mloc = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + mfma_row(r, hi);
            S[r] = (key < len) ? S[r] * a.scale : -INFINITY;
            mloc = fmaxf(mloc, S[r]);
        }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
        const float m_new = fmaxf(m_run, mloc);
        const float alpha = expf(m_run - m_new);
        float lsum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            S[r] = expf(S[r] - m_new);
            lsum += S[r];
        }
        lsum += __shfl_xor(lsum, 32);
        l_run = l_run * alpha + lsum;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float ar = __shfl(alpha, mfma_row(r, hi));
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) O[dt][r] *= ar;
        }
        const float* vp = base + 2 * a.D + j;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int vr = min(k0 + mfma_row(r, hi), len - 1);
            const float* vrow = vp + (long)vr * ld;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
                O[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(S[r], vrow[32 * dt], O[dt], 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int q = q0 + mfma_row(r, hi);
        const float lr = __shfl(l_run, mfma_row(r, hi));
        if (q < len) {
            float* o = a.out + (long)(start + q) * a.ldo + h * DK + j;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) o[32 * dt] = O[dt][r] / lr;
        }
    }
}

// Split-fp16 variant (default math): the two contractions of k_attention as 3-term split-fp16 MFMA sums
// (v_mfma_f32_32x32x16_f16, fp32 accumulate; softmax and all sums in fp32): 72 x 32-cycle MFMAs per 32-key
// tile instead of 192 x 64-cycle ones.  Q is split once per wave; K, P and V tiles are split in registers.
//   S^T = K . Q^T : A = K rows (lane: key j, k = d0 + 8*hi + e -> 32 contiguous bytes), B = Q rows, same k
//   O   = P . V   : A = P, element e of k-step s is accumulator register 8*s + e of S (K-permutation chaining),
//                   B = V[key(8*s + e, hi)][32*dt + j]
typedef _Float16 at_f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 at_pkh2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void at_split8(const float (&v)[8], at_f16x8& hi, at_f16x8& lo) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        // hi = v_cvt_pkrtz (round toward zero, saturating at +-65504); x - hi exactly by v_fma_mix_f32 on the packed
        // high part; lo = fp16_rne(x - hi): |x - hi - lo| <= 2^-21 |x|
        const at_pkh2 h = __builtin_amdgcn_cvt_pkrtz(v[2 * p], v[2 * p + 1]);
        const unsigned hu = __builtin_bit_cast(unsigned, h);
        float l0, l1;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hu), "v"(v[2 * p]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hu), "v"(v[2 * p + 1]));
        hi[2 * p] = (_Float16)h[0];
        hi[2 * p + 1] = (_Float16)h[1];
        lo[2 * p] = (_Float16)l0;
        lo[2 * p + 1] = (_Float16)l1;
    }
}
// split of 2^k * x (block scaling, pk_split.h)
__device__ __forceinline__ void at_split8s(const float (&v)[8], float s, at_f16x8& hi, at_f16x8& lo) {
    float t[8];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        f32x2 u = {v[2 * p], v[2 * p + 1]};
        u *= s;
        t[2 * p] = u[0];
        t[2 * p + 1] = u[1];
    }
    at_split8(t, hi, lo);
}
// the three block scales of an (utterance, head) and the constants that undo them
struct AtScales {
    float sq, sk, sv;   // 2^kq, 2^kk, 2^kv
    float cs;           // softmax scale / (2^kq 2^kk): S^T accumulators -> logits
    float co;           // 1 / 2^kv: the 2^14 of P cancels against the row sum of the same P
};
__device__ __forceinline__ AtScales at_scales(const AttnArgs& a, int b, int h, int heads) {
    const unsigned* m = a.amax + ((long)b * heads + h) * 3;
    const int kq = blk_scale_exp(m[0]), kk = blk_scale_exp(m[1]), kv = blk_scale_exp(m[2]);
    AtScales s;
    s.sq = pow2f(kq);
    s.sk = pow2f(kk);
    s.sv = pow2f(kv);
    s.cs = a.scale * pow2f(-kq) * pow2f(-kk);
    s.co = pow2f(-kv);
    return s;
}
// Magnitude bounds instead of passes over the activations (see Dense): from the row maxima ham[] that k_layernorm
// leaves for its output h,
//   per (utterance, head): |q|, |k|, |v| <= max_r ham[r] * c1 + c0   -> the attention kernels' block maxima
//   per row: |ctx[r, :]| <= max_head bound_v (a convex combination of the utterance's value rows)
struct QkvBoundC {
    float c1[3 * FS2_MAX_HEADS], c0[3 * FS2_MAX_HEADS];
};
__global__ __launch_bounds__(256) void k_fs2_seg_bounds(const float* __restrict__ ham, const int* __restrict__ seg_start,
                                                        const int* __restrict__ seg_len, int heads, QkvBoundC c,
                                                        unsigned* __restrict__ segb, float* __restrict__ ctx_bound) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const int start = seg_start[b], len = seg_len[b];
    float m = 0.f;
    for (int r = threadIdx.x; r < len; r += 256) m = fmaxf(m, ham[start + r]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float vb = 0.f;
    for (int hd = 0; hd < heads; ++hd) vb = fmaxf(vb, fmaf(m, c.c1[2 * FS2_MAX_HEADS + hd], c.c0[2 * FS2_MAX_HEADS + hd]));
    if (threadIdx.x < 3 * heads) {
        const int part = threadIdx.x / heads, hd = threadIdx.x % heads;
        segb[((long)b * heads + hd) * 3 + part] =
            __float_as_uint(fmaf(m, c.c1[part * FS2_MAX_HEADS + hd], c.c0[part * FS2_MAX_HEADS + hd]));
    }
    for (int r = threadIdx.x; r < len; r += 256) ctx_bound[start + r] = vb;
}
// per row: |relu(conv(h) + b)[r, :]| <= max_tap ham[r + tap] * c1 + c0 (gap rows: 0)
__global__ __launch_bounds__(256) void k_fs2_row_bounds(const float* __restrict__ ham, const int* __restrict__ row_utt,
                                                        int rows, int pad, float c1, float c0,
                                                        float* __restrict__ out) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    float m = 0.f;
    for (int t = -pad; t <= pad; ++t) m = fmaxf(m, ham[r + t]);
    out[r] = row_utt[r] >= 0 ? fmaf(m, c1, c0) : 0.f;
}

// One 32-key tile of the online softmax of the split-fp16 attention kernels (round 6: the vector diet of VERDICT r5 #5).
// S holds the raw S^T accumulators of this lane's query (16 keys per half wave); c2 = cs * log2(e) > 0 turns them into
// logits in units of log2.  On return S = 2^14 p (the block scale of the P operand folded into the exponent), m_run / l_run are
// updated (l_run in the same 2^14 units) and the factor the running sums shrink by is returned.  Per element one v_fma and
// one v_exp_f32 (before: a multiply, a subtract and libm's expf, and a multiply by 2^14); the key mask only in an utterance's
// last tile.  Everything is per QUERY = per lane: with O accumulated transposed (O^T = V^T P^T, below) the rescale needs no
// cross-lane traffic at all (before: 16 ds_bpermute per tile to bring alpha to the accumulator rows).
__device__ __forceinline__ float at_softmax_tile(f32x16& S, float c2, int k0, int hi, int len, float& m_run, float& l_run) {
    if (k0 + 32 > len) {   // (uniform: an utterance's last tile only)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (k0 + mfma_row(r, hi) >= len) S[r] = -INFINITY;
    }
    float mloc = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, S[r]);
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32)) * c2;
    const float m_new = fmaxf(m_run, mloc);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // first tile: 2^-inf = 0
    const float off = (float)PK_UNIT_EXP - m_new;
    float lsum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        S[r] = __builtin_amdgcn_exp2f(fmaf(S[r], c2, off));
        lsum += S[r];
    }
    lsum += __shfl_xor(lsum, 32);
    l_run = fmaf(l_run, alpha, lsum);
    m_run = m_new;
    return alpha;
}
// O^T tile rows are value channels, columns queries: this lane's query row goes out as 4-float pieces
template <int DT>
__device__ __forceinline__ void at_store_out(const f32x16 (&O)[DT], float f, float* o_row, int hi) {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            f32x4 v = {O[dt][4 * rq] * f, O[dt][4 * rq + 1] * f, O[dt][4 * rq + 2] * f, O[dt][4 * rq + 3] * f};
            *reinterpret_cast<f32x4*>(o_row + 32 * dt + 8 * rq + 4 * hi) = v;   // channels mfma_row(4 rq .. 4 rq + 3, hi)
        }
}

__device__ __forceinline__ f32x16 at_mfma3(at_f16x8 ah, at_f16x8 al, at_f16x8 bh, at_f16x8 bl, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0);
}

template <int DK>
__global__ __launch_bounds__(256, 1) void k_attention_h3(AttnArgs a) {
    constexpr int KS = DK / 16;  // k-steps of the QK^T product
    constexpr int DT = DK / 32;  // 32-wide tiles of the value dimension
    const int b = blockIdx.z, h = blockIdx.y;
    const int len = a.seg_len[b], start = a.seg_start[b];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q0 = (blockIdx.x * 4 + wave) * 32;
    if (q0 >= len) return;
    const int j = lane & 31, hi = lane >> 5;
    const long ld = a.ld;
    const float* base = a.qkv + (long)start * ld + h * DK;
    const AtScales sc = at_scales(a, b, h, gridDim.y);
    const float c2 = sc.cs * 1.4426950408889634f;   // accumulator units -> log2 units (at_softmax_tile)

    at_f16x8 qh[KS], ql[KS];
    {
        const int qr = min(q0 + j, len - 1);
        const float* qp = base + (long)qr * ld + 8 * hi;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(qp + 16 * ks);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(qp + 16 * ks + 4);
            const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            at_split8s(v, sc.sq, qh[ks], ql[ks]);
        }
    }
    f32x16 O[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    for (int k0 = 0; k0 < len; k0 += 32) {
        f32x16 S;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.f;
        {
            const int kr = min(k0 + j, len - 1);
            const float* kp = base + a.D + (long)kr * ld + 8 * hi;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(kp + 16 * ks);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(kp + 16 * ks + 4);
                const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                at_f16x8 kh, kl;
                at_split8s(v, sc.sk, kh, kl);
                S = at_mfma3(kh, kl, qh[ks], ql[ks], S);
            }
        }
        const float alpha = at_softmax_tile(S, c2, k0, hi, len, m_run, l_run);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) O[dt][r] *= alpha;   // O^T: this lane's query in every register
        const float* vp = base + 2 * a.D + j;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            float pv[8];
            long voff[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                pv[e] = S[8 * s2 + e];   // 2^14 p
                voff[e] = (long)min(k0 + mfma_row(8 * s2 + e, hi), len - 1) * ld;
            }
            at_f16x8 ph, pl;
            at_split8(pv, ph, pl);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                float vv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) vv[e] = vp[voff[e] + 32 * dt];
                at_f16x8 vh, vl;
                at_split8s(vv, sc.sv, vh, vl);
                O[dt] = at_mfma3(vh, vl, ph, pl, O[dt]);   // O^T += V^T P^T
            }
        }
    }
    if (q0 + j < len)   // O^T = 2^14 2^kv sum p v, l_run = 2^14 sum p: the 2^14 cancels
        at_store_out<DT>(O, sc.co / l_run, a.out + (long)(start + q0 + j) * a.ldo + h * DK, hi);
}

// LDS-staged version of k_attention_h3: one workgroup = 4 waves = 4 query tiles (128 queries) of one
// (utterance, head).  Every 32-key tile of K and V is loaded from HBM/L2 ONCE per workgroup, split into fp16
// (hi, lo) parts by the loading threads and parked in LDS in MFMA fragment order, so the inner loop of a wave is
// ds_read_b128 + MFMA only (the per-wave version loads and splits each tile four times over).  48 KB of LDS:
//   Kf[ks][part][lane]   A fragments of S^T = K . Q^T : 8 halves = K[key = lane&31][16*ks + 8*(lane>>5) + e]
//   Vf[s2][dt][part][lane] B fragments of O = P . V   : 8 halves = V[key(8*s2 + e, lane>>5)][32*dt + (lane&31)]
// The next tile's global loads are issued before the current tile's MFMAs (registers), stored after them.
// PIPE: two sets of fragment buffers -- tile t + 1 is split and stored while tile t is being multiplied (its VALU work fills
// the MFMA shadow instead of standing between two barriers), one barrier per tile instead of two.
constexpr int ATT_THREADS = 256;
// NTHR: 64 x (query tiles per workgroup).  With the two buffer sets one workgroup fits a CU, so the launch runs in rounds of
// n_cu workgroups: the launcher picks 4 or 8 query tiles per workgroup (640-frame utterances, 32 x 2 (utterance, head)
// pairs: 4 tiles -> 320 workgroups = two rounds on 256 CUs, 8 tiles -> 192 = one).
template <int DK, bool PIPE = false, int NTHR = ATT_THREADS>
__global__ __launch_bounds__(NTHR, 1) void k_attention_h3_lds(AttnArgs a) {
    constexpr int KS = DK / 16;
    constexpr int DT = DK / 32;
    constexpr int NT = NTHR;
    constexpr int KG = (32 * (DK / 8) + NT - 1) / NT;      // 8-float groups of the K tile per thread (3 for DK = 192)
    constexpr int VG = (2 * DT * 64 + NT - 1) / NT;        // V fragment lanes per thread (3 for DK = 192)
    // 65 slots per 64-lane fragment block: neighbouring loader threads write different k-steps of the same key, i.e.
    // blocks 2 KB apart -- the same banks without the pad (PMC r01: 48 % of this kernel's LDS cycles were conflicts)
    constexpr int KP = 65;
    constexpr int KSZ = KS * 2 * KP, VSZ = 2 * DT * 2 * 64, NB = PIPE ? 2 : 1;
    __shared__ __attribute__((aligned(16))) at_f16x8 Kf[NB * KSZ];
    __shared__ __attribute__((aligned(16))) at_f16x8 Vf[NB * VSZ];
    const int b = blockIdx.z, h = blockIdx.y;
    const int len = a.seg_len[b], start = a.seg_start[b];
    if ((int)blockIdx.x * (NT / 2) >= len) return;    // uniform over the workgroup
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int q0 = (blockIdx.x * (NT / 64) + wave) * 32;
    const int j = lane & 31, hi = lane >> 5;
    const long ld = a.ld;
    const unsigned ld4 = (unsigned)a.ld * 4u;   // bytes per row
    const float* base = a.qkv + (long)start * ld + h * DK;
    const char* const kbase = reinterpret_cast<const char*>(base + a.D);
    const char* const vbase = reinterpret_cast<const char*>(base + 2 * a.D);
    const AtScales sc = at_scales(a, b, h, gridDim.y);
    const float c2 = sc.cs * 1.4426950408889634f;   // accumulator units -> log2 units (at_softmax_tile)

    at_f16x8 qh[KS], ql[KS];
    {
        const int qr = min(q0 + j, len - 1);
        const float* qp = base + (long)qr * ld + 8 * hi;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(qp + 16 * ks);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(qp + 16 * ks + 4);
            const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            at_split8s(v, sc.sq, qh[ks], ql[ks]);
        }
    }
    f32x16 O[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    float kreg[KG][8], vreg[VG][8];
    // (tl = the thread index the staging coordinates derive from: the loop passes it on top of an opaque zero renewed per key tile,
    // so that the dozen thread-invariant offsets are recomputed -- a few integer instructions per 72 MFMAs -- instead of being
    // hoisted out of the loop, spilled there (17 registers in the 8-tile kernel) and reloaded in every iteration; round 5)
    auto load_tile = [&](int k0, int tl) {
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            const int idx = min(tl + NT * g, 32 * (DK / 8) - 1);   // (key, 8-float group) of the K tile
            const int key = idx / (DK / 8), grp = idx % (DK / 8);
            // (round 6: 32-bit offsets inside the utterance -- a scalar base + an unsigned byte offset instead of a 64-bit
            // multiply-add per address; pk_fft_run_attention checks that an utterance's rows fit)
            const char* kp = kbase + ((unsigned)min(k0 + key, len - 1) * ld4 + 32u * grp);
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(kp);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(kp + 16);
            kreg[g][0] = v0[0]; kreg[g][1] = v0[1]; kreg[g][2] = v0[2]; kreg[g][3] = v0[3];
            kreg[g][4] = v1[0]; kreg[g][5] = v1[1]; kreg[g][6] = v1[2]; kreg[g][7] = v1[3];
        }
#pragma unroll
        for (int g = 0; g < VG; ++g) {
            const int idx = min(tl + NT * g, 2 * DT * 64 - 1);     // (s2, dt, fragment lane) of the V tile
            const int fl = idx & 63, dt = (idx >> 6) % DT, s2 = idx / (64 * DT);
            const unsigned vo = (unsigned)(32 * dt + (fl & 31)) * 4u;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                vreg[g][e] = *reinterpret_cast<const float*>(vbase + ((unsigned)min(k0 + mfma_row(8 * s2 + e, fl >> 5), len - 1) * ld4 + vo));
        }
    };
    auto store_tile = [&](int buf, int tl) {
        at_f16x8* kf = Kf + buf * KSZ;
        at_f16x8* vf = Vf + buf * VSZ;
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            const int idx = tl + NT * g;
            if (idx >= 32 * (DK / 8)) break;
            const int key = idx / (DK / 8), grp = idx % (DK / 8);
            at_f16x8 fh, fl_;
            at_split8s(kreg[g], sc.sk, fh, fl_);
            const int ks = grp >> 1, fl = key + 32 * (grp & 1);
            kf[(ks * 2 + 0) * KP + fl] = fh;
            kf[(ks * 2 + 1) * KP + fl] = fl_;
        }
#pragma unroll
        for (int g = 0; g < VG; ++g) {
            const int idx = tl + NT * g;
            if (idx >= 2 * DT * 64) break;
            const int fl = idx & 63, dt = (idx >> 6) % DT, s2 = idx / (64 * DT);
            at_f16x8 fh, fl_;
            at_split8s(vreg[g], sc.sv, fh, fl_);
            vf[((s2 * DT + dt) * 2 + 0) * 64 + fl] = fh;
            vf[((s2 * DT + dt) * 2 + 1) * 64 + fl] = fl_;
        }
    };

    load_tile(0, tid);
    if (PIPE) {
        store_tile(0, tid);
        load_tile(32, tid);   // (rows are clamped to the utterance: a tile beyond its end is loaded and stored, never multiplied)
    }
    for (int k0 = 0, it = 0; k0 < len; k0 += 32, ++it) {
        const int cur = PIPE ? (it & 1) : 0;
        int oz = 0;
        if (NTHR > ATT_THREADS) asm volatile("" : "+s"(oz));   // (the 8-tile kernel: 256 registers; the 4-tile kernels have 512 and keep their code)
        const int tl = tid + oz;
        __syncthreads();          // every wave is done with the previous tile's fragments (PIPE: and sees this tile's)
        if (PIPE) {
            store_tile(cur ^ 1, tl);
            load_tile(k0 + 64, tl);
        } else {
            store_tile(0, tl);
            __syncthreads();
            if (k0 + 32 < len) load_tile(k0 + 32, tl);
        }
        const at_f16x8* kf = Kf + cur * KSZ;
        const at_f16x8* vf = Vf + cur * VSZ;
        f32x16 S;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            S = at_mfma3(kf[(ks * 2 + 0) * KP + lane], kf[(ks * 2 + 1) * KP + lane], qh[ks], ql[ks], S);
        const float alpha = at_softmax_tile(S, c2, k0, hi, len, m_run, l_run);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) O[dt][r] *= alpha;   // O^T: this lane's query in every register
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            float pv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) pv[e] = S[8 * s2 + e];   // 2^14 p (p <= 1: fixed block scale, folded into the exponent)
            at_f16x8 ph, pl;
            at_split8(pv, ph, pl);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)   // O^T += V^T P^T: the V fragment is the A operand, P (this lane's query) the B operand
                O[dt] = at_mfma3(vf[((s2 * DT + dt) * 2 + 0) * 64 + lane], vf[((s2 * DT + dt) * 2 + 1) * 64 + lane], ph, pl, O[dt]);
        }
    }
    if (q0 >= len) return;
    if (q0 + j < len)   // O^T = 2^14 2^kv sum p v, l_run = 2^14 sum p: the 2^14 cancels
        at_store_out<DT>(O, sc.co / l_run, a.out + (long)(start + q0 + j) * a.ldo + h * DK, hi);
}

// Predictor heads: Linear(C -> 1) per row (+ masked_fill) and, for the duration
// predictor in inference, clip(round(exp(x) - offset), min=0) and the alpha speed
// scaling round(d * alpha)  (duration_predictor.py:95-103, length_regulator.py:85-88;
// paddle.round = half away from zero).
__device__ __forceinline__ float round_half_away(float x) { return copysignf(floorf(fabsf(x) + 0.5f), x); }

__global__ __launch_bounds__(256) void k_rowdot(const float* __restrict__ h, int C, const float* __restrict__ w,
                                                float bias, const int* __restrict__ row_utt, int rows,
                                                int duration_mode, float offset, float alpha,
                                                float* __restrict__ out) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    if (row_utt[r] < 0) {
        if (lane == 0) out[r] = 0.f;
        return;
    }
    const float* x = h + (long)r * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s = fmaf(x[c], w[c], s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    s += bias;
    if (duration_mode) {
        s = fmaxf(round_half_away(expf(s) - offset), 0.f);
        if (alpha != 1.0f) s = round_half_away(s * alpha);
    }
    if (lane == 0) out[r] = s;
}

// Inclusive prefix sum of the integer durations of each utterance; frames[b] = total.
__global__ __launch_bounds__(256) void k_cumsum(const float* __restrict__ dur, const int* __restrict__ seg_start,
                                                const int* __restrict__ seg_len, int* __restrict__ cum,
                                                int* __restrict__ frames) {
    __shared__ int sh[256];
    const int b = blockIdx.x, start = seg_start[b], len = seg_len[b];
    int carry = 0;
    for (int base = 0; base < len; base += 256) {
        const int t = base + threadIdx.x;
        int v = (t < len) ? (int)dur[start + t] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            int add = ((int)threadIdx.x >= o) ? sh[threadIdx.x - o] : 0;
            __syncthreads();
            sh[threadIdx.x] += add;
            __syncthreads();
        }
        if (t < len) cum[start + t] = carry + sh[threadIdx.x];
        carry += sh[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) frames[b] = carry;
}

// Length regulator + variance embeddings + decoder positional encoding, fused:
//   hs2  = hs + (e * w_e + b_e) + (p * w_p + b_p)            fastspeech2.py:426-430 (k=1 convs)
//   up[l] = hs2[token(l)]                                    length_regulator.py:46-66 (row repeat)
//   x[l]  = up[l] * xscale + alpha_dec * PE[l]               decoder embed, fastspeech2.py:250-266
// token(l) = first t with cum[t] > l (binary search in the utterance's prefix sums).
__global__ __launch_bounds__(128) void k_regulate(
    const float* __restrict__ hs, const float* __restrict__ p_out, const float* __restrict__ e_out,
    const float* __restrict__ wp, const float* __restrict__ bp, const float* __restrict__ we,
    const float* __restrict__ be, const int* __restrict__ cum, const int* __restrict__ tseg_start,
    const int* __restrict__ tseg_len, const int* __restrict__ frow_utt, const int* __restrict__ frow_pos,
    const float* __restrict__ pe, float alpha_dec, float xscale, int d, float* __restrict__ x,
    float* __restrict__ hs_up_dbg) {
    const int r = blockIdx.x;
    const int b = frow_utt[r];
    float* xo = x + (long)r * d;
    if (b < 0) {
        for (int c = threadIdx.x; c < d; c += blockDim.x) xo[c] = 0.f;
        return;
    }
    const int l = frow_pos[r];
    const int* cu = cum + tseg_start[b];
    int lo = 0, hi = tseg_len[b] - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cu[mid] > l) hi = mid; else lo = mid + 1;
    }
    const int tr = tseg_start[b] + lo;
    const float pv = p_out[tr], ev = e_out[tr];
    const float* src = hs + (long)tr * d;
    const float* pp = pe + (long)l * d;
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
        const float e_emb = fmaf(ev, we[c], be[c]);
        const float p_emb = fmaf(pv, wp[c], bp[c]);
        const float up = (src[c] + e_emb) + p_emb;
        if (hs_up_dbg) hs_up_dbg[(long)r * d + c] = up;
        xo[c] = up * xscale + alpha_dec * pp[c];
    }
}

}  // namespace

// ================================================================== host side
// Speaker vector per utterance: v[b] = normalize(e_b) . W + bias, e_b = spembs[b] or table[spk_id[b]]
// (zero for the padding id 0); normalize = x / max(||x||_2, 1e-12) (F.normalize, fastspeech2.py:575,580).
// W is [D][A] row-major (the whole spk_projection for "add", its last D rows for "concat").
__global__ __launch_bounds__(256) void k_spk_vec(const long long* __restrict__ spk_id, const float* __restrict__ spembs,
                                                 const float* __restrict__ table, const float* __restrict__ W,
                                                 const float* __restrict__ bias, int D, int A,
                                                 float* __restrict__ v) {
    extern __shared__ float e[];   // D floats + 1
    const int b = blockIdx.x;
    const float* src = spembs ? spembs + (long)b * D : table + (long)spk_id[b] * D;
    const bool zero = !spembs && spk_id[b] == 0;
    float ss = 0.f;
    for (int i = threadIdx.x; i < D; i += blockDim.x) {
        const float x = zero ? 0.f : src[i];
        e[i] = x;
        ss += x * x;
    }
    __shared__ float red[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float inv = 1.0f / fmaxf(sqrtf(red[0] + red[1] + red[2] + red[3]), 1e-12f);
    for (int c = threadIdx.x; c < A; c += blockDim.x) {
        float acc = 0.f;
        for (int i = 0; i < D; ++i) acc = fmaf(e[i] * inv, W[(long)i * A + c], acc);
        v[(long)b * A + c] = acc + bias[c];
    }
}

// y[r] = x[r] + v[row_utt[r]] on the rows of the timeline that belong to an utterance
__global__ __launch_bounds__(256) void k_add_rowvec(const float* __restrict__ x, const float* __restrict__ v,
                                                    const int* __restrict__ row_utt, int rows, int A,
                                                    float* __restrict__ y) {
    const int r = blockIdx.x;
    const int b = row_utt[r];
    if (b < 0) return;
    for (int c = threadIdx.x; c < A; c += blockDim.x) y[(long)r * A + c] = x[(long)r * A + c] + v[(long)b * A + c];
}

// reduction_factor r > 1: feat_out gives r frames per decoder row ([frame 0 | ... | frame r-1], fastspeech2.py:457:
// .reshape((B, -1, odim))).  Frame row q of the frame timeline (utterance u, position p) takes columns (p % r) * O .. of
// decoder row dec_seg_start[u] + p / r; gap rows are zeroed when rowmap == NULL (the postnet convolves over them).
__global__ __launch_bounds__(128) void k_fs2_unfold_r(const float* __restrict__ wide, int O, int r,
                                                      const int* __restrict__ dec_seg_start, const int* __restrict__ row_utt,
                                                      const int* __restrict__ row_pos, const int* __restrict__ rowmap,
                                                      const float* __restrict__ cscale, const float* __restrict__ cshift,
                                                      float* __restrict__ dst) {
    const long q = blockIdx.x;
    const int u = row_utt[q];
    const long o = rowmap ? rowmap[q] : q;
    if (o < 0) return;
    if (u < 0) {
        if (!rowmap)
            for (int c = threadIdx.x; c < O; c += blockDim.x) dst[o * O + c] = 0.f;
        return;
    }
    const int p = row_pos[q];
    const float* s = wide + ((long)(dec_seg_start[u] + p / r) * r + p % r) * O;
    for (int c = threadIdx.x; c < O; c += blockDim.x) {
        float v = s[c];
        if (cscale) v = v * cscale[c] + cshift[c];
        dst[o * O + c] = v;
    }
}

// hs[r] += ptone[tone[r]] on the rows of the timeline that belong to an utterance
__global__ __launch_bounds__(256) void k_add_tone(float* __restrict__ hs, const float* __restrict__ ptone,
                                                  const int* __restrict__ tone, const int* __restrict__ row_utt,
                                                  int A) {
    const int r = blockIdx.x;
    if (row_utt[r] < 0) return;
    const float* src = ptone + (long)tone[r] * A;
    for (int c = threadIdx.x; c < A; c += blockDim.x) hs[(long)r * A + c] += src[c];
}

typedef pk_fft_dense Dense;
typedef pk_fft_layer FftLayer;
typedef pk_fft_timeline Timeline;

struct Predictor {
    std::vector<Dense> conv;
    std::vector<size_t> ln_g, ln_b;
    size_t lin_w;
    float lin_b;
    int chans;
};

// hs[r] += v[utterance of r] on the rows of a timeline (v: [B][adim])
int pk_fft_add_rowvec(pk_fft_core* h, const pk_fft_timeline& tl, const float* d_vec, float* hs) {
    PK_LAUNCH(h->ctx, "fft_add_rowvec", k_add_rowvec, dim3(tl.rows), dim3(256), 0, hs, d_vec, tl.d_row_utt(), tl.rows, h->adim, hs);
    return PK_OK;
}

// _integrate_with_spk_embed (fastspeech2.py:560-586, transformer_tts.py:725-755) on the rows of a timeline:
//   "add":    hs += normalize(e_b) . W + bias
//   "concat": hs  = hs . W[:A] + (normalize(e_b) . W[A:] + bias)      (hs_proj = the [A][A] part; tmp: A-wide rows)
// e_b = d_spembs[b] (B x D) or table[d_spk_id[b]].
int pk_fft_run_speaker(pk_fft_core* h, const pk_fft_timeline& tl, const long long* d_spk_id, const float* d_spembs,
                       size_t table, size_t w, size_t bias, const pk_fft_dense* hs_proj, int D, pk_dbuf& d_vec, float* hs,
                       float* tmp) {
    pk_ctx* ctx = h->ctx;
    const int A = h->adim, B = tl.B;
    PK_TRY(d_vec.reserve((size_t)B * A * sizeof(float)));
    PK_LAUNCH(ctx, "fft_spk_vec", k_spk_vec, dim3(B), dim3(256), (size_t)D * sizeof(float), d_spk_id, d_spembs, h->W(table),
              h->W(w), h->W(bias), D, A, d_vec.as<float>());
    const float* src = hs;
    if (hs_proj) {
        PK_TRY(pk_fft_run_dense(h, "fft_gemm_spk_proj", *hs_proj, hs, A, tmp, A, tl.rows, PK_ACT_NONE, nullptr, 0,
                                tl.d_row_utt()));
        src = tmp;
    }
    PK_LAUNCH(ctx, "fft_add_rowvec", k_add_rowvec, dim3(tl.rows), dim3(256), 0, src, d_vec.as<float>(), tl.d_row_utt(), tl.rows,
              A, hs);
    return PK_OK;
}

struct pk_fs2 : pk_fft_core {
    pk_fs2_cfg cfg;
    pk_param_map params;
    bool finalized = false;
    int gapr = 2;
    // weights (arena, math mode, positional table and the FFT-stack buffers: pk_fft_core)
    size_t emb_table = 0, enc_after_g = 0, enc_after_b = 0, dec_after_g = 0, dec_after_b = 0;
    float alpha_enc = 1.f, alpha_dec = 1.f, xscale = 1.f;
    std::vector<FftLayer> enc, dec;
    Predictor dur, pitch, energy;
    size_t pitch_w = 0, pitch_b = 0, energy_w = 0, energy_b = 0;
    Dense feat_out;
    std::vector<Dense> postnet;
    size_t tone_table = 0;                         // [num_tones][A]: tone_projection(normalize(embedding row))
    std::vector<long long> cond_tone;              // conditioning of the next encode (pk_fs2_set_tones)
    pk_dbuf d_tone;
    size_t spk_table = 0, spk_w = 0, spk_b = 0;   // embedding table, [D][A] speaker part of spk_projection, bias
    Dense spk_hs;                                  // "concat": the [A][A] hidden-state part of spk_projection
    std::vector<long long> cond_spk;               // conditioning of the next encode (pk_fs2_set_speakers)
    std::vector<float> cond_emb;
    int cond_B = 0;
    pk_dbuf d_spk_id, d_spk_emb, d_spk_vec;
    size_t out_scale = 0, out_shift = 0;
    bool has_out_affine = false;
    std::vector<float> h_out_scale, h_out_shift;
    // per-call state
    Timeline tl_tok, tl_frm, tl_frm2;   // tokens, decoder rows, mel frames (= decoder rows unless reduction_factor > 1)
    pk_dbuf d_wide, d_rowmap2;
    pk_dbuf d_pamax;   // row maxima of the predictors' LayerNorm outputs
    pk_dbuf d_hsp, d_hsam, d_pp, d_ppam;   // planes path of the predictors: the encoder output as planes, a layer's work planes
    char* hsp = nullptr;
    unsigned* hsam = nullptr;
    bool hs_planes_valid = false;           // ... converted once per pk_fs2_encode
    pk_dbuf d_tok, d_p1, d_p2, d_hs, d_pout, d_eout, d_dout, d_cum, d_frames,
        d_before, d_q1, d_q2, d_rowmap, d_dbg_up, d_zs, d_mel_stage;
    std::vector<int> frames;   // per utterance, result of encode
    bool encoded = false;
    bool debug = false;
};

static const int LEAD = PK_FFT_LEAD;  // rows of margin in front of every activation buffer

int pk_fft_act_reserve(pk_dbuf& buf, int rows, int C) {
    const size_t r = (size_t)((rows + PK_GEMM_BM - 1) / PK_GEMM_BM) * PK_GEMM_BM + 2 * LEAD;
    return buf.reserve(r * C * sizeof(float));
}

int pk_fft_build_timeline(pk_ctx* ctx, Timeline& tl, const int* lens, int B, int gapr) {
    tl.B = B;
    tl.seg_start.resize(B);
    tl.seg_len.assign(lens, lens + B);
    int r = gapr;
    for (int b = 0; b < B; ++b) {
        tl.seg_start[b] = r;
        r += lens[b] + gapr;
    }
    tl.rows = r;
    tl.rows_alloc = ((r + PK_GEMM_BM - 1) / PK_GEMM_BM) * PK_GEMM_BM;
    tl.row_utt.assign(tl.rows_alloc, -1);
    tl.row_pos.assign(tl.rows_alloc, 0);
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < lens[b]; ++t) {
            tl.row_utt[tl.seg_start[b] + t] = b;
            tl.row_pos[tl.seg_start[b] + t] = t;
        }
    std::vector<int> tab;
    tab.insert(tab.end(), tl.seg_start.begin(), tl.seg_start.end());
    tab.insert(tab.end(), tl.seg_len.begin(), tl.seg_len.end());
    tab.insert(tab.end(), tl.row_utt.begin(), tl.row_utt.end());
    tab.insert(tab.end(), tl.row_pos.begin(), tl.row_pos.end());
    return pk_upload(ctx, tl.d_tab, tab.data(), tab.size() * sizeof(int));
}

extern "C" int pk_fs2_create(pk_ctx* ctx, const pk_fs2_cfg* cfg, pk_fs2** out) {
    if (!ctx || !cfg || !out) PK_FAIL(PK_EINVAL, "pk_fs2_create: NULL argument");
    *out = nullptr;
    const pk_fs2_cfg& c = *cfg;
    if (c.idim <= 0 || c.odim <= 0 || c.adim <= 0 || c.aheads <= 0)
        PK_FAIL(PK_EINVAL, "FastSpeech2: idim/odim/adim/aheads must be positive");
    if (c.adim % c.aheads != 0) PK_FAIL(PK_ESHAPE, "FastSpeech2: adim %% aheads != 0 (attention.py:40)");
    const int dk = c.adim / c.aheads;
    if (dk != 64 && dk != 96 && dk != 128 && dk != 192)
        PK_FAIL(PK_EUNSUPPORTED, "FastSpeech2: head size %d not built (64/96/128/192)", dk);
    if (c.adim % 64 != 0 || c.adim > 64 * LN_MAXPER)
        PK_FAIL(PK_EUNSUPPORTED, "FastSpeech2: adim must be a multiple of 64, <= %d", 64 * LN_MAXPER);
    if (c.reduction_factor < 1 || c.reduction_factor > 16) PK_FAIL(PK_EUNSUPPORTED, "FastSpeech2: reduction_factor must be in [1, 16]");
    if (c.pitch_embed_kernel_size != 1 || c.energy_embed_kernel_size != 1)
        PK_FAIL(PK_EUNSUPPORTED, "FastSpeech2: pitch/energy_embed_kernel_size must be 1 (all reference recipes)");
    if (c.tone_embed_dim < 0 || c.num_tones < 0) PK_FAIL(PK_EINVAL, "FastSpeech2: negative tone sizes");
    if (c.tone_embed_dim > 0 && c.num_tones <= 0) PK_FAIL(PK_EINVAL, "FastSpeech2: tone_embed_dim needs num_tones");
    if (c.tone_embed_dim > 0 && c.tone_embed_integration_type != 0)
        PK_FAIL(PK_EUNSUPPORTED, "FastSpeech2: tone_embed_integration_type 'concat' is not implemented "
                                 "(the reference's branch cannot broadcast 1-D tone ids, fastspeech2.py:606-610)");
    if (c.positionwise_layer_type < 0 || c.positionwise_layer_type > 2)
        PK_FAIL(PK_EUNSUPPORTED, "Support only linear or conv1d. (encoder.py:169)");
    if (c.spk_embed_dim < 0 || c.num_speakers < 0) PK_FAIL(PK_EINVAL, "FastSpeech2: negative speaker sizes");
    if (c.spk_embed_dim > 0 && c.spk_embed_integration_type != 0 && c.spk_embed_integration_type != 1)
        PK_FAIL(PK_EUNSUPPORTED, "support only add or concat. (fastspeech2.py:584)");
    if (c.spk_embed_dim > 8192) PK_FAIL(PK_EUNSUPPORTED, "FastSpeech2: spk_embed_dim > 8192");
    if (c.postnet_layers > 0 && !c.use_batch_norm)
        PK_FAIL(PK_EUNSUPPORTED, "FastSpeech2: postnet without batch norm not implemented");
    const int ks[] = {c.positionwise_conv_kernel_size, c.duration_predictor_kernel_size,
                      c.pitch_predictor_kernel_size, c.energy_predictor_kernel_size,
                      c.postnet_layers > 0 ? c.postnet_filts : 1};
    int gapr = 1;
    for (int k : ks) {
        if (k < 1 || k % 2 == 0 || k > 15) PK_FAIL(PK_EUNSUPPORTED, "FastSpeech2: conv kernel size %d unsupported", k);
        gapr = std::max(gapr, (k - 1) / 2);
    }
    const int chans[] = {c.adim, c.eunits, c.dunits, c.duration_predictor_chans, c.pitch_predictor_chans,
                         c.energy_predictor_chans, c.postnet_layers > 0 ? c.postnet_chans : 16, c.odim};
    for (int ch : chans)
        if (ch % PK_GEMM_BK != 0) PK_FAIL(PK_EUNSUPPORTED, "FastSpeech2: channel count %d not a multiple of 16", ch);
    const int pch[] = {c.duration_predictor_chans, c.pitch_predictor_chans, c.energy_predictor_chans};
    for (int ch : pch)
        if (ch % 64 != 0 || ch > 64 * LN_MAXPER)
            PK_FAIL(PK_EUNSUPPORTED, "FastSpeech2: predictor channels must be a multiple of 64, <= %d", 64 * LN_MAXPER);
    pk_fs2* h = new pk_fs2();
    h->ctx = ctx;
    h->cfg = c;
    h->adim = c.adim;
    h->aheads = c.aheads;
    h->attn_lds = pk_prof_env("PK_FS2_ATTN_NO_LDS") == nullptr;
    h->no_bounds = pk_prof_env("PK_FS2_NO_BOUNDS") != nullptr;
    h->gapr = gapr;
    h->ffn_planes_min_blocks = FFNP_MIN_BLOCKS;
    if (const char* e = pk_prof_env("PK_FS2_FFN_PLANES")) h->ffn_planes = e[0] != '0';
    if (const char* e = pk_prof_env("PK_FS2_FFN_PLANES_MIN_BLOCKS")) h->ffn_planes_min_blocks = atoi(e);
    if (const char* e = pk_prof_env("PK_FFNP_VARIANT")) h->ffnp_variant = atoi(e);
    if (const char* e = pk_prof_env("PK_FS2_ATTN_WAVES")) h->attn_waves = atoi(e);
    if (const char* e = pk_prof_env("PK_FS2_MATH")) h->math = strcmp(e, "f32") == 0 ? PK_GEMM_MATH_F32 : PK_GEMM_MATH_F16X3;
    if (gapr > LEAD) { delete h; PK_FAIL(PK_EUNSUPPORTED, "conv kernel too wide"); }
    *out = h;
    return PK_OK;
}

extern "C" int pk_fs2_set_param(pk_fs2* h, const char* name, const float* data, const int64_t* shape,
                                int32_t ndim) {
    if (!h) PK_FAIL(PK_EINVAL, "pk_fs2_set_param: handle is NULL");
    h->finalized = false;
    return pk_store_param(h->params, name, data, shape, ndim);
}

extern "C" int pk_fs2_set_normalizer(pk_fs2* h, const float* mu, const float* sigma, int32_t n) {
    if (!h) PK_FAIL(PK_EINVAL, "pk_fs2_set_normalizer: handle is NULL");
    if (!mu && !sigma) {
        h->has_out_affine = false;
    } else {
        if (!mu || !sigma || n != h->cfg.odim) PK_FAIL(PK_ESHAPE, "normalizer needs mu and sigma of odim elements");
        h->h_out_scale.assign(sigma, sigma + n);
        h->h_out_shift.assign(mu, mu + n);
        h->has_out_affine = true;
    }
    h->finalized = false;
    return PK_OK;
}

typedef pk_fft_arena Arena;

// c1 = max over columns [n0, n1) of sum_k |W[k, n]|, c0 = max |bias[n]| (see Dense)
static void dense_bound(const std::vector<float>& kn, const std::vector<float>* bias, int K, int N, int n0, int n1,
                        float& c1, float& c0) {
    double m1 = 0.0, m0 = 0.0;
    for (int n = n0; n < n1; ++n) {
        double s = 0.0;
        for (int k = 0; k < K; ++k) s += std::fabs((double)kn[(size_t)k * N + n]);
        m1 = std::max(m1, s);
        if (bias) m0 = std::max(m0, std::fabs((double)(*bias)[n]));
    }
    c1 = (float)(m1 * (1.0 + 1e-6));
    c0 = (float)(m0 * (1.0 + 1e-6));
}

int pk_fft_add_dense_kn(Arena& ar, const std::vector<float>& kn, const std::vector<float>* bias, int Cin, int taps,
                 int N, Dense& d) {
    std::vector<float> packed;
    pk_gemm_pack(kn.data(), Cin * taps, N, packed);
    d.w = ar.put(packed);
    if (ar.v16 && Cin % PK_GEMM_HBK == 0) {
        std::vector<uint16_t> ph;
        pk_gemm_pack_h3(kn.data(), Cin * taps, N, ph);
        d.wh = ar.put16(ph);
    }
    d.b = bias ? ar.put(*bias) : (size_t)-1;
    d.Cin = Cin;
    d.N = N;
    d.taps = taps;
    d.pad = (taps - 1) / 2;
    dense_bound(kn, bias, Cin * taps, N, 0, N, d.c1, d.c0);
    // the 256-channel convs of the variance predictors and of the postnets (384 | 256 -> 256, k = 3 | 5) also get the planes
    // kernel's fragments (ffnp_conv256_launch)
    if (ar.v16 && ffnp_conv256_supports(Cin, N, taps)) {
        std::vector<float> ws;
        d.wp = ffnp_pack(kn.data(), Cin, N, FFNP_NQ2, *ar.v16, ws, taps);
        d.wp1 = ffnp_pack(kn.data(), Cin, N, 1, *ar.v16, ws, taps);   // (one tile per wave: short timelines)
        d.wps = ar.put(ws);
    }
    return PK_OK;
}

int pk_fft_add_conv(Arena& ar, const pk_param_map& P, const std::string& base, int Cout, int Cin, int k, bool bias,
             Dense& d, int planes) {
    std::vector<float> w, kn, b;
    PK_TRY(pk_get_weight(P, base, {Cout, Cin, k}, w));
    pk_conv_to_kn(w.data(), Cout, Cin, k, kn);
    if (bias) PK_TRY(pk_get_vector(P, base + ".bias", Cout, b));
    PK_TRY(pk_fft_add_dense_kn(ar, kn, bias ? &b : nullptr, Cin, k, Cout, d));
    if (planes && ar.v16 && k == FFNP_TAPS) {
        std::vector<float> ws;
        d.wp = ffnp_pack(kn.data(), Cin, Cout, planes == 1 ? FFNP_NQ1 : FFNP_NQ2, *ar.v16, ws);
        if (planes == 1) d.wp4 = ffnp_pack(kn.data(), Cin, Cout, FFNP_NQ2, *ar.v16, ws);   // (the same scales: per 32 channels)
        d.wp1 = ffnp_pack(kn.data(), Cin, Cout, 1, *ar.v16, ws);
        d.wps = ar.put(ws);
    }
    return PK_OK;
}

int pk_fft_add_vec(Arena& ar, const pk_param_map& P, const std::string& name, int n, size_t& off) {
    std::vector<float> v;
    PK_TRY(pk_get_vector(P, name, n, v));
    off = ar.put(v);
    return PK_OK;
}

int pk_fft_add_linear(Arena& ar, const pk_param_map& P, const std::string& base, int Cin, int N, Dense& d) {
    std::vector<float> w, b;
    PK_TRY(pk_get_weight(P, base, {Cin, N}, w));   // Linear weight [in, out]
    PK_TRY(pk_get_vector(P, base + ".bias", N, b));
    return pk_fft_add_dense_kn(ar, w, &b, Cin, 1, N, d);
}

int pk_fft_add_conv_bn(Arena& ar, const pk_param_map& P, const std::string& conv_base, const std::string& bn_base,
                       int Cout, int Cin, int k, Dense& d, bool conv_bias) {
    std::vector<float> w, g, b, mean, var, kn, bias(Cout), cb(Cout, 0.f);
    PK_TRY(pk_get_weight(P, conv_base, {Cout, Cin, k}, w));
    if (conv_bias) PK_TRY(pk_get_vector(P, conv_base + ".bias", Cout, cb));
    PK_TRY(pk_get_vector(P, bn_base + ".weight", Cout, g));
    PK_TRY(pk_get_vector(P, bn_base + ".bias", Cout, b));
    PK_TRY(pk_get_vector(P, bn_base + "._mean", Cout, mean));
    PK_TRY(pk_get_vector(P, bn_base + "._variance", Cout, var));
    // fold BatchNorm1D (eval, eps 1e-5) into the bias-free conv (tacotron2/decoder.py:133-147)
    const size_t per = (size_t)Cin * k;
    for (int o = 0; o < Cout; ++o) {
        const double s = (double)g[o] / std::sqrt((double)var[o] + 1e-5);
        for (size_t i = 0; i < per; ++i) w[o * per + i] = (float)((double)w[o * per + i] * s);
        bias[o] = (float)((double)b[o] + ((double)cb[o] - (double)mean[o]) * s);
    }
    pk_conv_to_kn(w.data(), Cout, Cin, k, kn);
    return pk_fft_add_dense_kn(ar, kn, &bias, Cin, k, Cout, d);
}

int pk_fft_add_postnet(Arena& ar, const pk_param_map& P, const std::string& prefix, int n_layers, int odim, int chans,
                       int filts, std::vector<Dense>& out) {
    out.resize(n_layers);
    for (int j = 0; j < n_layers; ++j) {
        const int cin = j == 0 ? odim : chans;
        const int cout = j == n_layers - 1 ? odim : chans;
        const std::string p = prefix + ".postnet." + std::to_string(j);
        PK_TRY(pk_fft_add_conv_bn(ar, P, p + ".0", p + ".1", cout, cin, filts, out[j]));
    }
    return PK_OK;
}

int pk_fft_add_stack(Arena& ar, const pk_param_map& P, const std::string& prefix, int n_layers, int A, int units,
                  int k, int ff_type, int heads, std::vector<FftLayer>& out, size_t& after_g, size_t& after_b,
                  bool normalize_before, bool concat_after) {
    out.resize(n_layers);
    // pre-norm stacks with conv feed-forward layers of the built shape also get the planes-kernel fragments (pk_ffn_planes.h)
    const bool planes = normalize_before && ffnp_supports(A, units, k, k);
    for (int l = 0; l < n_layers; ++l) {
        const std::string p = prefix + ".encoders." + std::to_string(l);
        FftLayer& L = out[l];
        PK_TRY(pk_fft_add_vec(ar, P, p + ".norm1.weight", A, L.ln1_g));
        PK_TRY(pk_fft_add_vec(ar, P, p + ".norm1.bias", A, L.ln1_b));
        PK_TRY(pk_fft_add_vec(ar, P, p + ".norm2.weight", A, L.ln2_g));
        PK_TRY(pk_fft_add_vec(ar, P, p + ".norm2.bias", A, L.ln2_b));
        // fused q|k|v projection: Linear weights are [in, out] (paddle)
        std::vector<float> wq, wk, wv, bq, bk, bv, kn((size_t)A * 3 * A), bias(3 * A);
        PK_TRY(pk_get_weight(P, p + ".self_attn.linear_q", {A, A}, wq));
        PK_TRY(pk_get_weight(P, p + ".self_attn.linear_k", {A, A}, wk));
        PK_TRY(pk_get_weight(P, p + ".self_attn.linear_v", {A, A}, wv));
        PK_TRY(pk_get_vector(P, p + ".self_attn.linear_q.bias", A, bq));
        PK_TRY(pk_get_vector(P, p + ".self_attn.linear_k.bias", A, bk));
        PK_TRY(pk_get_vector(P, p + ".self_attn.linear_v.bias", A, bv));
        for (int i = 0; i < A; ++i)
            for (int o = 0; o < A; ++o) {
                kn[(size_t)i * 3 * A + o] = wq[(size_t)i * A + o];
                kn[(size_t)i * 3 * A + A + o] = wk[(size_t)i * A + o];
                kn[(size_t)i * 3 * A + 2 * A + o] = wv[(size_t)i * A + o];
            }
        for (int o = 0; o < A; ++o) {
            bias[o] = bq[o];
            bias[A + o] = bk[o];
            bias[2 * A + o] = bv[o];
        }
        PK_TRY(pk_fft_add_dense_kn(ar, kn, &bias, A, 1, 3 * A, L.qkv));
        if (planes && !concat_after && ar.v16 && (3 * A) % (32 * FFNP_NQL) == 0)
        {
            std::vector<float> ws;
            L.qkv.wp = ffnp_pack(kn.data(), A, 3 * A, FFNP_NQL, *ar.v16, ws, 1);
            L.qkv.wps = ar.put(ws);
        }
        for (int part = 0; part < 3; ++part)
            for (int hd = 0; hd < heads && hd < FS2_MAX_HEADS; ++hd)
                dense_bound(kn, &bias, A, 3 * A, part * A + hd * (A / heads), part * A + (hd + 1) * (A / heads),
                            L.qkv_c1[part * FS2_MAX_HEADS + hd], L.qkv_c0[part * FS2_MAX_HEADS + hd]);
        std::vector<float> wo, bo;
        PK_TRY(pk_get_weight(P, p + ".self_attn.linear_out", {A, A}, wo));
        PK_TRY(pk_get_vector(P, p + ".self_attn.linear_out.bias", A, bo));
        PK_TRY(pk_fft_add_dense_kn(ar, wo, &bo, A, 1, A, L.out));
        if (planes && !concat_after && ar.v16 && A % (32 * FFNP_NQ2) == 0) {
            std::vector<float> ws;
            L.out.wp = ffnp_pack(wo.data(), A, A, FFNP_NQ2, *ar.v16, ws, 1);
            L.out.wps = ar.put(ws);
        }
        L.concat = concat_after;
        if (concat_after) {
            // concat_linear: Linear(2A -> A) on cat(x, attention output) = x . W[:A] + att . W[A:] + b (encoder_layer.py:103-106)
            std::vector<float> w, b;
            PK_TRY(pk_get_weight(P, p + ".concat_linear", {2 * A, A}, w));
            PK_TRY(pk_get_vector(P, p + ".concat_linear.bias", A, b));
            std::vector<float> wx(w.begin(), w.begin() + (size_t)A * A), wa(w.begin() + (size_t)A * A, w.end());
            PK_TRY(pk_fft_add_dense_kn(ar, wx, &b, A, 1, A, L.cat_x));
            PK_TRY(pk_fft_add_dense_kn(ar, wa, nullptr, A, 1, A, L.cat_a));
        }
        // position-wise layer (encoder.py:145-170): conv1d = (k, k), conv1d-linear = (k, Linear), linear = 2 x Linear
        if (ff_type == 1) {
            std::vector<float> w, b;
            PK_TRY(pk_get_weight(P, p + ".feed_forward.w_1", {A, units}, w));
            PK_TRY(pk_get_vector(P, p + ".feed_forward.w_1.bias", units, b));
            PK_TRY(pk_fft_add_dense_kn(ar, w, &b, A, 1, units, L.ffn1));
        } else {
            PK_TRY(pk_fft_add_conv(ar, P, p + ".feed_forward.w_1", units, A, k, true, L.ffn1, planes && ff_type == 0 ? 1 : 0));
        }
        if (ff_type == 0) {
            PK_TRY(pk_fft_add_conv(ar, P, p + ".feed_forward.w_2", A, units, k, true, L.ffn2, planes ? 2 : 0));
        } else {
            std::vector<float> w, b;
            PK_TRY(pk_get_weight(P, p + ".feed_forward.w_2", {units, A}, w));
            PK_TRY(pk_get_vector(P, p + ".feed_forward.w_2.bias", A, b));
            PK_TRY(pk_fft_add_dense_kn(ar, w, &b, units, 1, A, L.ffn2));
        }
    }
    if (normalize_before) {   // after_norm exists only then (encoder.py:142-143)
        PK_TRY(pk_fft_add_vec(ar, P, prefix + ".after_norm.weight", A, after_g));
        PK_TRY(pk_fft_add_vec(ar, P, prefix + ".after_norm.bias", A, after_b));
    }
    return PK_OK;
}

namespace {
int add_predictor(Arena& ar, const pk_param_map& P, const std::string& prefix, int n_layers, int A, int chans,
                  int k, Predictor& pr) {
    pr.conv.resize(n_layers);
    pr.ln_g.resize(n_layers);
    pr.ln_b.resize(n_layers);
    pr.chans = chans;
    for (int j = 0; j < n_layers; ++j) {
        const std::string p = prefix + ".conv." + std::to_string(j);
        PK_TRY(pk_fft_add_conv(ar, P, p + ".0", chans, j == 0 ? A : chans, k, true, pr.conv[j]));
        PK_TRY(pk_fft_add_vec(ar, P, p + ".2.weight", chans, pr.ln_g[j]));
        PK_TRY(pk_fft_add_vec(ar, P, p + ".2.bias", chans, pr.ln_b[j]));
    }
    PK_TRY(pk_fft_add_vec(ar, P, prefix + ".linear.weight", chans, pr.lin_w));
    std::vector<float> b;
    PK_TRY(pk_get_vector(P, prefix + ".linear.bias", 1, b));
    pr.lin_b = b[0];
    return PK_OK;
}
}  // namespace

int pk_fft_ensure_pe(pk_fft_core* h, int need) {
    if (need <= h->max_len) return PK_OK;
    pk_ctx* ctx = h->ctx;
    const int d = h->adim;
    int n = std::max(need, 1024);
    n = std::max(n, h->max_len * 2);
    if (!h->d_div.p) {
        // div_term = exp(arange(0, d, 2) * -(log(10000)/d)) in float32 (embedding.py:56-58)
        std::vector<float> div(d / 2);
        const float cst = (float)(-(std::log(10000.0) / d));
        for (int i = 0; i < d / 2; ++i) div[i] = expf((float)(2 * i) * cst);
        PK_TRY(pk_upload(ctx, h->d_div, div.data(), div.size() * sizeof(float)));
    }
    PK_TRY(h->d_pe.reserve((size_t)n * d * sizeof(float)));
    PK_LAUNCH(ctx, "fs2_build_pe", k_build_pe, dim3(n), dim3(128), 0, h->d_pe.as<float>(), h->d_div.as<float>(), n, d);
    h->max_len = n;
    return PK_OK;
}

extern "C" int pk_fs2_finalize(pk_fs2* h) {
    if (!h) PK_FAIL(PK_EINVAL, "pk_fs2_finalize: handle is NULL");
    pk_ctx* ctx = h->ctx;
    PK_DEVICE(ctx->device);
    const pk_fs2_cfg& c = h->cfg;
    const pk_param_map& P = h->params;
    const int A = c.adim;
    h->arena_h.clear();
    h->arena16_h.clear();
    Arena ar{h->arena_h, &h->arena16_h};
    {
        std::vector<float> t;
        PK_TRY(pk_get_weight(P, "encoder.embed.0", {c.idim, A}, t));
        for (int i = 0; i < A; ++i) t[i] = 0.f;  // nn.Embedding(padding_idx=0): id 0 -> zero row
        h->emb_table = ar.put(t);
    }
    std::vector<float> al;
    if (c.use_scaled_pos_enc) {
        PK_TRY(pk_get_vector(P, "encoder.embed.1.alpha", 1, al));
        h->alpha_enc = al[0];
        PK_TRY(pk_get_vector(P, "decoder.embed.0.alpha", 1, al));
        h->alpha_dec = al[0];
        h->xscale = 1.f;
    } else {
        h->alpha_enc = h->alpha_dec = 1.f;
        h->xscale = std::sqrt((float)A);  // PositionalEncoding.forward embedding.py:78
    }
    PK_TRY(pk_fft_add_stack(ar, P, "encoder", c.elayers, A, c.eunits, c.positionwise_conv_kernel_size, c.positionwise_layer_type, c.aheads, h->enc,
                         h->enc_after_g, h->enc_after_b, c.encoder_normalize_before != 0, c.encoder_concat_after != 0));
    PK_TRY(pk_fft_add_stack(ar, P, "decoder", c.dlayers, A, c.dunits, c.positionwise_conv_kernel_size, c.positionwise_layer_type, c.aheads, h->dec,
                         h->dec_after_g, h->dec_after_b, c.decoder_normalize_before != 0, c.decoder_concat_after != 0));
    PK_TRY(add_predictor(ar, P, "duration_predictor", c.duration_predictor_layers, A, c.duration_predictor_chans,
                         c.duration_predictor_kernel_size, h->dur));
    PK_TRY(add_predictor(ar, P, "pitch_predictor", c.pitch_predictor_layers, A, c.pitch_predictor_chans,
                         c.pitch_predictor_kernel_size, h->pitch));
    PK_TRY(add_predictor(ar, P, "energy_predictor", c.energy_predictor_layers, A, c.energy_predictor_chans,
                         c.energy_predictor_kernel_size, h->energy));
    PK_TRY(pk_fft_add_vec(ar, P, "pitch_embed.0.weight", A, h->pitch_w));
    PK_TRY(pk_fft_add_vec(ar, P, "pitch_embed.0.bias", A, h->pitch_b));
    PK_TRY(pk_fft_add_vec(ar, P, "energy_embed.0.weight", A, h->energy_w));
    PK_TRY(pk_fft_add_vec(ar, P, "energy_embed.0.bias", A, h->energy_b));
    {
        std::vector<float> w, b;
        const int OR = c.odim * c.reduction_factor;   // feat_out: adim -> odim * reduction_factor (fastspeech2.py:271)
        PK_TRY(pk_get_weight(P, "feat_out", {A, OR}, w));
        PK_TRY(pk_get_vector(P, "feat_out.bias", OR, b));
        PK_TRY(pk_fft_add_dense_kn(ar, w, &b, A, 1, OR, h->feat_out));
    }
    PK_TRY(pk_fft_add_postnet(ar, P, "postnet", c.postnet_layers, c.odim, c.postnet_chans, c.postnet_filts, h->postnet));
    if (c.tone_embed_dim > 0) {
        // "add": hs[t] += Linear(F.normalize(E[tone[t]])) is a function of the tone id alone -> one table
        const int Dt = c.tone_embed_dim;
        std::vector<float> e, w, b, tab((size_t)c.num_tones * A);
        PK_TRY(pk_get_weight(P, "tone_embedding_table", {c.num_tones, Dt}, e));
        PK_TRY(pk_get_weight(P, "tone_projection", {Dt, A}, w));
        PK_TRY(pk_get_vector(P, "tone_projection.bias", A, b));
        for (int k = 0; k < c.num_tones; ++k) {
            double ss = 0.0;
            if (k != 0)   // nn.Embedding(padding_idx=0) returns zeros for id 0
                for (int i = 0; i < Dt; ++i) ss += (double)e[(size_t)k * Dt + i] * e[(size_t)k * Dt + i];
            const double inv = 1.0 / std::max(std::sqrt(ss), 1e-12);
            for (int o = 0; o < A; ++o) {
                double acc = 0.0;
                if (k != 0)
                    for (int i = 0; i < Dt; ++i) acc += (double)e[(size_t)k * Dt + i] * inv * w[(size_t)i * A + o];
                tab[(size_t)k * A + o] = (float)(acc + b[o]);
            }
        }
        h->tone_table = ar.put(tab);
    }
    if (c.spk_embed_dim > 0) {
        const int D = c.spk_embed_dim;
        std::vector<float> t, w, b;
        if (c.num_speakers > 0) {
            PK_TRY(pk_get_weight(P, "spk_embedding_table", {c.num_speakers, D}, t));
            h->spk_table = ar.put(t);
        }
        PK_TRY(pk_get_vector(P, "spk_projection.bias", A, b));
        h->spk_b = ar.put(b);
        if (c.spk_embed_integration_type == 0) {
            PK_TRY(pk_get_weight(P, "spk_projection", {D, A}, w));
            h->spk_w = ar.put(w);
        } else {
            // Linear(adim + D -> adim) on concat([hs, e]) = hs . W[:adim] + e . W[adim:] (+ bias, added with e's part)
            PK_TRY(pk_get_weight(P, "spk_projection", {A + D, A}, w));
            std::vector<float> whs(w.begin(), w.begin() + (size_t)A * A), wsp(w.begin() + (size_t)A * A, w.end());
            PK_TRY(pk_fft_add_dense_kn(ar, whs, nullptr, A, 1, A, h->spk_hs));
            h->spk_w = ar.put(wsp);
        }
    }
    if (h->has_out_affine) {
        h->out_scale = ar.put(h->h_out_scale);
        h->out_shift = ar.put(h->h_out_shift);
    }
    PK_TRY(pk_upload(ctx, h->arena, h->arena_h.data(), h->arena_h.size() * sizeof(float)));
    h->arena_h.clear();
    h->arena_h.shrink_to_fit();
    if (!h->arena16_h.empty())
        PK_TRY(pk_upload(ctx, h->arena16, h->arena16_h.data(), h->arena16_h.size() * sizeof(uint16_t)));
    h->arena16_h.clear();
    h->arena16_h.shrink_to_fit();
    PK_TRY(pk_fft_ensure_pe(h, 1024));
    h->finalized = true;
    h->encoded = false;
    return PK_OK;
}

int pk_fft_run_dense(pk_fft_core* h, const char* name, const Dense& d, const float* A, int lda, float* C, int ldc,
                     int rows, int act, const float* res, int ldr, const int* rowvalid, const float* a_amax) {
    pk_gemm_args g;
    g.a_amax = a_amax;   // row maxima of A when its producer left them (k_layernorm), else computed by the launcher
    g.A = A;
    g.lda = lda;
    g.Wp = h->W(d.w);
    g.Wh = d.wh == (size_t)-1 ? nullptr : h->arena16.as<uint16_t>() + d.wh;
    g.math = h->math;
    g.bias = d.b == (size_t)-1 ? nullptr : h->W(d.b);
    g.res = res;
    g.ldr = ldr;
    g.C = C;
    g.ldc = ldc;
    g.rowvalid = rowvalid;
    g.M = rows;
    g.N = d.N;
    g.Cin = d.Cin;
    g.taps = d.taps;
    g.pad = d.pad;
    g.act = act;
    return pk_gemm_launch(h->ctx, name, g);
}

int pk_fft_run_layernorm(pk_fft_core* h, const float* x, size_t g, size_t b, const Timeline& tl, int C, float* y,
                         float* amax) {
    PK_LAUNCH(h->ctx, "fs2_layernorm", k_layernorm, dim3(pk_div_up(tl.rows, 4)), dim3(256), 0, x, h->W(g), h->W(b),
              tl.d_row_utt(), tl.rows, C, 1e-5f, y, amax);
    return PK_OK;
}

int pk_fft_run_attention(pk_fft_core* h, const Timeline& tl, const float* qkv, float* out,
                         const unsigned* seg_bounds) {
    const int A = h->adim, heads = h->aheads, dk = A / heads;
    int maxlen = 0;
    for (int l : tl.seg_len) maxlen = std::max(maxlen, l);
    AttnArgs a;
    a.qkv = qkv;
    a.ld = 3 * A;
    a.out = out;
    a.ldo = A;
    a.seg_start = tl.d_seg_start();
    a.seg_len = tl.d_seg_len();
    a.D = A;
    a.scale = (float)(1.0 / std::sqrt((double)dk));
    a.amax = seg_bounds;   // bounds on |q|, |k|, |v| per (utterance, head) from k_fs2_seg_bounds, when the caller has them
    if (h->math == PK_GEMM_MATH_F16X3 && !seg_bounds) {   // else: the block maxima themselves, one pass over qkv
        pk_ctx_scratch* sc = pk_ctx_get_scratch(h->ctx);
        const size_t nb = (size_t)tl.B * heads * 3 * sizeof(unsigned);
        PK_TRY(sc->attn_amax.reserve(nb));
        PK_HIP(hipMemsetAsync(sc->attn_amax.p, 0, nb, h->ctx->stream));
        PK_LAUNCH(h->ctx, "fs2_qkv_amax", k_qkv_amax, dim3(pk_div_up(maxlen, 32), heads, tl.B), dim3(256), 0, qkv,
                  3 * A, tl.d_seg_start(), tl.d_seg_len(), A, dk, sc->attn_amax.as<unsigned>());
        a.amax = sc->attn_amax.as<unsigned>();
    }
    dim3 grid(pk_div_up(maxlen, 128), heads, tl.B);
    if ((long)maxlen * 3 * A * 4 >= (1L << 32)) PK_FAIL(PK_EUNSUPPORTED, "attention: an utterance of %d rows exceeds the 32-bit row offsets", maxlen);
    if (h->math == PK_GEMM_MATH_F16X3 && h->attn_lds) {
        dim3 g2(pk_div_up(maxlen, ATT_THREADS / 2), heads, tl.B);
        // (measured, 32 x 640 frames: 180 -> 170 us per decoder launch; PK_FS2_ATTN_PIPE=0: the two-barrier loop)
        static const bool pipe = !(pk_prof_env("PK_FS2_ATTN_PIPE") && pk_prof_env("PK_FS2_ATTN_PIPE")[0] == '0');
        if (pipe && dk == 192) {
            // query tiles per workgroup: 4, or 8 where that saves rounds of n_cu workgroups.  Measured per decoder launch (32 x 2
            // pairs of 20 query tiles): 4 tiles 171 us (320 workgroups, two rounds), 5: 148, 6: 138, 8: 132 (192 workgroups);
            // from 5 waves on the kernel has 256 registers instead of 512 and spills 17-34 of them, a workgroup alone takes
            // 1.55 x as long -- the encoder's 4-tile utterances stay with 4 (24 vs 30 us).
            const int wenv = h->attn_waves;   // measurement override ("attn_waves" option)
            const long r4 = pk_div_up((long)pk_div_up(maxlen, 128) * heads * tl.B, h->ctx->n_cu);
            const long r8 = pk_div_up((long)pk_div_up(maxlen, 256) * heads * tl.B, h->ctx->n_cu);
            int best = 100 * r4 <= 155 * r8 ? 4 : 8;
            if (wenv == 4 || wenv == 8) best = wenv;
            const dim3 gw(pk_div_up(maxlen, 32 * best), heads, tl.B);
            auto go = [&](auto kern) -> int {
                PK_LAUNCH(h->ctx, "fs2_attention_h3", kern, gw, dim3(64 * best), 0, a);
                return PK_OK;
            };
            return best == 8 ? go(k_attention_h3_lds<192, true, 512>) : go(k_attention_h3_lds<192, true, 256>);
        }
        switch (dk) {
            case 64: PK_LAUNCH(h->ctx, "fs2_attention_h3", k_attention_h3_lds<64>, g2, dim3(ATT_THREADS), 0, a); break;
            case 96: PK_LAUNCH(h->ctx, "fs2_attention_h3", k_attention_h3_lds<96>, g2, dim3(ATT_THREADS), 0, a); break;
            case 128: PK_LAUNCH(h->ctx, "fs2_attention_h3", k_attention_h3_lds<128>, g2, dim3(ATT_THREADS), 0, a); break;
            case 192: PK_LAUNCH(h->ctx, "fs2_attention_h3", k_attention_h3_lds<192>, g2, dim3(ATT_THREADS), 0, a); break;
            default: PK_FAIL(PK_EUNSUPPORTED, "attention head size %d", dk);
        }
        return PK_OK;
    }
    if (h->math == PK_GEMM_MATH_F16X3) {
        switch (dk) {
            case 64: PK_LAUNCH(h->ctx, "fs2_attention_h3", k_attention_h3<64>, grid, dim3(256), 0, a); break;
            case 96: PK_LAUNCH(h->ctx, "fs2_attention_h3", k_attention_h3<96>, grid, dim3(256), 0, a); break;
            case 128: PK_LAUNCH(h->ctx, "fs2_attention_h3", k_attention_h3<128>, grid, dim3(256), 0, a); break;
            case 192: PK_LAUNCH(h->ctx, "fs2_attention_h3", k_attention_h3<192>, grid, dim3(256), 0, a); break;
            default: PK_FAIL(PK_EUNSUPPORTED, "attention head size %d", dk);
        }
        return PK_OK;
    }
    switch (dk) {
        case 64: PK_LAUNCH(h->ctx, "fs2_attention", k_attention<64>, grid, dim3(256), 0, a); break;
        case 96: PK_LAUNCH(h->ctx, "fs2_attention", k_attention<96>, grid, dim3(256), 0, a); break;
        case 128: PK_LAUNCH(h->ctx, "fs2_attention", k_attention<128>, grid, dim3(256), 0, a); break;
        case 192: PK_LAUNCH(h->ctx, "fs2_attention", k_attention<192>, grid, dim3(256), 0, a); break;
        default: PK_FAIL(PK_EUNSUPPORTED, "attention head size %d", dk);
    }
    return PK_OK;
}

// N FFT blocks + after_norm on the timeline tl; x is updated in place, result in hs.
// Post-norm blocks (normalize_before = False, encoder_layer.py:64-115): x = norm1(x + att(x)); x = norm2(x + ffn(x)); no
// after_norm.  The LayerNorms ping-pong between the two A-wide row buffers (their output is the next residual stream);
// operand scales of the split-fp16 GEMMs come from passes over the data (the LayerNorm-based magnitude bounds of the
// pre-norm path do not apply to layer 0's input).
// Several small clears as ONE launch.  A hipMemsetAsync is a kernel launch of its own, and a FastSpeech2 call issued 43 of them (rocprofv3,
// round 6): the six clears of a stack and the three of a planes work buffer each go out together now.
namespace {
constexpr int ZERO_LIST_MAX = 8;
struct ZeroList {
    unsigned* p[ZERO_LIST_MAX];
    unsigned words[ZERO_LIST_MAX];
};
__global__ __launch_bounds__(256) void k_zero_list(ZeroList z) {
    unsigned* p = z.p[blockIdx.y];
    const unsigned n = z.words[blockIdx.y];
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) p[i] = 0u;
}
struct ZeroItems {
    ZeroList z;
    int n = 0;
    int add(void* p, size_t bytes) {
        if (!p || bytes == 0) return PK_OK;
        // (whole 32-bit words: every user is an array of floats / unsigneds; a pk_dbuf's capacity may end in a few spare bytes)
        if (n == ZERO_LIST_MAX || bytes > ((size_t)1 << 33)) PK_FAIL(PK_EINVAL, "zero_list: %d entries / %zu bytes", n, bytes);
        z.p[n] = static_cast<unsigned*>(p);
        z.words[n++] = (unsigned)(bytes >> 2);
        return PK_OK;
    }
    int launch(pk_ctx* ctx) {
        if (n == 0) return PK_OK;
        for (int i = n; i < ZERO_LIST_MAX; ++i) { z.p[i] = nullptr; z.words[i] = 0; }
        PK_LAUNCH(ctx, "zero_list", k_zero_list, dim3(32, n), dim3(256), 0, z);
        return PK_OK;
    }
};
}  // namespace

static int run_stack_postnorm(pk_fft_core* h, const std::vector<FftLayer>& layers, const Timeline& tl, int units, float* hs_out) {
    const int A = h->adim;
    PK_TRY(pk_fft_act_reserve(h->d_h, tl.rows, A));
    PK_TRY(pk_fft_act_reserve(h->d_qkv, tl.rows, 3 * A));
    PK_TRY(pk_fft_act_reserve(h->d_ctx, tl.rows, A));
    PK_TRY(pk_fft_act_reserve(h->d_f, tl.rows, units));
    PK_TRY(pk_fft_act_reserve(h->d_cat, tl.rows, A));
    float* cur = pk_fft_act_ptr(h->d_x, A);
    float* alt = pk_fft_act_ptr(h->d_h, A);
    float* qkv = pk_fft_act_ptr(h->d_qkv, 3 * A);
    float* ctxb = pk_fft_act_ptr(h->d_ctx, A);
    float* f = pk_fft_act_ptr(h->d_f, units);
    float* t = pk_fft_act_ptr(h->d_cat, A);
    const int* rv = tl.d_row_utt();
    if (layers.empty()) {
        PK_HIP(hipMemcpyAsync(hs_out, cur, (size_t)tl.rows * A * sizeof(float), hipMemcpyDeviceToDevice, h->ctx->stream));
        return PK_OK;
    }
    for (size_t li = 0; li < layers.size(); ++li) {
        const FftLayer& L = layers[li];
        PK_TRY(pk_fft_run_dense(h, "fs2_gemm_qkv", L.qkv, cur, A, qkv, 3 * A, tl.rows, PK_ACT_NONE, nullptr, 0, nullptr));
        PK_TRY(pk_fft_run_attention(h, tl, qkv, ctxb, nullptr));
        if (L.concat) {
            // alt = cur + concat_linear(cat(cur, att)); a GEMM must not write the rows it reads, hence the second buffer
            PK_TRY(pk_fft_run_dense(h, "fs2_gemm_attn_out", L.out, ctxb, A, t, A, tl.rows, PK_ACT_NONE, nullptr, 0, nullptr));
            PK_TRY(pk_fft_run_dense(h, "fs2_gemm_concat_x", L.cat_x, cur, A, alt, A, tl.rows, PK_ACT_NONE, cur, A, nullptr));
            PK_TRY(pk_fft_run_dense(h, "fs2_gemm_concat_a", L.cat_a, t, A, alt, A, tl.rows, PK_ACT_NONE, alt, A, nullptr));
            PK_TRY(pk_fft_run_layernorm(h, alt, L.ln1_g, L.ln1_b, tl, A, cur));
        } else {
            PK_TRY(pk_fft_run_dense(h, "fs2_gemm_attn_out", L.out, ctxb, A, cur, A, tl.rows, PK_ACT_NONE, cur, A, nullptr));
            PK_TRY(pk_fft_run_layernorm(h, cur, L.ln1_g, L.ln1_b, tl, A, alt));
            std::swap(cur, alt);
        }
        PK_TRY(pk_fft_run_dense(h, "fs2_conv_ffn1", L.ffn1, cur, A, f, units, tl.rows, PK_ACT_RELU, nullptr, 0, rv));
        PK_TRY(pk_fft_run_dense(h, "fs2_conv_ffn2", L.ffn2, f, units, cur, A, tl.rows, PK_ACT_NONE, cur, A, nullptr));
        float* dst = li + 1 == layers.size() ? hs_out : alt;
        PK_TRY(pk_fft_run_layernorm(h, cur, L.ln2_g, L.ln2_b, tl, A, dst));
        std::swap(cur, alt);   // (after the last layer the pointers are dead)
    }
    return PK_OK;
}

int pk_fft_run_stack(pk_fft_core* h, const std::vector<FftLayer>& layers, size_t after_g, size_t after_b,
                         const Timeline& tl, int units, float* hs_out, bool normalize_before) {
    if (!normalize_before) return run_stack_postnorm(h, layers, tl, units, hs_out);
    const int A = h->adim;
    PK_TRY(pk_fft_act_reserve(h->d_h, tl.rows, A));
    PK_TRY(pk_fft_act_reserve(h->d_qkv, tl.rows, 3 * A));
    PK_TRY(pk_fft_act_reserve(h->d_ctx, tl.rows, A));
    PK_TRY(pk_fft_act_reserve(h->d_f, tl.rows, units));
    float* x = pk_fft_act_ptr(h->d_x, A);
    float* hh = pk_fft_act_ptr(h->d_h, A);
    float* qkv = pk_fft_act_ptr(h->d_qkv, 3 * A);
    float* ctxb = pk_fft_act_ptr(h->d_ctx, A);
    float* f = pk_fft_act_ptr(h->d_f, units);
    const int* rv = tl.d_row_utt();
    // row maxima of the LayerNorm outputs, left by k_layernorm for the split-fp16 GEMMs that read them (rows outside
    // the timeline: zero)
    // ... and magnitude bounds derived from them for the tensors in between (see Dense): no pass over qkv, the
    // attention output or the FFN hidden activations is needed for the operand scales
    float *ham = nullptr, *cbnd = nullptr, *fbnd = nullptr;
    unsigned* segb = nullptr;
    ZeroItems zl;   // the per-run clears below, one launch
    const int heads = h->aheads;
    const bool bounds = h->math == PK_GEMM_MATH_F16X3 && heads <= FS2_MAX_HEADS && !h->no_bounds;
    if (h->math == PK_GEMM_MATH_F16X3) {
        PK_TRY(pk_fft_act_reserve(h->d_lnamax, tl.rows, 1));
        PK_TRY(zl.add(h->d_lnamax.p, h->d_lnamax.cap));
        ham = pk_fft_act_ptr(h->d_lnamax, 1);
    }
    if (bounds) {
        PK_TRY(pk_fft_act_reserve(h->d_cbnd, tl.rows, 1));
        PK_TRY(pk_fft_act_reserve(h->d_fbnd, tl.rows, 1));
        PK_TRY(h->d_segb.reserve((size_t)tl.B * heads * 3 * sizeof(unsigned)));
        PK_TRY(zl.add(h->d_cbnd.p, h->d_cbnd.cap));
        PK_TRY(zl.add(h->d_fbnd.p, h->d_fbnd.cap));
        cbnd = pk_fft_act_ptr(h->d_cbnd, 1);
        fbnd = pk_fft_act_ptr(h->d_fbnd, 1);
        segb = h->d_segb.as<unsigned>();
    }
    // norm2 and the two feed-forward convs on pre-split planes (pk_ffn_planes.h) where pk_fft_add_stack packed for them
    const bool planes = h->math == PK_GEMM_MATH_F16X3 && !layers.empty() && layers[0].ffn1.wp != (size_t)-1 &&
                        layers[0].ffn2.wp != (size_t)-1 && tl.rows_alloc % FFNP_BLK == 0 &&
                        h->ffn_planes && tl.rows_alloc / FFNP_BLK >= h->ffn_planes_min_blocks;
    const int nblk = tl.rows_alloc / FFNP_BLK;
    char *hp = nullptr, *fp = nullptr;
    unsigned *hpam = nullptr, *fpam = nullptr;
    if (planes) {
        // one block / one element of margin on either side (the +-1 taps of the edge tiles); the leading block is zero from the
        // allocation on, the maxima are cleared per run (what lies behind the last block only reaches gap rows)
        pk_dbuf* pb[2] = {&h->d_hp, &h->d_fp};
        const int pc[2] = {A, units};
        for (int i = 0; i < 2; ++i) {
            const void* p0 = pb[i]->p;
            PK_TRY(pb[i]->reserve(ffnp_plane_bytes(nblk, pc[i])));
            if (pb[i]->p != p0) PK_HIP(hipMemsetAsync(pb[i]->p, 0, pb[i]->cap, h->ctx->stream));
        }
        PK_TRY(h->d_pam.reserve((size_t)2 * (tl.rows_alloc + 2) * sizeof(unsigned)));
        PK_TRY(zl.add(h->d_pam.p, (size_t)2 * (tl.rows_alloc + 2) * sizeof(unsigned)));
        hp = h->d_hp.as<char>() + (size_t)A * 128;
        fp = h->d_fp.as<char>() + (size_t)units * 128;
        // the margin block BEHIND the last block may hold planes of an earlier, longer timeline: cleared per run, so that the
        // +1 taps of the last tile read zeros whatever ran before (the leading margin block is never written)
        PK_TRY(zl.add(hp + (size_t)nblk * A * 128, (size_t)A * 128));
        PK_TRY(zl.add(fp + (size_t)nblk * units * 128, (size_t)units * 128));
        hpam = h->d_pam.as<unsigned>() + 1;   // row maxima (fp32 bits), one element of margin on either side
        fpam = hpam + tl.rows_alloc + 2;
    }
    PK_TRY(zl.launch(h->ctx));
    for (const FftLayer& L : layers) {
        const bool qkv_planes = planes && bounds && !L.concat && L.qkv.wp != (size_t)-1;
        if (qkv_planes) {
            // norm1 -> planes, the fused q | k | v projection on them; the segment bounds from the row maxima it leaves
            PK_TRY(ffnp_layernorm_launch(h->ctx, x, h->W(L.ln1_g), h->W(L.ln1_b), rv, nblk, A, 1e-5f, hp, hpam));
            FfnpConv c;
            memset(&c, 0, sizeof(c));
            c.nblk = nblk;
            c.row_utt = rv;
            c.w = h->arena16.as<uint16_t>() + L.qkv.wp;
            c.bias = L.qkv.b == (size_t)-1 ? nullptr : h->W(L.qkv.b);
            c.wscale = h->W(L.qkv.wps); c.Cin = A; c.N = 3 * A;
            c.in = hp; c.in_amax = hpam;
            c.x = qkv; c.ldx = 3 * A;
            PK_TRY(ffnp_linear_launch(h->ctx, "fs2_gemm_qkv_planes", c));
        } else {
            PK_TRY(pk_fft_run_layernorm(h, x, L.ln1_g, L.ln1_b, tl, A, hh, ham));
            PK_TRY(pk_fft_run_dense(h, "fs2_gemm_qkv", L.qkv, hh, A, qkv, 3 * A, tl.rows, PK_ACT_NONE, nullptr, 0, nullptr, ham));
        }
        if (bounds) {
            QkvBoundC qc;
            memcpy(qc.c1, L.qkv_c1, sizeof(qc.c1));
            memcpy(qc.c0, L.qkv_c0, sizeof(qc.c0));
            PK_LAUNCH(h->ctx, "fs2_bounds", k_fs2_seg_bounds, dim3(tl.B), dim3(256), 0,
                      qkv_planes ? reinterpret_cast<const float*>(hpam) : ham, tl.d_seg_start(), tl.d_seg_len(), heads, qc, segb,
                      cbnd);
        }
        PK_TRY(pk_fft_run_attention(h, tl, qkv, ctxb, segb));
        if (L.concat) {
            // x = x + concat_linear(cat(norm1(x), att)) (encoder_layer.py:103-106)
            PK_TRY(pk_fft_act_reserve(h->d_cat, tl.rows, A));
            float* t = pk_fft_act_ptr(h->d_cat, A);
            PK_TRY(pk_fft_run_dense(h, "fs2_gemm_attn_out", L.out, ctxb, A, t, A, tl.rows, PK_ACT_NONE, nullptr, 0, nullptr, cbnd));
            PK_TRY(pk_fft_run_dense(h, "fs2_gemm_concat_x", L.cat_x, hh, A, x, A, tl.rows, PK_ACT_NONE, x, A, nullptr, ham));
            PK_TRY(pk_fft_run_dense(h, "fs2_gemm_concat_a", L.cat_a, t, A, x, A, tl.rows, PK_ACT_NONE, x, A, nullptr));
        } else if (qkv_planes && L.out.wp != (size_t)-1) {
            // x += attention output . W_out + b on the planes kernel, its operand the fp32 rows the attention kernel wrote
            // (split in registers with the row bound |ctx| <= max|v| as scale)
            FfnpConv c;
            memset(&c, 0, sizeof(c));
            c.nblk = nblk;
            c.row_utt = rv;
            c.w = h->arena16.as<uint16_t>() + L.out.wp;
            c.wscale = h->W(L.out.wps);
            c.bias = L.out.b == (size_t)-1 ? nullptr : h->W(L.out.b);
            c.Cin = A; c.N = A;
            c.in = ctxb; c.ldin = A; c.in_amax = reinterpret_cast<const unsigned*>(cbnd);
            c.x = x; c.ldx = A;
            PK_TRY(ffnp_linear_launch(h->ctx, "fs2_gemm_attn_out_planes", c));
        } else
        PK_TRY(pk_fft_run_dense(h, "fs2_gemm_attn_out", L.out, ctxb, A, x, A, tl.rows, PK_ACT_NONE, x, A, nullptr, cbnd));
        if (planes) {
            PK_TRY(ffnp_layernorm_launch(h->ctx, x, h->W(L.ln2_g), h->W(L.ln2_b), rv, nblk, A, 1e-5f, hp, hpam));
            FfnpConv c;
            memset(&c, 0, sizeof(c));
            c.nblk = nblk;
            c.row_utt = rv;
            c.w = h->arena16.as<uint16_t>() + L.ffn1.wp;
            c.w4 = L.ffn1.wp4 == (size_t)-1 ? nullptr : h->arena16.as<uint16_t>() + L.ffn1.wp4;
            c.w1 = L.ffn1.wp1 == (size_t)-1 ? nullptr : h->arena16.as<uint16_t>() + L.ffn1.wp1;
            c.bias = L.ffn1.b == (size_t)-1 ? nullptr : h->W(L.ffn1.b);
            c.wscale = h->W(L.ffn1.wps); c.Cin = A; c.N = units;
            c.in = hp; c.in_amax = hpam;
            c.out = fp; c.out_amax = fpam; c.c1 = L.ffn1.c1; c.c0 = L.ffn1.c0;
            c.variant = h->ffnp_variant;
            c.one_max = h->ffn_one_tile_max;
            PK_TRY(ffnp_conv_launch(h->ctx, "fs2_conv_ffn1_planes", c));
            c.w = h->arena16.as<uint16_t>() + L.ffn2.wp;
            c.w4 = nullptr;
            c.w1 = L.ffn2.wp1 == (size_t)-1 ? nullptr : h->arena16.as<uint16_t>() + L.ffn2.wp1;
            c.bias = L.ffn2.b == (size_t)-1 ? nullptr : h->W(L.ffn2.b);
            c.wscale = h->W(L.ffn2.wps); c.Cin = units; c.N = A;
            c.in = fp; c.in_amax = fpam;
            c.out = nullptr; c.out_amax = nullptr;
            c.x = x; c.ldx = A;
            PK_TRY(ffnp_conv_launch(h->ctx, "fs2_conv_ffn2_planes", c));
            continue;
        }
        PK_TRY(pk_fft_run_layernorm(h, x, L.ln2_g, L.ln2_b, tl, A, hh, ham));
        PK_TRY(pk_fft_run_dense(h, "fs2_conv_ffn1", L.ffn1, hh, A, f, units, tl.rows, PK_ACT_RELU, nullptr, 0, rv, ham));
        if (bounds)
            PK_LAUNCH(h->ctx, "fs2_bounds", k_fs2_row_bounds, dim3(pk_div_up(tl.rows, 256)), dim3(256), 0, ham, rv,
                      tl.rows, L.ffn1.pad, L.ffn1.c1, L.ffn1.c0, fbnd);
        PK_TRY(pk_fft_run_dense(h, "fs2_conv_ffn2", L.ffn2, f, units, x, A, tl.rows, PK_ACT_NONE, x, A, nullptr, fbnd));
    }
    PK_TRY(pk_fft_run_layernorm(h, x, after_g, after_b, tl, A, hs_out));
    return PK_OK;
}

// A planes work buffer of C channels for a timeline of nblk blocks with its row maxima: one block / FFNP_AM_MARGIN elements of
// ZERO margin on either side (what the edge tiles' taps read), the rest written by the producers.
constexpr int FFNP_AM_MARGIN = 4;
static int planes_buf(pk_fft_core* h, pk_dbuf& buf, pk_dbuf& am, int nblk, int C, char** planes, unsigned** amax) {
    const size_t pb = ffnp_plane_bytes(nblk, C), ab = ((size_t)nblk * FFNP_BLK + 2 * FFNP_AM_MARGIN) * sizeof(unsigned);
    PK_TRY(buf.reserve(pb));
    PK_TRY(am.reserve(ab));
    ZeroItems zl;
    PK_TRY(zl.add(buf.p, (size_t)C * 128));                                            // leading margin block
    PK_TRY(zl.add(buf.as<char>() + (size_t)(nblk + 1) * C * 128, (size_t)C * 128));   // trailing
    PK_TRY(zl.add(am.p, ab));
    PK_TRY(zl.launch(h->ctx));
    *planes = buf.as<char>() + (size_t)C * 128;
    *amax = am.as<unsigned>() + FFNP_AM_MARGIN;
    return PK_OK;
}

static FfnpConv conv256_args(pk_fft_core* h, const Dense& d, int nblk, const int* row_utt, const void* in, const unsigned* in_amax) {
    FfnpConv c;
    memset(&c, 0, sizeof(c));
    c.nblk = nblk;
    c.row_utt = row_utt;
    c.w = h->arena16.as<uint16_t>() + d.wp;
    c.w1 = d.wp1 == (size_t)-1 ? nullptr : h->arena16.as<uint16_t>() + d.wp1;
    c.one_max = h->ffn_one_tile_max;
    c.variant = h->ffnp_variant;
    c.bias = d.b == (size_t)-1 ? nullptr : h->W(d.b);
    c.wscale = h->W(d.wps);
    c.Cin = d.Cin;
    c.N = d.N;
    c.in = in;
    c.in_amax = in_amax;
    return c;
}

int pk_fft_run_postnet(pk_fft_core* h, const char* name, const std::vector<Dense>& postnet, const float* before, int odim,
                       int chans, const Timeline& tl, pk_dbuf& q1, pk_dbuf& q2, float* d_out, const int* out_rowmap,
                       const float* cscale, const float* cshift) {
    const int n = (int)postnet.size();
    PK_TRY(pk_fft_act_reserve(q1, tl.rows, chans));
    PK_TRY(pk_fft_act_reserve(q2, tl.rows, chans));
    const float* in = before;
    int ldin = odim;
    // operand scale of the layers that read tanh outputs: |tanh| <= 1 is its own bound -- a constant array of ones in place of a
    // k_row_amax pass per layer (refilled only when the buffer grows)
    const float* ones = nullptr;
    if (h->math == PK_GEMM_MATH_F16X3 && n > 1) {
        const void* p0 = h->d_ones.p;
        PK_TRY(pk_fft_act_reserve(h->d_ones, tl.rows, 1));
        if (h->d_ones.p != p0)
            PK_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(h->d_ones.p), 0x3f800000, h->d_ones.cap / 4, h->ctx->stream));
        ones = pk_fft_act_ptr(h->d_ones, 1);
    }
    // Round 4: the middle layers (256 -> 256 channels, k = 5, BatchNorm folded, tanh) on the planes kernel: the first layer's
    // fp32 rows -> planes, layers 1 .. n - 3 planes -> planes (|tanh| <= 1: scale 2^13, row maximum 1), layer n - 2 planes ->
    // fp32 rows for the last layer (whose epilogue -- residual, ZScore, row map -- stays the tile GEMM's)
    bool planes = h->math == PK_GEMM_MATH_F16X3 && h->ffn_planes && n >= 4 && tl.rows_alloc % FFNP_BLK == 0 && chans == 256;
    for (int j = 1; j + 1 < n; ++j)
        planes = planes && postnet[j].wp != (size_t)-1 && postnet[j].wps != (size_t)-1 && postnet[j].Cin == 256 && postnet[j].taps == 5;
    for (int j = 0; j < n; ++j) {
        const Dense& d = postnet[j];
        const bool last = j == n - 1;
        float* outb = pk_fft_act_ptr((j & 1) ? q2 : q1, chans);
        if (planes && j == 1) {
            const int nblk = tl.rows_alloc / FFNP_BLK;
            const int* rv = tl.d_row_utt();
            char* pl[2];
            unsigned* am[2];
            PK_TRY(planes_buf(h, h->d_pnp[0], h->d_pnam[0], nblk, chans, &pl[0], &am[0]));
            PK_TRY(planes_buf(h, h->d_pnp[1], h->d_pnam[1], nblk, chans, &pl[1], &am[1]));
            PK_TRY(ffnp_layernorm_launch(h->ctx, in, nullptr, nullptr, rv, nblk, chans, 0.f, pl[0], am[0]));
            int cur = 0;
            for (; j + 1 < n; ++j) {
                FfnpConv c = conv256_args(h, postnet[j], nblk, rv, pl[cur], am[cur]);
                const std::string nm = std::string(name) + "_planes";
                if (j + 2 < n) {   // planes -> planes
                    c.out = pl[cur ^ 1];
                    c.out_amax = am[cur ^ 1];
                    c.c1 = 0.f;
                    c.c0 = 1.f;
                    PK_TRY(ffnp_conv256_launch(h->ctx, nm.c_str(), c, 5, 1));
                    cur ^= 1;
                } else {           // planes -> fp32 rows, the last layer's input
                    outb = pk_fft_act_ptr((j & 1) ? q2 : q1, chans);
                    c.x = outb;
                    c.ldx = chans;
                    PK_TRY(ffnp_conv256_launch(h->ctx, nm.c_str(), c, 5, 1));
                    in = outb;
                    ldin = chans;
                }
            }
            --j;   // j == n - 2 was the last planes layer: the loop header moves on to n - 1
            continue;
        }
        pk_gemm_args g;
        g.A = in; g.lda = ldin; g.Wp = h->W(d.w); g.bias = h->W(d.b);
        g.Wh = d.wh == (size_t)-1 ? nullptr : h->arena16.as<uint16_t>() + d.wh; g.math = h->math;
        g.M = tl.rows; g.N = d.N; g.Cin = d.Cin; g.taps = d.taps; g.pad = d.pad;
        g.rowvalid = tl.d_row_utt();
        if (j > 0) g.a_amax = ones;
        if (!last) {
            g.C = outb; g.ldc = chans; g.act = PK_ACT_TANH;
        } else {
            g.C = d_out; g.ldc = odim; g.act = PK_ACT_NONE; g.res = before; g.ldr = odim;
            g.cscale = cscale; g.cshift = cshift; g.out_rowmap = out_rowmap;
        }
        PK_TRY(pk_gemm_launch(h->ctx, name, g));
        in = outb;
        ldin = chans;
    }
    return PK_OK;
}

int pk_fft_embed(pk_fft_core* h, const char* name, const int* d_tok, const Timeline& tl, size_t table, float alpha,
                 float xscale, float* x) {
    PK_LAUNCH(h->ctx, name, k_embed, dim3(tl.rows), dim3(128), 0, d_tok, tl.d_row_utt(), tl.d_row_pos(), h->W(table),
              h->d_pe.as<float>(), alpha, xscale, h->adim, x);
    return PK_OK;
}

int pk_fft_layernorm_rows(pk_fft_core* h, const float* x, size_t g, size_t b, const int* d_row_utt, int rows, int C,
                          float* y, float* amax) {
    if (C % 64 != 0 || C > 64 * LN_MAXPER) PK_FAIL(PK_EUNSUPPORTED, "LayerNorm: %d channels (multiple of 64, <= %d)", C, 64 * LN_MAXPER);
    PK_LAUNCH(h->ctx, "fft_layernorm", k_layernorm, dim3(pk_div_up(rows, 4)), dim3(256), 0, x, h->W(g), h->W(b),
              d_row_utt, rows, C, 1e-5f, y, amax);
    return PK_OK;
}

static int run_predictor(pk_fs2* h, const Predictor& pr, const Timeline& tl, const float* hs, int duration_mode,
                         float alpha, float* out) {
    const int A = h->cfg.adim;
    PK_TRY(pk_fft_act_reserve(h->d_p1, tl.rows, pr.chans));
    PK_TRY(pk_fft_act_reserve(h->d_p2, tl.rows, pr.chans));
    float* p1 = pk_fft_act_ptr(h->d_p1, pr.chans);
    float* p2 = pk_fft_act_ptr(h->d_p2, pr.chans);
    const float* in = hs;
    int ldin = A;
    // operand scales of the split-fp16 convs: the row maxima come out of the LayerNorm that produces the rows (no pass of
    // their own: 6 of the 9 k_row_amax launches of a batch); the buffer is zero outside the rows it writes (margins, padding)
    float* pam = nullptr;
    if (h->math == PK_GEMM_MATH_F16X3) {
        PK_TRY(pk_fft_act_reserve(h->d_pamax, tl.rows, 1));
        PK_HIP(hipMemsetAsync(h->d_pamax.p, 0, h->d_pamax.cap, h->ctx->stream));
        pam = pk_fft_act_ptr(h->d_pamax, 1);
    }
    // Round 4: the convs on the planes kernel (csrc/ffn_planes.hip, 384 | 256 -> 256 channels, k = 3 | 5): encoder output ->
    // planes once per batch (shared by the three predictors), conv -> fp32 rows (ReLU), LayerNorm -> planes for the next conv
    // (k_ffn_ln_planes) or -> fp32 rows for the head after the last one.  Same arithmetic class as the tile GEMM path
    // (block-scaled split-fp16, one scale per row), half its time.
    bool planes = h->math == PK_GEMM_MATH_F16X3 && h->ffn_planes && tl.rows_alloc % FFNP_BLK == 0 && !pr.conv.empty();
    for (const Dense& d : pr.conv) planes = planes && d.wp != (size_t)-1 && d.wps != (size_t)-1;
    if (planes) {
        const int nblk = tl.rows_alloc / FFNP_BLK;
        const int* rv = tl.d_row_utt();
        if (!h->hs_planes_valid) {
            PK_TRY(planes_buf(h, h->d_hsp, h->d_hsam, nblk, A, &h->hsp, &h->hsam));
            PK_TRY(ffnp_layernorm_launch(h->ctx, hs, nullptr, nullptr, rv, nblk, A, 0.f, h->hsp, h->hsam));
            h->hs_planes_valid = true;
        }
        char* pp = nullptr;
        unsigned* ppam = nullptr;
        PK_TRY(planes_buf(h, h->d_pp, h->d_ppam, nblk, pr.chans, &pp, &ppam));
        const void* cin = h->hsp;
        const unsigned* cam = h->hsam;
        for (size_t j = 0; j < pr.conv.size(); ++j) {
            FfnpConv c = conv256_args(h, pr.conv[j], nblk, rv, cin, cam);
            c.x = p1;
            c.ldx = pr.chans;
            PK_TRY(ffnp_conv256_launch(h->ctx, "fs2_conv_predictor_planes", c, pr.conv[j].taps, 2));
            if (j + 1 < pr.conv.size()) {
                PK_TRY(ffnp_layernorm_launch(h->ctx, p1, h->W(pr.ln_g[j]), h->W(pr.ln_b[j]), rv, nblk, pr.chans, 1e-5f, pp, ppam));
                cin = pp;
                cam = ppam;
            } else {
                PK_TRY(pk_fft_run_layernorm(h, p1, pr.ln_g[j], pr.ln_b[j], tl, pr.chans, p2));
            }
        }
        in = p2;
        ldin = pr.chans;
    }
    const float* in_amax = nullptr;
    for (size_t j = 0; j < pr.conv.size() && !planes; ++j) {
        PK_TRY(pk_fft_run_dense(h, "fs2_conv_predictor", pr.conv[j], in, ldin, p1, pr.chans, tl.rows, PK_ACT_RELU, nullptr, 0,
                         nullptr, in_amax));
        PK_TRY(pk_fft_run_layernorm(h, p1, pr.ln_g[j], pr.ln_b[j], tl, pr.chans, p2, pam));
        in = p2;
        ldin = pr.chans;
        in_amax = pam;
    }
    PK_LAUNCH(h->ctx, "fs2_rowdot", k_rowdot, dim3(pk_div_up(tl.rows, 4)), dim3(256), 0, in, ldin, h->W(pr.lin_w),
              pr.lin_b, tl.d_row_utt(), tl.rows, duration_mode, 1.0f, alpha, out);
    return PK_OK;
}

extern "C" int pk_fs2_encode(pk_fs2* h, const int64_t* ids, const int32_t* tok_lens, int32_t B, float alpha,
                             int32_t* out_frames) {
    if (!h || !ids || !tok_lens || !out_frames) PK_FAIL(PK_EINVAL, "pk_fs2_encode: NULL argument");
    if (!h->finalized) PK_FAIL(PK_ESTATE, "pk_fs2_encode: call pk_fs2_finalize first");
    if (B <= 0) PK_FAIL(PK_EINVAL, "pk_fs2_encode: batch size must be positive");
    if (!(alpha > 0.f)) PK_FAIL(PK_ESHAPE, "LengthRegulator: alpha must be > 0 (length_regulator.py:86)");
    pk_ctx* ctx = h->ctx;
    PK_DEVICE(ctx->device);
    const pk_fs2_cfg& c = h->cfg;
    const int A = c.adim;
    int maxT = 0;
    long sumT = 0;
    for (int b = 0; b < B; ++b) {
        if (tok_lens[b] <= 0) PK_FAIL(PK_EINVAL, "pk_fs2_encode: utterance %d has %d tokens", b, tok_lens[b]);
        maxT = std::max(maxT, tok_lens[b]);
        sumT += tok_lens[b];
    }
    h->encoded = false;
    // Per-call conditioning (pk_fs2_set_speakers / pk_fs2_set_tones) is taken off the handle FIRST, so that no
    // exit path -- error or success -- leaves it behind for an unrelated later encode, and its counts are
    // validated before anything is launched.
    const int condB = h->cond_B;
    std::vector<long long> cond_spk, cond_tone;
    std::vector<float> cond_emb;
    cond_spk.swap(h->cond_spk);
    cond_emb.swap(h->cond_emb);
    cond_tone.swap(h->cond_tone);
    h->cond_B = 0;
    if (c.spk_embed_dim > 0 && condB > 0 && condB != B)
        PK_FAIL(PK_ESHAPE, "pk_fs2_encode: speakers were set for %d utterances, batch has %d", condB, B);
    if (c.tone_embed_dim > 0 && !cond_tone.empty() && (long)cond_tone.size() != sumT)
        PK_FAIL(PK_ESHAPE, "pk_fs2_encode: %zu tone ids for %ld tokens", cond_tone.size(), sumT);
    PK_TRY(pk_fft_build_timeline(ctx, h->tl_tok, tok_lens, B, h->gapr));
    Timeline& tl = h->tl_tok;
    PK_TRY(pk_fft_ensure_pe(h, maxT));
    // token ids on the row timeline
    {
        std::vector<int> tok(tl.rows_alloc, 0);
        long o = 0;
        for (int b = 0; b < B; ++b)
            for (int t = 0; t < tok_lens[b]; ++t, ++o) {
                const int64_t id = ids[o];
                if (id < 0 || id >= c.idim) PK_FAIL(PK_EINVAL, "pk_fs2_encode: token id %lld out of [0,%d)", (long long)id, c.idim);
                tok[tl.seg_start[b] + t] = (int)id;
            }
        PK_TRY(pk_upload(ctx, h->d_tok, tok.data(), tok.size() * sizeof(int)));
    }
    PK_TRY(pk_fft_act_reserve(h->d_x, tl.rows, A));
    PK_TRY(pk_fft_act_reserve(h->d_hs, tl.rows, A));
    float* x = pk_fft_act_ptr(h->d_x, A);
    float* hs = pk_fft_act_ptr(h->d_hs, A);
    PK_TRY(pk_fft_embed(h, "fs2_embed", h->d_tok.as<int>(), tl, h->emb_table, h->alpha_enc, h->xscale, x));
    PK_TRY(pk_fft_run_stack(h, h->enc, h->enc_after_g, h->enc_after_b, tl, c.eunits, hs, c.encoder_normalize_before != 0));
    // speaker embedding (:396-402)
    if (c.spk_embed_dim > 0 && condB > 0) {
        const int D = c.spk_embed_dim;
        const bool ext = !cond_emb.empty();
        const long long* d_id = nullptr;
        const float* d_emb = nullptr;
        if (ext) {
            PK_TRY(pk_upload(ctx, h->d_spk_emb, cond_emb.data(), cond_emb.size() * sizeof(float)));
            d_emb = h->d_spk_emb.as<float>();
        } else {
            PK_TRY(pk_upload(ctx, h->d_spk_id, cond_spk.data(), cond_spk.size() * sizeof(long long)));
            d_id = h->d_spk_id.as<long long>();
        }
        PK_TRY(pk_fft_run_speaker(h, tl, d_id, d_emb, h->spk_table, h->spk_w, h->spk_b,
                                  c.spk_embed_integration_type == 1 ? &h->spk_hs : nullptr, D, h->d_spk_vec, hs, x));
    }
    // tone embedding (:404-408)
    if (c.tone_embed_dim > 0 && !cond_tone.empty()) {
        const std::vector<long long>& ct = cond_tone;
        std::vector<int> tn(tl.rows_alloc, 0);
        long o = 0;
        for (int b = 0; b < B; ++b)
            for (int t = 0; t < tok_lens[b]; ++t, ++o) tn[tl.seg_start[b] + t] = (int)ct[o];
        PK_TRY(pk_upload(ctx, h->d_tone, tn.data(), tn.size() * sizeof(int)));
        PK_LAUNCH(ctx, "fs2_add_tone", k_add_tone, dim3(tl.rows), dim3(256), 0, hs, h->W(h->tone_table),
                  h->d_tone.as<int>(), tl.d_row_utt(), A);
    }
    // variance adaptor
    PK_TRY(h->d_pout.reserve((size_t)tl.rows_alloc * 4));
    PK_TRY(h->d_eout.reserve((size_t)tl.rows_alloc * 4));
    PK_TRY(h->d_dout.reserve((size_t)tl.rows_alloc * 4));
    PK_TRY(h->d_cum.reserve((size_t)tl.rows_alloc * 4));
    PK_TRY(h->d_frames.reserve((size_t)B * 4));
    h->hs_planes_valid = false;
    PK_TRY(run_predictor(h, h->pitch, tl, hs, 0, 1.f, h->d_pout.as<float>()));
    PK_TRY(run_predictor(h, h->energy, tl, hs, 0, 1.f, h->d_eout.as<float>()));
    PK_TRY(run_predictor(h, h->dur, tl, hs, 1, alpha, h->d_dout.as<float>()));
    PK_LAUNCH(ctx, "fs2_cumsum", k_cumsum, dim3(B), dim3(256), 0, h->d_dout.as<float>(), tl.d_seg_start(),
              tl.d_seg_len(), h->d_cum.as<int>(), h->d_frames.as<int>());
    h->frames.resize(B);
    PK_HIP(hipMemcpyAsync(h->frames.data(), h->d_frames.p, (size_t)B * 4, hipMemcpyDeviceToHost, ctx->stream));
    PK_HIP(hipStreamSynchronize(ctx->stream));
    for (int b = 0; b < B; ++b) out_frames[b] = h->frames[b] * c.reduction_factor;   // mel frames; h->frames: decoder rows
    h->encoded = true;
    return PK_OK;
}

extern "C" int pk_fs2_decode(pk_fs2* h, float* mel_out, int32_t flags) {
    if (!h || !mel_out) PK_FAIL(PK_EINVAL, "pk_fs2_decode: NULL argument");
    if (!h->encoded) PK_FAIL(PK_ESTATE, "pk_fs2_decode: call pk_fs2_encode first");
    pk_ctx* ctx = h->ctx;
    PK_DEVICE(ctx->device);
    const pk_fs2_cfg& c = h->cfg;
    const int A = c.adim, B = h->tl_tok.B;
    // frame timeline; utterances with 0 frames get an empty segment
    std::vector<int> lens(h->frames);
    int maxL = 0;
    long sumL = 0;
    for (int b = 0; b < B; ++b) {
        maxL = std::max(maxL, lens[b]);
        sumL += lens[b];
    }
    if (sumL == 0) return PK_OK;
    PK_TRY(pk_fft_build_timeline(ctx, h->tl_frm, lens.data(), B, h->gapr));
    Timeline& tl = h->tl_frm;
    PK_TRY(pk_fft_ensure_pe(h, maxL));
    // packed output row of each timeline row
    {
        std::vector<int> rowmap(tl.rows_alloc, -1);
        int o = 0;
        for (int b = 0; b < B; ++b)
            for (int l = 0; l < lens[b]; ++l) rowmap[tl.seg_start[b] + l] = o++;
        PK_TRY(pk_upload(ctx, h->d_rowmap, rowmap.data(), rowmap.size() * sizeof(int)));
    }
    // d_hs keeps the encoder output (token rate) until k_regulate has consumed it; d_x was the
    // encoder's residual stream and is free for the decoder.
    PK_TRY(pk_fft_act_reserve(h->d_x, tl.rows, A));
    float* x = pk_fft_act_ptr(h->d_x, A);
    const float* hs_tok = pk_fft_act_ptr(h->d_hs, A);
    float* up_dbg = nullptr;
    if (h->debug) {
        PK_TRY(pk_fft_act_reserve(h->d_dbg_up, tl.rows, A));
        up_dbg = pk_fft_act_ptr(h->d_dbg_up, A);
    }
    PK_LAUNCH(ctx, "fs2_regulate", k_regulate, dim3(tl.rows), dim3(128), 0, hs_tok, h->d_pout.as<float>(),
              h->d_eout.as<float>(), h->W(h->pitch_w), h->W(h->pitch_b), h->W(h->energy_w), h->W(h->energy_b),
              h->d_cum.as<int>(), h->tl_tok.d_seg_start(), h->tl_tok.d_seg_len(), tl.d_row_utt(), tl.d_row_pos(),
              h->d_pe.as<float>(), h->alpha_dec, h->xscale, A, x, up_dbg);
    PK_TRY(pk_fft_act_reserve(h->d_zs, tl.rows, A));
    float* zs = pk_fft_act_ptr(h->d_zs, A);
    PK_TRY(pk_fft_run_stack(h, h->dec, h->dec_after_g, h->dec_after_b, tl, c.dunits, zs, c.decoder_normalize_before != 0));
    // feat_out (+ row mask: the postnet convolves over it)
    PK_TRY(pk_fft_act_reserve(h->d_before, tl.rows, c.odim));
    float* before = pk_fft_act_ptr(h->d_before, c.odim);
    float* d_out = mel_out;
    if (flags & PK_HOST_IO) {
        PK_TRY(h->d_mel_stage.reserve((size_t)sumL * c.odim * 4));
        d_out = h->d_mel_stage.as<float>();
    }
    const bool denorm = h->has_out_affine && (flags & PK_APPLY_NORMALIZER);   // FastSpeech2Inference (:668-671)
    const float* cs = denorm ? h->W(h->out_scale) : nullptr;
    const float* ch = denorm ? h->W(h->out_shift) : nullptr;
    if (c.reduction_factor > 1) {
        // feat_out -> (rows, odim * r), unfolded onto a timeline of rows * r frames for the postnet (:457-464)
        const int RF = c.reduction_factor, O = c.odim;
        PK_TRY(pk_fft_act_reserve(h->d_wide, tl.rows, O * RF));
        float* wide = pk_fft_act_ptr(h->d_wide, O * RF);
        PK_TRY(pk_fft_run_dense(h, "fs2_gemm_feat_out", h->feat_out, zs, A, wide, O * RF, tl.rows, PK_ACT_NONE, nullptr, 0, nullptr));
        std::vector<int> flens(B);
        long sumF = 0;
        for (int b = 0; b < B; ++b) {
            flens[b] = lens[b] * RF;
            sumF += flens[b];
        }
        PK_TRY(pk_fft_build_timeline(ctx, h->tl_frm2, flens.data(), B, h->gapr));
        Timeline& tf = h->tl_frm2;
        {
            std::vector<int> rowmap(tf.rows_alloc, -1);
            int o = 0;
            for (int b = 0; b < B; ++b)
                for (int l = 0; l < flens[b]; ++l) rowmap[tf.seg_start[b] + l] = o++;
            PK_TRY(pk_upload(ctx, h->d_rowmap2, rowmap.data(), rowmap.size() * sizeof(int)));
        }
        if (flags & PK_HOST_IO) {
            PK_TRY(h->d_mel_stage.reserve((size_t)sumF * O * 4));
            d_out = h->d_mel_stage.as<float>();
        }
        if (c.postnet_layers == 0) {
            PK_LAUNCH(ctx, "fs2_unfold", k_fs2_unfold_r, dim3(tf.rows), dim3(128), 0, wide, O, RF, tl.d_seg_start(), tf.d_row_utt(),
                      tf.d_row_pos(), h->d_rowmap2.as<int>(), cs, ch, d_out);
        } else {
            PK_TRY(pk_fft_act_reserve(h->d_before, tf.rows, O));
            float* before2 = pk_fft_act_ptr(h->d_before, O);
            PK_LAUNCH(ctx, "fs2_unfold", k_fs2_unfold_r, dim3(tf.rows), dim3(128), 0, wide, O, RF, tl.d_seg_start(), tf.d_row_utt(),
                      tf.d_row_pos(), (const int*)nullptr, (const float*)nullptr, (const float*)nullptr, before2);
            PK_TRY(pk_fft_run_postnet(h, "fs2_conv_postnet", h->postnet, before2, O, c.postnet_chans, tf, h->d_q1, h->d_q2, d_out,
                                      h->d_rowmap2.as<int>(), cs, ch));
        }
        if (flags & PK_HOST_IO) {
            PK_HIP(hipMemcpyAsync(mel_out, d_out, (size_t)sumF * O * 4, hipMemcpyDeviceToHost, ctx->stream));
            PK_HIP(hipStreamSynchronize(ctx->stream));
        }
        return PK_OK;
    }
    if (c.postnet_layers == 0) {
        pk_gemm_args g;
        g.A = zs; g.lda = A; g.Wp = h->W(h->feat_out.w); g.bias = h->W(h->feat_out.b);
        g.Wh = h->feat_out.wh == (size_t)-1 ? nullptr : h->arena16.as<uint16_t>() + h->feat_out.wh; g.math = h->math;
        g.C = d_out; g.ldc = c.odim; g.rowvalid = tl.d_row_utt(); g.cscale = cs; g.cshift = ch;
        g.out_rowmap = h->d_rowmap.as<int>(); g.M = tl.rows; g.N = c.odim; g.Cin = A; g.taps = 1; g.pad = 0;
        PK_TRY(pk_gemm_launch(ctx, "fs2_gemm_feat_out", g));
    } else {
        PK_TRY(pk_fft_run_dense(h, "fs2_gemm_feat_out", h->feat_out, zs, A, before, c.odim, tl.rows, PK_ACT_NONE, nullptr, 0,
                         tl.d_row_utt()));
        // after = before + postnet(before)  (:463-464), then ZScore.inverse (FastSpeech2Inference :670)
        PK_TRY(pk_fft_run_postnet(h, "fs2_conv_postnet", h->postnet, before, c.odim, c.postnet_chans, tl, h->d_q1, h->d_q2,
                                  d_out, h->d_rowmap.as<int>(), cs, ch));
    }
    if (flags & PK_HOST_IO) {
        PK_HIP(hipMemcpyAsync(mel_out, d_out, (size_t)sumL * c.odim * 4, hipMemcpyDeviceToHost, ctx->stream));
        PK_HIP(hipStreamSynchronize(ctx->stream));
    }
    return PK_OK;
}

extern "C" int pk_fs2_set_speakers(pk_fs2* h, const int64_t* spk_id, const float* spembs, int32_t B) {
    if (!h) PK_FAIL(PK_EINVAL, "pk_fs2_set_speakers: handle is NULL");
    h->cond_B = 0;
    h->cond_spk.clear();
    h->cond_emb.clear();
    if (!spk_id && !spembs) return PK_OK;
    const pk_fs2_cfg& c = h->cfg;
    if (c.spk_embed_dim <= 0) return PK_OK;   // a single-speaker model ignores speakers, like the reference (:396)
    if (B <= 0) PK_FAIL(PK_EINVAL, "pk_fs2_set_speakers: batch size must be positive");
    if (spembs) {
        h->cond_emb.assign(spembs, spembs + (size_t)B * c.spk_embed_dim);
    } else {
        if (c.num_speakers <= 0) PK_FAIL(PK_ESTATE, "pk_fs2_set_speakers: the model has no spk_embedding_table");
        for (int b = 0; b < B; ++b)
            if (spk_id[b] < 0 || spk_id[b] >= c.num_speakers)
                PK_FAIL(PK_EINVAL, "pk_fs2_set_speakers: speaker id %lld out of [0,%d)", (long long)spk_id[b], c.num_speakers);
        h->cond_spk.assign(spk_id, spk_id + B);
    }
    h->cond_B = B;
    return PK_OK;
}

extern "C" int pk_fs2_set_tones(pk_fs2* h, const int64_t* tone_id, int64_t n) {
    if (!h) PK_FAIL(PK_EINVAL, "pk_fs2_set_tones: handle is NULL");
    h->cond_tone.clear();
    if (!tone_id || h->cfg.tone_embed_dim <= 0) return PK_OK;   // a model without tones ignores them (:404)
    if (n <= 0) PK_FAIL(PK_EINVAL, "pk_fs2_set_tones: n must be positive");
    for (int64_t i = 0; i < n; ++i)
        if (tone_id[i] < 0 || tone_id[i] >= h->cfg.num_tones)
            PK_FAIL(PK_EINVAL, "pk_fs2_set_tones: tone id %lld out of [0,%d)", (long long)tone_id[i], h->cfg.num_tones);
    h->cond_tone.assign(tone_id, tone_id + n);
    return PK_OK;
}

extern "C" int pk_fs2_set_math(pk_fs2* h, int32_t mode) {
    if (!h) PK_FAIL(PK_EINVAL, "pk_fs2_set_math: handle is NULL");
    if (mode != PK_GEMM_MATH_F32 && mode != PK_GEMM_MATH_F16X3) PK_FAIL(PK_EINVAL, "pk_fs2_set_math: unknown mode %d", mode);
    h->math = mode;
    return PK_OK;
}

// shared by pk_fs2_set_option and pk_tts_set_option: the options of an FFT stack
int pk_fft_set_option(pk_fft_core* h, const char* key, int64_t value, const char* who) {
    if (!h || !key) PK_FAIL(PK_EINVAL, "%s: NULL argument", who);
    if (strcmp(key, "ffn_planes") == 0) h->ffn_planes = value != 0;
    else if (strcmp(key, "ffn_planes_min_blocks") == 0) h->ffn_planes_min_blocks = (int)std::max<int64_t>(0, value);
    else if (strcmp(key, "ffn_one_tile_max") == 0) h->ffn_one_tile_max = (int)std::max<int64_t>(0, std::min<int64_t>(value, 1 << 20));
    else if (strcmp(key, "ffnp_variant") == 0) {
        if (value != 0 && value != 44 && value != 48 && value != 84 && value != 88) PK_FAIL(PK_EINVAL, "%s: ffnp_variant %lld (0, 44, 48, 84, 88)", who, (long long)value);
        h->ffnp_variant = (int)value;
    } else if (strcmp(key, "attn_waves") == 0) {
        if (value != 0 && value != 4 && value != 8) PK_FAIL(PK_EINVAL, "%s: attn_waves %lld (0, 4, 8)", who, (long long)value);
        h->attn_waves = (int)value;
    } else PK_FAIL(PK_EINVAL, "%s: unknown option '%s'", who, key);
    return PK_OK;
}

extern "C" int pk_fs2_set_option(pk_fs2* h, const char* key, int64_t value) {
    return pk_fft_set_option(h, key, value, "pk_fs2_set_option");
}

extern "C" int pk_fs2_set_debug(pk_fs2* h, int32_t on) {
    if (!h) PK_FAIL(PK_EINVAL, "pk_fs2_set_debug: handle is NULL");
    h->debug = on != 0;
    return PK_OK;
}

extern "C" int pk_fs2_debug_read(pk_fs2* h, int32_t what, int32_t b, float* host_out, int64_t n_floats) {
    if (!h || !host_out) PK_FAIL(PK_EINVAL, "pk_fs2_debug_read: NULL argument");
    if (!h->encoded) PK_FAIL(PK_ESTATE, "pk_fs2_debug_read: nothing has run");
    pk_ctx* ctx = h->ctx;
    PK_DEVICE(ctx->device);
    const int A = h->cfg.adim;
    const Timeline* tl = &h->tl_tok;
    const float* src = nullptr;
    int C = A;
    switch (what) {
        case 0: src = pk_fft_act_ptr(h->d_hs, A); break;                           // encoder output hs (T, adim)
        case 1: src = h->d_pout.as<float>(); C = 1; break;                  // pitch (T,)
        case 2: src = h->d_eout.as<float>(); C = 1; break;                  // energy (T,)
        case 3: src = h->d_dout.as<float>(); C = 1; break;                  // durations (T,)
        case 4: tl = &h->tl_frm; src = pk_fft_act_ptr(h->d_dbg_up, A); break;      // length-regulated hs (L, adim)
        case 5: tl = &h->tl_frm; src = pk_fft_act_ptr(h->d_zs, A); break;      // decoder output zs (L, adim)
        case 6: tl = h->cfg.reduction_factor > 1 ? &h->tl_frm2 : &h->tl_frm; src = pk_fft_act_ptr(h->d_before, h->cfg.odim); C = h->cfg.odim; break;  // before_outs
        default: PK_FAIL(PK_EINVAL, "pk_fs2_debug_read: unknown tap %d", what);
    }
    if (b < 0 || b >= tl->B) PK_FAIL(PK_EINVAL, "pk_fs2_debug_read: utterance out of range");
    if (what == 4 && !h->debug) PK_FAIL(PK_ESTATE, "pk_fs2_debug_read: tap 4 needs pk_fs2_set_debug(1) before decode");
    const long n = (long)tl->seg_len[b] * C;
    if (n_floats != n) PK_FAIL(PK_ESHAPE, "pk_fs2_debug_read: expected %ld floats, got %lld", n, (long long)n_floats);
    PK_HIP(hipStreamSynchronize(ctx->stream));
    if (n > 0)
        PK_HIP(hipMemcpy(host_out, src + (long)tl->seg_start[b] * C, n * sizeof(float), hipMemcpyDeviceToHost));
    return PK_OK;
}

extern "C" void pk_fs2_destroy(pk_fs2* h) {
    if (!h) return;
    pk_device_guard _dg(h->ctx->device);
    (void)hipStreamSynchronize(h->ctx->stream);
    h->release_core();
    pk_dbuf* bufs[] = {&h->d_hsp, &h->d_hsam, &h->d_pp, &h->d_ppam, &h->d_pamax, &h->d_tok, &h->d_p1, &h->d_p2, &h->d_hs, &h->d_pout, &h->d_eout, &h->d_dout, &h->d_cum, &h->d_frames,
                       &h->d_tone, &h->d_spk_id, &h->d_spk_emb, &h->d_spk_vec, &h->d_before, &h->d_q1, &h->d_q2, &h->d_rowmap, &h->d_dbg_up, &h->d_zs, &h->d_mel_stage,
                       &h->d_wide, &h->d_rowmap2};
    for (auto* b : bufs) b->release();
    h->tl_tok.release();
    h->tl_frm.release();
    h->tl_frm2.release();
    delete h;
}
