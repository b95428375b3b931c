// tts.hip -- TransformerTTS inference on gfx950: kernels + pk_tts_* entry points.
//
// Reference: parakeet/models/transformer_tts/transformer_tts.py TransformerTTS.inference :511-647,
//   Decoder.forward_one_step   parakeet/modules/fastspeech2_transformer/decoder.py:190-227
//   DecoderLayer.forward       parakeet/modules/fastspeech2_transformer/decoder_layer.py:74-158 (cache branch)
//   MultiHeadedAttention       parakeet/modules/fastspeech2_transformer/attention.py:51-156
//   DecoderPrenet (Prenet)     parakeet/modules/tacotron2/decoder.py:62-81   (dropout stays on at inference)
//   EncoderPrenet              parakeet/modules/tacotron2/encoder.py:150-176
//   Postnet                    parakeet/modules/tacotron2/decoder.py:127-198
//
// The text encoder is the same `Encoder` class FastSpeech2 uses: it runs on the shared row-timeline machinery of
// pk_fft.h (fs2.hip).  The decoder is autoregressive; what the engine does with the reference's loop:
//
//  * B utterances are decoded in lockstep, one frame per step for every utterance.  All per-frame tensors are
//    POSITION-MAJOR: row = pos * B + b, so the rows of steps 1..s are the contiguous prefix [0, s*B) and every GEMM
//    of a step is one launch over a contiguous row range with no per-step index tables.
//  * The reference re-applies decoder.embed (prenet with always-on dropout -> Linear -> positional encoding) to the
//    WHOLE prefix at every step (decoder.py:210) and caches layer outputs only (:213-218).  Layer 0 therefore sees
//    s freshly re-dropped rows at step s: the engine recomputes prenet, embedding, norm1 and the K/V projection of
//    layer 0 for the s*B prefix rows each step (one GEMM each).  For layers >= 1 the inputs of old rows are cached
//    layer outputs, so their K/V rows never change: they are projected once, when the row is new (a KV cache the
//    reference does not have: it re-projects the whole prefix in every layer at every step).
//  * with a cache the reference computes the last query row only and its mask row is all ones
//    (decoder_layer.py:110-120): k_tts_attn_step is one query per (utterance, head) over s keys; the same kernel
//    serves the encoder-decoder attention over the T_b memory rows, whose K/V projections are computed once per call.
//  * the per-step products of the new rows (B rows, one per utterance) run on the row GEMM of pk_rowgemm.h (weights
//    streamed once, exact fp32 FMA, LayerNorm fused as a prologue, residual in the epilogue): 8 launches per decoder
//    layer and step; PK_AR_ROWGEMM=0 selects the tile GEMM of gemm.hip + separate LayerNorm launches instead.
//  * an utterance that has stopped keeps being stepped (its rows are ignored) until all have; the stop state lives
//    on the device and is polled every few steps.
//
// Dropout: include/pk_synth.h "dropout stream".
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <string>
#include <vector>

#include "pk_ar.h"
#include "pk_fft.h"
#include "pk_gst.h"
#include "pk_rowgemm.h"

namespace {

typedef pk_fft_dense Dense;
typedef pk_fft_timeline Timeline;

// ---------------------------------------------------------------------------------------------- kernels

// out[r][c] = alpha * pe[r / B][c]: the positional term of decoder.embed in position-major rows, added by the
// epilogue of the embedding GEMM (ScaledPositionalEncoding.forward embedding.py:111-126)
__global__ __launch_bounds__(128) void k_tts_pe_pos_major(const float* __restrict__ pe, float alpha, int B, int A,
                                                          float* __restrict__ out) {
    const long r = blockIdx.x;
    const float* p = pe + (r / B) * A;
    for (int c = threadIdx.x; c < A; c += blockDim.x) out[r * A + c] = alpha * p[c];
}

// the same term on a row timeline (encoder conv prenet path): gap rows zero
__global__ __launch_bounds__(128) void k_tts_pe_timeline(const float* __restrict__ pe, float alpha,
                                                         const int* __restrict__ row_utt,
                                                         const int* __restrict__ row_pos, int A,
                                                         float* __restrict__ out) {
    const long r = blockIdx.x;
    const bool valid = row_utt[r] >= 0;
    const float* p = pe + (long)(valid ? row_pos[r] : 0) * A;
    for (int c = threadIdx.x; c < A; c += blockDim.x) out[r * A + c] = valid ? alpha * p[c] : 0.f;
}

// x[r] = table[tok[r]] (row 0 of the table is zero: padding_idx), gap rows zero
__global__ __launch_bounds__(128) void k_tts_lookup(const int* __restrict__ tok, const int* __restrict__ row_utt,
                                                    const float* __restrict__ table, int E, float* __restrict__ x) {
    const long r = blockIdx.x;
    const bool valid = row_utt[r] >= 0;
    const float* e = table + (long)(valid ? tok[r] : 0) * E;
    for (int c = threadIdx.x; c < E; c += blockDim.x) x[r * E + c] = valid ? e[c] : 0.f;
}

// x = relu(t) + peb: the tail of the "linear" decoder input layer (Linear -> LayerNorm -> Dropout(eval) -> ReLU ->
// positional encoding, decoder.py:112-118); n4 float4 elements
__global__ __launch_bounds__(256) void k_tts_relu_add(const float4* __restrict__ t, const float4* __restrict__ peb, long n4,
                                                      float4* __restrict__ x) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 a = t[i], p = peb[i];
    x[i] = make_float4(fmaxf(a.x, 0.f) + p.x, fmaxf(a.y, 0.f) + p.y, fmaxf(a.z, 0.f) + p.z, fmaxf(a.w, 0.f) + p.w);
}

// One decoding step of MultiHeadedAttention (attention.py:133-156) for ONE query row per utterance:
//   grid (heads, B), 256 threads.  Key / value row j of utterance b is row kbase[b] + j * kstride (ld = ldkv).
//   self-attention: K/V = the layer's projected prefix rows (position-major: kbase = b, kstride = B, n = step);
//   encoder-decoder attention: K/V = the projected memory rows of the token timeline (kbase = seg_start, kstride 1,
//   klen = T_b), whose softmax weights are the att_ws the reference returns (:623-636).
struct AttnStep {
    const float* q;     // query rows [B][ldq], head h at column h * dk
    int ldq;
    const float* K;
    const float* V;
    int ldkv;
    const int* kbase;   // per utterance, NULL: b
    const int* klen;    // per utterance, NULL: n
    int kstride, n, dk;
    float scale;
    float* out;         // [B][ldo]
    int ldo;
    float* att;         // NULL or base of the attention-weight store
    const long* att_off;  // per utterance offset (floats) of its (layers, heads, cap_b, T_b) block
    const int* att_cap;   // per utterance cap_b
    int layer, step;      // step counted from 0
    // k_tts_attn_step64<NB, true>: the query is not read but PROJECTED here -- q[b] = LN?(qx[b]) . W[:, head's 64 columns] + bias
    // (W as pk_rowgemm_pack tiles, K = qK <= 512): the encoder-decoder attention's linear_q without a launch of its own
    const float* qx = nullptr;
    int ldqx = 0, qK = 0;
    const float* qW = nullptr;
    const float* qb = nullptr;
    const float* q_ln_g = nullptr;
    const float* q_ln_b = nullptr;
    float q_eps = 1e-5f;
};

__global__ __launch_bounds__(256) void k_tts_attn_step(AttnStep a) {
    extern __shared__ float sm[];
    const int head = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int dk = a.dk;
    const int n = a.klen ? a.klen[b] : a.n;
    const long base = a.kbase ? a.kbase[b] : b;
    float* qs = sm;           // dk (16-byte aligned: dk % 4 == 0)
    float* red = sm + dk;     // 1024 partial sums (256 threads x float4) + 8 reduction slots
    float* sc = red + 1032;   // n scores -> probabilities
    const float* qp = a.q + (long)b * a.ldq + head * dk;
    // scores: 16 lanes x float4 per key row, 4 keys per wave, 16 keys per pass; NB passes per iteration with all of their
    // (clamped, unconditional) loads issued before the first use -- at the LJSpeech head size (64) that is 16 key rows per lane
    // in flight, 256 keys per iteration (round 4; 4 passes before: a 640-key step was ten dependent round trips).  The query
    // is staged after the first batch has been requested: its load and the barrier wait under the key loads.
    const int sub = lane & 15, kq = lane >> 4, nv = dk >> 2;
    const float qreg = tid < dk ? qp[tid] * a.scale : 0.f;   // (dk <= 192 < 256 threads)
    auto score_batch = [&](auto nb_tag, int j0) {
        constexpr int NB = decltype(nb_tag)::value;
        float4 kv[NB];
        const float4* kp[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int j = min(j0 + u * 16 + wave * 4 + kq, n - 1);
            kp[u] = reinterpret_cast<const float4*>(a.K + (base + (long)j * a.kstride) * a.ldkv + head * dk);
            kv[u] = kp[u][min(sub, nv - 1)];
        }
        if (j0 == 0) {   // (uniform)
            if (tid < dk) qs[tid] = qreg;
            __syncthreads();
        }
        float s4[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const float4 q0 = *reinterpret_cast<const float4*>(qs + 4 * min(sub, nv - 1));
            float s = sub < nv ? fmaf(kv[u].x, q0.x, fmaf(kv[u].y, q0.y, fmaf(kv[u].z, q0.z, kv[u].w * q0.w))) : 0.f;
#pragma unroll
            for (int it = 1; it < 3; ++it) {   // dk <= 192: at most 3 float4 per lane
                const int c4 = sub + 16 * it;
                if (16 * it < nv) {            // block-uniform
                    const float4 k2 = kp[u][min(c4, nv - 1)];
                    const float4 qv = *reinterpret_cast<const float4*>(qs + 4 * min(c4, nv - 1));
                    const float t = fmaf(k2.x, qv.x, fmaf(k2.y, qv.y, fmaf(k2.z, qv.z, k2.w * qv.w)));
                    s += c4 < nv ? t : 0.f;
                }
            }
            s4[u] = s;
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            float s = s4[u];
            s += __shfl_xor(s, 8);
            s += __shfl_xor(s, 4);
            s += __shfl_xor(s, 2);
            s += __shfl_xor(s, 1);
            const int j = j0 + u * 16 + wave * 4 + kq;
            if (j < n && sub == 0) sc[j] = s;
        }
    };
    if (nv <= 16) {
        for (int j0 = 0; j0 < n; j0 += 256) score_batch(std::integral_constant<int, 16>(), j0);
    } else {
        for (int j0 = 0; j0 < n; j0 += 64) score_batch(std::integral_constant<int, 4>(), j0);
    }
    if (n <= 0) {   // (never: every utterance has at least one key) keep the barrier count uniform
        if (tid < dk) qs[tid] = qreg;
        __syncthreads();
    }
    __syncthreads();
    // softmax over the n keys
    float m = -INFINITY;
    for (int j = tid; j < n; j += 256) m = fmaxf(m, sc[j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) red[1024 + wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[1024], red[1025]), fmaxf(red[1026], red[1027]));
    float sum = 0.f;
    for (int j = tid; j < n; j += 256) {
        const float p = expf(sc[j] - m);
        sc[j] = p;
        sum += p;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) red[1028 + wave] = sum;
    __syncthreads();   // also publishes the probabilities in sc[]
    sum = (red[1028] + red[1029]) + (red[1030] + red[1031]);
    const float inv = 1.f / sum;
    // context: a thread owns 4 consecutive head dimensions (one float4 of a value row), groups of dk / 4 threads walk
    // the keys G apart with 8 loads in flight (the loop is bound by load latency: 30 us with dk threads per row and
    // 4 loads in flight at 640 keys)
    const int G = 256 / nv;
    const int g = tid / nv, c4 = tid - g * nv;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g < G) {
        const float* vp = a.V + head * dk + 4 * c4;
#pragma unroll 16
        for (int j = g; j < n; j += G) {
            const float4 vv = *reinterpret_cast<const float4*>(vp + (base + (long)j * a.kstride) * a.ldkv);
            const float p = sc[j];
            acc.x = fmaf(p, vv.x, acc.x);
            acc.y = fmaf(p, vv.y, acc.y);
            acc.z = fmaf(p, vv.z, acc.z);
            acc.w = fmaf(p, vv.w, acc.w);
        }
    }
    if (g < G) *reinterpret_cast<float4*>(red + g * dk + 4 * c4) = acc;
    __syncthreads();
    if (tid < dk) {
        float o = 0.f;
        for (int gg = 0; gg < G; ++gg) o += red[gg * dk + tid];
        a.out[(long)b * a.ldo + head * dk + tid] = o * inv;
    }
    if (a.att) {
        const int cap = a.att_cap[b];
        if (a.step < cap) {
            float* ap = a.att + a.att_off[b] + (((long)a.layer * gridDim.x + head) * cap + a.step) * n;
            for (int j = tid; j < n; j += 256) ap[j] = sc[j] * inv;
        }
    }
}

// The same step for 64-wide heads (the LJSpeech recipe: adim 512, 8 heads) and at most 16 NB keys, ONE trip to memory for the
// keys and one for the values (round 4: the general kernel above walks the keys in passes of 256 and the values 16 deep --
// seven dependent round trips at 640 keys, 21 us per launch of which 3 are the gap between two launches).  Thread (g = tid / 16,
// sub = tid % 16) owns float4 column sub of the key AND value rows g, g + 16, ...: all NB key loads are requested at once, the
// dot products are summed over the row's 16 lanes by DPP rotations (every lane of the group ends up with the score), the value
// loads are requested before the maximum is exchanged, the probabilities never leave the registers, and the partial contexts
// of the 16 groups and the wave sums of the normaliser cross one barrier together (two barriers in all).
template <int CTRL>
__device__ __forceinline__ float tts_dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// QP: the query is projected in the kernel (AttnStep::qx ...): thread (c = tid % 64, part = tid / 64) owns output column c and a
// quarter of K -- its 32 float4 of the packed weights are requested with the key loads, the row is normalised by wave 0 (8 values
// per lane, two-pass statistics by shuffles) into LDS, and the four partial dot products meet in LDS: three barriers and
// 128 KB of L2-resident weights per workgroup instead of a row-GEMM launch (9 us of the decoder's per-layer chain).
template <int NB, bool QP = false>
__global__ __launch_bounds__(256) void k_tts_attn_step64(AttnStep a) {
    __shared__ __attribute__((aligned(16))) float red[16 * 64];
    __shared__ float wmax[4], wsum[4];
    __shared__ __attribute__((aligned(16))) float xs[QP ? 512 : 4];
    __shared__ __attribute__((aligned(16))) float qs[64];
    const int head = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = lane & 15, g = tid >> 4;
    const int n = a.klen ? a.klen[b] : a.n;
    const long base = a.kbase ? a.kbase[b] : b;
    float4 q4;
    if (!QP) q4 = reinterpret_cast<const float4*>(a.q + (long)b * a.ldq + head * 64)[sub];
    const long rs = (long)a.kstride * a.ldkv;
    const float* kp = a.K + base * a.ldkv + head * 64 + 4 * sub;
    const float* vp = a.V + base * a.ldkv + head * 64 + 4 * sub;
    float4 kv[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) kv[u] = *reinterpret_cast<const float4*>(kp + (long)min(16 * u + g, n - 1) * rs);
    if (QP) {
        const int K = a.qK, K4 = ((K + 15) / 16) * 4;   // float4 rows of a packed 16-column tile (K padded to 16)
        const int c = tid & 63, part = tid >> 6;
        // weights of column c: tile (head * 64 + c) / 16, column c % 16; this thread's k-quads part * 32 .. + 31 (clamped: zeros
        // beyond K in the pack, and the row is zero there too)
        const float4* wq = reinterpret_cast<const float4*>(a.qW) + ((long)(head * 4 + (c >> 4)) * K4) * 16 + (c & 15);
        float4 w4[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) w4[i] = wq[(long)min(part * 32 + i, K4 - 1) * 16];
        const float qbias = a.qb ? a.qb[head * 64 + c] : 0.f;
        if (wave == 0) {   // the row, normalised: lane l holds x[8 l .. 8 l + 7]
            const float* xr = a.qx + (long)b * a.ldqx;
            float4 x0 = make_float4(0.f, 0.f, 0.f, 0.f), x1 = x0;
            if (8 * lane < K) x0 = *reinterpret_cast<const float4*>(xr + 8 * lane);
            if (8 * lane + 4 < K) x1 = *reinterpret_cast<const float4*>(xr + 8 * lane + 4);
            if (a.q_ln_g) {
                float t = ((x0.x + x0.y) + (x0.z + x0.w)) + ((x1.x + x1.y) + (x1.z + x1.w));
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
                const float mean = t / (float)K;
                const bool in0 = 8 * lane < K, in1 = 8 * lane + 4 < K;
                float d, qq = 0.f;
                d = x0.x - mean; qq += in0 ? d * d : 0.f;  d = x0.y - mean; qq += in0 ? d * d : 0.f;
                d = x0.z - mean; qq += in0 ? d * d : 0.f;  d = x0.w - mean; qq += in0 ? d * d : 0.f;
                d = x1.x - mean; qq += in1 ? d * d : 0.f;  d = x1.y - mean; qq += in1 ? d * d : 0.f;
                d = x1.z - mean; qq += in1 ? d * d : 0.f;  d = x1.w - mean; qq += in1 ? d * d : 0.f;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) qq += __shfl_xor(qq, o);
                const float rstd = 1.0f / sqrtf(qq / (float)K + a.q_eps);
                float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0, b0 = g0, b1 = g0;
                if (in0) { g0 = *reinterpret_cast<const float4*>(a.q_ln_g + 8 * lane); b0 = *reinterpret_cast<const float4*>(a.q_ln_b + 8 * lane); }
                if (in1) { g1 = *reinterpret_cast<const float4*>(a.q_ln_g + 8 * lane + 4); b1 = *reinterpret_cast<const float4*>(a.q_ln_b + 8 * lane + 4); }
                x0.x = in0 ? (x0.x - mean) * rstd * g0.x + b0.x : 0.f;  x0.y = in0 ? (x0.y - mean) * rstd * g0.y + b0.y : 0.f;
                x0.z = in0 ? (x0.z - mean) * rstd * g0.z + b0.z : 0.f;  x0.w = in0 ? (x0.w - mean) * rstd * g0.w + b0.w : 0.f;
                x1.x = in1 ? (x1.x - mean) * rstd * g1.x + b1.x : 0.f;  x1.y = in1 ? (x1.y - mean) * rstd * g1.y + b1.y : 0.f;
                x1.z = in1 ? (x1.z - mean) * rstd * g1.z + b1.z : 0.f;  x1.w = in1 ? (x1.w - mean) * rstd * g1.w + b1.w : 0.f;
            }
            *reinterpret_cast<float4*>(xs + 8 * lane) = x0;
            *reinterpret_cast<float4*>(xs + 8 * lane + 4) = x1;
        }
        __syncthreads();
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int kq = part * 32 + i;   // k = 4 kq .. 4 kq + 3 (xs is zero beyond K: 512 slots, K <= 512)
            const float4 xv = *reinterpret_cast<const float4*>(xs + 4 * min(kq, 127));
            const float t = fmaf(w4[i].x, xv.x, fmaf(w4[i].y, xv.y, fmaf(w4[i].z, xv.z, w4[i].w * xv.w)));
            acc += kq < K4 ? t : 0.f;
        }
        red[part * 64 + c] = acc;
        __syncthreads();
        if (tid < 64) qs[tid] = (((red[tid] + red[64 + tid]) + red[128 + tid]) + red[192 + tid] + qbias) * a.scale;
        __syncthreads();
        q4 = *reinterpret_cast<const float4*>(qs + 4 * sub);
    } else {
        q4.x *= a.scale; q4.y *= a.scale; q4.z *= a.scale; q4.w *= a.scale;
    }
    float sc[NB];
    float m = -INFINITY;
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        float t = fmaf(kv[u].x, q4.x, fmaf(kv[u].y, q4.y, fmaf(kv[u].z, q4.z, kv[u].w * q4.w)));
        t = tts_dpp_add<0x128>(t);   // row_ror:8, 4, 2, 1: the sum of the row's 16 lanes in every one of them
        t = tts_dpp_add<0x124>(t);
        t = tts_dpp_add<0x122>(t);
        t = tts_dpp_add<0x121>(t);
        sc[u] = 16 * u + g < n ? t : -INFINITY;
        m = fmaxf(m, sc[u]);
    }
    float4 vv[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) vv[u] = *reinterpret_cast<const float4*>(vp + (long)min(16 * u + g, n - 1) * rs);
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    if (lane == 0) wmax[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    float sum = 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const float p = expf(sc[u] - m);   // (keys beyond n: exp(-inf) = 0)
        sc[u] = p;
        sum += p;
        acc.x = fmaf(p, vv[u].x, acc.x);
        acc.y = fmaf(p, vv[u].y, acc.y);
        acc.z = fmaf(p, vv[u].z, acc.z);
        acc.w = fmaf(p, vv[u].w, acc.w);
    }
    sum += __shfl_xor(sum, 16);   // (the 16 lanes of a row hold the same values: lane 0 sums its wave's four rows)
    sum += __shfl_xor(sum, 32);
    if (lane == 0) wsum[wave] = sum;
    *reinterpret_cast<float4*>(red + g * 64 + 4 * sub) = acc;
    __syncthreads();
    const float inv = 1.f / ((wsum[0] + wsum[1]) + (wsum[2] + wsum[3]));
    if (tid < 64) {
        float o = 0.f;
#pragma unroll
        for (int gg = 0; gg < 16; ++gg) o += red[gg * 64 + tid];
        a.out[(long)b * a.ldo + head * 64 + tid] = o * inv;
    }
    if (a.att) {
        const int cap = a.att_cap[b];
        if (a.step < cap && sub == 0) {
            float* ap = a.att + a.att_off[b] + (((long)a.layer * gridDim.x + head) * cap + a.step) * n;
#pragma unroll
            for (int u = 0; u < NB; ++u)
                if (16 * u + g < n) ap[16 * u + g] = sc[u] * inv;
        }
    }
}


// prob_out + sigmoid + the stop rule of :638-642, one wave per utterance.  len[b] == 0 while utterance b runs.
// ln_g != NULL: z is the decoder's last row BEFORE after_norm, and the LayerNorm (eps 1e-5, two-pass in the wave) happens here --
// the feat_out row GEMM normalises the same row in its own prologue, so after_norm needs no launch of its own (A <= 1024).
__global__ __launch_bounds__(256) void k_tts_stop(const float* __restrict__ z, int A, const float* __restrict__ w,
                                                  float bias, int B, int step, float thr,
                                                  const int* __restrict__ minlen, const int* __restrict__ maxlen,
                                                  float* __restrict__ probs, int* __restrict__ len,
                                                  int* __restrict__ ndone, const float* __restrict__ ln_g,
                                                  const float* __restrict__ ln_b) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    float s = 0.f;
    if (ln_g) {
        float v[16];
        float t = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int c = lane + 64 * e;
            v[e] = c < A ? z[(long)b * A + c] : 0.f;
            t += v[e];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
        const float mean = t / (float)A;
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float d = lane + 64 * e < A ? v[e] - mean : 0.f;
            q += d * d;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
        const float rstd = 1.0f / sqrtf(q / (float)A + 1e-5f);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int c = lane + 64 * e;
            if (c < A) s = fmaf((v[e] - mean) * rstd * ln_g[c] + ln_b[c], w[c], s);
        }
    } else
    for (int c = lane; c < A; c += 64) s = fmaf(z[(long)b * A + c], w[c], s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) {
        const float p = 1.f / (1.f + expf(-(s + bias)));
        probs[(long)(step - 1) * B + b] = p;
        if (len[b] == 0 && (p >= thr || step >= maxlen[b]) && step >= minlen[b]) {
            len[b] = step;
            atomicAdd(ndone, 1);
        }
    }
}

// reduction_factor r > 1: prob_out has r outputs per step (w [A][r]); the utterance ends at a step where ANY of them
// reaches the threshold (:638-642).  probs[((step - 1) * B + b) * r + k].
__global__ __launch_bounds__(256) void k_tts_stop_r(const float* __restrict__ z, int A, const float* __restrict__ w,
                                                    const float* __restrict__ bias, int r, int B, int step, float thr,
                                                    const int* __restrict__ minlen, const int* __restrict__ maxlen,
                                                    float* __restrict__ probs, int* __restrict__ len,
                                                    int* __restrict__ ndone) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    bool any = false;
    for (int k = 0; k < r; ++k) {
        float s = 0.f;
        for (int c = lane; c < A; c += 64) s = fmaf(z[(long)b * A + c], w[(long)c * r + k], s);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float p = 1.f / (1.f + expf(-(s + bias[k])));
        if (lane == 0) probs[((long)(step - 1) * B + b) * r + k] = p;
        any = any || p >= thr;
    }
    if (lane == 0 && len[b] == 0 && (any || step >= maxlen[b]) && step >= minlen[b]) {
        len[b] = step;
        atomicAdd(ndone, 1);
    }
}

// k_ar_gather for r frames per step: frame p of utterance u is the C-wide slice p % r of step row (p / r + off) * B + u
__global__ __launch_bounds__(128) void k_tts_gather_r(const float* __restrict__ src, int C, int B, int r, int off,
                                                      const int* __restrict__ row_utt, const int* __restrict__ row_pos,
                                                      const int* __restrict__ rowmap, const float* __restrict__ cscale,
                                                      const float* __restrict__ cshift, float* __restrict__ dst) {
    const long q = blockIdx.x;
    const int u = row_utt[q];
    const long o = rowmap ? rowmap[q] : q;
    if (o < 0) return;
    if (u < 0) {
        if (!rowmap)
            for (int c = threadIdx.x; c < C; c += blockDim.x) dst[o * C + c] = 0.f;
        return;
    }
    const int p = row_pos[q];
    const float* s = src + (((long)(p / r + off) * B + u) * r + p % r) * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float v = s[c];
        if (cscale) v = v * cscale[c] + cshift[c];
        dst[o * C + c] = v;
    }
}

struct RowW {   // a layer as pk_rowgemm_pack tiles of its [K][N] matrix (+ bias) for the row GEMM
    size_t w = 0, b = (size_t)-1;
    size_t kn = (size_t)-1;   // the same matrix row-major [K][N] (k_ar_prenet_embed, pk_ar.h), when asked for
    int K = 0, N = 0;
};

struct DecLayer {
    size_t ln1_g, ln1_b, ln2_g, ln2_b, ln3_g, ln3_b;
    Dense qkv, out, src_q, src_kv, src_out, ffn1, ffn2;
    RowW r_qkv, r_out, r_src_q, r_src_out, r_ffn1, r_ffn2;
    RowW r_cat1_x, r_cat1_a, r_cat2_x, r_cat2_a;   // concat_linear1 / 2 (decoder_layer.py:66-68): input half (+ bias), attention half
};
}  // namespace

struct pk_tts : pk_fft_core {
    pk_tts_cfg cfg;
    pk_param_map params;
    bool finalized = false, inferred = false;
    int gapr = 1;
    bool dropout = true;
    bool kv_prefix = false;            // "kv_prefix" option (pk_tts_set_option), see pk_tts_infer
    bool overlap_prefix = true;        // "overlap_prefix": the NEXT step's prefix work (prenet .. layer-0 q|k|v of the rows that
                                       // exist already) on a side stream under this step's layer chain, see pk_tts_infer
    bool fuse_prenet = true;           // "fuse_prenet": prenet x 2 + input layer of a step's new rows in one launch (k_ar_prenet_embed, pk_ar.h)
    bool fuse_src_q = true;            // "fuse_src_q": the encoder-decoder attention projects its query itself (k_tts_attn_step64<8, true>)
    int side_cu_mask = 0;              // "overlap_cu_mask": 0 = an unmasked low-priority side stream (the loop's stream at the most urgent priority), 1 = the side stream on every other CU, 2 = ... and the loop's stream on the rest
    hipStream_t own_main = nullptr;    // the decoding loop's own stream (see pk_tts_infer), ordered against the caller's by ev_io
    hipEvent_t ev_io = nullptr;
    hipStream_t side = nullptr;        // ... the side stream and the two events that order it against the loop's stream
    hipEvent_t ev_main = nullptr, ev_side = nullptr;
    pk_dbuf d2_p0, d2_p1, d2_x0, d2_t, d2_ham, d2_pam, d2_qkv0;   // ... and the second set of the buffers it fills
    // weights
    size_t emb_table = 0;
    float alpha_enc = 1.f, alpha_dec = 1.f;
    float xscale = 1.f;                // PositionalEncoding (use_scaled_pos_enc=False): x * sqrt(adim) + pe (embedding.py:78)
    size_t dlin_ln_g = 0, dlin_ln_b = 0;   // LayerNorm of the "linear" decoder input layer (dprenet_layers == 0)
    std::vector<Dense> eprenet;
    Dense eprenet_lin;
    std::vector<pk_fft_layer> enc;
    size_t enc_after_g = 0, enc_after_b = 0, dec_after_g = 0, dec_after_b = 0;
    std::vector<Dense> dprenet;
    Dense dlin, feat_out;
    RowW r_feat_out;
    std::vector<RowW> r_dprenet;   // the decoder prenet's layers and the input Linear as row-GEMM layers: the NEW row block of a
    RowW r_dlin;                   // step goes through them (4 launches of 32 rows instead of 7 tile-GEMM launches)
    Dense kv0;    // layer 0's self-attention k | v only ([A][2A]) and its q as a row-GEMM layer: PK_TTS_KV_PREFIX (see pk_tts_infer)
    RowW r_q0;
    std::vector<DecLayer> dec;
    size_t prob_w = 0, prob_bv = 0;   // prob_out weight [A][r] and bias [r]
    float prob_b = 0.f;
    std::vector<Dense> postnet;
    size_t spk_w = 0, spk_b = 0;      // speaker part of `projection` ([D][A]) and its bias (:313-317)
    Dense spk_hs;                      // "concat": the [A][A] hidden-state part
    std::vector<float> cond_emb;       // speaker embeddings of the next infer (pk_tts_set_speakers)
    int cond_B = 0;
    pk_dbuf d_spk_emb, d_spk_vec;
    pk_gst gst;                        // global style tokens (use_gst)
    std::vector<float> cond_speech;    // reference spectrograms of the next infer (pk_tts_set_style_reference)
    std::vector<int> cond_speech_lens;
    pk_dbuf d_style;
    size_t out_scale = 0, out_shift = 0;
    bool has_out_affine = false;
    std::vector<float> h_out_scale, h_out_shift;
    // per call
    Timeline tl_tok, tl_frm;
    int B = 0, Lcap = 0, steps = 0;
    bool keep_att = false;
    std::vector<int> T, len, cap, frames;   // per utterance: tokens (+ eos), decoder steps, step capacity, frames = steps * r
    std::vector<long> att_off;
    long att_total = 0;
    pk_dbuf d_tok, d_e1, d_e2, d_tpe, d_hs, d_valid, d_y, d_p0, d_p1, d_x0, d_t, d_ham, d_pam, d_peb, d_rt, d_rc, d_rx, d_rq,
        d_rf, d_rz, d_ra, d_rn, d_probs, d_state, d_seeds, d_att, d_attoff, d_before, d_q1, d_q2, d_rowmap, d_stage, d_stage2;
    std::vector<pk_dbuf> d_qkv_l, d_xc_l, d_mkv_l;
};

// ---------------------------------------------------------------------------------------------- create / params
extern "C" int pk_tts_create(pk_ctx* ctx, const pk_tts_cfg* cfg, pk_tts** out) {
    if (!ctx || !cfg || !out) PK_FAIL(PK_EINVAL, "pk_tts_create: NULL argument");
    *out = nullptr;
    const pk_tts_cfg& c = *cfg;
    if (c.idim <= 1 || c.odim <= 0 || c.adim <= 0 || c.aheads <= 0)
        PK_FAIL(PK_EINVAL, "TransformerTTS: idim/odim/adim/aheads must be positive");
    if (c.adim % c.aheads != 0) PK_FAIL(PK_ESHAPE, "TransformerTTS: adim %% aheads != 0 (attention.py:40)");
    const int dk = c.adim / c.aheads;
    if (dk != 64 && dk != 96 && dk != 128 && dk != 192)
        PK_FAIL(PK_EUNSUPPORTED, "TransformerTTS: head size %d not built (64/96/128/192)", dk);
    if (c.adim % 64 != 0 || c.adim > 64 * PK_FFT_LN_MAXPER)
        PK_FAIL(PK_EUNSUPPORTED, "TransformerTTS: adim must be a multiple of 64, <= %d", 64 * PK_FFT_LN_MAXPER);
    if (c.reduction_factor < 1 || c.reduction_factor > 16) PK_FAIL(PK_EUNSUPPORTED, "TransformerTTS: reduction_factor must be in [1, 16]");
    if ((!c.decoder_normalize_before || c.decoder_concat_after) && pk_prof_env("PK_AR_ROWGEMM") && atoi(pk_prof_env("PK_AR_ROWGEMM")) == 0)
        PK_FAIL(PK_EUNSUPPORTED, "TransformerTTS: post-norm / concat_after decoder blocks run on the row-GEMM path only");
    if (c.spk_embed_dim < 0 || c.spk_embed_dim > 8192) PK_FAIL(PK_EUNSUPPORTED, "TransformerTTS: spk_embed_dim must be in [0, 8192]");
    if (c.spk_embed_dim > 0 && c.spk_embed_integration_type != 0 && c.spk_embed_integration_type != 1)
        PK_FAIL(PK_EUNSUPPORTED, "support only add or concat. (transformer_tts.py:753)");
    pk_gst_cfg gc;
    if (c.use_gst) {
        gc.idim = c.odim;   // the style encoder reads a mel spectrogram (:300)
        gc.tokens = c.gst_tokens; gc.token_dim = c.adim; gc.heads = c.gst_heads;
        gc.conv_layers = c.gst_conv_layers; gc.conv_kernel_size = c.gst_conv_kernel_size; gc.conv_stride = c.gst_conv_stride;
        gc.gru_layers = c.gst_gru_layers; gc.gru_units = c.gst_gru_units;
        for (int i = 0; i < PK_GST_MAX_CONV; ++i) gc.conv_chans[i] = c.gst_conv_chans[i];
        PK_TRY(pk_gst_check(gc));
    }
    if (c.dprenet_layers < 0) PK_FAIL(PK_EINVAL, "TransformerTTS: dprenet_layers must be >= 0");
    if (c.dprenet_layers > 0 && (c.dprenet_units % 16 != 0 || c.dprenet_units <= 0))
        PK_FAIL(PK_EUNSUPPORTED, "TransformerTTS: dprenet_units must be a positive multiple of 16");
    if (c.elayers < 0 || c.dlayers <= 0) PK_FAIL(PK_EINVAL, "TransformerTTS: elayers >= 0, dlayers > 0");
    if (c.positionwise_layer_type < 0 || c.positionwise_layer_type > 2)
        PK_FAIL(PK_EUNSUPPORTED, "Support only linear or conv1d. (encoder.py:169)");
    if (c.postnet_layers > 0 && !c.use_batch_norm)
        PK_FAIL(PK_EUNSUPPORTED, "TransformerTTS: postnet without batch norm not implemented");
    if (c.eprenet_conv_layers > 0 && !c.use_batch_norm)
        PK_FAIL(PK_EUNSUPPORTED, "TransformerTTS: encoder prenet without batch norm not implemented");
    if (c.eprenet_conv_layers < 0 || c.postnet_layers < 0) PK_FAIL(PK_EINVAL, "TransformerTTS: negative layer count");
    const int ks[] = {c.positionwise_conv_kernel_size, c.postnet_layers > 0 ? c.postnet_filts : 1,
                      c.eprenet_conv_layers > 0 ? c.eprenet_conv_filts : 1};
    int gapr = 1;
    for (int k : ks) {
        if (k < 1 || k % 2 == 0 || k > PK_GEMM_MAX_TAPS)
            PK_FAIL(PK_EUNSUPPORTED, "TransformerTTS: conv kernel size %d unsupported", k);
        gapr = std::max(gapr, (k - 1) / 2);
    }
    if (gapr > PK_FFT_LEAD) PK_FAIL(PK_EUNSUPPORTED, "TransformerTTS: conv kernel too wide");
    const int chans[] = {c.adim, c.eunits, c.dunits, c.odim, c.postnet_layers > 0 ? c.postnet_chans : 16,
                         c.eprenet_conv_layers > 0 ? c.eprenet_conv_chans : 16,
                         c.eprenet_conv_layers > 0 ? c.embed_dim : 16};
    for (int ch : chans)
        if (ch <= 0 || ch % PK_GEMM_BK != 0)
            PK_FAIL(PK_EUNSUPPORTED, "TransformerTTS: channel count %d not a positive multiple of 16", ch);
    pk_tts* h = new pk_tts();
    h->ctx = ctx;
    h->cfg = c;
    h->adim = c.adim;
    h->aheads = c.aheads;
    h->gapr = gapr;
    h->gst.cfg = gc;
    if (const char* e = pk_prof_env("PK_TTS_MATH")) h->math = strcmp(e, "f32") == 0 ? PK_GEMM_MATH_F32 : PK_GEMM_MATH_F16X3;
    *out = h;
    return PK_OK;
}

extern "C" int pk_tts_set_param(pk_tts* h, const char* name, const float* data, const int64_t* shape, int32_t ndim) {
    if (!h) PK_FAIL(PK_EINVAL, "pk_tts_set_param: handle is NULL");
    h->finalized = false;
    return pk_store_param(h->params, name, data, shape, ndim);
}

extern "C" int pk_tts_set_normalizer(pk_tts* h, const float* mu, const float* sigma, int32_t n) {
    if (!h) PK_FAIL(PK_EINVAL, "pk_tts_set_normalizer: handle is NULL");
    if (!mu && !sigma) {
        h->has_out_affine = false;
    } else {
        if (!mu || !sigma || n != h->cfg.odim) PK_FAIL(PK_ESHAPE, "normalizer needs mu and sigma of odim elements");
        h->h_out_scale.assign(sigma, sigma + n);   // ZScore.inverse: x * sigma + mu (normalizer.py:30-33)
        h->h_out_shift.assign(mu, mu + n);
        h->has_out_affine = true;
    }
    h->finalized = false;
    return PK_OK;
}

extern "C" int pk_tts_set_math(pk_tts* h, int32_t mode) {
    if (!h) PK_FAIL(PK_EINVAL, "pk_tts_set_math: handle is NULL");
    if (mode != PK_GEMM_MATH_F32 && mode != PK_GEMM_MATH_F16X3) PK_FAIL(PK_EINVAL, "pk_tts_set_math: unknown mode %d", mode);
    h->math = mode;
    return PK_OK;
}

int pk_fft_set_option(pk_fft_core* h, const char* key, int64_t value, const char* who);   // fs2.hip
extern "C" int pk_tts_set_option(pk_tts* h, const char* key, int64_t value) {
    if (!h || !key) PK_FAIL(PK_EINVAL, "pk_tts_set_option: NULL argument");
    if (strcmp(key, "kv_prefix") == 0) {
        h->kv_prefix = value != 0;
        return PK_OK;
    }
    if (strcmp(key, "overlap_prefix") == 0) {
        h->overlap_prefix = value != 0;
        return PK_OK;
    }
    if (strcmp(key, "fuse_prenet") == 0) {
        h->fuse_prenet = value != 0;
        return PK_OK;
    }
    if (strcmp(key, "fuse_src_q") == 0) {
        h->fuse_src_q = value != 0;
        return PK_OK;
    }
    if (strcmp(key, "overlap_cu_mask") == 0) {   // (takes effect when the side stream is created: before the first inference)
        h->side_cu_mask = value < 0 ? 0 : (value > 2 ? 2 : (int)value);
        return PK_OK;
    }
    return pk_fft_set_option(h, key, value, "pk_tts_set_option");
}

extern "C" int pk_tts_set_speakers(pk_tts* h, const float* spembs, int32_t B) {
    if (!h) PK_FAIL(PK_EINVAL, "pk_tts_set_speakers: handle is NULL");
    h->cond_emb.clear();
    h->cond_B = 0;
    if (!spembs) return PK_OK;
    if (h->cfg.spk_embed_dim <= 0) PK_FAIL(PK_ESTATE, "pk_tts_set_speakers: the model has no speaker embedding (spk_embed_dim=None)");
    if (B <= 0) PK_FAIL(PK_EINVAL, "pk_tts_set_speakers: batch size must be positive");
    h->cond_emb.assign(spembs, spembs + (size_t)B * h->cfg.spk_embed_dim);
    h->cond_B = B;
    return PK_OK;
}

extern "C" int pk_tts_set_style_reference(pk_tts* h, const float* speech, const int32_t* lens, int32_t B) {
    if (!h) PK_FAIL(PK_EINVAL, "pk_tts_set_style_reference: handle is NULL");
    h->cond_speech.clear();
    h->cond_speech_lens.clear();
    if (!speech) return PK_OK;
    if (!h->cfg.use_gst) PK_FAIL(PK_ESTATE, "pk_tts_set_style_reference: the model has no style encoder (use_gst=False)");
    if (!lens || B <= 0) PK_FAIL(PK_EINVAL, "pk_tts_set_style_reference: lens / batch size");
    size_t total = 0;
    for (int b = 0; b < B; ++b) {
        if (lens[b] <= 0) PK_FAIL(PK_EINVAL, "pk_tts_set_style_reference: reference %d has %d frames", b, lens[b]);
        total += (size_t)lens[b];
    }
    h->cond_speech.assign(speech, speech + total * h->cfg.odim);
    h->cond_speech_lens.assign(lens, lens + B);
    return PK_OK;
}

extern "C" int pk_tts_set_dropout(pk_tts* h, int32_t on) {
    if (!h) PK_FAIL(PK_EINVAL, "pk_tts_set_dropout: handle is NULL");
    h->dropout = on != 0;
    return PK_OK;
}

namespace {
// Linear (weight [in, out] + bias) with both multiplied by `scale`: (x . W + b) * scale as one dense layer
int add_linear_scaled(pk_fft_arena& ar, const pk_param_map& P, const std::string& base, int K, int N, float scale, Dense& d) {
    if (scale == 1.f) return pk_fft_add_linear(ar, P, base, K, N, d);
    std::vector<float> w, b;
    PK_TRY(pk_get_weight(P, base, {K, N}, w));
    PK_TRY(pk_get_vector(P, base + ".bias", N, b));
    for (float& v : w) v *= scale;
    for (float& v : b) v *= scale;
    return pk_fft_add_dense_kn(ar, w, &b, K, 1, N, d);
}

// k | v projections of one attention module fused into one [A][2A] dense layer
int add_kv(pk_fft_arena& ar, const pk_param_map& P, const std::string& p, int A, Dense& d) {
    std::vector<float> wk, wv, bk, bv, kn((size_t)A * 2 * A), bias(2 * A);
    PK_TRY(pk_get_weight(P, p + ".linear_k", {A, A}, wk));
    PK_TRY(pk_get_weight(P, p + ".linear_v", {A, A}, wv));
    PK_TRY(pk_get_vector(P, p + ".linear_k.bias", A, bk));
    PK_TRY(pk_get_vector(P, p + ".linear_v.bias", A, bv));
    for (int i = 0; i < A; ++i)
        for (int o = 0; o < A; ++o) {
            kn[(size_t)i * 2 * A + o] = wk[(size_t)i * A + o];
            kn[(size_t)i * 2 * A + A + o] = wv[(size_t)i * A + o];
        }
    for (int o = 0; o < A; ++o) {
        bias[o] = bk[o];
        bias[A + o] = bv[o];
    }
    return pk_fft_add_dense_kn(ar, kn, &bias, A, 1, 2 * A, d);
}

int add_row_linear(pk_fft_arena& ar, const pk_param_map& P, const std::string& base, int K, int N, RowW& r, float scale = 1.f,
                   bool keep_kn = false) {
    std::vector<float> w, b;
    PK_TRY(pk_get_weight(P, base, {K, N}, w));   // Linear weight [in, out] = [K][N]
    PK_TRY(pk_get_vector(P, base + ".bias", N, b));
    if (scale != 1.f) {   // (as add_linear_scaled: the sqrt(adim) of PositionalEncoding folded into the layer)
        for (float& v : w) v *= scale;
        for (float& v : b) v *= scale;
    }
    std::vector<float> wt;
    pk_rowgemm_pack(w.data(), K, N, wt);
    r.w = ar.put(wt);
    r.b = ar.put(b);
    if (keep_kn) r.kn = ar.put(w);
    r.K = K;
    r.N = N;
    return PK_OK;
}

// concat_linear{1,2}: Linear(2A -> A) on cat(x, attention output) as two row-GEMM layers, x . W[:A] + b and att . W[A:]
int add_row_concat(pk_fft_arena& ar, const pk_param_map& P, const std::string& base, int A, RowW& rx, RowW& ra) {
    std::vector<float> w, b, wt;
    PK_TRY(pk_get_weight(P, base, {2 * A, A}, w));
    PK_TRY(pk_get_vector(P, base + ".bias", A, b));
    pk_rowgemm_pack(w.data(), A, A, wt);
    rx.w = ar.put(wt);
    rx.b = ar.put(b);
    rx.K = A;
    rx.N = A;
    pk_rowgemm_pack(w.data() + (size_t)A * A, A, A, wt);
    ra.w = ar.put(wt);
    ra.b = (size_t)-1;
    ra.K = A;
    ra.N = A;
    return PK_OK;
}

int add_qkv(pk_fft_arena& ar, const pk_param_map& P, const std::string& p, int A, Dense& d, RowW& r) {
    std::vector<float> wq, wk, wv, bq, bk, bv, kn((size_t)A * 3 * A), bias(3 * A);
    PK_TRY(pk_get_weight(P, p + ".linear_q", {A, A}, wq));
    PK_TRY(pk_get_weight(P, p + ".linear_k", {A, A}, wk));
    PK_TRY(pk_get_weight(P, p + ".linear_v", {A, A}, wv));
    PK_TRY(pk_get_vector(P, p + ".linear_q.bias", A, bq));
    PK_TRY(pk_get_vector(P, p + ".linear_k.bias", A, bk));
    PK_TRY(pk_get_vector(P, p + ".linear_v.bias", A, bv));
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
    std::vector<float> wt;
    pk_rowgemm_pack(kn.data(), A, 3 * A, wt);
    r.w = ar.put(wt);
    r.b = ar.put(bias);
    r.K = A;
    r.N = 3 * A;
    return pk_fft_add_dense_kn(ar, kn, &bias, A, 1, 3 * A, d);
}
}  // namespace

extern "C" int pk_tts_finalize(pk_tts* h) {
    if (!h) PK_FAIL(PK_EINVAL, "pk_tts_finalize: handle is NULL");
    pk_ctx* ctx = h->ctx;
    PK_DEVICE(ctx->device);
    const pk_tts_cfg& c = h->cfg;
    const pk_param_map& P = h->params;
    const int A = c.adim;
    h->arena_h.clear();
    h->arena16_h.clear();
    pk_fft_arena ar{h->arena_h, &h->arena16_h};
    std::vector<float> al;
    // ScaledPositionalEncoding: x + alpha * pe; PositionalEncoding: x * sqrt(adim) + pe (embedding.py:78,125).  The
    // scale is folded into the Linear that produces x (or applied by the embedding lookup)
    h->xscale = c.use_scaled_pos_enc ? 1.f : std::sqrt((float)A);
    // encoder input layer (:258-277)
    if (c.eprenet_conv_layers > 0) {
        std::vector<float> t;
        PK_TRY(pk_get_weight(P, "encoder.embed.0.0.embed", {c.idim, c.embed_dim}, t));
        for (int i = 0; i < c.embed_dim; ++i) t[i] = 0.f;   // nn.Embedding(padding_idx=0): id 0 -> zero row
        h->emb_table = ar.put(t);
        h->eprenet.resize(c.eprenet_conv_layers);
        for (int i = 0; i < c.eprenet_conv_layers; ++i) {
            const std::string p = "encoder.embed.0.0.convs." + std::to_string(i);
            PK_TRY(pk_fft_add_conv_bn(ar, P, p + ".0", p + ".1", c.eprenet_conv_chans,
                                      i == 0 ? c.embed_dim : c.eprenet_conv_chans, c.eprenet_conv_filts, h->eprenet[i]));
        }
        PK_TRY(add_linear_scaled(ar, P, "encoder.embed.0.1", c.eprenet_conv_chans, A, h->xscale, h->eprenet_lin));
    } else {
        std::vector<float> t;
        PK_TRY(pk_get_weight(P, "encoder.embed.0", {c.idim, A}, t));
        for (int i = 0; i < A; ++i) t[i] = 0.f;
        h->emb_table = ar.put(t);
    }
    h->alpha_enc = 1.f;
    if (c.use_scaled_pos_enc) {
        PK_TRY(pk_get_vector(P, "encoder.embed.1.alpha", 1, al));
        h->alpha_enc = al[0];
    }
    PK_TRY(pk_fft_add_stack(ar, P, "encoder", c.elayers, A, c.eunits, c.positionwise_conv_kernel_size,
                            c.positionwise_layer_type, c.aheads, h->enc, h->enc_after_g, h->enc_after_b,
                            c.encoder_normalize_before != 0, c.encoder_concat_after != 0));
    if (c.use_gst) PK_TRY(pk_gst_finalize(ar, P, "gst", h->gst));
    if (c.spk_embed_dim > 0) {
        // `projection` (:313-317): Linear(D, adim) for "add", Linear(adim + D, adim) on concat([hs, e]) for "concat"
        const int D = c.spk_embed_dim;
        std::vector<float> w, b;
        PK_TRY(pk_get_vector(P, "projection.bias", A, b));
        h->spk_b = ar.put(b);
        if (c.spk_embed_integration_type == 0) {
            PK_TRY(pk_get_weight(P, "projection", {D, A}, w));
            h->spk_w = ar.put(w);
        } else {
            PK_TRY(pk_get_weight(P, "projection", {A + D, A}, w));
            std::vector<float> whs(w.begin(), w.begin() + (size_t)A * A), wsp(w.begin() + (size_t)A * A, w.end());
            PK_TRY(pk_fft_add_dense_kn(ar, whs, nullptr, A, 1, A, h->spk_hs));
            h->spk_w = ar.put(wsp);
        }
    }
    // decoder input layer: Sequential(Sequential(Prenet, Linear), ScaledPositionalEncoding) (:311-321, decoder.py:124-127)
    h->dprenet.resize(c.dprenet_layers);
    for (int j = 0; j < c.dprenet_layers; ++j)
        PK_TRY(pk_fft_add_linear(ar, P, "decoder.embed.0.0.prenet." + std::to_string(j) + ".0",
                                 j == 0 ? c.odim : c.dprenet_units, c.dprenet_units, h->dprenet[j]));
    h->r_dprenet.assign(c.dprenet_layers, RowW());
    for (int j = 0; j < c.dprenet_layers; ++j)
        PK_TRY(add_row_linear(ar, P, "decoder.embed.0.0.prenet." + std::to_string(j) + ".0", j == 0 ? c.odim : c.dprenet_units,
                              c.dprenet_units, h->r_dprenet[j], 1.f, true));
    h->alpha_dec = 1.f;
    if (c.dprenet_layers > 0) {
        PK_TRY(add_linear_scaled(ar, P, "decoder.embed.0.1", c.dprenet_units, A, h->xscale, h->dlin));
        PK_TRY(add_row_linear(ar, P, "decoder.embed.0.1", c.dprenet_units, A, h->r_dlin, h->xscale, true));
        if (c.use_scaled_pos_enc) {
            PK_TRY(pk_get_vector(P, "decoder.embed.1.alpha", 1, al));
            h->alpha_dec = al[0];
        }
    } else {
        // input_layer "linear" (decoder.py:112-118): Sequential(Linear(odim, adim), LayerNorm, Dropout, ReLU, pos_enc);
        // relu(s y) = s relu(y) for s > 0: the sqrt(adim) of PositionalEncoding goes into the LayerNorm's affine
        PK_TRY(pk_fft_add_linear(ar, P, "decoder.embed.0", c.odim, A, h->dlin));
        std::vector<float> g, b;
        PK_TRY(pk_get_vector(P, "decoder.embed.1.weight", A, g));
        PK_TRY(pk_get_vector(P, "decoder.embed.1.bias", A, b));
        for (float& v : g) v *= h->xscale;
        for (float& v : b) v *= h->xscale;
        h->dlin_ln_g = ar.put(g);
        h->dlin_ln_b = ar.put(b);
        if (c.use_scaled_pos_enc) {
            PK_TRY(pk_get_vector(P, "decoder.embed.4.alpha", 1, al));
            h->alpha_dec = al[0];
        }
    }
    h->dec.resize(c.dlayers);
    for (int l = 0; l < c.dlayers; ++l) {
        const std::string p = "decoder.decoders." + std::to_string(l);
        DecLayer& L = h->dec[l];
        PK_TRY(pk_fft_add_vec(ar, P, p + ".norm1.weight", A, L.ln1_g));
        PK_TRY(pk_fft_add_vec(ar, P, p + ".norm1.bias", A, L.ln1_b));
        PK_TRY(pk_fft_add_vec(ar, P, p + ".norm2.weight", A, L.ln2_g));
        PK_TRY(pk_fft_add_vec(ar, P, p + ".norm2.bias", A, L.ln2_b));
        PK_TRY(pk_fft_add_vec(ar, P, p + ".norm3.weight", A, L.ln3_g));
        PK_TRY(pk_fft_add_vec(ar, P, p + ".norm3.bias", A, L.ln3_b));
        PK_TRY(add_qkv(ar, P, p + ".self_attn", A, L.qkv, L.r_qkv));
        PK_TRY(pk_fft_add_linear(ar, P, p + ".self_attn.linear_out", A, A, L.out));
        PK_TRY(pk_fft_add_linear(ar, P, p + ".src_attn.linear_q", A, A, L.src_q));
        PK_TRY(add_kv(ar, P, p + ".src_attn", A, L.src_kv));
        PK_TRY(pk_fft_add_linear(ar, P, p + ".src_attn.linear_out", A, A, L.src_out));
        PK_TRY(pk_fft_add_linear(ar, P, p + ".feed_forward.w_1", A, c.dunits, L.ffn1));   // PositionwiseFeedForward
        PK_TRY(pk_fft_add_linear(ar, P, p + ".feed_forward.w_2", c.dunits, A, L.ffn2));
        PK_TRY(add_row_linear(ar, P, p + ".self_attn.linear_out", A, A, L.r_out));
        PK_TRY(add_row_linear(ar, P, p + ".src_attn.linear_q", A, A, L.r_src_q));
        PK_TRY(add_row_linear(ar, P, p + ".src_attn.linear_out", A, A, L.r_src_out));
        PK_TRY(add_row_linear(ar, P, p + ".feed_forward.w_1", A, c.dunits, L.r_ffn1));
        PK_TRY(add_row_linear(ar, P, p + ".feed_forward.w_2", c.dunits, A, L.r_ffn2));
        if (c.decoder_concat_after) {
            PK_TRY(add_row_concat(ar, P, p + ".concat_linear1", A, L.r_cat1_x, L.r_cat1_a));
            PK_TRY(add_row_concat(ar, P, p + ".concat_linear2", A, L.r_cat2_x, L.r_cat2_a));
        }
    }
    PK_TRY(add_kv(ar, P, "decoder.decoders.0.self_attn", A, h->kv0));
    PK_TRY(add_row_linear(ar, P, "decoder.decoders.0.self_attn.linear_q", A, A, h->r_q0));
    if (c.decoder_normalize_before) {   // after_norm exists only then (decoder.py:168-169)
        PK_TRY(pk_fft_add_vec(ar, P, "decoder.after_norm.weight", A, h->dec_after_g));
        PK_TRY(pk_fft_add_vec(ar, P, "decoder.after_norm.bias", A, h->dec_after_b));
    }
    // feat_out: adim -> odim * reduction_factor, prob_out: adim -> reduction_factor (:348-349)
    PK_TRY(pk_fft_add_linear(ar, P, "feat_out", A, c.odim * c.reduction_factor, h->feat_out));
    PK_TRY(add_row_linear(ar, P, "feat_out", A, c.odim * c.reduction_factor, h->r_feat_out));
    {
        std::vector<float> w, b;
        PK_TRY(pk_get_weight(P, "prob_out", {A, c.reduction_factor}, w));
        PK_TRY(pk_get_vector(P, "prob_out.bias", c.reduction_factor, b));
        h->prob_w = ar.put(w);
        h->prob_bv = ar.put(b);
        h->prob_b = b[0];
    }
    PK_TRY(pk_fft_add_postnet(ar, P, "postnet", c.postnet_layers, c.odim, c.postnet_chans, c.postnet_filts, h->postnet));
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
    h->d_qkv_l.resize(c.dlayers);
    h->d_xc_l.resize(c.dlayers);
    h->d_mkv_l.resize(c.dlayers);
    h->finalized = true;
    h->inferred = false;
    return PK_OK;
}

// ---------------------------------------------------------------------------------------------- inference
namespace {
constexpr int SLACK = 2 * PK_GEMM_BM;   // rows a GEMM tile may read beyond the rows it was asked for

int rows_reserve(pk_dbuf& buf, long rows, int C) { return pk_fft_act_reserve(buf, (int)(rows + SLACK), C); }

int attn_step(pk_tts* h, const char* name, const AttnStep& a, int heads, int B, int nmax) {
    if (a.qx) {   // the query projected in the kernel: 64-wide heads, at most 256 keys, K <= 512 (the callers check)
        if (nmax <= 128) PK_LAUNCH(h->ctx, name, (k_tts_attn_step64<8, true>), dim3(heads, B), dim3(256), 0, a);
        else PK_LAUNCH(h->ctx, name, (k_tts_attn_step64<16, true>), dim3(heads, B), dim3(256), 0, a);
        return PK_OK;
    }
    if (a.dk == 64 && nmax <= 640 && a.ldkv % 4 == 0 && a.ldq % 4 == 0) {   // (keys beyond 640: the general kernel)
        const int nb = (nmax + 15) / 16;
        if (nb <= 8) PK_LAUNCH(h->ctx, name, k_tts_attn_step64<8>, dim3(heads, B), dim3(256), 0, a);
        else if (nb <= 16) PK_LAUNCH(h->ctx, name, k_tts_attn_step64<16>, dim3(heads, B), dim3(256), 0, a);
        else if (nb <= 24) PK_LAUNCH(h->ctx, name, k_tts_attn_step64<24>, dim3(heads, B), dim3(256), 0, a);
        else if (nb <= 32) PK_LAUNCH(h->ctx, name, k_tts_attn_step64<32>, dim3(heads, B), dim3(256), 0, a);
        else PK_LAUNCH(h->ctx, name, k_tts_attn_step64<40>, dim3(heads, B), dim3(256), 0, a);
        return PK_OK;
    }
    const size_t smem = (size_t)(a.dk + 1032 + nmax + 4) * sizeof(float);
    if (smem > 60 * 1024) PK_FAIL(PK_EUNSUPPORTED, "TransformerTTS: %d attention keys exceed the step kernel's LDS budget", nmax);
    PK_LAUNCH(h->ctx, name, k_tts_attn_step, dim3(heads, B), dim3(256), smem, a);
    return PK_OK;
}

int encode(pk_tts* h, const int64_t* ids, const int32_t* tok_lens, int B, const std::vector<float>& spembs,
           const std::vector<float>& speech, const std::vector<int>& speech_lens) {
    pk_ctx* ctx = h->ctx;
    const pk_tts_cfg& c = h->cfg;
    const int A = c.adim;
    PK_TRY(pk_fft_build_timeline(ctx, h->tl_tok, h->T.data(), B, h->gapr));
    Timeline& tl = h->tl_tok;
    {
        std::vector<int> tok(tl.rows_alloc, 0);
        long o = 0;
        for (int b = 0; b < B; ++b) {
            for (int t = 0; t < tok_lens[b]; ++t, ++o) {
                const int64_t id = ids[o];
                if (id < 0 || id >= c.idim) PK_FAIL(PK_EINVAL, "pk_tts_infer: token id %lld out of [0,%d)", (long long)id, c.idim);
                tok[tl.seg_start[b] + t] = (int)id;
            }
            tok[tl.seg_start[b] + tok_lens[b]] = c.idim - 1;   // <eos> (:563-565)
        }
        PK_TRY(pk_upload(ctx, h->d_tok, tok.data(), tok.size() * sizeof(int)));
    }
    PK_TRY(pk_fft_act_reserve(h->d_x, tl.rows, A));
    PK_TRY(pk_fft_act_reserve(h->d_hs, tl.rows, A));
    float* x = pk_fft_act_ptr(h->d_x, A);
    float* hs = pk_fft_act_ptr(h->d_hs, A);
    if (c.eprenet_conv_layers > 0) {
        const int E = c.embed_dim, Cc = c.eprenet_conv_chans;
        PK_TRY(pk_fft_act_reserve(h->d_e1, tl.rows, std::max(E, Cc)));
        PK_TRY(pk_fft_act_reserve(h->d_e2, tl.rows, std::max(E, Cc)));
        PK_TRY(pk_fft_act_reserve(h->d_tpe, tl.rows, A));
        // the activation buffers are shared between widths: pointers are taken with the width actually stored
        float* cur = pk_fft_act_ptr(h->d_e1, E);
        PK_HIP(hipMemsetAsync(h->d_e1.p, 0, h->d_e1.cap, ctx->stream));   // margins read by the k > 1 taps
        PK_HIP(hipMemsetAsync(h->d_e2.p, 0, h->d_e2.cap, ctx->stream));
        PK_LAUNCH(ctx, "tts_lookup", k_tts_lookup, dim3(tl.rows), dim3(128), 0, h->d_tok.as<int>(), tl.d_row_utt(),
                  h->W(h->emb_table), E, cur);
        int ldin = E;
        for (int i = 0; i < c.eprenet_conv_layers; ++i) {
            // Conv1D(no bias) -> BatchNorm1D -> ReLU (-> Dropout: eval) (tacotron2/encoder.py:98-110); gap rows -> 0
            float* nxt = pk_fft_act_ptr((i & 1) ? h->d_e1 : h->d_e2, Cc);
            PK_TRY(pk_fft_run_dense(h, "tts_conv_eprenet", h->eprenet[i], cur, ldin, nxt, Cc, tl.rows, PK_ACT_RELU, nullptr,
                                    0, tl.d_row_utt()));
            cur = nxt;
            ldin = Cc;
        }
        float* tpe = pk_fft_act_ptr(h->d_tpe, A);
        PK_LAUNCH(ctx, "tts_pe", k_tts_pe_timeline, dim3(tl.rows), dim3(128), 0, h->d_pe.as<float>(), h->alpha_enc,
                  tl.d_row_utt(), tl.d_row_pos(), A, tpe);
        // Linear + x + alpha * pe
        PK_TRY(pk_fft_run_dense(h, "tts_gemm_eprenet_lin", h->eprenet_lin, cur, ldin, x, A, tl.rows, PK_ACT_NONE, tpe, A,
                                tl.d_row_utt()));
    } else {
        PK_TRY(pk_fft_embed(h, "tts_embed", h->d_tok.as<int>(), tl, h->emb_table, h->alpha_enc, h->xscale, x));
    }
    PK_TRY(pk_fft_run_stack(h, h->enc, h->enc_after_g, h->enc_after_b, tl, c.eunits, hs, c.encoder_normalize_before != 0));
    if (c.use_gst) {
        // hs = hs + gst(speech).unsqueeze(1) (:586-588)
        PK_TRY(h->d_style.reserve((size_t)B * A * sizeof(float)));
        PK_TRY(pk_gst_run(h, h->gst, speech.data(), speech_lens.data(), B, h->d_style.as<float>()));
        PK_TRY(pk_fft_add_rowvec(h, tl, h->d_style.as<float>(), hs));
    }
    if (c.spk_embed_dim > 0) {
        // hs = _integrate_with_spk_embed(hs, spembs) (:591-593, :725-755); the residual stream x is free by now
        PK_TRY(pk_upload(ctx, h->d_spk_emb, spembs.data(), spembs.size() * sizeof(float)));
        PK_TRY(pk_fft_run_speaker(h, tl, nullptr, h->d_spk_emb.as<float>(), 0, h->spk_w, h->spk_b,
                                  c.spk_embed_integration_type == 1 ? &h->spk_hs : nullptr, c.spk_embed_dim, h->d_spk_vec, hs, x));
    }
    // encoder-decoder attention: K | V of the memory, once per decoder layer
    for (int l = 0; l < c.dlayers; ++l) {
        PK_TRY(pk_fft_act_reserve(h->d_mkv_l[l], tl.rows, 2 * A));
        PK_TRY(pk_fft_run_dense(h, "tts_gemm_mem_kv", h->dec[l].src_kv, hs, A, pk_fft_act_ptr(h->d_mkv_l[l], 2 * A), 2 * A,
                                tl.rows, PK_ACT_NONE, nullptr, 0, nullptr));
    }
    return PK_OK;
}
}  // namespace

extern "C" int pk_tts_infer(pk_tts* h, const int64_t* ids, const int32_t* tok_lens, int32_t B, double threshold,
                            double minlenratio, double maxlenratio, const uint64_t* seeds, int32_t flags,
                            int32_t* out_frames) {
    if (!h) PK_FAIL(PK_EINVAL, "pk_tts_infer: NULL argument");
    // the per-call conditioning is consumed by this call, whatever happens next
    std::vector<float> spembs;
    spembs.swap(h->cond_emb);
    const int condB = h->cond_B;
    h->cond_B = 0;
    std::vector<float> speech;
    std::vector<int> speech_lens;
    speech.swap(h->cond_speech);
    speech_lens.swap(h->cond_speech_lens);
    if (!ids || !tok_lens || !out_frames) PK_FAIL(PK_EINVAL, "pk_tts_infer: NULL argument");
    if (!h->finalized) PK_FAIL(PK_ESTATE, "pk_tts_infer: call pk_tts_finalize first");
    if (B <= 0) PK_FAIL(PK_EINVAL, "pk_tts_infer: batch size must be positive");
    if (h->cfg.spk_embed_dim > 0 && condB != B)
        PK_FAIL(PK_EINVAL, "pk_tts_infer: the model integrates a speaker embedding into the encoder output (:591-593): "
                           "pk_tts_set_speakers needs %d rows, got %d", B, condB);
    if (h->cfg.use_gst && (int)speech_lens.size() != B)
        PK_FAIL(PK_EINVAL, "pk_tts_infer: the model adds a style embedding of a reference spectrogram to the encoder output "
                           "(:586-588): pk_tts_set_style_reference needs %d spectrograms, got %d", B, (int)speech_lens.size());
    if (!(minlenratio >= 0.0) || !(maxlenratio >= 0.0)) PK_FAIL(PK_EINVAL, "pk_tts_infer: length ratios must be >= 0");
    pk_ctx* ctx = h->ctx;
    PK_DEVICE(ctx->device);
    const pk_tts_cfg& c = h->cfg;
    const int A = c.adim, H = c.aheads, dk = A / H, O = c.odim, J = c.dprenet_layers, U = J > 0 ? c.dprenet_units : 16;
    const int RF = c.reduction_factor, OR = O * RF;   // a decoder step emits RF frames: its Y row is [frame 0 | ... | frame RF-1]
    h->inferred = false;
    h->B = B;
    h->keep_att = (flags & PK_TTS_KEEP_ATT) != 0;
    h->T.resize(B);
    h->cap.resize(B);
    std::vector<int> minlen(B), maxlen(B);
    int maxT = 0, Lcap = 1;
    for (int b = 0; b < B; ++b) {
        if (tok_lens[b] < 0) PK_FAIL(PK_EINVAL, "pk_tts_infer: utterance %d has %d tokens", b, tok_lens[b]);
        h->T[b] = tok_lens[b] + 1;                                   // with <eos>
        maxlen[b] = (int)((double)h->T[b] * maxlenratio / (double)RF);      // :597-598, counted in decoder steps
        minlen[b] = (int)((double)h->T[b] * minlenratio / (double)RF);
        h->cap[b] = std::max(1, std::max(maxlen[b], minlen[b]));
        maxT = std::max(maxT, h->T[b]);
        Lcap = std::max(Lcap, h->cap[b]);
    }
    h->Lcap = Lcap;
    const long rowsCap = (long)Lcap * B;
    if (rowsCap + B + SLACK > 0x3fffffff) PK_FAIL(PK_EUNSUPPORTED, "pk_tts_infer: %ld decoder rows", rowsCap);
    PK_TRY(pk_fft_ensure_pe(h, std::max(maxT, Lcap)));
    PK_TRY(encode(h, ids, tok_lens, B, spembs, speech, speech_lens));
    const Timeline& tlk = h->tl_tok;
    // ---- decoder state
    PK_TRY(rows_reserve(h->d_y, rowsCap + B, OR));
    PK_TRY(rows_reserve(h->d_p0, rowsCap, U));
    PK_TRY(rows_reserve(h->d_p1, rowsCap, U));
    PK_TRY(rows_reserve(h->d_x0, rowsCap, A));
    PK_TRY(rows_reserve(h->d_t, rowsCap, A));
    PK_TRY(rows_reserve(h->d_ham, rowsCap, 1));
    PK_TRY(rows_reserve(h->d_pam, 2 * (rowsCap + SLACK), 1));
    PK_HIP(hipMemsetAsync(h->d_pam.p, 0, h->d_pam.cap, ctx->stream));   // (rows beyond the prefix are read as scales of unused tile rows)
    PK_TRY(rows_reserve(h->d_peb, rowsCap, A));
    // Overlap of the per-step prefix work with the layer chain (option "overlap_prefix").  decoder.embed and layer 0's q | k | v are
    // recomputed for EVERY prefix row at every step (fresh prenet dropout, decoder.py:210), but for step s + 1 only the newest
    // row block depends on step s: the blocks 0 .. s - 1 are issued on a side stream while step s's 47 small dependent launches
    // run on the main one (they leave most of the chip idle), into a second set of buffers (step parity); at step s + 1 the main
    // stream waits for them and adds the new block.  Row results do not depend on how the rows are grouped into launches
    // (per-row operand scales), so the spectrogram is the sequential path's bit for bit.
    const bool overlap = h->overlap_prefix && !h->kv_prefix;
    if (overlap) {
        PK_TRY(rows_reserve(h->d2_p0, rowsCap, U));
        PK_TRY(rows_reserve(h->d2_p1, rowsCap, U));
        PK_TRY(rows_reserve(h->d2_x0, rowsCap, A));
        PK_TRY(rows_reserve(h->d2_t, rowsCap, A));
        PK_TRY(rows_reserve(h->d2_ham, rowsCap, 1));
        PK_TRY(rows_reserve(h->d2_pam, 2 * (rowsCap + SLACK), 1));
        PK_HIP(hipMemsetAsync(h->d2_pam.p, 0, h->d2_pam.cap, ctx->stream));
        PK_TRY(rows_reserve(h->d2_qkv0, rowsCap, 3 * A));
        if (!h->side) {
            // "overlap_cu_mask" >= 1: the side stream gets HALF of the CUs (hipExtStreamCreateWithCUMask) -- with the row GEMM and
            // step-attention kernels of round 3 this was worth 4 % (the prefix GEMMs are grids of a thousand workgroups that fill
            // the chip for 30 - 90 us at a time); with round 4's kernels and stream priorities it no longer is (see below).
            uint32_t mask[16];
            const int words = std::min(16, (ctx->n_cu + 31) / 32);
            for (int i = 0; i < words; ++i) mask[i] = 0x55555555u;   // every other CU, all XCDs / shader engines alike
            // (a CU-masked stream is a BLOCKING stream -- it synchronises implicitly with the NULL stream, which is what torch's
            // default stream is -- so the loop itself moves to a stream of the engine's own; the caller's stream only waits
            // for it at the end)
            // The loop's own stream.  "overlap_cu_mask" 2 puts it on the OTHER half of the CUs, so that no workgroup of the chain
            // shares a CU with the prefix GEMMs.  Round 4, one box, us per step (profiles/r04_tts_options_ab.txt): everything in
            // order 568; overlapped 518 - 520 with no masks (side stream at the least, this stream at the most urgent priority),
            // 528 - 530 with the side stream masked, 528 - 534 with complementary masks: where the workgroups run is not what the
            // overlap loses -- the chain's kernels are latency chains through a memory system the prefix GEMMs keep busy.  No
            // masks is the default.
            if (h->side_cu_mask >= 2) {
                uint32_t other[16];
                for (int i = 0; i < words; ++i) other[i] = 0xAAAAAAAAu;
                if (hipExtStreamCreateWithCUMask(&h->own_main, (uint32_t)words, other) != hipSuccess) {
                    (void)hipGetLastError();
                    h->own_main = nullptr;
                }
            }
            if (!h->own_main) {   // (else: at the most urgent priority)
                int lo = 0, hi = 0;
                (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
                PK_HIP(hipStreamCreateWithPriority(&h->own_main, hipStreamNonBlocking, hi));
            }
            PK_HIP(hipEventCreateWithFlags(&h->ev_io, hipEventDisableTiming));
            if (h->side_cu_mask == 0 || hipExtStreamCreateWithCUMask(&h->side, (uint32_t)words, mask) != hipSuccess) {
                (void)hipGetLastError();
                int lo = 0, hi = 0;
                (void)hipDeviceGetStreamPriorityRange(&lo, &hi);   // (lo = the numerically largest = least urgent)
                PK_HIP(hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, lo));
            }
            PK_HIP(hipEventCreateWithFlags(&h->ev_main, hipEventDisableTiming));
            PK_HIP(hipEventCreateWithFlags(&h->ev_side, hipEventDisableTiming));
        }
    }
    for (int l = 0; l < c.dlayers; ++l) {
        PK_TRY(rows_reserve(h->d_qkv_l[l], rowsCap, 3 * A));
        PK_TRY(rows_reserve(h->d_xc_l[l], rowsCap, A));
    }
    pk_dbuf* rowbufs[] = {&h->d_rt, &h->d_rc, &h->d_rx, &h->d_rq, &h->d_rz, &h->d_ra, &h->d_rn};
    for (pk_dbuf* rb : rowbufs) PK_TRY(rows_reserve(*rb, B, A));
    PK_TRY(rows_reserve(h->d_rf, B, c.dunits));
    PK_TRY(h->d_probs.reserve((size_t)(rowsCap + B) * RF * sizeof(float)));
    // rows of a "timeline" whose every row is valid, for the LayerNorm launcher
    {
        const size_t nvalid = (size_t)(rowsCap + B + SLACK);
        PK_TRY(h->d_valid.reserve(nvalid * sizeof(int)));
        PK_HIP(hipMemsetAsync(h->d_valid.p, 0, nvalid * sizeof(int), ctx->stream));
    }
    // state block: [len B][minlen B][maxlen B][cap B][ndone 1]
    {
        std::vector<int> st(4 * (size_t)B + 1, 0);
        for (int b = 0; b < B; ++b) {
            st[B + b] = minlen[b];
            st[2 * B + b] = maxlen[b];
            st[3 * B + b] = h->cap[b];
        }
        PK_TRY(pk_upload(ctx, h->d_state, st.data(), st.size() * sizeof(int)));
    }
    int* d_len = h->d_state.as<int>();
    const int* d_minlen = d_len + B;
    const int* d_maxlen = d_len + 2 * B;
    const int* d_cap = d_len + 3 * B;
    int* d_ndone = d_len + 4 * B;
    const unsigned long long* d_seeds = nullptr;
    if (seeds) {
        PK_TRY(pk_upload(ctx, h->d_seeds, seeds, (size_t)B * sizeof(uint64_t)));
        d_seeds = h->d_seeds.as<unsigned long long>();
    }
    float* att = nullptr;
    const long* d_attoff = nullptr;
    h->att_off.assign(B, 0);
    h->att_total = 0;
    if (h->keep_att) {
        for (int b = 0; b < B; ++b) {
            h->att_off[b] = h->att_total;
            h->att_total += (long)c.dlayers * H * h->cap[b] * h->T[b];
        }
        PK_TRY(h->d_att.reserve((size_t)h->att_total * sizeof(float)));
        PK_TRY(pk_upload(ctx, h->d_attoff, h->att_off.data(), (size_t)B * sizeof(long)));
        att = h->d_att.as<float>();
        d_attoff = h->d_attoff.as<long>();
    }
    float* Y = pk_fft_act_ptr(h->d_y, OR);
    struct PrefixSet {   // what the prefix work of one step writes (two sets under "overlap_prefix": step parity)
        float *P[2], *X0, *Tn, *ham, *pam[2], *QKV0;
    };
    PrefixSet sets[2];
    sets[0] = {{pk_fft_act_ptr(h->d_p0, U), pk_fft_act_ptr(h->d_p1, U)}, pk_fft_act_ptr(h->d_x0, A), pk_fft_act_ptr(h->d_t, A),
               pk_fft_act_ptr(h->d_ham, 1), {pk_fft_act_ptr(h->d_pam, 1), pk_fft_act_ptr(h->d_pam, 1) + rowsCap + SLACK},
               pk_fft_act_ptr(h->d_qkv_l[0], 3 * A)};
    sets[1] = sets[0];
    if (overlap)
        sets[1] = {{pk_fft_act_ptr(h->d2_p0, U), pk_fft_act_ptr(h->d2_p1, U)}, pk_fft_act_ptr(h->d2_x0, A), pk_fft_act_ptr(h->d2_t, A),
                   pk_fft_act_ptr(h->d2_ham, 1), {pk_fft_act_ptr(h->d2_pam, 1), pk_fft_act_ptr(h->d2_pam, 1) + rowsCap + SLACK},
                   pk_fft_act_ptr(h->d2_qkv0, 3 * A)};
    float* PEB = pk_fft_act_ptr(h->d_peb, A);
    float* rt = pk_fft_act_ptr(h->d_rt, A);
    float* rc = pk_fft_act_ptr(h->d_rc, A);
    float* rx = pk_fft_act_ptr(h->d_rx, A);
    float* rq = pk_fft_act_ptr(h->d_rq, A);
    float* rz = pk_fft_act_ptr(h->d_rz, A);
    float* ra = pk_fft_act_ptr(h->d_ra, A);
    float* rn = pk_fft_act_ptr(h->d_rn, A);
    const bool post = !c.decoder_normalize_before, cat = c.decoder_concat_after != 0;
    float* rf = pk_fft_act_ptr(h->d_rf, c.dunits);
    const int* valid = h->d_valid.as<int>();
    // the decoding loop on the engine's own stream (overlap only): everything issued so far on the caller's stream first
    struct StreamGuard {   // (restores the context's stream on every return path)
        pk_ctx* c;
        hipStream_t user;
        ~StreamGuard() { c->stream = user; }
    } sguard{ctx, ctx->stream};
    if (overlap) {
        PK_HIP(hipEventRecord(h->ev_io, sguard.user));
        PK_HIP(hipStreamWaitEvent(h->own_main, h->ev_io, 0));
        ctx->stream = h->own_main;
    }
    PK_HIP(hipMemsetAsync(Y, 0, (size_t)B * OR * sizeof(float), ctx->stream));   // ys = zeros(1, 1, odim) (:601-602)
    PK_LAUNCH(ctx, "tts_pe", k_tts_pe_pos_major, dim3((unsigned)rowsCap), dim3(128), 0, h->d_pe.as<float>(),
              h->alpha_dec, B, A, PEB);
    const unsigned thr = h->dropout ? pk_dropout_threshold(0.5) : 0u;   // F.dropout's default p (decoder.py:80)
    const float dscale = 2.0f;
    const float att_scale = (float)(1.0 / std::sqrt((double)dk));
    static const int poll = pk_prof_env("PK_TTS_POLL") ? std::max(1, atoi(pk_prof_env("PK_TTS_POLL"))) : 4;
    const bool use_ham = h->math == PK_GEMM_MATH_F16X3;
    static const bool use_rg = pk_prof_env("PK_AR_ROWGEMM") ? atoi(pk_prof_env("PK_AR_ROWGEMM")) != 0 : true;
    // Experiment, off by default (not yet measured on a GPU): layer 0's query is needed for the NEW rows only, so the
    // prefix GEMM can project k | v alone (2/3 of its work) and the B new queries come from a row GEMM.
    const bool kv_prefix = h->kv_prefix;
    // y = [LayerNorm(x)] . W + b [ReLU] [+ res] for the B new rows of a step (pk_rowgemm.h)
    auto rowgemm = [&](const char* name, const RowW& w, const float* x, int ldx, float* y, int ldy, int act, const float* res,
                       int ldr, size_t ln_g, size_t ln_b, bool ln) -> int {
        pk_rowgemm_args g;
        g.x = x; g.ldx = ldx; g.Wt = h->W(w.w); g.bias = w.b == (size_t)-1 ? nullptr : h->W(w.b); g.y = y; g.ldy = ldy; g.M = B; g.K = w.K; g.N = w.N;
        g.act = act; g.res = res; g.ldr = ldr;
        if (ln) { g.ln_g = h->W(ln_g); g.ln_b = h->W(ln_b); }
        return pk_rowgemm_launch(ctx, name, g);
    };
    // The prefix work of step `st` for the rows [r0, r0 + n) (whole row blocks: r0, n multiples of B): prenet (+ dropout, stream
    // position st (st - 1) / 2 + row block), the input layer, the positional encoding, layer 0's norm1 and q | k | v.  Every
    // kernel here works row by row, so any split of the prefix into calls gives the same rows.  Runs on ctx->stream.
    auto prefix_rows = [&](int st, long r0, long n, PrefixSet& S) -> int {
        if (n <= 0) return PK_OK;
        const float* in = Y + (RF - 1) * O + r0 * OR;   // the LAST frame of every step's output is the next input (:619-621)
        int ldin = OR;
        const float* in_amax = nullptr;
        const unsigned long long base = (unsigned long long)st * (unsigned long long)(st - 1) / 2ull + (unsigned long long)(r0 / B);
        for (int j = 0; j < J; ++j) {
            float* o = S.P[j & 1] + r0 * U;
            PK_TRY(pk_fft_run_dense(h, "tts_gemm_prenet", h->dprenet[j], in, ldin, o, U, (int)n, PK_ACT_RELU, nullptr, 0, nullptr, in_amax));
            // the row maxima of what the next GEMM reads come out of the dropout kernel (one wave per row at 256 units)
            float* o_amax = (h->dropout && use_ham && U == 256) ? S.pam[j & 1] + r0 : nullptr;
            if (h->dropout)
                PK_LAUNCH(ctx, "tts_dropout", k_ar_dropout, dim3(pk_div_up(n * (U / 4), 256)), dim3(256), 0, o, U, (int)n, U,
                          B, base, J, j, d_seeds, thr, dscale, o_amax);
            in = o;
            ldin = U;
            in_amax = o_amax;
        }
        float* x0 = S.X0 + r0 * A;
        float* tn = S.Tn + r0 * A;
        float* hm = S.ham + r0;
        if (J > 0) {
            PK_TRY(pk_fft_run_dense(h, "tts_gemm_embed", h->dlin, in, ldin, x0, A, (int)n, PK_ACT_NONE, PEB + r0 * A, A, nullptr, in_amax));
        } else {
            PK_TRY(pk_fft_run_dense(h, "tts_gemm_embed", h->dlin, in, ldin, x0, A, (int)n, PK_ACT_NONE, nullptr, 0, nullptr));
            PK_TRY(pk_fft_layernorm_rows(h, x0, h->dlin_ln_g, h->dlin_ln_b, valid, (int)n, A, tn, nullptr));
            const long n4 = n * (A / 4);
            PK_LAUNCH(ctx, "tts_relu_pe", k_tts_relu_add, dim3(pk_div_up(n4, 256)), dim3(256), 0,
                      reinterpret_cast<const float4*>(tn), reinterpret_cast<const float4*>(PEB + r0 * A), n4, reinterpret_cast<float4*>(x0));
        }
        // layer 0: norm1 and q | k | v of the rows
        float* qkv0 = S.QKV0 + r0 * 3 * A;
        if (!post && kv_prefix && use_rg) {   // (never overlapped: r0 == 0, n == all rows of step st)
            PK_TRY(pk_fft_layernorm_rows(h, x0, h->dec[0].ln1_g, h->dec[0].ln1_b, valid, (int)n, A, tn, use_ham ? hm : nullptr));
            PK_TRY(pk_fft_run_dense(h, "tts_gemm_kv0", h->kv0, tn, A, qkv0 + A, 3 * A, (int)n, PK_ACT_NONE, nullptr, 0, nullptr,
                                    use_ham ? hm : nullptr));
            const long nr0 = (long)(st - 1) * B;
            if (r0 <= nr0 && nr0 < r0 + n)   // (the call that holds the new row block)
                PK_TRY(rowgemm("tts_row_q0", h->r_q0, S.X0 + nr0 * A, A, S.QKV0 + nr0 * 3 * A, 3 * A, PK_ACT_NONE, nullptr, 0,
                               h->dec[0].ln1_g, h->dec[0].ln1_b, true));
        } else if (!post) {
            PK_TRY(pk_fft_layernorm_rows(h, x0, h->dec[0].ln1_g, h->dec[0].ln1_b, valid, (int)n, A, tn, use_ham ? hm : nullptr));
            PK_TRY(pk_fft_run_dense(h, "tts_gemm_qkv0", h->dec[0].qkv, tn, A, qkv0, 3 * A, (int)n, PK_ACT_NONE, nullptr, 0, nullptr,
                                    use_ham ? hm : nullptr));
        } else {   // post-norm: the self-attention reads the un-normalised rows (decoder_layer.py:104-106)
            PK_TRY(pk_fft_run_dense(h, "tts_gemm_qkv0", h->dec[0].qkv, x0, A, qkv0, 3 * A, (int)n, PK_ACT_NONE, nullptr, 0, nullptr));
        }
        return PK_OK;
    };
    // The same for the NEW row block of step `st` on the row GEMM (exact fp32 FMA; the layer chain's kernels): prenet layers with
    // ReLU + dropout in the epilogue, the input Linear + positional encoding, layer 0's norm1 + q | k | v -- J + 2 launches of
    // B rows (the tile GEMM needs 20 - 60 us for such a problem).  Not with concat_after blocks (they read the normed rows).
    const bool fast_new = use_rg && J > 0 && !cat && !kv_prefix && (int)h->r_dprenet.size() == J;
    auto prefix_new = [&](int st, PrefixSet& S) -> int {
        const long r0 = (long)(st - 1) * B;
        if (!fast_new) return prefix_rows(st, r0, B, S);
        const float* in = Y + (RF - 1) * O + r0 * OR;
        int ldin = OR;
        auto phase_ok = [](int N) { return N % 4 == 0 && N / 4 <= 512 && 512 % (N / 4) == 0; };   // (column groups x K parts = 512 threads)
        if (h->fuse_prenet && J == 2 && phase_ok(U) && phase_ok(A) && U <= 512 && O <= 512 && h->r_dprenet[0].kn != (size_t)-1 &&
            h->r_dlin.kn != (size_t)-1) {
            // prenet x 2 + input layer + positional encoding of the new row block in one launch (k_ar_prenet_embed, pk_ar.h)
            pk_prenet_embed pe;
            pe.y = in; pe.ldy = ldin; pe.O = O; pe.U = U; pe.A = A; pe.B = B;
            pe.w1 = h->W(h->r_dprenet[0].kn); pe.b1 = h->W(h->r_dprenet[0].b);
            pe.w2 = h->W(h->r_dprenet[1].kn); pe.b2 = h->W(h->r_dprenet[1].b);
            pe.we = h->W(h->r_dlin.kn); pe.be = h->W(h->r_dlin.b);
            pe.peb = PEB + r0 * A; pe.ldpe = A;
            pe.x0 = S.X0 + r0 * A; pe.ldx0 = A;
            pe.dropout = h->dropout ? 1 : 0;
            pe.base = (unsigned long long)st * (unsigned long long)(st - 1) / 2ull + (unsigned long long)(st - 1);
            pe.J = J; pe.seeds = d_seeds; pe.thr = thr; pe.scale = dscale;
            PK_LAUNCH(ctx, "tts_prenet_embed", k_ar_prenet_embed, dim3(B), dim3(512), 0, pe);
            PK_TRY(rowgemm("tts_row_qkv", h->dec[0].r_qkv, S.X0 + r0 * A, A, S.QKV0 + r0 * 3 * A, 3 * A, PK_ACT_NONE, nullptr, 0,
                           h->dec[0].ln1_g, h->dec[0].ln1_b, !post));
            return PK_OK;
        }
        for (int j = 0; j < J; ++j) {
            const RowW& w = h->r_dprenet[j];
            pk_rowgemm_args g;
            g.x = in; g.ldx = ldin; g.Wt = h->W(w.w); g.bias = h->W(w.b); g.y = S.P[j & 1] + r0 * U; g.ldy = U; g.M = B; g.K = w.K; g.N = w.N;
            g.act = PK_ACT_RELU;
            if (h->dropout) {
                g.dropout = 1;
                g.drop_base = (unsigned long long)st * (unsigned long long)(st - 1) / 2ull + (unsigned long long)(st - 1);
                g.drop_J = J; g.drop_j = j; g.drop_seeds = d_seeds; g.drop_thr = thr; g.drop_scale = dscale;
            }
            PK_TRY(pk_rowgemm_launch(ctx, "tts_row_prenet", g));
            in = g.y;
            ldin = U;
        }
        PK_TRY(rowgemm("tts_row_embed", h->r_dlin, in, ldin, S.X0 + r0 * A, A, PK_ACT_NONE, PEB + r0 * A, A, 0, 0, false));
        PK_TRY(rowgemm("tts_row_qkv", h->dec[0].r_qkv, S.X0 + r0 * A, A, S.QKV0 + r0 * 3 * A, 3 * A, PK_ACT_NONE, nullptr, 0,
                       h->dec[0].ln1_g, h->dec[0].ln1_b, !post));
        return PK_OK;
    };
    int s = 0;
    for (s = 1; s <= Lcap; ++s) {
        const int R = s * B;
        const long nr = (long)(s - 1) * B;   // first new row
        // decoder.embed on the whole prefix (decoder.py:210) and layer 0's norm1 + q | k | v of every prefix row: prefix_rows().
        // Sequential: all R rows here.  Overlapped: the blocks 0 .. s - 2 of THIS step were issued on the side stream during the
        // previous step; the new block follows here, and the next step's old blocks go to the side stream now.
        PrefixSet& S = sets[s & 1];
        float* const X0 = S.X0;
        float* const Tn = S.Tn;
        float* const ham = S.ham;   // (rows 0 .. B - 1 double as scratch of the tile-GEMM variant of the layer chain)
        (void)ham;
        if (!overlap) {
            PK_TRY(prefix_rows(s, 0, nr, S));
            PK_TRY(prefix_new(s, S));
        } else {
            if (s > 1) PK_HIP(hipStreamWaitEvent(ctx->stream, h->ev_side, 0));   // this step's old blocks (issued last step)
            PK_HIP(hipEventRecord(h->ev_main, ctx->stream));                     // step s - 1 is complete on the main stream
            PK_TRY(prefix_new(s, S));
            if (s + 1 <= Lcap) {
                // step s + 1's blocks 0 .. s - 1 (inputs Y[0 .. s - 1]: all known) into the other set: free since step s - 1 ended
                hipStream_t main_stream = ctx->stream;
                PK_HIP(hipStreamWaitEvent(h->side, h->ev_main, 0));
                ctx->stream = h->side;
                const int st = prefix_rows(s + 1, 0, R, sets[(s + 1) & 1]);
                ctx->stream = main_stream;
                PK_TRY(st);
                PK_HIP(hipEventRecord(h->ev_side, h->side));
            }
        }
        for (int l = 0; l < c.dlayers; ++l) {
            const DecLayer& L = h->dec[l];
            const float* xin = (l == 0 ? X0 : pk_fft_act_ptr(h->d_xc_l[l - 1], A)) + nr * A;
            float* qkv = l == 0 ? S.QKV0 : pk_fft_act_ptr(h->d_qkv_l[l], 3 * A);
            float* xc_new = pk_fft_act_ptr(h->d_xc_l[l], A) + nr * A;
            if (post || cat) {
                // post-norm and / or concat_after blocks: the same kernels in the order of decoder_layer.py:104-151.
                // tq = the new rows of tgt (normed with pre-norm blocks), needed on its own only by concat_linear1
                const float* tq = xin;
                if (!post && cat) {
                    if (l == 0) tq = Tn + nr * A;
                    else {
                        PK_TRY(pk_fft_layernorm_rows(h, xin, L.ln1_g, L.ln1_b, valid, B, A, rt, nullptr));
                        tq = rt;
                    }
                }
                if (l > 0)
                    PK_TRY(rowgemm("tts_row_qkv", L.r_qkv, xin, A, qkv + nr * 3 * A, 3 * A, PK_ACT_NONE, nullptr, 0, L.ln1_g, L.ln1_b, !post));
                AttnStep a;
                memset(&a, 0, sizeof(a));
                a.q = qkv + nr * 3 * A; a.ldq = 3 * A;
                a.K = qkv + A; a.V = qkv + 2 * A; a.ldkv = 3 * A;
                a.kbase = nullptr; a.klen = nullptr; a.kstride = B; a.n = s; a.dk = dk; a.scale = att_scale;
                a.out = rc; a.ldo = A;
                PK_TRY(attn_step(h, "tts_attn_self", a, H, B, s));
                float* x1 = post ? rn : rx;   // x after the self-attention sub-block, before its post-norm
                if (cat) {
                    PK_TRY(rowgemm("tts_row_attn_out", L.r_out, rc, A, ra, A, PK_ACT_NONE, nullptr, 0, 0, 0, false));
                    PK_TRY(rowgemm("tts_row_concat1", L.r_cat1_x, tq, A, x1, A, PK_ACT_NONE, xin, A, 0, 0, false));
                    PK_TRY(rowgemm("tts_row_concat1", L.r_cat1_a, ra, A, x1, A, PK_ACT_NONE, x1, A, 0, 0, false));
                } else {
                    PK_TRY(rowgemm("tts_row_attn_out", L.r_out, rc, A, x1, A, PK_ACT_NONE, xin, A, 0, 0, false));
                }
                if (post) PK_TRY(pk_fft_layernorm_rows(h, rn, L.ln1_g, L.ln1_b, valid, B, A, rx, nullptr));
                // rx = x; encoder-decoder attention on x2 = norm2(x) (pre-norm) or x (post-norm)
                const float* x2 = rx;
                if (!post && cat) {
                    PK_TRY(pk_fft_layernorm_rows(h, rx, L.ln2_g, L.ln2_b, valid, B, A, rt, nullptr));
                    x2 = rt;
                    PK_TRY(rowgemm("tts_row_src_q", L.r_src_q, rt, A, rq, A, PK_ACT_NONE, nullptr, 0, 0, 0, false));
                } else {
                    PK_TRY(rowgemm("tts_row_src_q", L.r_src_q, rx, A, rq, A, PK_ACT_NONE, nullptr, 0, L.ln2_g, L.ln2_b, !post));
                }
                const float* mkv = pk_fft_act_ptr(h->d_mkv_l[l], 2 * A);
                AttnStep a2;
                memset(&a2, 0, sizeof(a2));
                a2.q = rq; a2.ldq = A;
                a2.K = mkv; a2.V = mkv + A; a2.ldkv = 2 * A;
                a2.kbase = tlk.d_seg_start(); a2.klen = tlk.d_seg_len(); a2.kstride = 1; a2.n = 0; a2.dk = dk; a2.scale = att_scale;
                a2.out = rc; a2.ldo = A;
                a2.att = att; a2.att_off = d_attoff; a2.att_cap = d_cap; a2.layer = l; a2.step = s - 1;
                PK_TRY(attn_step(h, "tts_attn_src", a2, H, B, maxT));
                float* x3 = post ? rn : rx;
                if (cat) {
                    // x + concat_linear2(cat(x2, att)) accumulates in a buffer that is neither of the GEMMs' inputs (a row
                    // GEMM must not write the rows it reads): rn with post-norm blocks, rz (free until after_norm) otherwise
                    x3 = post ? rn : rz;
                    PK_TRY(rowgemm("tts_row_src_out", L.r_src_out, rc, A, ra, A, PK_ACT_NONE, nullptr, 0, 0, 0, false));
                    PK_TRY(rowgemm("tts_row_concat2", L.r_cat2_x, x2, A, x3, A, PK_ACT_NONE, rx, A, 0, 0, false));
                    PK_TRY(rowgemm("tts_row_concat2", L.r_cat2_a, ra, A, x3, A, PK_ACT_NONE, x3, A, 0, 0, false));
                } else {
                    PK_TRY(rowgemm("tts_row_src_out", L.r_src_out, rc, A, x3, A, PK_ACT_NONE, rx, A, 0, 0, false));
                }
                const float* x4 = x3;   // x after the second sub-block
                if (post) {
                    PK_TRY(pk_fft_layernorm_rows(h, x3, L.ln2_g, L.ln2_b, valid, B, A, rx, nullptr));
                    x4 = rx;
                }
                PK_TRY(rowgemm("tts_row_ffn1", L.r_ffn1, x4, A, rf, c.dunits, PK_ACT_RELU, nullptr, 0, L.ln3_g, L.ln3_b, !post));
                if (post) {
                    PK_TRY(rowgemm("tts_row_ffn2", L.r_ffn2, rf, c.dunits, rn, A, PK_ACT_NONE, x4, A, 0, 0, false));
                    PK_TRY(pk_fft_layernorm_rows(h, rn, L.ln3_g, L.ln3_b, valid, B, A, xc_new, nullptr));
                } else {
                    PK_TRY(rowgemm("tts_row_ffn2", L.r_ffn2, rf, c.dunits, xc_new, A, PK_ACT_NONE, x4, A, 0, 0, false));
                }
                continue;
            }
            if (l > 0) {
                if (use_rg) {
                    PK_TRY(rowgemm("tts_row_qkv", L.r_qkv, xin, A, qkv + nr * 3 * A, 3 * A, PK_ACT_NONE, nullptr, 0, L.ln1_g,
                                   L.ln1_b, true));
                } else {
                    PK_TRY(pk_fft_layernorm_rows(h, xin, L.ln1_g, L.ln1_b, valid, B, A, rt, use_ham ? ham : nullptr));
                    PK_TRY(pk_fft_run_dense(h, "tts_gemm_qkv", L.qkv, rt, A, qkv + nr * 3 * A, 3 * A, B, PK_ACT_NONE, nullptr,
                                            0, nullptr, use_ham ? ham : nullptr));
                }
            }
            AttnStep a;
            memset(&a, 0, sizeof(a));
            a.q = qkv + nr * 3 * A; a.ldq = 3 * A;
            a.K = qkv + A; a.V = qkv + 2 * A; a.ldkv = 3 * A;
            a.kbase = nullptr; a.klen = nullptr; a.kstride = B; a.n = s; a.dk = dk; a.scale = att_scale;
            a.out = rc; a.ldo = A;
            PK_TRY(attn_step(h, "tts_attn_self", a, H, B, s));
            const float* mkv = pk_fft_act_ptr(h->d_mkv_l[l], 2 * A);
            AttnStep a2;
            memset(&a2, 0, sizeof(a2));
            a2.q = rq; a2.ldq = A;
            a2.K = mkv; a2.V = mkv + A; a2.ldkv = 2 * A;
            a2.kbase = tlk.d_seg_start(); a2.klen = tlk.d_seg_len(); a2.kstride = 1; a2.n = 0; a2.dk = dk; a2.scale = att_scale;
            a2.out = rc; a2.ldo = A;
            a2.att = att; a2.att_off = d_attoff; a2.att_cap = d_cap; a2.layer = l; a2.step = s - 1;
            if (use_rg) {
                // x = residual + self_attn(...) (decoder_layer.py:127-128); x = residual + src_attn(norm2(x), memory)
                // (:132-141); x = residual + feed_forward(norm3(x)) (:145-148) -> the layer's cached output row
                PK_TRY(rowgemm("tts_row_attn_out", L.r_out, rc, A, rx, A, PK_ACT_NONE, xin, A, 0, 0, false));
                if (h->fuse_src_q && dk == 64 && maxT <= 256 && A <= 512 && A % 8 == 0) {
                    // linear_q of the encoder-decoder attention (norm2 in its prologue) inside the attention kernel
                    a2.q = nullptr;
                    a2.qx = rx; a2.ldqx = A; a2.qK = A;
                    a2.qW = h->W(L.r_src_q.w);
                    a2.qb = L.r_src_q.b == (size_t)-1 ? nullptr : h->W(L.r_src_q.b);
                    a2.q_ln_g = h->W(L.ln2_g); a2.q_ln_b = h->W(L.ln2_b);
                    PK_TRY(attn_step(h, "tts_attn_src_q", a2, H, B, maxT));
                } else {
                    PK_TRY(rowgemm("tts_row_src_q", L.r_src_q, rx, A, rq, A, PK_ACT_NONE, nullptr, 0, L.ln2_g, L.ln2_b, true));
                    PK_TRY(attn_step(h, "tts_attn_src", a2, H, B, maxT));
                }
                PK_TRY(rowgemm("tts_row_src_out", L.r_src_out, rc, A, rx, A, PK_ACT_NONE, rx, A, 0, 0, false));
                PK_TRY(rowgemm("tts_row_ffn1", L.r_ffn1, rx, A, rf, c.dunits, PK_ACT_RELU, nullptr, 0, L.ln3_g, L.ln3_b, true));
                PK_TRY(rowgemm("tts_row_ffn2", L.r_ffn2, rf, c.dunits, xc_new, A, PK_ACT_NONE, rx, A, 0, 0, false));
                continue;
            }
            PK_TRY(pk_fft_run_dense(h, "tts_gemm_attn_out", L.out, rc, A, rx, A, B, PK_ACT_NONE, xin, A, nullptr));
            PK_TRY(pk_fft_layernorm_rows(h, rx, L.ln2_g, L.ln2_b, valid, B, A, rt, use_ham ? ham : nullptr));
            PK_TRY(pk_fft_run_dense(h, "tts_gemm_src_q", L.src_q, rt, A, rq, A, B, PK_ACT_NONE, nullptr, 0, nullptr,
                                    use_ham ? ham : nullptr));
            PK_TRY(attn_step(h, "tts_attn_src", a2, H, B, maxT));
            PK_TRY(pk_fft_run_dense(h, "tts_gemm_src_out", L.src_out, rc, A, rx, A, B, PK_ACT_NONE, rx, A, nullptr));
            PK_TRY(pk_fft_layernorm_rows(h, rx, L.ln3_g, L.ln3_b, valid, B, A, rt, use_ham ? ham : nullptr));
            PK_TRY(pk_fft_run_dense(h, "tts_gemm_ffn1", L.ffn1, rt, A, rf, c.dunits, B, PK_ACT_RELU, nullptr, 0, nullptr,
                                    use_ham ? ham : nullptr));
            PK_TRY(pk_fft_run_dense(h, "tts_gemm_ffn2", L.ffn2, rf, c.dunits, xc_new, A, B, PK_ACT_NONE, rx, A, nullptr));
        }
        // after_norm of the last row, feat_out -> the next prefix row, prob_out -> stop state (:613-616, :638-642).
        // Row-GEMM path with pre-norm blocks and one frame per step: after_norm is the LayerNorm prologue of the feat_out row
        // GEMM and of the stop kernel (two launches instead of three); otherwise rz = the normalised row first.
        const float* xlast = pk_fft_act_ptr(h->d_xc_l[c.dlayers - 1], A) + nr * A;
        const bool fuse_after = !post && use_rg && RF == 1 && A <= 1024 && A <= PK_RG_KC;
        if (fuse_after && B <= PK_RG_ROWS) {
            // (one launch: the stop-token head rides on the feat_out row GEMM as an extra workgroup, pk_rowgemm.h)
            const RowW& w = h->r_feat_out;
            pk_rowgemm_args g;
            g.x = xlast; g.ldx = A; g.Wt = h->W(w.w); g.bias = w.b == (size_t)-1 ? nullptr : h->W(w.b); g.y = Y + (long)s * B * OR; g.ldy = OR;
            g.M = B; g.K = w.K; g.N = w.N; g.act = PK_ACT_NONE;
            g.ln_g = h->W(h->dec_after_g); g.ln_b = h->W(h->dec_after_b);
            g.stop_w = h->W(h->prob_w); g.stop_bias = h->prob_b; g.stop_thr = (float)threshold; g.stop_step = s;
            g.stop_minlen = d_minlen; g.stop_maxlen = d_maxlen; g.stop_probs = h->d_probs.as<float>(); g.stop_len = d_len; g.stop_ndone = d_ndone;
            PK_TRY(pk_rowgemm_launch(ctx, "tts_row_feat_out_stop", g));
        } else if (fuse_after) {
            PK_TRY(rowgemm("tts_row_feat_out", h->r_feat_out, xlast, A, Y + (long)s * B * OR, OR, PK_ACT_NONE, nullptr, 0, h->dec_after_g,
                           h->dec_after_b, true));
            PK_LAUNCH(ctx, "tts_stop", k_tts_stop, dim3(pk_div_up(B, 4)), dim3(256), 0, xlast, A, h->W(h->prob_w), h->prob_b, B, s,
                      (float)threshold, d_minlen, d_maxlen, h->d_probs.as<float>(), d_len, d_ndone, h->W(h->dec_after_g),
                      h->W(h->dec_after_b));
        } else {
        if (!post)
            PK_TRY(pk_fft_layernorm_rows(h, xlast, h->dec_after_g, h->dec_after_b, valid, B, A, rz, use_ham ? ham : nullptr));
        else   // no after_norm with post-norm blocks (decoder.py:220-221): the last layer's row as it is
            PK_HIP(hipMemcpyAsync(rz, xlast, (size_t)B * A * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
        if (use_rg)
            PK_TRY(rowgemm("tts_row_feat_out", h->r_feat_out, rz, A, Y + (long)s * B * OR, OR, PK_ACT_NONE, nullptr, 0, 0, 0, false));
        else
            PK_TRY(pk_fft_run_dense(h, "tts_gemm_feat_out", h->feat_out, rz, A, Y + (long)s * B * OR, OR, B, PK_ACT_NONE, nullptr,
                                    0, nullptr, use_ham ? ham : nullptr));
        if (RF == 1)
            PK_LAUNCH(ctx, "tts_stop", k_tts_stop, dim3(pk_div_up(B, 4)), dim3(256), 0, rz, A, h->W(h->prob_w), h->prob_b, B, s,
                      (float)threshold, d_minlen, d_maxlen, h->d_probs.as<float>(), d_len, d_ndone, (const float*)nullptr,
                      (const float*)nullptr);
        else
            PK_LAUNCH(ctx, "tts_stop", k_tts_stop_r, dim3(pk_div_up(B, 4)), dim3(256), 0, rz, A, h->W(h->prob_w),
                      h->W(h->prob_bv), RF, B, s, (float)threshold, d_minlen, d_maxlen, h->d_probs.as<float>(), d_len, d_ndone);
        }
        if (s % poll == 0 || s == Lcap) {
            int ndone = 0;
            PK_HIP(hipMemcpyAsync(&ndone, d_ndone, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            PK_HIP(hipStreamSynchronize(ctx->stream));
            if (ndone >= B) break;
        }
    }
    if (overlap) {
        PK_HIP(hipStreamSynchronize(h->side));   // (the prefix of a step that never ran may still be in flight)
        PK_HIP(hipEventRecord(h->ev_io, h->own_main));
        PK_HIP(hipStreamWaitEvent(sguard.user, h->ev_io, 0));   // what the caller issues next sees the finished decode
    }
    h->steps = std::min(s, Lcap);
    h->len.resize(B);
    PK_HIP(hipMemcpyAsync(h->len.data(), d_len, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    PK_HIP(hipStreamSynchronize(ctx->stream));
    for (int b = 0; b < B; ++b) {
        if (h->len[b] <= 0 || h->len[b] > h->steps)
            PK_FAIL(PK_EHIP, "pk_tts_infer: utterance %d did not stop within %d steps (internal error)", b, h->steps);
    }
    h->frames.resize(B);
    for (int b = 0; b < B; ++b) out_frames[b] = h->frames[b] = h->len[b] * RF;
    h->inferred = true;
    return PK_OK;
}

extern "C" int pk_tts_read(pk_tts* h, float* mel_out, float* probs_out, float* att_out, int32_t flags) {
    if (!h || !mel_out) PK_FAIL(PK_EINVAL, "pk_tts_read: NULL argument");
    if (!h->inferred) PK_FAIL(PK_ESTATE, "pk_tts_read: call pk_tts_infer first");
    if (att_out && !h->keep_att) PK_FAIL(PK_ESTATE, "pk_tts_read: attention weights need PK_TTS_KEEP_ATT at pk_tts_infer");
    pk_ctx* ctx = h->ctx;
    PK_DEVICE(ctx->device);
    const pk_tts_cfg& c = h->cfg;
    const int B = h->B, O = c.odim, H = c.aheads, rf = c.reduction_factor;
    long total = 0;
    for (int b = 0; b < B; ++b) total += h->frames[b];
    PK_TRY(pk_fft_build_timeline(ctx, h->tl_frm, h->frames.data(), B, h->gapr));
    Timeline& tl = h->tl_frm;
    {
        std::vector<int> rowmap(tl.rows_alloc, -1);
        int o = 0;
        for (int b = 0; b < B; ++b)
            for (int l = 0; l < h->frames[b]; ++l) rowmap[tl.seg_start[b] + l] = o++;
        PK_TRY(pk_upload(ctx, h->d_rowmap, rowmap.data(), rowmap.size() * sizeof(int)));
    }
    const bool host = (flags & PK_HOST_IO) != 0;
    float* d_mel = mel_out;
    if (host) {
        PK_TRY(h->d_stage.reserve((size_t)total * O * sizeof(float)));
        d_mel = h->d_stage.as<float>();
    }
    const bool denorm = h->has_out_affine && (flags & PK_APPLY_NORMALIZER);   // TransformerTTSInference (:757-767)
    const float* cs = denorm ? h->W(h->out_scale) : nullptr;
    const float* ch = denorm ? h->W(h->out_shift) : nullptr;
    const float* Y = pk_fft_act_ptr(h->d_y, O * rf);
    // frames of the steps' output rows: position-major step rows [frame 0 | ... | frame rf-1], step s at row s * B + b
    auto gather = [&](const float* src, int C, int off, const int* rowmap, const float* scale, const float* shift, float* dst) -> int {
        if (rf == 1)
            PK_LAUNCH(ctx, "tts_gather", k_ar_gather, dim3(tl.rows), dim3(128), 0, src, C, B, off, tl.d_row_utt(), tl.d_row_pos(),
                      rowmap, scale, shift, dst);
        else
            PK_LAUNCH(ctx, "tts_gather", k_tts_gather_r, dim3(tl.rows), dim3(128), 0, src, C, B, rf, off, tl.d_row_utt(),
                      tl.d_row_pos(), rowmap, scale, shift, dst);
        return PK_OK;
    };
    if (c.postnet_layers == 0) {
        PK_TRY(gather(Y, O, 1, h->d_rowmap.as<int>(), cs, ch, d_mel));
    } else {
        // outs on a frame timeline with zero gap rows, then outs + postnet(outs) (:644-648)
        PK_TRY(pk_fft_act_reserve(h->d_before, tl.rows, O));
        PK_HIP(hipMemsetAsync(h->d_before.p, 0, h->d_before.cap, ctx->stream));
        float* before = pk_fft_act_ptr(h->d_before, O);
        PK_TRY(gather(Y, O, 1, nullptr, nullptr, nullptr, before));
        PK_TRY(pk_fft_run_postnet(h, "tts_conv_postnet", h->postnet, before, O, c.postnet_chans, tl, h->d_q1, h->d_q2, d_mel,
                                  h->d_rowmap.as<int>(), cs, ch));
    }
    if (host) PK_HIP(hipMemcpyAsync(mel_out, d_mel, (size_t)total * O * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    if (probs_out) {
        float* d_p = probs_out;
        if (host) {
            PK_TRY(h->d_stage2.reserve((size_t)total * sizeof(float)));
            d_p = h->d_stage2.as<float>();
        }
        PK_TRY(gather(h->d_probs.as<float>(), 1, 0, h->d_rowmap.as<int>(), nullptr, nullptr, d_p));
        if (host) PK_HIP(hipMemcpyAsync(probs_out, d_p, (size_t)total * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    }
    if (att_out) {
        // per utterance (dlayers, heads, L_b, T_b) out of the (dlayers, heads, cap_b, T_b) store
        long o = 0;
        for (int b = 0; b < B; ++b) {
            const size_t width = (size_t)h->len[b] * h->T[b] * sizeof(float);
            const size_t spitch = (size_t)h->cap[b] * h->T[b] * sizeof(float);
            PK_HIP(hipMemcpy2DAsync(att_out + o, width, h->d_att.as<float>() + h->att_off[b], spitch, width,
                                    (size_t)c.dlayers * H, host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice,
                                    ctx->stream));
            o += (long)c.dlayers * H * h->len[b] * h->T[b];
        }
    }
    if (host) PK_HIP(hipStreamSynchronize(ctx->stream));
    return PK_OK;
}

/* what: 0 = encoder output hs (T_b, adim), 1 = outs before the postnet (L_b, odim), 2 = last decoder layer's output
 * rows (L_b, adim). */
extern "C" int pk_tts_debug_read(pk_tts* h, int32_t what, int32_t b, float* host_out, int64_t n_floats) {
    if (!h || !host_out) PK_FAIL(PK_EINVAL, "pk_tts_debug_read: NULL argument");
    if (!h->inferred) PK_FAIL(PK_ESTATE, "pk_tts_debug_read: nothing has run");
    if (b < 0 || b >= h->B) PK_FAIL(PK_EINVAL, "pk_tts_debug_read: utterance out of range");
    pk_ctx* ctx = h->ctx;
    PK_DEVICE(ctx->device);
    const int A = h->cfg.adim, O = h->cfg.odim, B = h->B;
    PK_HIP(hipStreamSynchronize(ctx->stream));
    if (what == 0) {
        const long n = (long)h->T[b] * A;
        if (n_floats != n) PK_FAIL(PK_ESHAPE, "pk_tts_debug_read: expected %ld floats, got %lld", n, (long long)n_floats);
        PK_HIP(hipMemcpy(host_out, pk_fft_act_ptr(h->d_hs, A) + (long)h->tl_tok.seg_start[b] * A, n * sizeof(float),
                         hipMemcpyDeviceToHost));
        return PK_OK;
    }
    if (what != 1 && what != 2) PK_FAIL(PK_EINVAL, "pk_tts_debug_read: unknown tap %d", what);
    const int C = what == 1 ? O * h->cfg.reduction_factor : A;   // a step's row holds its reduction_factor frames
    const float* src = what == 1 ? pk_fft_act_ptr(h->d_y, C) + (long)B * C
                                 : pk_fft_act_ptr(h->d_xc_l[h->cfg.dlayers - 1], A);
    const long n = (long)h->len[b] * C;
    if (n_floats != n) PK_FAIL(PK_ESHAPE, "pk_tts_debug_read: expected %ld floats, got %lld", n, (long long)n_floats);
    // rows (p * B + b) of a position-major array: one strided copy
    PK_HIP(hipMemcpy2D(host_out, (size_t)C * sizeof(float), src + (long)b * C, (size_t)B * C * sizeof(float),
                       (size_t)C * sizeof(float), (size_t)h->len[b], hipMemcpyDeviceToHost));
    return PK_OK;
}

extern "C" void pk_tts_destroy(pk_tts* h) {
    if (!h) return;
    pk_device_guard _dg(h->ctx->device);
    (void)hipStreamSynchronize(h->ctx->stream);
    h->release_core();
    pk_gst_release(h->gst);
    pk_dbuf* bufs[] = {&h->d_style, &h->d_spk_emb, &h->d_spk_vec, &h->d_tok, &h->d_e1, &h->d_e2, &h->d_tpe, &h->d_hs, &h->d_valid, &h->d_y, &h->d_p0, &h->d_p1,
                       &h->d_x0, &h->d_t, &h->d_ham, &h->d_pam, &h->d_peb, &h->d_rt, &h->d_rc, &h->d_rx, &h->d_rq, &h->d_rf, &h->d_rz,
                       &h->d_ra, &h->d_rn, &h->d_probs, &h->d_state, &h->d_seeds, &h->d_att, &h->d_attoff, &h->d_before, &h->d_q1, &h->d_q2,
                       &h->d_rowmap, &h->d_stage, &h->d_stage2};
    for (auto* b : bufs) b->release();
    for (auto& b : h->d_qkv_l) b.release();
    {
        pk_dbuf* b2[] = {&h->d2_p0, &h->d2_p1, &h->d2_x0, &h->d2_t, &h->d2_ham, &h->d2_pam, &h->d2_qkv0};
        for (pk_dbuf* b : b2) b->release();
        if (h->side) {
            (void)hipStreamSynchronize(h->side);
            (void)hipEventDestroy(h->ev_main);
            (void)hipEventDestroy(h->ev_side);
            (void)hipStreamDestroy(h->side);
            (void)hipStreamSynchronize(h->own_main);
            (void)hipEventDestroy(h->ev_io);
            (void)hipStreamDestroy(h->own_main);
        }
    }
    for (auto& b : h->d_xc_l) b.release();
    for (auto& b : h->d_mkv_l) b.release();
    h->tl_tok.release();
    h->tl_frm.release();
    delete h;
}
