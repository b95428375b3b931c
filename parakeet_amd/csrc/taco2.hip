// taco2.hip -- Tacotron2 inference on gfx950: kernels + pk_taco_* entry points.
//
// Reference: parakeet/models/tacotron2.py
//   Tacotron2.infer :781-840, Tacotron2Encoder.forward :216-241, Tacotron2Decoder.infer :474-541,
//   Tacotron2Decoder._decode :378-417, _initialize_decoder_states :352-376, DecoderPreNet.forward :61-79,
//   DecoderPostNet.forward :147-171; parakeet/modules/attention.py LocationSensitiveAttention.forward :300-348;
//   parakeet/modules/conv.py Conv1dBatchNorm :186-260.
//
// Encoder: the text batch is one channels-last row timeline (as FastSpeech2, pk_fft.h): embedding (+ tones), every
// Conv1dBatchNorm -> ReLU as ONE implicit-conv GEMM (conv bias and the batch norm folded into weights / bias), the
// bidirectional LSTM as one GEMM for the input projections of every step and both directions, followed by
// k_taco_lstm_seq: one workgroup per (utterance, direction) that keeps h, c and the gate vector in LDS and walks the
// sequence, reading the k-major recurrent matrix from L2.
//
// Decoder: B utterances in lockstep, one frame per step.  The LSTMCells' operands are laid out so that each cell is ONE
// GEMM per step: the cell input is a row [x | context | h] that the producing kernels write in place (prenet GEMM ->
// first slice, attention kernel -> context slice, the cell's own pointwise kernel -> h slice), multiplied by
// [W_ih^T ; W_hh^T] with bias b_ih + b_hh.  The per-step products have B rows (one per utterance): they run on the
// row GEMM (pk_rowgemm.h: weights streamed once, fp32 FMA, ReLU + prenet dropout in the epilogue) unless
// PK_AR_ROWGEMM=0 selects the tile GEMM.  LocationSensitiveAttention is three launches (see LsaArgs).
// Stop rules run on the device (k_taco_stop); a finished utterance keeps being stepped and is ignored.
//
// LSTM semantics [paddle-semantics, from Paddle's API documentation]: gate order i, f, g, o along the 4H axis.
// Dropout: include/pk_synth.h "dropout stream"; only the decoder prenet's stays on at inference (:76-79).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "pk_ar.h"
#include "pk_fft.h"
#include "pk_rowgemm.h"

namespace {
typedef pk_fft_dense Dense;
typedef pk_fft_timeline Timeline;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// x[r] = E[id[r]] + (tone[r] != 0 ? Et[tone[r]] : 0)   (:807-810; embedding_tones has padding_idx 0), gap rows zero
__global__ __launch_bounds__(128) void k_taco_embed(const int* __restrict__ tok, const int* __restrict__ tone,
                                                    const int* __restrict__ row_utt, const float* __restrict__ etab,
                                                    const float* __restrict__ ttab, int E, float* __restrict__ x) {
    const long r = blockIdx.x;
    const bool valid = row_utt[r] >= 0;
    const int id = valid ? tok[r] : 0;
    const int tn = (valid && tone) ? tone[r] : 0;
    for (int c = threadIdx.x; c < E; c += blockDim.x) {
        float v = 0.f;
        if (valid) v = etab[(long)id * E + c];
        if (tn != 0) v += ttab[(long)tn * E + c];
        x[r * E + c] = v;
    }
}

// memory[r] = [encoder_outputs[r] | global_condition[utterance of r]] (:816-821), gap rows zero
__global__ __launch_bounds__(128) void k_taco_concat_global(const float* __restrict__ enc, int E, const float* __restrict__ g,
                                                            int G, const int* __restrict__ row_utt, float* __restrict__ mem) {
    const long r = blockIdx.x;
    const int b = row_utt[r];
    for (int c = threadIdx.x; c < E + G; c += blockDim.x) {
        float v = 0.f;
        if (b >= 0) v = c < E ? enc[r * E + c] : g[(long)b * G + (c - E)];
        mem[r * (E + G) + c] = v;
    }
}

// One direction of one utterance of nn.LSTM: grid (B, 2), dir 1 walks the sequence backwards.
//   xg   [rows][8H]: x_t W_ih^T + b_ih + b_hh for (forward | backward), gate order i, f, g, o
//   whhT [2][H][4H]: recurrent weights, k-major (coalesced over the gate index)
//   out  [rows][2H]: (forward | backward) hidden states  (:239)
// LDS: h[H] | c[H] | gates[4H].
__global__ __launch_bounds__(1024) void k_taco_lstm_seq(const float* __restrict__ xg, const float* __restrict__ whhT,
                                                        const int* __restrict__ seg_start,
                                                        const int* __restrict__ seg_len, int H,
                                                        float* __restrict__ out) {
    extern __shared__ float sm[];
    float* hs = sm;
    float* cs = sm + H;
    float* gs = sm + 2 * H;
    const int b = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x, nt = blockDim.x;
    const int T = seg_len[b], s0 = seg_start[b], G = 4 * H;
    const float* W = whhT + (long)dir * H * G;
    for (int u = tid; u < H; u += nt) {
        hs[u] = 0.f;
        cs[u] = 0.f;
    }
    __syncthreads();
    for (int step = 0; step < T; ++step) {
        const long row = s0 + (dir ? T - 1 - step : step);
        for (int g = tid; g < G; g += nt) {
            float a0 = xg[row * (2 * G) + dir * G + g], a1 = 0.f;
            int k = 0;
            for (; k + 1 < H; k += 2) {
                a0 = fmaf(W[(long)k * G + g], hs[k], a0);
                a1 = fmaf(W[(long)(k + 1) * G + g], hs[k + 1], a1);
            }
            if (k < H) a0 = fmaf(W[(long)k * G + g], hs[k], a0);
            gs[g] = a0 + a1;
        }
        __syncthreads();
        for (int u = tid; u < H; u += nt) {
            const float i = sigmoidf_(gs[u]), f = sigmoidf_(gs[H + u]), g = tanhf(gs[2 * H + u]), o = sigmoidf_(gs[3 * H + u]);
            const float c = f * cs[u] + i * g;
            const float h = o * tanhf(c);
            cs[u] = c;
            hs[u] = h;
            out[row * (2 * H) + dir * H + u] = h;
        }
        __syncthreads();
    }
}

// LSTMCell pointwise part for the decoder cells: gates [B][4H] (i, f, g, o) and c [B][H] -> c, h; h goes to two
// destinations (slices of the operand rows of the GEMMs that consume it).
__global__ __launch_bounds__(256) void k_taco_lstm_point(const float* __restrict__ gates, float* __restrict__ c, int H,
                                                         int B, float* __restrict__ h1, int ld1,
                                                         float* __restrict__ h2, int ld2) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= (long)B * H) return;
    const int b = (int)(q / H), u = (int)(q - (long)b * H);
    const float* g = gates + (long)b * 4 * H;
    const float i = sigmoidf_(g[u]), f = sigmoidf_(g[H + u]), gg = tanhf(g[2 * H + u]), o = sigmoidf_(g[3 * H + u]);
    const float cn = f * c[q] + i * gg;
    const float h = o * tanhf(cn);
    c[q] = cn;
    h1[(long)b * ld1 + u] = h;
    h2[(long)b * ld2 + u] = h;
}

// One step of LocationSensitiveAttention.forward (attention.py:300-348) + the state updates of _decode (:387-397), as
// three launches so that the whole chip works on it (one workgroup per utterance used 32 of 256 CUs and took 480 us):
//   processed_query = query_layer(attention_hidden)          a row GEMM (pk_rowgemm.h)
//   k_taco_lsa_energy   one WAVE per memory row of the token timeline:
//        alignment[t] = value(tanh(location_layer(location_conv(cat))[t] + processed_key[t] + processed_query))
//   k_taco_lsa_ctx      per utterance (x column blocks): softmax over its T memory rows, the new / cumulative weights,
//        the alignment row, attention_context = weights^T . memory written to the three operand rows that consume it
struct LsaArgs {
    int Da, E, F, K;                  // K = location kernel size (odd), F <= 64, Da <= 256
    const float* pq;                  // processed query [B][Da]
    const float* Wconv;               // location_conv [F][2][K]
    const float* Wloc;                // location_layer [F][Da]
    const float* v;                   // value [Da]
    const float* pkey;                // processed memory [rows][Da]
    const float* mem;                 // memory [rows][E]
    const int* row_utt;               // token timeline
    const int* row_pos;
    const int* seg_start;
    const int* seg_len;
    int rows;
    float* energy;                    // [rows]
    float* attw;                      // [rows]: attention_weights (previous -> new)
    float* cum;                       // [rows]: attention_weights_cum
    float* ctx1; int ld1;             // three copies of the context vector
    float* ctx2; int ld2;
    float* ctx3; int ld3;
    float* align;                     // alignment store
    const long* align_off;            // per utterance offset; row `step` of (cap, T_b)
    int step;
};

__global__ __launch_bounds__(256) void k_taco_lsa_energy(LsaArgs a) {
    __shared__ float win[4][2][64];   // per wave: the attw / cum window t - pad .. t + pad
    __shared__ float loc[4][64];      // per wave: location_conv output of its row
    // (Round 4: location_layer and location_conv weights staged in LDS per workgroup instead of read through L1 inside the
    // loops: 15.2 -> 15.6 us with the first, 18.5 with both -- a workgroup is four memory rows, the staging costs what it saves
    // and the 52 KB halve the resident workgroups.  Left as it was.)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = blockIdx.x * 4 + wave;
    const int b = r < a.rows ? a.row_utt[r] : -1;
    const bool valid = b >= 0;   // wave-uniform
    const int K = a.K, pad = (K - 1) / 2, F = a.F, Da = a.Da;
    int t = 0, T = 0;
    long s0 = 0;
    if (valid) {
        t = a.row_pos[r];
        T = a.seg_len[b];
        s0 = a.seg_start[b];
        if (lane < K) {
            const int tt = t + lane - pad;
            const bool in = tt >= 0 && tt < T;   // zero padding of the conv at the utterance's ends
            win[wave][0][lane] = in ? a.attw[s0 + tt] : 0.f;
            win[wave][1][lane] = in ? a.cum[s0 + tt] : 0.f;
        }
    }
    __syncthreads();
    if (valid && lane < F) {
        const float* wc = a.Wconv + (long)lane * 2 * K;
        float acc = 0.f;
#pragma unroll 8
        for (int k = 0; k < K; ++k) {   // (unrolled: eight pairs of weight loads in flight instead of one)
            acc = fmaf(wc[k], win[wave][0][k], acc);
            acc = fmaf(wc[K + k], win[wave][1][k], acc);
        }
        loc[wave][lane] = acc;
    }
    __syncthreads();
    float e = 0.f;
    if (valid) {
        const float* pq = a.pq + (long)b * Da;
        for (int d = lane; d < Da; d += 64) {
            const float vd = a.v[d], kd = a.pkey[(long)r * Da + d], qd = pq[d];   // (requested before the f loop)
            float pl = 0.f;
#pragma unroll 8
            for (int f = 0; f < F; ++f) pl = fmaf(loc[wave][f], a.Wloc[(long)f * Da + d], pl);
            e = fmaf(vd, tanhf(pl + kd + qd), e);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o);
    if (valid && lane == 0) a.energy[r] = e;
}

// grid (B, ceil(E / 256)), 256 threads: every block redoes the (cheap) softmax of its utterance, block column 0 also
// stores the weights; thread c of block column y owns context column y * 256 + c.
__global__ __launch_bounds__(256) void k_taco_lsa_ctx(LsaArgs a) {
    extern __shared__ float sm[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = a.seg_len[b];
    const long s0 = a.seg_start[b];
    float* red = sm;          // 8
    float* sc = sm + 8;       // T
    float m = -INFINITY;
    for (int t = tid; t < T; t += 256) {
        const float e = a.energy[s0 + t];
        sc[t] = e;
        m = fmaxf(m, e);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int t = tid; t < T; t += 256) {
        const float p = expf(sc[t] - m);
        sc[t] = p;
        sum += p;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();   // also publishes sc[]
    sum = (red[4] + red[5]) + (red[6] + red[7]);
    const float inv = 1.f / sum;
    if (blockIdx.y == 0) {
        // new weights, cumulative weights (:397), alignment row
        float* al = a.align ? a.align + a.align_off[b] + (long)a.step * T : nullptr;
        for (int t = tid; t < T; t += 256) {
            const float w = sc[t] * inv;
            a.attw[s0 + t] = w;
            a.cum[s0 + t] += w;
            if (al) al[t] = w;
        }
    }
    // attention_context = weights^T . memory  (:342-343)
    const int c = blockIdx.y * 256 + tid;
    if (c < a.E) {
        const float* mp = a.mem + s0 * a.E + c;
        // (sixteen memory rows in flight: with four the 129-token walk was 32 dependent round trips)
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int t0 = 0; t0 < T; t0 += 16) {
            float mv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) mv[i] = mp[(long)min(t0 + i, T - 1) * a.E];
#pragma unroll
            for (int i = 0; i < 16; i += 4) {
                a0 = fmaf(t0 + i < T ? sc[t0 + i] : 0.f, mv[i], a0);
                a1 = fmaf(t0 + i + 1 < T ? sc[t0 + i + 1] : 0.f, mv[i + 1], a1);
                a2 = fmaf(t0 + i + 2 < T ? sc[t0 + i + 2] : 0.f, mv[i + 2], a2);
                a3 = fmaf(t0 + i + 3 < T ? sc[t0 + i + 3] : 0.f, mv[i + 3], a3);
            }
        }
        const float acc = ((a0 + a1) + (a2 + a3)) * inv;
        a.ctx1[(long)b * a.ld1 + c] = acc;
        a.ctx2[(long)b * a.ld2 + c] = acc;
        a.ctx3[(long)b * a.ld3 + c] = acc;
    }
}

// stop_layer + the rules that end a run (:515-528), one wave per utterance.  len[b] == 0 while utterance b runs.
//   with a stop token: sigmoid(stop_logit) > 0.5;
//   without: the argmax of this step's alignment sits on the last memory position for the first time -> remember the
//   step; again at a step more than 20 later -> end;
//   always: step + 1 == max_steps.
__global__ __launch_bounds__(256) void k_taco_stop(const float* __restrict__ hc, int ld, int n, const float* __restrict__ w,
                                                   float bias, int use_stop, int B, int step, int max_steps,
                                                   const float* __restrict__ attw, const int* __restrict__ seg_start,
                                                   const int* __restrict__ seg_len, float* __restrict__ logits,
                                                   int* __restrict__ len, int* __restrict__ first_hit,
                                                   int* __restrict__ ndone) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    bool end = false;
    if (use_stop) {
        float s = 0.f;
        for (int c = lane; c < n; c += 64) s = fmaf(hc[(long)b * ld + c], w[c], s);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        s += bias;
        if (lane == 0) logits[(long)step * B + b] = s;
        end = sigmoidf_(s) > 0.5f;
    } else {
        // argmax with the first index winning ties (paddle.argmax)
        const int T = seg_len[b];
        const float* p = attw + seg_start[b];
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int t = lane; t < T; t += 64) {
            const float v = p[t];
            if (v > best) { best = v; bi = t; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o);
            const int oi = __shfl_xor(bi, o);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (bi == T - 1) {
            const int fh = first_hit[b];
            if (fh < 0) {
                if (lane == 0) first_hit[b] = step;
            } else if (step > fh + 20) {
                end = true;
            }
        }
    }
    if (step + 1 >= max_steps) end = true;
    if (lane == 0 && end && len[b] == 0) {
        len[b] = step + 1;
        atomicAdd(ndone, 1);
    }
}
}  // namespace

struct pk_taco : pk_fft_core {
    pk_taco_cfg cfg;
    pk_param_map params;
    bool finalized = false, inferred = false;
    int gapr = 1;
    bool dropout = true;
    // weights
    size_t emb = 0, temb = 0, whhT = 0, Wconv = 0, Wloc = 0, vvec = 0, stop_w = 0;
    float stop_b = 0.f;
    std::vector<Dense> econv, postnet;
    Dense lstm_in, key_layer, pre1, pre2, att_rnn, dec_rnn, proj;
    struct RowW {
        size_t w = 0, b = (size_t)-1;
        int K = 0, N = 0;
    } rw_pre1, rw_pre2, rw_att, rw_dec, rw_proj, rw_q;   // the same layers as pk_rowgemm_pack tiles for the row GEMM
    size_t pre1_kn = (size_t)-1, pre2_kn = (size_t)-1;   // the prenet's matrices row-major [K][N] (k_ar_prenet_embed)
    // per call
    Timeline tl_tok, tl_frm;
    int B = 0, cap = 0, steps = 0, maxT = 0;
    std::vector<int> T, len;
    std::vector<long> align_off;
    std::vector<float> cond_g;   // global condition of the next infer (pk_taco_set_global_condition)
    int cond_B = 0;
    pk_dbuf d_gc, d_enc;
    pk_dbuf d_tok, d_tone, d_e1, d_e2, d_xg, d_mem, d_pkey, d_attw, d_cum, d_in1, d_in1b, d_in2, d_in2b, d_in3, d_p1, d_gates, d_catt,
        d_cdec, d_zero, d_y, d_pq, d_energy, d_logits, d_state, d_seeds, d_align, d_alignoff, d_before, d_q1, d_q2, d_rowmap, d_stage,
        d_stage2;
};

namespace {
int find_param(const pk_param_map& P, const std::vector<std::string>& names, int64_t rows, int64_t cols,
               std::vector<float>& out) {
    for (const std::string& n : names) {
        auto it = P.find(n);
        if (it == P.end()) continue;
        if (it->second.numel() != rows * cols)
            PK_FAIL(PK_ESHAPE, "parameter %s has %lld elements, expected %lld x %lld", n.c_str(),
                    (long long)it->second.numel(), (long long)rows, (long long)cols);
        out = it->second.data;
        return PK_OK;
    }
    PK_FAIL(PK_ESTATE, "parameter %s was never set", names.empty() ? "?" : names.back().c_str());
}

// One LSTMCell as one dense layer on the operand row [x (in) | h (H)]: kn = [W_ih^T ; W_hh^T], bias = b_ih + b_hh
int add_cell(pk_fft_arena& ar, const pk_param_map& P, const std::string& p, int in, int H, Dense& d,
             pk_taco::RowW& rw) {
    std::vector<float> wih, whh, bih, bhh;
    PK_TRY(find_param(P, {p + ".weight_ih"}, 4 * H, in, wih));
    PK_TRY(find_param(P, {p + ".weight_hh"}, 4 * H, H, whh));
    PK_TRY(find_param(P, {p + ".bias_ih"}, 4 * H, 1, bih));
    PK_TRY(find_param(P, {p + ".bias_hh"}, 4 * H, 1, bhh));
    const int K = in + H, N = 4 * H;
    std::vector<float> kn((size_t)K * N), bias(N);
    for (int g = 0; g < N; ++g) {
        for (int k = 0; k < in; ++k) kn[(size_t)k * N + g] = wih[(size_t)g * in + k];
        for (int k = 0; k < H; ++k) kn[(size_t)(in + k) * N + g] = whh[(size_t)g * H + k];
        bias[g] = bih[g] + bhh[g];
    }
    {
        // the row GEMM finishes the cell in its epilogue: gate columns regrouped so that a 16-column workgroup holds
        // i | f | g | o of four units (pk_rowgemm_lstm_perm)
        std::vector<int> perm;
        pk_rowgemm_lstm_perm(H, perm);
        std::vector<float> knp((size_t)K * N), bp(N), wt;
        for (int c = 0; c < N; ++c) {
            bp[c] = bias[perm[c]];
            for (int k = 0; k < K; ++k) knp[(size_t)k * N + c] = kn[(size_t)k * N + perm[c]];
        }
        pk_rowgemm_pack(knp.data(), K, N, wt);
        rw.w = ar.put(wt);
        rw.b = ar.put(bp);
        rw.K = K;
        rw.N = N;
    }
    return pk_fft_add_dense_kn(ar, kn, &bias, K, 1, N, d);
}
}  // namespace

extern "C" int pk_taco_create(pk_ctx* ctx, const pk_taco_cfg* cfg, pk_taco** out) {
    if (!ctx || !cfg || !out) PK_FAIL(PK_EINVAL, "pk_taco_create: NULL argument");
    *out = nullptr;
    const pk_taco_cfg& c = *cfg;
    if (c.vocab_size <= 0 || c.n_tones < 0 || c.d_mels <= 0) PK_FAIL(PK_EINVAL, "Tacotron2: vocab_size / d_mels must be positive");
    // Tacotron2.infer hands the (B, T, d_mels * r) decoder output straight to the postnet (:822-826), whose first
    // convolution has d_mels input channels: with r > 1 the reference's own inference path cannot run
    if (c.reduction_factor != 1) PK_FAIL(PK_EUNSUPPORTED, "Tacotron2: reduction_factor != 1 not implemented");
    if (c.d_global_condition < 0 || c.d_global_condition % PK_GEMM_BK != 0)
        PK_FAIL(PK_EUNSUPPORTED, "Tacotron2: d_global_condition must be a multiple of 16 (or 0 = None)");
    if (c.d_encoder <= 0 || c.d_encoder % 32 != 0) PK_FAIL(PK_EUNSUPPORTED, "Tacotron2: d_encoder must be a positive multiple of 32");
    const int sizes[] = {c.d_mels, c.d_prenet, c.d_attention_rnn, c.d_decoder_rnn, c.d_postnet, c.d_attention};
    for (int s : sizes)
        if (s <= 0 || s % PK_GEMM_BK != 0) PK_FAIL(PK_EUNSUPPORTED, "Tacotron2: size %d not a positive multiple of 16", s);
    if (c.d_attention > 256) PK_FAIL(PK_EUNSUPPORTED, "Tacotron2: d_attention > 256");
    if (c.attention_filters <= 0 || c.attention_filters > 64) PK_FAIL(PK_EUNSUPPORTED, "Tacotron2: attention_filters must be in [1, 64]");
    if (c.attention_kernel_size < 1 || c.attention_kernel_size % 2 == 0 || c.attention_kernel_size > 63)
        PK_FAIL(PK_EUNSUPPORTED, "Tacotron2: attention_kernel_size must be odd and <= 63");
    if (c.encoder_conv_layers < 0 || c.postnet_conv_layers < 1) PK_FAIL(PK_EINVAL, "Tacotron2: layer counts");
    const int ks[] = {c.encoder_conv_layers > 0 ? c.encoder_kernel_size : 1, c.postnet_kernel_size};
    int gapr = 1;
    for (int k : ks) {
        if (k < 1 || k % 2 == 0 || k > PK_GEMM_MAX_TAPS) PK_FAIL(PK_EUNSUPPORTED, "Tacotron2: conv kernel size %d unsupported", k);
        gapr = std::max(gapr, (k - 1) / 2);
    }
    if (gapr > PK_FFT_LEAD) PK_FAIL(PK_EUNSUPPORTED, "Tacotron2: conv kernel too wide");
    if (!(c.p_prenet_dropout >= 0.f) || !(c.p_prenet_dropout < 1.f)) PK_FAIL(PK_EINVAL, "Tacotron2: p_prenet_dropout must be in [0, 1)");
    if ((size_t)(6 * (c.d_encoder / 2)) * sizeof(float) > 60 * 1024) PK_FAIL(PK_EUNSUPPORTED, "Tacotron2: d_encoder too large for the LSTM kernel");
    pk_taco* h = new pk_taco();
    h->ctx = ctx;
    h->cfg = c;
    h->adim = c.d_encoder;
    h->aheads = 1;
    h->gapr = gapr;
    if (const char* e = pk_prof_env("PK_TACO_MATH")) h->math = strcmp(e, "f32") == 0 ? PK_GEMM_MATH_F32 : PK_GEMM_MATH_F16X3;
    *out = h;
    return PK_OK;
}

extern "C" int pk_taco_set_param(pk_taco* h, const char* name, const float* data, const int64_t* shape, int32_t ndim) {
    if (!h) PK_FAIL(PK_EINVAL, "pk_taco_set_param: handle is NULL");
    h->finalized = false;
    return pk_store_param(h->params, name, data, shape, ndim);
}

extern "C" int pk_taco_set_math(pk_taco* h, int32_t mode) {
    if (!h) PK_FAIL(PK_EINVAL, "pk_taco_set_math: handle is NULL");
    if (mode != PK_GEMM_MATH_F32 && mode != PK_GEMM_MATH_F16X3) PK_FAIL(PK_EINVAL, "pk_taco_set_math: unknown mode %d", mode);
    h->math = mode;
    return PK_OK;
}

extern "C" int pk_taco_set_dropout(pk_taco* h, int32_t on) {
    if (!h) PK_FAIL(PK_EINVAL, "pk_taco_set_dropout: handle is NULL");
    h->dropout = on != 0;
    return PK_OK;
}

extern "C" int pk_taco_set_global_condition(pk_taco* h, const float* g, int32_t B) {
    if (!h) PK_FAIL(PK_EINVAL, "pk_taco_set_global_condition: handle is NULL");
    h->cond_g.clear();
    h->cond_B = 0;
    if (!g) return PK_OK;
    if (h->cfg.d_global_condition <= 0) PK_FAIL(PK_ESTATE, "pk_taco_set_global_condition: the model has no global condition");
    if (B <= 0) PK_FAIL(PK_EINVAL, "pk_taco_set_global_condition: batch size must be positive");
    h->cond_g.assign(g, g + (size_t)B * h->cfg.d_global_condition);
    h->cond_B = B;
    return PK_OK;
}

extern "C" int pk_taco_finalize(pk_taco* h) {
    if (!h) PK_FAIL(PK_EINVAL, "pk_taco_finalize: handle is NULL");
    pk_ctx* ctx = h->ctx;
    PK_DEVICE(ctx->device);
    const pk_taco_cfg& c = h->cfg;
    const pk_param_map& P = h->params;
    const int E = c.d_encoder, Hh = E / 2, M = c.d_mels, Pn = c.d_prenet, Ha = c.d_attention_rnn, Hd = c.d_decoder_rnn,
              Da = c.d_attention, F = c.attention_filters, K = c.attention_kernel_size;
    const int Eg = E + c.d_global_condition;   // width of the decoder's memory rows (:668-669)
    h->arena_h.clear();
    h->arena16_h.clear();
    pk_fft_arena ar{h->arena_h, &h->arena16_h};
    std::vector<float> t;
    PK_TRY(pk_get_weight(P, "embedding", {c.vocab_size, E}, t));
    h->emb = ar.put(t);
    if (c.n_tones > 0) {
        PK_TRY(pk_get_weight(P, "embedding_tones", {c.n_tones, E}, t));
        for (int i = 0; i < E; ++i) t[i] = 0.f;   // padding_idx=0
        h->temb = ar.put(t);
    }
    h->econv.resize(c.encoder_conv_layers);
    for (int i = 0; i < c.encoder_conv_layers; ++i) {
        const std::string p = "encoder.conv_batchnorms." + std::to_string(i);
        PK_TRY(pk_fft_add_conv_bn(ar, P, p + ".conv", p + ".bn", E, E, c.encoder_kernel_size, h->econv[i], true));
    }
    {
        // nn.LSTM(d_hidden, d_hidden / 2, direction="bidirectional") (:209-211): input projections of both directions
        // as one [E][8 Hh] dense layer, recurrent matrices k-major
        const int G = 4 * Hh;
        std::vector<float> kn((size_t)E * 2 * G), bias(2 * G), whhT((size_t)2 * Hh * G);
        const char* sfx[2] = {"", "_reverse"};
        const char* cell[2] = {"cell_fw", "cell_bw"};
        for (int d = 0; d < 2; ++d) {
            const std::string a = std::string("encoder.lstm."), cl = a + "0." + cell[d] + ".";
            std::vector<float> wih, whh, bih, bhh;
            PK_TRY(find_param(P, {a + "weight_ih_l0" + sfx[d], cl + "weight_ih"}, G, E, wih));
            PK_TRY(find_param(P, {a + "weight_hh_l0" + sfx[d], cl + "weight_hh"}, G, Hh, whh));
            PK_TRY(find_param(P, {a + "bias_ih_l0" + sfx[d], cl + "bias_ih"}, G, 1, bih));
            PK_TRY(find_param(P, {a + "bias_hh_l0" + sfx[d], cl + "bias_hh"}, G, 1, bhh));
            for (int g = 0; g < G; ++g) {
                for (int k = 0; k < E; ++k) kn[(size_t)k * 2 * G + d * G + g] = wih[(size_t)g * E + k];
                for (int k = 0; k < Hh; ++k) whhT[((size_t)d * Hh + k) * G + g] = whh[(size_t)g * Hh + k];
                bias[d * G + g] = bih[g] + bhh[g];
            }
        }
        PK_TRY(pk_fft_add_dense_kn(ar, kn, &bias, E, 1, 2 * G, h->lstm_in));
        h->whhT = ar.put(whhT);
    }
    {
        std::vector<float> w;
        PK_TRY(pk_get_weight(P, "decoder.attention_layer.key_layer", {Eg, Da}, w));
        PK_TRY(pk_fft_add_dense_kn(ar, w, nullptr, Eg, 1, Da, h->key_layer));
        PK_TRY(pk_get_weight(P, "decoder.prenet.linear1", {M, Pn}, w));
        PK_TRY(pk_fft_add_dense_kn(ar, w, nullptr, M, 1, Pn, h->pre1));
        std::vector<float> wt;
        pk_rowgemm_pack(w.data(), M, Pn, wt);
        h->rw_pre1.w = ar.put(wt); h->rw_pre1.K = M; h->rw_pre1.N = Pn;
        h->pre1_kn = ar.put(w);
        PK_TRY(pk_get_weight(P, "decoder.prenet.linear2", {Pn, Pn}, w));
        PK_TRY(pk_fft_add_dense_kn(ar, w, nullptr, Pn, 1, Pn, h->pre2));
        pk_rowgemm_pack(w.data(), Pn, Pn, wt);
        h->rw_pre2.w = ar.put(wt); h->rw_pre2.K = Pn; h->rw_pre2.N = Pn;
        h->pre2_kn = ar.put(w);
        PK_TRY(pk_get_weight(P, "decoder.attention_layer.query_layer", {Ha, Da}, w));
        pk_rowgemm_pack(w.data(), Ha, Da, wt);
        h->rw_q.w = ar.put(wt); h->rw_q.K = Ha; h->rw_q.N = Da;
        PK_TRY(pk_get_weight(P, "decoder.attention_layer.value", {Da, 1}, w));
        h->vvec = ar.put(w);
        PK_TRY(pk_get_weight(P, "decoder.attention_layer.location_conv", {F, 2, K}, w));
        h->Wconv = ar.put(w);
        PK_TRY(pk_get_weight(P, "decoder.attention_layer.location_layer", {F, Da}, w));
        h->Wloc = ar.put(w);
    }
    PK_TRY(add_cell(ar, P, "decoder.attention_rnn", Pn + Eg, Ha, h->att_rnn, h->rw_att));   // input [prenet | context] (:380)
    PK_TRY(add_cell(ar, P, "decoder.decoder_rnn", Ha + Eg, Hd, h->dec_rnn, h->rw_dec));     // input [attention_hidden | context] (:400-401)
    PK_TRY(pk_fft_add_linear(ar, P, "decoder.linear_projection", Hd + Eg, M, h->proj));   // [decoder_hidden | context] (:409-412)
    {
        std::vector<float> w, b;
        PK_TRY(pk_get_weight(P, "decoder.linear_projection", {Hd + Eg, M}, w));
        PK_TRY(pk_get_vector(P, "decoder.linear_projection.bias", M, b));
        std::vector<float> wt;
        pk_rowgemm_pack(w.data(), Hd + Eg, M, wt);
        h->rw_proj.w = ar.put(wt); h->rw_proj.b = ar.put(b); h->rw_proj.K = Hd + Eg; h->rw_proj.N = M;
    }
    if (c.use_stop_token) {
        std::vector<float> w, b;
        PK_TRY(pk_get_weight(P, "decoder.stop_layer", {Hd + Eg, 1}, w));
        PK_TRY(pk_get_vector(P, "decoder.stop_layer.bias", 1, b));
        h->stop_w = ar.put(w);
        h->stop_b = b[0];
    }
    h->postnet.resize(c.postnet_conv_layers);
    for (int j = 0; j < c.postnet_conv_layers; ++j) {
        const int n = c.postnet_conv_layers;
        const std::string p = "postnet.conv_batchnorms." + std::to_string(j);
        PK_TRY(pk_fft_add_conv_bn(ar, P, p + ".conv", p + ".bn", j == n - 1 ? M : c.d_postnet, j == 0 ? M : c.d_postnet,
                                  c.postnet_kernel_size, h->postnet[j], true));
    }
    PK_TRY(pk_upload(ctx, h->arena, h->arena_h.data(), h->arena_h.size() * sizeof(float)));
    h->arena_h.clear();
    h->arena_h.shrink_to_fit();
    if (!h->arena16_h.empty())
        PK_TRY(pk_upload(ctx, h->arena16, h->arena16_h.data(), h->arena16_h.size() * sizeof(uint16_t)));
    h->arena16_h.clear();
    h->arena16_h.shrink_to_fit();
    h->finalized = true;
    h->inferred = false;
    return PK_OK;
}

namespace {
constexpr int SLACK = 2 * PK_GEMM_BM;   // rows a GEMM tile may read beyond the rows it was asked for
int rows_reserve(pk_dbuf& buf, long rows, int C) { return pk_fft_act_reserve(buf, (int)(rows + SLACK), C); }
}  // namespace

extern "C" int pk_taco_infer(pk_taco* h, const int64_t* ids, const int64_t* tones, const int32_t* tok_lens, int32_t B,
                             int32_t max_decoder_steps, const uint64_t* seeds, int32_t flags, int32_t* out_frames) {
    (void)flags;
    if (!h) PK_FAIL(PK_EINVAL, "pk_taco_infer: NULL argument");
    // the per-call conditioning is consumed by this call, whatever happens next
    std::vector<float> cond_g;
    cond_g.swap(h->cond_g);
    const int condB = h->cond_B;
    h->cond_B = 0;
    if (!ids || !tok_lens || !out_frames) PK_FAIL(PK_EINVAL, "pk_taco_infer: NULL argument");
    if (!h->finalized) PK_FAIL(PK_ESTATE, "pk_taco_infer: call pk_taco_finalize first");
    if (B <= 0) PK_FAIL(PK_EINVAL, "pk_taco_infer: batch size must be positive");
    if (max_decoder_steps <= 0) PK_FAIL(PK_EINVAL, "pk_taco_infer: max_decoder_steps must be positive");
    pk_ctx* ctx = h->ctx;
    PK_DEVICE(ctx->device);
    const pk_taco_cfg& c = h->cfg;
    if (c.n_tones > 0 && !tones) PK_FAIL(PK_EINVAL, "pk_taco_infer: the model has a tone embedding, tones are required (:809-810)");
    if (c.n_tones <= 0 && tones) PK_FAIL(PK_ESTATE, "pk_taco_infer: the model has no tone embedding");
    const int E = c.d_encoder, Hh = E / 2, M = c.d_mels, Pn = c.d_prenet, Ha = c.d_attention_rnn, Hd = c.d_decoder_rnn,
              Da = c.d_attention, G = c.d_global_condition, Eg = E + G;
    if (G > 0 && condB != B)
        PK_FAIL(PK_EINVAL, "pk_taco_infer: the model concatenates a global condition to the encoder outputs (:816-821): "
                           "pk_taco_set_global_condition needs %d rows, got %d", B, condB);
    h->inferred = false;
    h->B = B;
    h->T.assign(tok_lens, tok_lens + B);
    int maxT = 0;
    for (int b = 0; b < B; ++b) {
        if (tok_lens[b] <= 0) PK_FAIL(PK_EINVAL, "pk_taco_infer: utterance %d has %d tokens", b, tok_lens[b]);
        maxT = std::max(maxT, tok_lens[b]);
    }
    h->maxT = maxT;
    const int cap = max_decoder_steps;
    h->cap = cap;
    // ---- encoder (:807-811)
    PK_TRY(pk_fft_build_timeline(ctx, h->tl_tok, tok_lens, B, h->gapr));
    Timeline& tl = h->tl_tok;
    {
        std::vector<int> tx(tl.rows_alloc, 0), tn(tl.rows_alloc, 0);
        long o = 0;
        for (int b = 0; b < B; ++b)
            for (int t = 0; t < tok_lens[b]; ++t, ++o) {
                if (ids[o] < 0 || ids[o] >= c.vocab_size)
                    PK_FAIL(PK_EINVAL, "pk_taco_infer: token id %lld out of [0,%d)", (long long)ids[o], c.vocab_size);
                tx[tl.seg_start[b] + t] = (int)ids[o];
                if (tones) {
                    if (tones[o] < 0 || tones[o] >= c.n_tones)
                        PK_FAIL(PK_EINVAL, "pk_taco_infer: tone id %lld out of [0,%d)", (long long)tones[o], c.n_tones);
                    tn[tl.seg_start[b] + t] = (int)tones[o];
                }
            }
        PK_TRY(pk_upload(ctx, h->d_tok, tx.data(), tx.size() * sizeof(int)));
        if (tones) PK_TRY(pk_upload(ctx, h->d_tone, tn.data(), tn.size() * sizeof(int)));
    }
    PK_TRY(pk_fft_act_reserve(h->d_e1, tl.rows, E));
    PK_TRY(pk_fft_act_reserve(h->d_e2, tl.rows, E));
    PK_TRY(pk_fft_act_reserve(h->d_xg, tl.rows, 8 * Hh));
    PK_TRY(pk_fft_act_reserve(h->d_mem, tl.rows, Eg));
    if (G > 0) PK_TRY(pk_fft_act_reserve(h->d_enc, tl.rows, E));
    PK_TRY(pk_fft_act_reserve(h->d_pkey, tl.rows, Da));
    PK_HIP(hipMemsetAsync(h->d_e1.p, 0, h->d_e1.cap, ctx->stream));   // margins read by the k > 1 taps
    PK_HIP(hipMemsetAsync(h->d_e2.p, 0, h->d_e2.cap, ctx->stream));
    PK_HIP(hipMemsetAsync(h->d_mem.p, 0, h->d_mem.cap, ctx->stream));
    float* cur = pk_fft_act_ptr(h->d_e1, E);
    float* mem = pk_fft_act_ptr(h->d_mem, Eg);
    float* enc_out = G > 0 ? pk_fft_act_ptr(h->d_enc, E) : mem;
    PK_LAUNCH(ctx, "taco_embed", k_taco_embed, dim3(tl.rows), dim3(128), 0, h->d_tok.as<int>(),
              tones ? h->d_tone.as<int>() : (const int*)nullptr, tl.d_row_utt(), h->W(h->emb),
              c.n_tones > 0 ? h->W(h->temb) : (const float*)nullptr, E, cur);
    for (int i = 0; i < c.encoder_conv_layers; ++i) {
        // relu(Conv1dBatchNorm(x)); dropout is off in eval (:233-237); gap rows -> 0 = the conv's zero padding
        float* nxt = pk_fft_act_ptr((i & 1) ? h->d_e1 : h->d_e2, E);
        PK_TRY(pk_fft_run_dense(h, "taco_conv_encoder", h->econv[i], cur, E, nxt, E, tl.rows, PK_ACT_RELU, nullptr, 0,
                                tl.d_row_utt()));
        cur = nxt;
    }
    float* xg = pk_fft_act_ptr(h->d_xg, 8 * Hh);
    PK_TRY(pk_fft_run_dense(h, "taco_gemm_lstm_in", h->lstm_in, cur, E, xg, 8 * Hh, tl.rows, PK_ACT_NONE, nullptr, 0, nullptr));
    {
        const int nthreads = std::min(1024, ((4 * Hh + 63) / 64) * 64);
        PK_LAUNCH(ctx, "taco_lstm_seq", k_taco_lstm_seq, dim3(B, 2), dim3(nthreads), (size_t)6 * Hh * sizeof(float), xg,
                  h->W(h->whhT), tl.d_seg_start(), tl.d_seg_len(), Hh, enc_out);
    }
    if (G > 0) {
        PK_TRY(pk_upload(ctx, h->d_gc, cond_g.data(), cond_g.size() * sizeof(float)));
        PK_LAUNCH(ctx, "taco_concat_global", k_taco_concat_global, dim3(tl.rows), dim3(128), 0, enc_out, E, h->d_gc.as<float>(), G,
                  tl.d_row_utt(), mem);
    }
    float* pkey = pk_fft_act_ptr(h->d_pkey, Da);
    PK_TRY(pk_fft_run_dense(h, "taco_gemm_key", h->key_layer, mem, Eg, pkey, Da, tl.rows, PK_ACT_NONE, nullptr, 0, nullptr));   // :376
    // ---- decoder state (:352-376): zeros
    const int K1 = Pn + Eg + Ha, K2 = Ha + Eg + Hd, K3 = Hd + Eg;
    // the LSTM operand rows are double buffered: a cell's GEMM reads [x | context | h(t-1)] from one buffer while its
    // epilogue (and the attention kernel) write h(t) / context(t) for the NEXT step into the other
    PK_TRY(rows_reserve(h->d_in1, B, K1));
    PK_TRY(rows_reserve(h->d_in1b, B, K1));
    PK_TRY(rows_reserve(h->d_in2, B, K2));
    PK_TRY(rows_reserve(h->d_in2b, B, K2));
    PK_TRY(rows_reserve(h->d_in3, B, K3));
    PK_TRY(rows_reserve(h->d_p1, B, Pn));
    PK_TRY(rows_reserve(h->d_gates, B, 4 * std::max(Ha, Hd)));
    PK_TRY(rows_reserve(h->d_zero, B, M));
    PK_TRY(rows_reserve(h->d_y, (long)cap * B, M));
    PK_TRY(h->d_catt.reserve((size_t)B * Ha * sizeof(float)));
    PK_TRY(h->d_cdec.reserve((size_t)B * Hd * sizeof(float)));
    PK_TRY(h->d_attw.reserve((size_t)tl.rows_alloc * sizeof(float)));
    PK_TRY(h->d_cum.reserve((size_t)tl.rows_alloc * sizeof(float)));
    PK_TRY(h->d_logits.reserve((size_t)cap * B * sizeof(float)));
    PK_TRY(rows_reserve(h->d_pq, B, Da));
    PK_TRY(h->d_energy.reserve((size_t)tl.rows_alloc * sizeof(float)));
    pk_dbuf* zbufs[] = {&h->d_in1, &h->d_in1b, &h->d_in2, &h->d_in2b, &h->d_in3, &h->d_zero, &h->d_catt, &h->d_cdec, &h->d_attw, &h->d_cum};
    for (pk_dbuf* z : zbufs) PK_HIP(hipMemsetAsync(z->p, 0, z->cap, ctx->stream));
    // state block: [len B][first_hit B][ndone 1]
    {
        std::vector<int> st(2 * (size_t)B + 1, 0);
        for (int b = 0; b < B; ++b) st[B + b] = -1;
        PK_TRY(pk_upload(ctx, h->d_state, st.data(), st.size() * sizeof(int)));
    }
    int* d_len = h->d_state.as<int>();
    int* d_first = d_len + B;
    int* d_ndone = d_len + 2 * B;
    const unsigned long long* d_seeds = nullptr;
    if (seeds) {
        PK_TRY(pk_upload(ctx, h->d_seeds, seeds, (size_t)B * sizeof(uint64_t)));
        d_seeds = h->d_seeds.as<unsigned long long>();
    }
    h->align_off.assign(B, 0);
    long align_total = 0;
    for (int b = 0; b < B; ++b) {
        h->align_off[b] = align_total;
        align_total += (long)cap * tok_lens[b];
    }
    PK_TRY(h->d_align.reserve((size_t)align_total * sizeof(float)));
    PK_TRY(pk_upload(ctx, h->d_alignoff, h->align_off.data(), (size_t)B * sizeof(long)));
    float* in1_[2] = {pk_fft_act_ptr(h->d_in1, K1), pk_fft_act_ptr(h->d_in1b, K1)};
    float* in2_[2] = {pk_fft_act_ptr(h->d_in2, K2), pk_fft_act_ptr(h->d_in2b, K2)};
    float* in3 = pk_fft_act_ptr(h->d_in3, K3);
    float* p1 = pk_fft_act_ptr(h->d_p1, Pn);
    float* gates = pk_fft_act_ptr(h->d_gates, 4 * std::max(Ha, Hd));
    float* Y = pk_fft_act_ptr(h->d_y, M);
    const float* zero = pk_fft_act_ptr(h->d_zero, M);
    const float p = c.p_prenet_dropout;
    const bool drop = h->dropout && p > 0.f;
    const unsigned thr = pk_dropout_threshold((double)p);
    const float dscale = 1.0f / (1.0f - p);
    const size_t lsa_smem = (size_t)(8 + maxT + 4) * sizeof(float);
    if (lsa_smem > 60 * 1024) PK_FAIL(PK_EUNSUPPORTED, "pk_taco_infer: %d tokens exceed the attention kernel's LDS budget", maxT);
    static const int poll = pk_prof_env("PK_TACO_POLL") ? std::max(1, atoi(pk_prof_env("PK_TACO_POLL"))) : 8;
    static const bool use_rg = pk_prof_env("PK_AR_ROWGEMM") ? atoi(pk_prof_env("PK_AR_ROWGEMM")) != 0 : true;
    float* pq = pk_fft_act_ptr(h->d_pq, Da);
    // row GEMM of one of the per-step layers (B rows)
    // LSTMCell on the row GEMM: gates = [x | context | h] . [W_ih^T ; W_hh^T] + b, cell finished in the epilogue
    auto rowlstm = [&](const char* name, const pk_taco::RowW& w, const float* x, int ldx, float* cstate, int H, float* h1,
                       int ld1, float* h2, int ld2) -> int {
        pk_rowgemm_args g;
        g.x = x; g.ldx = ldx; g.Wt = h->W(w.w); g.bias = h->W(w.b); g.M = B; g.K = w.K; g.N = w.N;
        g.lstm_c = cstate; g.lstm_H = H; g.lstm_h1 = h1; g.lstm_ld1 = ld1; g.lstm_h2 = h2; g.lstm_ld2 = ld2;
        return pk_rowgemm_launch(ctx, name, g);
    };
    auto rowgemm = [&](const char* name, const pk_taco::RowW& w, const float* x, int ldx, float* y, int ldy, int act,
                       int drop_layer, unsigned long long step) -> int {
        pk_rowgemm_args g;
        g.x = x; g.ldx = ldx; g.Wt = h->W(w.w); g.bias = w.b == (size_t)-1 ? nullptr : h->W(w.b);
        g.y = y; g.ldy = ldy; g.M = B; g.K = w.K; g.N = w.N; g.act = act;
        if (drop_layer >= 0 && drop) {
            g.dropout = 1; g.drop_base = step; g.drop_J = 2; g.drop_j = drop_layer; g.drop_seeds = d_seeds;
            g.drop_thr = thr; g.drop_scale = dscale;
        }
        return pk_rowgemm_launch(ctx, name, g);
    };
    int i = 0;
    for (i = 0; i < cap; ++i) {
        // query = prenet(previous mel_output) (:499-500, :538); the first query is zeros (:493-497)
        const float* q = i == 0 ? zero : Y + (long)(i - 1) * B * M;
        float* in1 = in1_[i & 1];          // this step's attention-LSTM operand rows
        float* in1n = in1_[(i & 1) ^ 1];   // the next step's: context(t) and attention_hidden(t) go there
        float* in2 = in2_[i & 1];
        float* in2n = in2_[(i & 1) ^ 1];
        if (use_rg && Pn % 4 == 0 && Pn / 4 <= 512 && 512 % (Pn / 4) == 0 && Pn <= 512 && M <= 512 && K1 % 4 == 0) {
            // both prenet layers in one launch (pk_ar.h k_ar_prenet_embed without the input layer)
            pk_prenet_embed pe;
            memset(&pe, 0, sizeof(pe));
            pe.y = q; pe.ldy = M; pe.O = M; pe.U = Pn; pe.A = Pn; pe.B = B;
            pe.w1 = h->W(h->pre1_kn); pe.w2 = h->W(h->pre2_kn);
            pe.x0 = in1; pe.ldx0 = K1;
            pe.dropout = drop ? 1 : 0; pe.base = (unsigned long long)i; pe.J = 2; pe.seeds = d_seeds; pe.thr = thr; pe.scale = dscale;
            PK_LAUNCH(ctx, "taco_prenet", k_ar_prenet_embed, dim3(B), dim3(512), 0, pe);
            PK_TRY(rowlstm("taco_row_att_rnn", h->rw_att, in1, K1, h->d_catt.as<float>(), Ha, in1n + Pn + Eg, K1, in2, K2));
        } else if (use_rg) {
            PK_TRY(rowgemm("taco_row_prenet", h->rw_pre1, q, M, p1, Pn, PK_ACT_RELU, 0, (unsigned long long)i));
            PK_TRY(rowgemm("taco_row_prenet", h->rw_pre2, p1, Pn, in1, K1, PK_ACT_RELU, 1, (unsigned long long)i));
            // attention_rnn (:380-385) on [prenet | context | attention_hidden]; h -> the two operand rows that read it
            PK_TRY(rowlstm("taco_row_att_rnn", h->rw_att, in1, K1, h->d_catt.as<float>(), Ha, in1n + Pn + Eg, K1, in2, K2));
        } else {
            PK_TRY(pk_fft_run_dense(h, "taco_gemm_prenet", h->pre1, q, M, p1, Pn, B, PK_ACT_RELU, nullptr, 0, nullptr));
            if (drop)
                PK_LAUNCH(ctx, "taco_dropout", k_ar_dropout, dim3(pk_div_up((long)B * (Pn / 4), 256)), dim3(256), 0, p1, Pn, B,
                          Pn, B, (unsigned long long)i, 2, 0, d_seeds, thr, dscale, (float*)nullptr);
            PK_TRY(pk_fft_run_dense(h, "taco_gemm_prenet", h->pre2, p1, Pn, in1, K1, B, PK_ACT_RELU, nullptr, 0, nullptr));
            if (drop)
                PK_LAUNCH(ctx, "taco_dropout", k_ar_dropout, dim3(pk_div_up((long)B * (Pn / 4), 256)), dim3(256), 0, in1, K1, B,
                          Pn, B, (unsigned long long)i, 2, 1, d_seeds, thr, dscale, (float*)nullptr);
            PK_TRY(pk_fft_run_dense(h, "taco_gemm_att_rnn", h->att_rnn, in1, K1, gates, 4 * Ha, B, PK_ACT_NONE, nullptr, 0, nullptr));
            PK_LAUNCH(ctx, "taco_lstm_point", k_taco_lstm_point, dim3(pk_div_up((long)B * Ha, 256)), dim3(256), 0, gates,
                      h->d_catt.as<float>(), Ha, B, in1n + Pn + Eg, K1, in2, K2);
        }
        // location sensitive attention (:387-397): processed query, energies of every memory row, softmax + context
        PK_TRY(rowgemm("taco_row_query", h->rw_q, in2, K2, pq, Da, PK_ACT_NONE, -1, 0));
        LsaArgs a;
        memset(&a, 0, sizeof(a));
        a.Da = Da; a.E = Eg; a.F = c.attention_filters; a.K = c.attention_kernel_size;
        a.pq = pq; a.Wconv = h->W(h->Wconv); a.Wloc = h->W(h->Wloc); a.v = h->W(h->vvec);
        a.pkey = pkey; a.mem = mem;
        a.row_utt = tl.d_row_utt(); a.row_pos = tl.d_row_pos();
        a.seg_start = tl.d_seg_start(); a.seg_len = tl.d_seg_len(); a.rows = tl.rows;
        a.energy = h->d_energy.as<float>();
        a.attw = h->d_attw.as<float>(); a.cum = h->d_cum.as<float>();
        a.ctx1 = in1n + Pn; a.ld1 = K1;
        a.ctx2 = in2 + Ha; a.ld2 = K2;
        a.ctx3 = in3 + Hd; a.ld3 = K3;
        a.align = h->d_align.as<float>(); a.align_off = h->d_alignoff.as<long>(); a.step = i;
        PK_LAUNCH(ctx, "taco_lsa_energy", k_taco_lsa_energy, dim3(pk_div_up(tl.rows, 4)), dim3(256), 0, a);
        PK_LAUNCH(ctx, "taco_lsa_ctx", k_taco_lsa_ctx, dim3(B, pk_div_up(Eg, 256)), dim3(256), lsa_smem, a);
        // decoder_rnn (:399-403) on [attention_hidden | context | decoder_hidden]
        if (use_rg) {
            PK_TRY(rowlstm("taco_row_dec_rnn", h->rw_dec, in2, K2, h->d_cdec.as<float>(), Hd, in2n + Ha + Eg, K2, in3, K3));
        } else {
            PK_TRY(pk_fft_run_dense(h, "taco_gemm_dec_rnn", h->dec_rnn, in2, K2, gates, 4 * Hd, B, PK_ACT_NONE, nullptr, 0, nullptr));
            PK_LAUNCH(ctx, "taco_lstm_point", k_taco_lstm_point, dim3(pk_div_up((long)B * Hd, 256)), dim3(256), 0, gates,
                      h->d_cdec.as<float>(), Hd, B, in2n + Ha + Eg, K2, in3, K3);
        }
        // linear_projection on [decoder_hidden | context] (:409-413) -> this step's mel row; stop rules (:515-528)
        if (use_rg && c.use_stop_token && B <= PK_RG_ROWS) {
            // (one launch: the stop token rides on the projection's row GEMM as extra workgroups, pk_rowgemm.h stop_kind 1)
            pk_rowgemm_args g;
            g.x = in3; g.ldx = K3; g.Wt = h->W(h->rw_proj.w); g.bias = h->rw_proj.b == (size_t)-1 ? nullptr : h->W(h->rw_proj.b);
            g.y = Y + (long)i * B * M; g.ldy = M; g.M = B; g.K = h->rw_proj.K; g.N = h->rw_proj.N; g.act = PK_ACT_NONE;
            g.stop_w = h->W(h->stop_w); g.stop_bias = h->stop_b; g.stop_kind = 1; g.stop_step = i; g.stop_max_steps = cap;
            g.stop_probs = h->d_logits.as<float>(); g.stop_len = d_len; g.stop_ndone = d_ndone;
            PK_TRY(pk_rowgemm_launch(ctx, "taco_row_proj_stop", g));
        } else {
        if (use_rg)
            PK_TRY(rowgemm("taco_row_proj", h->rw_proj, in3, K3, Y + (long)i * B * M, M, PK_ACT_NONE, -1, 0));
        else
            PK_TRY(pk_fft_run_dense(h, "taco_gemm_proj", h->proj, in3, K3, Y + (long)i * B * M, M, B, PK_ACT_NONE, nullptr, 0, nullptr));
        PK_LAUNCH(ctx, "taco_stop", k_taco_stop, dim3(pk_div_up(B, 4)), dim3(256), 0, in3, K3, K3,
                  c.use_stop_token ? h->W(h->stop_w) : (const float*)nullptr, h->stop_b, c.use_stop_token ? 1 : 0, B, i, cap,
                  h->d_attw.as<float>(), tl.d_seg_start(), tl.d_seg_len(), h->d_logits.as<float>(), d_len, d_first, d_ndone);
        }
        if ((i + 1) % poll == 0 || i + 1 == cap) {
            int ndone = 0;
            PK_HIP(hipMemcpyAsync(&ndone, d_ndone, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            PK_HIP(hipStreamSynchronize(ctx->stream));
            if (ndone >= B) break;
        }
    }
    h->steps = std::min(i + 1, cap);
    h->len.resize(B);
    PK_HIP(hipMemcpyAsync(h->len.data(), d_len, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    PK_HIP(hipStreamSynchronize(ctx->stream));
    for (int b = 0; b < B; ++b) {
        if (h->len[b] <= 0 || h->len[b] > h->steps)
            PK_FAIL(PK_EHIP, "pk_taco_infer: utterance %d did not stop within %d steps (internal error)", b, h->steps);
        out_frames[b] = h->len[b];
    }
    h->inferred = true;
    return PK_OK;
}

extern "C" int pk_taco_read(pk_taco* h, float* mel_output, float* mel_outputs_postnet, float* alignments,
                            float* stop_logits, int32_t flags) {
    if (!h) PK_FAIL(PK_EINVAL, "pk_taco_read: handle is NULL");
    if (!h->inferred) PK_FAIL(PK_ESTATE, "pk_taco_read: call pk_taco_infer first");
    if (stop_logits && !h->cfg.use_stop_token) PK_FAIL(PK_ESTATE, "pk_taco_read: the model has no stop token");
    pk_ctx* ctx = h->ctx;
    PK_DEVICE(ctx->device);
    const pk_taco_cfg& c = h->cfg;
    const int B = h->B, M = c.d_mels;
    long total = 0;
    for (int b = 0; b < B; ++b) total += h->len[b];
    PK_TRY(pk_fft_build_timeline(ctx, h->tl_frm, h->len.data(), B, h->gapr));
    Timeline& tl = h->tl_frm;
    {
        std::vector<int> rowmap(tl.rows_alloc, -1);
        int o = 0;
        for (int b = 0; b < B; ++b)
            for (int l = 0; l < h->len[b]; ++l) rowmap[tl.seg_start[b] + l] = o++;
        PK_TRY(pk_upload(ctx, h->d_rowmap, rowmap.data(), rowmap.size() * sizeof(int)));
    }
    const bool host = (flags & PK_HOST_IO) != 0;
    const float* Y = pk_fft_act_ptr(h->d_y, M);
    const float* nof = nullptr;
    const int* noi = nullptr;
    if (mel_output) {
        float* d = mel_output;
        if (host) {
            PK_TRY(h->d_stage.reserve((size_t)total * M * sizeof(float)));
            d = h->d_stage.as<float>();
        }
        PK_LAUNCH(ctx, "taco_gather", k_ar_gather, dim3(tl.rows), dim3(128), 0, Y, M, B, 0, tl.d_row_utt(), tl.d_row_pos(),
                  h->d_rowmap.as<int>(), nof, nof, d);
        if (host) {
            PK_HIP(hipMemcpyAsync(mel_output, d, (size_t)total * M * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
            PK_HIP(hipStreamSynchronize(ctx->stream));   // d_stage is reused below
        }
    }
    if (mel_outputs_postnet) {
        // mel_outputs + postnet(mel_outputs) (:825-826): mel rows on a frame timeline with zero gap rows
        float* d = mel_outputs_postnet;
        if (host) {
            PK_TRY(h->d_stage.reserve((size_t)total * M * sizeof(float)));
            d = h->d_stage.as<float>();
        }
        PK_TRY(pk_fft_act_reserve(h->d_before, tl.rows, M));
        PK_HIP(hipMemsetAsync(h->d_before.p, 0, h->d_before.cap, ctx->stream));
        float* before = pk_fft_act_ptr(h->d_before, M);
        PK_LAUNCH(ctx, "taco_gather", k_ar_gather, dim3(tl.rows), dim3(128), 0, Y, M, B, 0, tl.d_row_utt(), tl.d_row_pos(), noi,
                  nof, nof, before);
        PK_TRY(pk_fft_run_postnet(h, "taco_conv_postnet", h->postnet, before, M, c.d_postnet, tl, h->d_q1, h->d_q2, d,
                                  h->d_rowmap.as<int>(), nullptr, nullptr));
        if (host) PK_HIP(hipMemcpyAsync(mel_outputs_postnet, d, (size_t)total * M * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    }
    if (stop_logits) {
        float* d = stop_logits;
        if (host) {
            PK_TRY(h->d_stage2.reserve((size_t)total * sizeof(float)));
            d = h->d_stage2.as<float>();
        }
        PK_LAUNCH(ctx, "taco_gather", k_ar_gather, dim3(tl.rows), dim3(128), 0, h->d_logits.as<float>(), 1, B, 0, tl.d_row_utt(),
                  tl.d_row_pos(), h->d_rowmap.as<int>(), nof, nof, d);
        if (host) PK_HIP(hipMemcpyAsync(stop_logits, d, (size_t)total * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    }
    if (alignments) {
        long o = 0;
        for (int b = 0; b < B; ++b) {
            const size_t n = (size_t)h->len[b] * h->T[b];
            PK_HIP(hipMemcpyAsync(alignments + o, h->d_align.as<float>() + h->align_off[b], n * sizeof(float),
                                  host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, ctx->stream));
            o += (long)n;
        }
    }
    if (host) PK_HIP(hipStreamSynchronize(ctx->stream));
    return PK_OK;
}

extern "C" int pk_taco_debug_read(pk_taco* h, int32_t what, int32_t b, float* host_out, int64_t n_floats) {
    if (!h || !host_out) PK_FAIL(PK_EINVAL, "pk_taco_debug_read: NULL argument");
    if (!h->inferred) PK_FAIL(PK_ESTATE, "pk_taco_debug_read: nothing has run");
    if (b < 0 || b >= h->B) PK_FAIL(PK_EINVAL, "pk_taco_debug_read: utterance out of range");
    if (what != 0) PK_FAIL(PK_EINVAL, "pk_taco_debug_read: unknown tap %d", what);
    pk_ctx* ctx = h->ctx;
    PK_DEVICE(ctx->device);
    const int E = h->cfg.d_encoder;
    const long n = (long)h->T[b] * E;
    if (n_floats != n) PK_FAIL(PK_ESHAPE, "pk_taco_debug_read: expected %ld floats, got %lld", n, (long long)n_floats);
    PK_HIP(hipStreamSynchronize(ctx->stream));
    const float* enc = h->cfg.d_global_condition > 0 ? pk_fft_act_ptr(h->d_enc, E) : pk_fft_act_ptr(h->d_mem, E);
    PK_HIP(hipMemcpy(host_out, enc + (long)h->tl_tok.seg_start[b] * E, n * sizeof(float), hipMemcpyDeviceToHost));
    return PK_OK;
}

extern "C" void pk_taco_destroy(pk_taco* h) {
    if (!h) return;
    pk_device_guard _dg(h->ctx->device);
    (void)hipStreamSynchronize(h->ctx->stream);
    h->release_core();
    pk_dbuf* bufs[] = {&h->d_gc, &h->d_enc, &h->d_tok, &h->d_tone, &h->d_e1, &h->d_e2, &h->d_xg, &h->d_mem, &h->d_pkey, &h->d_attw, &h->d_cum,
                       &h->d_in1, &h->d_in1b, &h->d_in2, &h->d_in2b, &h->d_in3, &h->d_p1, &h->d_gates, &h->d_catt, &h->d_cdec, &h->d_zero, &h->d_y, &h->d_pq, &h->d_energy,
                       &h->d_logits, &h->d_state, &h->d_seeds, &h->d_align, &h->d_alignoff, &h->d_before, &h->d_q1,
                       &h->d_q2, &h->d_rowmap, &h->d_stage, &h->d_stage2};
    for (auto* b : bufs) b->release();
    h->tl_tok.release();
    h->tl_frm.release();
    delete h;
}
