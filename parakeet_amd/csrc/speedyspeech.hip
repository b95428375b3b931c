// speedyspeech.hip -- SpeedySpeech inference on the engine's row-timeline GEMMs.
//
// Reference: parakeet/models/speedyspeech/speedyspeech.py
//   ResidualBlock :21-39, TextEmbedding :42-73, SpeedySpeechEncoder :76-105, DurationPredictor :108-118,
//   SpeedySpeechDecoder :121-139, SpeedySpeech.inference :178-218, SpeedySpeechInference :221-231,
//   expand modules/expansion.py:19-37, sinusoid_position_encoding modules/positional_encoding.py:20-39.
//
// Layout: as FastSpeech2 -- a batch of ragged utterances is one channels-last "row timeline" [rows][H] with
// GAP zero rows between utterances (GAP >= the largest one-sided reach of any convolution), so a
// Conv1D("same") is one implicit GEMM whose taps read row r + offset; rows that belong to no utterance are
// forced to zero in every epilogue, which is the reference's zero padding.  Every
// Conv1D -> ReLU -> BatchNorm1D(eval) sub-block is ONE GEMM launch (bias, ReLU, the batch-norm affine and,
// for the last sub-block, the block's residual all live in the epilogue).
//
// padding="same" semantics: see pk_ss_cfg.same_padding_resets_dilation in pk_synth.h.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "pk_gemm.h"

namespace {

// text (+ tone) embedding, padding_idx 0 for both tables (:42-73); gap rows -> 0
__global__ __launch_bounds__(128) void k_ss_embed(const int* __restrict__ text, const int* __restrict__ tone,
                                                  const int* __restrict__ row_utt, const float* __restrict__ etab,
                                                  const float* __restrict__ ttab, int H, float* __restrict__ x) {
    const int r = blockIdx.x;
    const bool valid = row_utt[r] >= 0;
    const int id = valid ? text[r] : 0;
    const int tn = (valid && tone) ? tone[r] : 0;
    for (int c = threadIdx.x; c < H; c += blockDim.x) {
        float v = 0.f;
        if (id != 0) v = etab[(long)id * H + c];
        if (tn != 0) v += ttab[(long)tn * H + c];
        x[(long)r * H + c] = v;
    }
}

// duration head: Linear(H, 1) (:113), then round(exp(.)) with paddle.round = half away from zero (:191-192);
// one wave per row.  dur (float, integer-valued) = 0 on gap rows.
__global__ __launch_bounds__(256) void k_ss_duration(const float* __restrict__ h, int H, const float* __restrict__ w,
                                                     float bias, const int* __restrict__ row_utt, int rows,
                                                     float* __restrict__ pred, float* __restrict__ dur) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    float s = 0.f;
    for (int c = lane; c < H; c += 64) s = fmaf(h[(long)r * H + c], w[c], s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) {
        const bool valid = row_utt[r] >= 0;
        const float p = s + bias;
        pred[r] = valid ? p : 0.f;
        dur[r] = valid ? floorf(expf(p) + 0.5f) : 0.f;   // exp > 0: half away from zero == floor(x + 0.5)
    }
}

// inclusive prefix sums of the durations per utterance + frame counts; one block per utterance
__global__ __launch_bounds__(256) void k_ss_cumsum(const float* __restrict__ dur, const int* __restrict__ seg_start,
                                                   const int* __restrict__ seg_len, int* __restrict__ cum,
                                                   int* __restrict__ frames) {
    __shared__ int part[256];
    const int b = blockIdx.x, s0 = seg_start[b], n = seg_len[b];
    const int per = (n + 255) / 256;
    const int lo = threadIdx.x * per, hi = min(n, lo + per);
    int sum = 0;
    for (int t = lo; t < hi; ++t) sum += (int)dur[s0 + t];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < 256; ++i) {
            const int v = part[i];
            part[i] = run;
            run += v;
        }
        frames[b] = run;
    }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int t = lo; t < hi; ++t) {
        run += (int)dur[s0 + t];
        cum[s0 + t] = run;
    }
}

// expand (:194-209: frame f of an utterance copies the token whose duration interval contains f; tokens with
// d < 1 own no frame) + sinusoid_position_encoding(t_dec, H) (:213-215), gap rows -> 0
__global__ __launch_bounds__(128) void k_ss_expand(const float* __restrict__ enc, const int* __restrict__ cum,
                                                   const int* __restrict__ tok_start, const int* __restrict__ tok_len,
                                                   const int* __restrict__ row_utt, const int* __restrict__ row_pos,
                                                   int H, float* __restrict__ x) {
    const int r = blockIdx.x;
    const int b = row_utt[r];
    if (b < 0) {
        for (int c = threadIdx.x; c < H; c += blockDim.x) x[(long)r * H + c] = 0.f;
        return;
    }
    const int f = row_pos[r], s0 = tok_start[b], n = tok_len[b];
    int lo = 0, hi = n - 1;   // first token with cum > f
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cum[s0 + mid] > f) hi = mid;
        else lo = mid + 1;
    }
    const float* src = enc + (long)(s0 + lo) * H;
    for (int c = threadIdx.x; c < H; c += blockDim.x) {
        const float channel = (float)(c & ~1);
        const float p = (float)f / powf(10000.0f, channel / (float)H);
        x[(long)r * H + c] = src[c] + ((c & 1) ? cosf(p) : sinf(p));
    }
}

struct Dense {
    size_t w = 0, b = (size_t)-1, wh = (size_t)-1, cs = (size_t)-1, ch = (size_t)-1;
    int Cin = 0, N = 0, k = 1, dilation = 1;
};

struct ResBlock {
    std::vector<Dense> sub;   // Conv1D -> ReLU -> BatchNorm1D each
};

struct Timeline {
    int B = 0, rows = 0, rows_alloc = 0;
    std::vector<int> seg_start, seg_len;
    pk_dbuf d_tab;   // [seg_start B][seg_len B][row_utt rows_alloc][row_pos rows_alloc]
    const int* d_seg_start() const { return d_tab.as<int>(); }
    const int* d_seg_len() const { return d_tab.as<int>() + B; }
    const int* d_row_utt() const { return d_tab.as<int>() + 2 * B; }
    const int* d_row_pos() const { return d_tab.as<int>() + 2 * B + rows_alloc; }
};

int build_timeline(pk_ctx* ctx, Timeline& tl, const int* lens, int B, int gap) {
    tl.B = B;
    tl.seg_start.resize(B);
    tl.seg_len.assign(lens, lens + B);
    int r = gap;
    for (int b = 0; b < B; ++b) {
        tl.seg_start[b] = r;
        r += lens[b] + gap;
    }
    tl.rows = r;
    tl.rows_alloc = ((r + PK_GEMM_BM - 1) / PK_GEMM_BM) * PK_GEMM_BM;
    std::vector<int> tab(2 * (size_t)B + 2 * (size_t)tl.rows_alloc, 0);
    for (int b = 0; b < B; ++b) {
        tab[b] = tl.seg_start[b];
        tab[B + b] = lens[b];
    }
    int* row_utt = tab.data() + 2 * B;
    int* row_pos = row_utt + tl.rows_alloc;
    std::fill(row_utt, row_utt + tl.rows_alloc, -1);
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < lens[b]; ++t) {
            row_utt[tl.seg_start[b] + t] = b;
            row_pos[tl.seg_start[b] + t] = t;
        }
    return pk_upload(ctx, tl.d_tab, tab.data(), tab.size() * sizeof(int));
}
}  // namespace

struct pk_ss {
    pk_ctx* ctx = nullptr;
    pk_ss_cfg cfg;
    pk_param_map params;
    bool finalized = false, encoded = false;
    int H = 0, gap = 1, lead = 8;
    int math = PK_GEMM_MATH_F16X3;
    std::vector<float> arena_h;
    std::vector<uint16_t> arena16_h;
    pk_dbuf arena, arena16;
    size_t text_tab = 0, tone_tab = 0, dur_w = 0;
    float dur_b = 0.f;
    Dense prenet, enc_post1, enc_post2, dec_post1, dec_out;
    size_t enc_bn_s = 0, enc_bn_t = 0;
    std::vector<ResBlock> enc_blocks, dur_blocks, dec_blocks;
    ResBlock dec_post2;
    bool has_out_affine = false;
    std::vector<float> h_out_scale, h_out_shift;
    // per call
    Timeline tl_tok, tl_frm;
    pk_dbuf d_text, d_tone, d_e, d_a, d_b, d_c, d_enc, d_pred, d_dur, d_cum, d_frames, d_rowmap, d_stage;
    std::vector<int> frames;

    const float* W(size_t off) const { return arena.as<float>() + off; }
};

namespace {
struct Arena {
    std::vector<float>& v;
    std::vector<uint16_t>& v16;
    size_t put(const std::vector<float>& x) {
        size_t o = (v.size() + 3) & ~(size_t)3;
        v.resize(o);
        v.insert(v.end(), x.begin(), x.end());
        return o;
    }
    size_t put16(const std::vector<uint16_t>& x) {
        size_t o = (v16.size() + 7) & ~(size_t)7;
        v16.resize(o);
        v16.insert(v16.end(), x.begin(), x.end());
        return o;
    }
};

int add_dense_kn(Arena& ar, const std::vector<float>& kn, const std::vector<float>* bias, int Cin, int k, int N,
                 Dense& d) {
    std::vector<float> packed;
    pk_gemm_pack(kn.data(), Cin * k, N, packed);
    d.w = ar.put(packed);
    if (Cin % PK_GEMM_HBK == 0) {
        std::vector<uint16_t> ph;
        pk_gemm_pack_h3(kn.data(), Cin * k, N, ph);
        d.wh = ar.put16(ph);
    }
    if (bias) d.b = ar.put(*bias);
    d.Cin = Cin;
    d.N = N;
    d.k = k;
    return PK_OK;
}

int add_linear(Arena& ar, const pk_param_map& P, const std::string& base, int Cin, int N, Dense& d) {
    std::vector<float> w, b;
    PK_TRY(pk_get_weight(P, base, {Cin, N}, w));   // Linear weight [in, out]
    PK_TRY(pk_get_vector(P, base + ".bias", N, b));
    return add_dense_kn(ar, w, &b, Cin, 1, N, d);
}

// BatchNorm1D eval (eps 1e-5) as y = x * s + t
int bn_affine(const pk_param_map& P, const std::string& base, int n, std::vector<float>& s, std::vector<float>& t) {
    std::vector<float> g, b, mean, var;
    PK_TRY(pk_get_vector(P, base + ".weight", n, g));
    PK_TRY(pk_get_vector(P, base + ".bias", n, b));
    PK_TRY(pk_get_vector(P, base + "._mean", n, mean));
    PK_TRY(pk_get_vector(P, base + "._variance", n, var));
    s.resize(n);
    t.resize(n);
    for (int i = 0; i < n; ++i) {
        const double sc = (double)g[i] / std::sqrt((double)var[i] + 1e-5);
        s[i] = (float)sc;
        t[i] = (float)((double)b[i] - (double)mean[i] * sc);
    }
    return PK_OK;
}

int add_res_block(Arena& ar, const pk_param_map& P, const std::string& base, int ch, int k, int dilation, int n,
                  ResBlock& rb) {
    rb.sub.resize(n);
    for (int j = 0; j < n; ++j) {
        const std::string p = base + ".blocks." + std::to_string(j);
        std::vector<float> w, kn, b, s, t;
        PK_TRY(pk_get_weight(P, p + ".0", {ch, ch, k}, w));
        PK_TRY(pk_get_vector(P, p + ".0.bias", ch, b));
        pk_conv_to_kn(w.data(), ch, ch, k, kn);
        PK_TRY(add_dense_kn(ar, kn, &b, ch, k, ch, rb.sub[j]));
        PK_TRY(bn_affine(P, p + ".2", ch, s, t));
        rb.sub[j].cs = ar.put(s);
        rb.sub[j].ch = ar.put(t);
        rb.sub[j].dilation = dilation;
    }
    return PK_OK;
}

// one-sided reaches of a Conv1D(padding="same"): rows before / after the output row
void same_reach(const pk_ss_cfg& c, int k, int dilation, int& d_eff, int& before, int& after) {
    d_eff = c.same_padding_resets_dilation ? 1 : dilation;
    const int pad_sum = d_eff * (k - 1);
    before = pad_sum / 2;
    after = pad_sum - before;
}

int launch_dense(pk_ss* h, const char* name, const Dense& d, const float* A, float* C, int ldc, int rows, int act,
                 const float* res, int res_pos, const int* rowvalid, const float* cscale, const float* cshift,
                 const int* out_rowmap) {
    pk_gemm_args g;
    g.A = A;
    g.lda = d.Cin;
    g.Wp = h->W(d.w);
    g.Wh = d.wh == (size_t)-1 ? nullptr : h->arena16.as<uint16_t>() + d.wh;
    g.math = h->math;
    g.bias = d.b == (size_t)-1 ? nullptr : h->W(d.b);
    g.res = res;
    g.ldr = d.N;
    g.res_pos = res_pos;
    g.C = C;
    g.ldc = ldc;
    g.rowvalid = rowvalid;
    g.cscale = cscale;
    g.cshift = cshift;
    g.out_rowmap = out_rowmap;
    g.M = rows;
    g.N = d.N;
    g.Cin = d.Cin;
    g.act = act;
    int d_eff, before, after;
    same_reach(h->cfg, d.k, d.dilation, d_eff, before, after);
    if (d.k > PK_GEMM_MAX_TAPS) PK_FAIL(PK_EUNSUPPORTED, "SpeedySpeech: kernel size %d > %d", d.k, PK_GEMM_MAX_TAPS);
    g.ntaps = d.k;
    for (int t = 0; t < d.k; ++t) {
        g.tap_off[t] = (long)(t * d_eff - before) * d.Cin;
        g.tap_w[t] = t;
    }
    g.wslabs_total = d.k * d.Cin / PK_GEMM_BK;
    return pk_gemm_launch(h->ctx, name, g);
}

// x + [conv -> ReLU -> BN] x n  (:21-39).  in != out; tmp is scratch for n == 2.
int run_res_block(pk_ss* h, const char* name, const ResBlock& rb, const float* in, float* tmp, float* out, int rows,
                  const int* rowvalid) {
    const float* cur = in;
    const int n = (int)rb.sub.size();
    for (int j = 0; j < n; ++j) {
        const Dense& d = rb.sub[j];
        const bool last = j == n - 1;
        PK_TRY(launch_dense(h, name, d, cur, last ? out : tmp, d.N, rows, PK_ACT_RELU, last ? in : nullptr,
                            PK_RES_AFTER_AFFINE, rowvalid, h->W(d.cs), h->W(d.ch), nullptr));
        cur = tmp;
    }
    return PK_OK;
}

int act_reserve(pk_ss* h, pk_dbuf& buf, int rows_alloc, int C) {
    return buf.reserve(((size_t)rows_alloc + 2 * (size_t)h->lead) * C * sizeof(float));
}
float* act_ptr(pk_ss* h, const pk_dbuf& buf, int C) { return buf.as<float>() + (size_t)h->lead * C; }
}  // namespace

extern "C" int pk_ss_create(pk_ctx* ctx, const pk_ss_cfg* cfg, pk_ss** out) {
    if (!ctx || !cfg || !out) PK_FAIL(PK_EINVAL, "pk_ss_create: NULL argument");
    *out = nullptr;
    const pk_ss_cfg& c = *cfg;
    if (c.vocab_size <= 0 || c.tone_size < 0) PK_FAIL(PK_EINVAL, "SpeedySpeech: vocab_size must be positive");
    const int H = c.encoder_hidden_size;
    if (H <= 0 || c.decoder_output_size <= 0) PK_FAIL(PK_EINVAL, "SpeedySpeech: sizes must be positive");
    if (c.duration_predictor_hidden_size != H || c.decoder_hidden_size != H)
        PK_FAIL(PK_ESHAPE, "SpeedySpeech: encoder (%d), duration predictor (%d) and decoder (%d) widths must agree "
                           "(the encodings feed both, speedyspeech.py:188-189,216)",
                H, c.duration_predictor_hidden_size, c.decoder_hidden_size);
    if (H % PK_GEMM_BK != 0) PK_FAIL(PK_EUNSUPPORTED, "SpeedySpeech: hidden size %d not a multiple of 16", H);
    if (c.n_encoder_dilations < 0 || c.n_encoder_dilations > 32 || c.n_decoder_dilations < 0 || c.n_decoder_dilations > 32)
        PK_FAIL(PK_EUNSUPPORTED, "SpeedySpeech: at most 32 residual blocks per stack");
    if (c.encoder_kernel_size < 1 || c.decoder_kernel_size < 1 || c.encoder_kernel_size > PK_GEMM_MAX_TAPS ||
        c.decoder_kernel_size > PK_GEMM_MAX_TAPS)
        PK_FAIL(PK_EUNSUPPORTED, "SpeedySpeech: kernel sizes must be in [1, %d]", PK_GEMM_MAX_TAPS);
    pk_ss* h = new pk_ss();
    h->ctx = ctx;
    h->cfg = c;
    h->H = H;
    // the widest one-sided reach decides the zero gap between utterances and the buffer margins
    int reach = 2;   // duration predictor k = 4: 1 before, 2 after
    auto upd = [&](int k, int d) {
        int de, b, a;
        same_reach(c, k, d, de, b, a);
        reach = std::max(reach, std::max(b, a));
    };
    for (int i = 0; i < c.n_encoder_dilations; ++i) {
        if (c.encoder_dilations[i] < 1) { delete h; PK_FAIL(PK_EINVAL, "SpeedySpeech: dilation must be >= 1"); }
        upd(c.encoder_kernel_size, c.encoder_dilations[i]);
    }
    for (int i = 0; i < c.n_decoder_dilations; ++i) {
        if (c.decoder_dilations[i] < 1) { delete h; PK_FAIL(PK_EINVAL, "SpeedySpeech: dilation must be >= 1"); }
        upd(c.decoder_kernel_size, c.decoder_dilations[i]);
    }
    upd(c.decoder_kernel_size, 1);
    h->gap = reach;
    h->lead = std::max(8, reach);
    if (const char* e = pk_prof_env("PK_SS_MATH")) h->math = strcmp(e, "f32") == 0 ? PK_GEMM_MATH_F32 : PK_GEMM_MATH_F16X3;
    *out = h;
    return PK_OK;
}

extern "C" int pk_ss_set_param(pk_ss* h, const char* name, const float* data, const int64_t* shape, int32_t ndim) {
    if (!h) PK_FAIL(PK_EINVAL, "pk_ss_set_param: handle is NULL");
    h->finalized = false;
    return pk_store_param(h->params, name, data, shape, ndim);
}

extern "C" int pk_ss_set_normalizer(pk_ss* h, const float* mu, const float* sigma, int32_t n) {
    if (!h) PK_FAIL(PK_EINVAL, "pk_ss_set_normalizer: handle is NULL");
    h->finalized = false;
    if (!mu && !sigma) {
        h->has_out_affine = false;
        return PK_OK;
    }
    if (!mu || !sigma) PK_FAIL(PK_EINVAL, "pk_ss_set_normalizer: mu and sigma must both be given");
    if (n != h->cfg.decoder_output_size) PK_FAIL(PK_ESHAPE, "pk_ss_set_normalizer: %d bins, model has %d", n, h->cfg.decoder_output_size);
    h->h_out_scale.assign(sigma, sigma + n);   // ZScore.inverse: x * sigma + mu (normalizer.py:30-33)
    h->h_out_shift.assign(mu, mu + n);
    h->has_out_affine = true;
    return PK_OK;
}

extern "C" int pk_ss_set_math(pk_ss* h, int32_t mode) {
    if (!h) PK_FAIL(PK_EINVAL, "pk_ss_set_math: handle is NULL");
    if (mode != PK_GEMM_MATH_F32 && mode != PK_GEMM_MATH_F16X3) PK_FAIL(PK_EINVAL, "pk_ss_set_math: unknown mode %d", mode);
    h->math = mode;
    return PK_OK;
}

extern "C" int pk_ss_finalize(pk_ss* h) {
    if (!h) PK_FAIL(PK_EINVAL, "pk_ss_finalize: handle is NULL");
    pk_ctx* ctx = h->ctx;
    PK_DEVICE(ctx->device);
    const pk_ss_cfg& c = h->cfg;
    const pk_param_map& P = h->params;
    const int H = h->H;
    h->arena_h.clear();
    h->arena16_h.clear();
    Arena ar{h->arena_h, h->arena16_h};
    std::vector<float> t;
    PK_TRY(pk_get_weight(P, "encoder.embedding.text_embedding", {c.vocab_size, H}, t));
    h->text_tab = ar.put(t);
    if (c.tone_size > 0) {
        PK_TRY(pk_get_weight(P, "encoder.embedding.tone_embedding", {c.tone_size, H}, t));
        h->tone_tab = ar.put(t);
    }
    PK_TRY(add_linear(ar, P, "encoder.prenet.0", H, H, h->prenet));
    h->enc_blocks.resize(c.n_encoder_dilations);
    for (int i = 0; i < c.n_encoder_dilations; ++i)
        PK_TRY(add_res_block(ar, P, "encoder.res_blocks." + std::to_string(i), H, c.encoder_kernel_size,
                             c.encoder_dilations[i], 2, h->enc_blocks[i]));
    PK_TRY(add_linear(ar, P, "encoder.postnet1.0", H, H, h->enc_post1));
    {
        std::vector<float> s, sh;
        PK_TRY(bn_affine(P, "encoder.postnet2.1", H, s, sh));
        h->enc_bn_s = ar.put(s);
        h->enc_bn_t = ar.put(sh);
    }
    PK_TRY(add_linear(ar, P, "encoder.postnet2.2", H, H, h->enc_post2));
    h->dur_blocks.resize(3);
    const int dk[3] = {4, 3, 1};   // speedyspeech.py:112-114
    for (int i = 0; i < 3; ++i)
        PK_TRY(add_res_block(ar, P, "duration_predictor.layers." + std::to_string(i), H, dk[i], 1, 1, h->dur_blocks[i]));
    {
        std::vector<float> w, b;
        PK_TRY(pk_get_weight(P, "duration_predictor.layers.3", {H, 1}, w));
        PK_TRY(pk_get_vector(P, "duration_predictor.layers.3.bias", 1, b));
        h->dur_w = ar.put(w);
        h->dur_b = b[0];
    }
    h->dec_blocks.resize(c.n_decoder_dilations);
    for (int i = 0; i < c.n_decoder_dilations; ++i)
        PK_TRY(add_res_block(ar, P, "decoder.res_blocks." + std::to_string(i), H, c.decoder_kernel_size,
                             c.decoder_dilations[i], 2, h->dec_blocks[i]));
    PK_TRY(add_linear(ar, P, "decoder.postnet1.0", H, H, h->dec_post1));
    PK_TRY(add_res_block(ar, P, "decoder.postnet2.0", H, c.decoder_kernel_size, 1, 2, h->dec_post2));
    PK_TRY(add_linear(ar, P, "decoder.postnet2.1", H, c.decoder_output_size, h->dec_out));
    if (h->has_out_affine) {
        h->dec_out.cs = ar.put(h->h_out_scale);
        h->dec_out.ch = ar.put(h->h_out_shift);
    }
    PK_TRY(pk_upload(ctx, h->arena, h->arena_h.data(), h->arena_h.size() * sizeof(float)));
    if (!h->arena16_h.empty())
        PK_TRY(pk_upload(ctx, h->arena16, h->arena16_h.data(), h->arena16_h.size() * sizeof(uint16_t)));
    h->arena_h.clear();
    h->arena_h.shrink_to_fit();
    h->arena16_h.clear();
    h->arena16_h.shrink_to_fit();
    h->finalized = true;
    h->encoded = false;
    return PK_OK;
}

extern "C" int pk_ss_encode(pk_ss* h, const int64_t* text, const int64_t* tones, const int32_t* tok_lens, int32_t B,
                            int32_t* out_frames) {
    if (!h || !text || !tok_lens || !out_frames) PK_FAIL(PK_EINVAL, "pk_ss_encode: NULL argument");
    if (!h->finalized) PK_FAIL(PK_ESTATE, "pk_ss_encode: call pk_ss_finalize first");
    if (B <= 0) PK_FAIL(PK_EINVAL, "pk_ss_encode: batch size must be positive");
    pk_ctx* ctx = h->ctx;
    PK_DEVICE(ctx->device);
    const pk_ss_cfg& c = h->cfg;
    const int H = h->H;
    if (tones && c.tone_size <= 0) PK_FAIL(PK_ESTATE, "pk_ss_encode: the model has no tone embedding");
    for (int b = 0; b < B; ++b)
        if (tok_lens[b] <= 0) PK_FAIL(PK_EINVAL, "pk_ss_encode: utterance %d has %d tokens", b, tok_lens[b]);
    h->encoded = false;
    PK_TRY(build_timeline(ctx, h->tl_tok, tok_lens, B, h->gap));
    Timeline& tl = h->tl_tok;
    {
        std::vector<int> tx(tl.rows_alloc, 0), tn(tl.rows_alloc, 0);
        long o = 0;
        for (int b = 0; b < B; ++b)
            for (int t = 0; t < tok_lens[b]; ++t, ++o) {
                if (text[o] < 0 || text[o] >= c.vocab_size)
                    PK_FAIL(PK_EINVAL, "pk_ss_encode: token id %lld out of [0,%d)", (long long)text[o], c.vocab_size);
                tx[tl.seg_start[b] + t] = (int)text[o];
                if (tones) {
                    if (tones[o] < 0 || tones[o] >= c.tone_size)
                        PK_FAIL(PK_EINVAL, "pk_ss_encode: tone id %lld out of [0,%d)", (long long)tones[o], c.tone_size);
                    tn[tl.seg_start[b] + t] = (int)tones[o];
                }
            }
        PK_TRY(pk_upload(ctx, h->d_text, tx.data(), tx.size() * sizeof(int)));
        if (tones) PK_TRY(pk_upload(ctx, h->d_tone, tn.data(), tn.size() * sizeof(int)));
    }
    pk_dbuf* bufs[] = {&h->d_e, &h->d_a, &h->d_b, &h->d_c, &h->d_enc};
    for (pk_dbuf* b : bufs) PK_TRY(act_reserve(h, *b, tl.rows_alloc, H));
    float *e = act_ptr(h, h->d_e, H), *xa = act_ptr(h, h->d_a, H), *xb = act_ptr(h, h->d_b, H),
          *xc = act_ptr(h, h->d_c, H), *enc = act_ptr(h, h->d_enc, H);
    const int* rv = tl.d_row_utt();
    const int rows = tl.rows;
    PK_LAUNCH(ctx, "ss_embed", k_ss_embed, dim3(tl.rows_alloc), dim3(128), 0, h->d_text.as<int>(),
              tones ? h->d_tone.as<int>() : (const int*)nullptr, rv, h->W(h->text_tab),
              c.tone_size > 0 ? h->W(h->tone_tab) : (const float*)nullptr, H, xa);
    // prenet: Linear + ReLU (:84-86) -> e ("embedding" of :100-103)
    PK_TRY(launch_dense(h, "ss_gemm_prenet", h->prenet, xa, e, H, rows, PK_ACT_RELU, nullptr, 0, rv, nullptr, nullptr,
                        nullptr));
    // res_blocks (:101)
    const float* cur = e;
    float* ping[2] = {xa, xb};
    int pp = 0;
    for (const ResBlock& rb : h->enc_blocks) {
        PK_TRY(run_res_block(h, "ss_conv_enc_block", rb, cur, xc, ping[pp], rows, rv));
        cur = ping[pp];
        pp ^= 1;
    }
    // x = embedding + postnet1(x); postnet2 = ReLU -> BN -> Linear (:102-103): the first two ride the epilogue
    PK_TRY(launch_dense(h, "ss_gemm_enc_post1", h->enc_post1, cur, xc, H, rows, PK_ACT_RELU, e, PK_RES_BEFORE_ACT, rv,
                        h->W(h->enc_bn_s), h->W(h->enc_bn_t), nullptr));
    PK_TRY(launch_dense(h, "ss_gemm_enc_post2", h->enc_post2, xc, enc, H, rows, PK_ACT_NONE, nullptr, 0, rv, nullptr,
                        nullptr, nullptr));
    // duration predictor (:108-118, :189-192)
    cur = enc;
    pp = 0;
    for (const ResBlock& rb : h->dur_blocks) {
        PK_TRY(run_res_block(h, "ss_conv_dur_block", rb, cur, xc, ping[pp], rows, rv));
        cur = ping[pp];
        pp ^= 1;
    }
    PK_TRY(h->d_pred.reserve((size_t)tl.rows_alloc * 4));
    PK_TRY(h->d_dur.reserve((size_t)tl.rows_alloc * 4));
    PK_TRY(h->d_cum.reserve((size_t)tl.rows_alloc * 4));
    PK_TRY(h->d_frames.reserve((size_t)B * 4));
    PK_LAUNCH(ctx, "ss_duration", k_ss_duration, dim3(pk_div_up(rows, 4)), dim3(256), 0, cur, H, h->W(h->dur_w),
              h->dur_b, rv, rows, h->d_pred.as<float>(), h->d_dur.as<float>());
    PK_LAUNCH(ctx, "ss_cumsum", k_ss_cumsum, dim3(B), dim3(256), 0, h->d_dur.as<float>(), tl.d_seg_start(),
              tl.d_seg_len(), h->d_cum.as<int>(), h->d_frames.as<int>());
    h->frames.resize(B);
    PK_HIP(hipMemcpyAsync(h->frames.data(), h->d_frames.p, (size_t)B * 4, hipMemcpyDeviceToHost, ctx->stream));
    PK_HIP(hipStreamSynchronize(ctx->stream));   // t_dec is data dependent (:194-196)
    for (int b = 0; b < B; ++b) out_frames[b] = h->frames[b];
    h->encoded = true;
    return PK_OK;
}

extern "C" int pk_ss_decode(pk_ss* h, float* mel_out, int32_t flags) {
    if (!h || !mel_out) PK_FAIL(PK_EINVAL, "pk_ss_decode: NULL argument");
    if (!h->encoded) PK_FAIL(PK_ESTATE, "pk_ss_decode: call pk_ss_encode first");
    pk_ctx* ctx = h->ctx;
    PK_DEVICE(ctx->device);
    const pk_ss_cfg& c = h->cfg;
    const int H = h->H, B = h->tl_tok.B, O = c.decoder_output_size;
    long total = 0;
    for (int b = 0; b < B; ++b) total += h->frames[b];
    if (total == 0) return PK_OK;
    PK_TRY(build_timeline(ctx, h->tl_frm, h->frames.data(), B, h->gap));
    Timeline& tl = h->tl_frm;
    // packed output rows: utterance-major, no gaps
    {
        std::vector<int> rowmap(tl.rows_alloc, -1);
        int o = 0;
        for (int b = 0; b < B; ++b)
            for (int f = 0; f < h->frames[b]; ++f) rowmap[tl.seg_start[b] + f] = o++;
        PK_TRY(pk_upload(ctx, h->d_rowmap, rowmap.data(), rowmap.size() * sizeof(int)));
    }
    pk_dbuf* bufs[] = {&h->d_e, &h->d_a, &h->d_b, &h->d_c};
    for (pk_dbuf* b : bufs) PK_TRY(act_reserve(h, *b, tl.rows_alloc, H));
    float *x0 = act_ptr(h, h->d_e, H), *xa = act_ptr(h, h->d_a, H), *xb = act_ptr(h, h->d_b, H),
          *xc = act_ptr(h, h->d_c, H);
    const float* enc = act_ptr(h, h->d_enc, H);
    const int* rv = tl.d_row_utt();
    const int rows = tl.rows;
    PK_LAUNCH(ctx, "ss_expand", k_ss_expand, dim3(tl.rows_alloc), dim3(128), 0, enc, h->d_cum.as<int>(),
              h->tl_tok.d_seg_start(), h->tl_tok.d_seg_len(), rv, tl.d_row_pos(), H, x0);
    const float* cur = x0;
    float* ping[2] = {xa, xb};
    int pp = 0;
    for (const ResBlock& rb : h->dec_blocks) {
        PK_TRY(run_res_block(h, "ss_conv_dec_block", rb, cur, xc, ping[pp], rows, rv));
        cur = ping[pp];
        pp ^= 1;
    }
    // x = x + postnet1(xx) (:136)
    float* y = ping[pp];
    PK_TRY(launch_dense(h, "ss_gemm_dec_post1", h->dec_post1, cur, y, H, rows, PK_ACT_NONE, x0, PK_RES_AFTER_ACT, rv,
                        nullptr, nullptr, nullptr));
    pp ^= 1;
    float* z = ping[pp];
    PK_TRY(run_res_block(h, "ss_conv_dec_block", h->dec_post2, y, xc, z, rows, rv));
    float* d_out = mel_out;
    if (flags & PK_HOST_IO) {
        PK_TRY(h->d_stage.reserve((size_t)total * O * 4));
        d_out = h->d_stage.as<float>();
    }
    PK_TRY(launch_dense(h, "ss_gemm_out", h->dec_out, z, d_out, O, rows, PK_ACT_NONE, nullptr, 0, rv,
                        (h->has_out_affine && (flags & PK_APPLY_NORMALIZER)) ? h->W(h->dec_out.cs) : nullptr,
                        (h->has_out_affine && (flags & PK_APPLY_NORMALIZER)) ? h->W(h->dec_out.ch) : nullptr, h->d_rowmap.as<int>()));
    if (flags & PK_HOST_IO) {
        PK_HIP(hipMemcpyAsync(mel_out, d_out, (size_t)total * O * 4, hipMemcpyDeviceToHost, ctx->stream));
        PK_HIP(hipStreamSynchronize(ctx->stream));
    }
    return PK_OK;
}

/* what: 0 = encodings (T_b, H), 1 = predicted log-durations (T_b), 2 = integer durations (T_b). */
extern "C" int pk_ss_debug_read(pk_ss* h, int32_t what, int32_t b, float* host_out, int64_t n_floats) {
    if (!h || !host_out) PK_FAIL(PK_EINVAL, "pk_ss_debug_read: NULL argument");
    if (!h->encoded) PK_FAIL(PK_ESTATE, "pk_ss_debug_read: no encode has run");
    if (b < 0 || b >= h->tl_tok.B) PK_FAIL(PK_EINVAL, "pk_ss_debug_read: utterance out of range");
    pk_ctx* ctx = h->ctx;
    PK_DEVICE(ctx->device);
    const int T = h->tl_tok.seg_len[b], s0 = h->tl_tok.seg_start[b], H = h->H;
    const float* src;
    long n;
    switch (what) {
        case 0: src = act_ptr(h, h->d_enc, H) + (size_t)s0 * H; n = (long)T * H; break;
        case 1: src = h->d_pred.as<float>() + s0; n = T; break;
        case 2: src = h->d_dur.as<float>() + s0; n = T; break;
        default: PK_FAIL(PK_EINVAL, "pk_ss_debug_read: unknown tap %d", what);
    }
    if (n_floats != n) PK_FAIL(PK_ESHAPE, "pk_ss_debug_read: expected %ld floats, got %lld", n, (long long)n_floats);
    PK_HIP(hipStreamSynchronize(ctx->stream));
    PK_HIP(hipMemcpy(host_out, src, (size_t)n * 4, hipMemcpyDeviceToHost));
    return PK_OK;
}

extern "C" void pk_ss_destroy(pk_ss* h) {
    if (!h) return;
    pk_device_guard _dg(h->ctx->device);
    (void)hipStreamSynchronize(h->ctx->stream);
    pk_dbuf* bufs[] = {&h->arena, &h->arena16, &h->d_text, &h->d_tone, &h->d_e, &h->d_a, &h->d_b, &h->d_c, &h->d_enc,
                       &h->d_pred, &h->d_dur, &h->d_cum, &h->d_frames, &h->d_rowmap, &h->d_stage, &h->tl_tok.d_tab,
                       &h->tl_frm.d_tab};
    for (auto* b : bufs) b->release();
    delete h;
}
