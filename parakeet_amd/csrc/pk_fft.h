// pk_fft.h -- the transformer ("FFT block") machinery shared by the models built from
// parakeet/modules/fastspeech2_transformer/: FastSpeech2 (fs2.hip, encoder and decoder are both `Encoder`
// stacks, fastspeech2.py:171,251) and TransformerTTS (tts.hip: the same `Encoder` class as its text encoder,
// transformer_tts.py:278-292).  Definitions live in fs2.hip; the kernels are documented there.
//
//   row timeline, Dense / FftLayer weight records, the weight arena      
//   pk_fft_add_*   finalize-time packing of Linear / Conv1D / FFT stacks / the tacotron2-style Postnet
//   pk_fft_run_*   launches on a timeline: dense layer, LayerNorm, self-attention, a whole pre-norm stack, Postnet
#pragma once
#include <string>
#include <vector>

#include "pk_gemm.h"

constexpr int PK_FFT_MAX_HEADS = 16;   // (q|k|v, head) magnitude-bound constants are passed to a kernel by value
constexpr int PK_FFT_LN_MAXPER = 8;    // k_layernorm: channels <= 64 * 8
constexpr int PK_FFT_LEAD = 8;         // rows of margin in front of every activation buffer

struct pk_fft_dense {
    size_t w = 0, b = 0;   // offsets (floats) into the weight arena; b == SIZE_MAX: no bias
    size_t wh = (size_t)-1;   // offset (halves) of the split-fp16 fragments, SIZE_MAX if Cin % 32 != 0
    size_t wp = (size_t)-1;   // offset (halves) of the planes-kernel fragments (pk_ffn_planes.h), SIZE_MAX if not packed
    size_t wp4 = (size_t)-1;  // first feed-forward conv: the same for the kernel with 4 tiles per wave (short timelines)
    size_t wp1 = (size_t)-1;  // both feed-forward convs: the same for the kernel with ONE tile per wave (an utterance or two)
    size_t wps = (size_t)-1;  // offset (floats) of their [N / 32] scale factors
    int Cin = 0, N = 0, taps = 1, pad = 0;
    // |y[r, n]| <= c1 * max|x[r + tap, :]| + c0 with c1 = max_n sum_k |W[k, n]|, c0 = max_n |bias[n]|: an upper
    // bound on the magnitude of this layer's output rows, used as the block maximum of the NEXT split-fp16 GEMM's
    // operand scale (pk_split.h) so that no pass over the activations is needed.  A bound that is loose by 2^k
    // only moves the scheme's error floor from 2^-39 to 2^(k-39) of the block maximum (fp32 itself: 2^-24).
    float c1 = 0.f, c0 = 0.f;
};

struct pk_fft_layer {
    size_t ln1_g, ln1_b, ln2_g, ln2_b;
    pk_fft_dense qkv, out, ffn1, ffn2;
    // concat_after (encoder_layer.py:103-106): x = residual + concat_linear(cat(x, self_attn(x))) as two dense layers,
    // the x half (with the bias) and the attention half of the [2A][A] weight
    bool concat = false;
    pk_fft_dense cat_x, cat_a;
    float qkv_c1[3 * PK_FFT_MAX_HEADS] = {0}, qkv_c0[3 * PK_FFT_MAX_HEADS] = {0};   // the same bound per (q|k|v, head)
};

struct pk_fft_timeline {
    int B = 0, rows = 0;
    std::vector<int> seg_start, seg_len, row_utt, row_pos;
    pk_dbuf d_tab;  // [seg_start B][seg_len B][row_utt rows_alloc][row_pos rows_alloc]
    const int* d_seg_start() const { return d_tab.as<int>(); }
    const int* d_seg_len() const { return d_tab.as<int>() + B; }
    const int* d_row_utt() const { return d_tab.as<int>() + 2 * B; }
    const int* d_row_pos() const { return d_tab.as<int>() + 2 * B + rows_alloc; }
    int rows_alloc = 0;
    void release() { d_tab.release(); }
};

struct pk_fft_arena {
    std::vector<float>& v;
    std::vector<uint16_t>* v16 = nullptr;
    size_t put(const std::vector<float>& x) {
        size_t o = (v.size() + 3) & ~(size_t)3;  // 16-byte alignment
        v.resize(o);
        v.insert(v.end(), x.begin(), x.end());
        return o;
    }
    size_t put16(const std::vector<uint16_t>& x) {
        size_t o = (v16->size() + 7) & ~(size_t)7;
        v16->resize(o);
        v16->insert(v16->end(), x.begin(), x.end());
        return o;
    }
};

// What a stack of FFT blocks needs from its owner: weights, math mode, the positional table and the activation
// buffers of one stack run.  pk_fs2 and pk_tts derive from it.
struct pk_fft_core {
    pk_ctx* ctx = nullptr;
    int adim = 0, aheads = 0;
    std::vector<float> arena_h;
    pk_dbuf arena;
    std::vector<uint16_t> arena16_h;
    pk_dbuf arena16;
    int math = PK_GEMM_MATH_F16X3;   // dense layers: 3-term split-fp16 MFMA (fp32-equivalent error) or exact fp32
    bool attn_lds = true;            // measurement switch (PK_FS2_ATTN_NO_LDS): per-wave K/V loads instead
    bool no_bounds = false;          // measurement switch (PK_FS2_NO_BOUNDS): block maxima by passes over the data
    // options (pk_fs2_set_option / pk_tts_set_option)
    bool ffn_planes = true;          // "ffn_planes": feed-forward convs on the planes kernels (pk_ffn_planes.h) where packed for them
    int ffn_planes_min_blocks = 0;   // "ffn_planes_min_blocks": timelines shorter than this many 32-row blocks stay on the tile GEMM
    int ffn_one_tile_max = 4096;     // "ffn_one_tile_max": the feed-forward convs run one 32-column tile per wave while blocks x N / 32 <= this
    int ffnp_variant = 0;            // "ffnp_variant": tiling override of the planes kernels (ffnp_conv_launch), 0 = by shape
    int attn_waves = 0;              // "attn_waves": 4 / 8 query tiles per attention workgroup, 0 = by shape
    int max_len = 0;                 // rows of the positional table
    pk_dbuf d_pe, d_div;
    pk_dbuf d_x, d_h, d_qkv, d_ctx, d_f, d_lnamax, d_cbnd, d_fbnd, d_segb, d_cat;
    pk_dbuf d_hp, d_fp, d_pam;       // planes of the norm2 output and of the hidden activations, their block maxima
    pk_dbuf d_ones;                  // a row of 1.0f: the operand bound of layers that read tanh outputs (pk_fft_run_postnet)
    pk_dbuf d_pnp[2], d_pnam[2];     // the postnet's middle layers on planes: two work buffers and their row maxima

    const float* W(size_t off) const { return arena.as<float>() + off; }
    void release_core() {
        pk_dbuf* bufs[] = {&arena, &arena16, &d_pe, &d_div, &d_x, &d_h, &d_qkv, &d_ctx, &d_f, &d_lnamax, &d_cbnd,
                           &d_fbnd, &d_segb, &d_cat, &d_hp, &d_fp, &d_pam, &d_ones, &d_pnp[0], &d_pnp[1], &d_pnam[0], &d_pnam[1]};
        for (pk_dbuf* b : bufs) b->release();
    }
};

int pk_fft_act_reserve(pk_dbuf& buf, int rows, int C);
static inline float* pk_fft_act_ptr(const pk_dbuf& buf, int C) { return buf.as<float>() + (size_t)PK_FFT_LEAD * C; }
int pk_fft_build_timeline(pk_ctx* ctx, pk_fft_timeline& tl, const int* lens, int B, int gapr);

// ---- finalize
int pk_fft_add_dense_kn(pk_fft_arena& ar, const std::vector<float>& kn, const std::vector<float>* bias, int Cin,
                        int taps, int N, pk_fft_dense& d);
int pk_fft_add_linear(pk_fft_arena& ar, const pk_param_map& P, const std::string& base, int Cin, int N,
                      pk_fft_dense& d);   // Linear weight [in, out] + bias
// planes: 1 / 2 = also pack the weights as the first / second feed-forward conv of the planes kernels (pk_ffn_planes.h)
int pk_fft_add_conv(pk_fft_arena& ar, const pk_param_map& P, const std::string& base, int Cout, int Cin, int k,
                    bool bias, pk_fft_dense& d, int planes = 0);
// Conv1D -> BatchNorm1D(eval, eps 1e-5) folded into one dense layer (tacotron2/decoder.py:133-147,
// tacotron2/encoder.py:98-110; with conv_bias: Conv1dBatchNorm, modules/conv.py:186-260): conv at `conv_base`,
// batch norm at `bn_base`
int pk_fft_add_conv_bn(pk_fft_arena& ar, const pk_param_map& P, const std::string& conv_base,
                       const std::string& bn_base, int Cout, int Cin, int k, pk_fft_dense& d, bool conv_bias = false);
int pk_fft_add_vec(pk_fft_arena& ar, const pk_param_map& P, const std::string& name, int n, size_t& off);
// `n_layers` EncoderLayers under prefix + ".encoders.{l}" and prefix + ".after_norm"; ff_type 0 conv1d, 1 linear,
// 2 conv1d-linear (encoder.py:145-170).  normalize_before = false: post-norm blocks, no after_norm (encoder.py:142-143);
// concat_after: every layer has a concat_linear (encoder_layer.py:61-62)
int pk_fft_add_stack(pk_fft_arena& ar, const pk_param_map& P, const std::string& prefix, int n_layers, int A,
                     int units, int k, int ff_type, int heads, std::vector<pk_fft_layer>& out, size_t& after_g,
                     size_t& after_b, bool normalize_before = true, bool concat_after = false);
// Postnet (modules/tacotron2/decoder.py:127-198) under prefix + ".postnet.{j}": Conv1D(no bias) + BatchNorm folded
int pk_fft_add_postnet(pk_fft_arena& ar, const pk_param_map& P, const std::string& prefix, int n_layers, int odim,
                       int chans, int filts, std::vector<pk_fft_dense>& out);

// ---- run (all on core->ctx->stream)
int pk_fft_ensure_pe(pk_fft_core* h, int need);
// x[r] = table[tok[r]] * xscale + alpha * PE[pos[r]]; gap rows zero
int pk_fft_embed(pk_fft_core* h, const char* name, const int* d_tok, const pk_fft_timeline& tl, size_t table,
                 float alpha, float xscale, float* x);
// LayerNorm (eps 1e-5) of `rows` rows of C channels; d_row_utt[r] < 0 marks rows to zero; amax (optional): max|y[r,:]|
int pk_fft_layernorm_rows(pk_fft_core* h, const float* x, size_t g, size_t b, const int* d_row_utt, int rows, int C,
                          float* y, float* amax = nullptr);
int pk_fft_run_dense(pk_fft_core* h, const char* name, const pk_fft_dense& d, const float* A, int lda, float* C,
                     int ldc, int rows, int act, const float* res, int ldr, const int* rowvalid,
                     const float* a_amax = nullptr);
int pk_fft_run_layernorm(pk_fft_core* h, const float* x, size_t g, size_t b, const pk_fft_timeline& tl, int C,
                         float* y, float* amax = nullptr);
int pk_fft_run_attention(pk_fft_core* h, const pk_fft_timeline& tl, const float* qkv, float* out,
                         const unsigned* seg_bounds = nullptr);
// N FFT blocks + after_norm on the timeline tl; the residual stream core->d_x is updated in place, result in hs_out
int pk_fft_run_stack(pk_fft_core* h, const std::vector<pk_fft_layer>& layers, size_t after_g, size_t after_b,
                     const pk_fft_timeline& tl, int units, float* hs_out, bool normalize_before = true);
// hs[r] += v[utterance of r] for the rows of a timeline that belong to an utterance; v: [B][adim]
int pk_fft_add_rowvec(pk_fft_core* h, const pk_fft_timeline& tl, const float* d_vec, float* hs);
// Speaker-embedding integration on the rows of a timeline ("add" / "concat"; hs_proj NULL = "add"); fs2.hip
int pk_fft_run_speaker(pk_fft_core* h, const pk_fft_timeline& tl, const long long* d_spk_id, const float* d_spembs,
                       size_t table, size_t w, size_t bias, const pk_fft_dense* hs_proj, int D, pk_dbuf& d_vec, float* hs,
                       float* tmp);
// after = before + postnet(before), then the optional per-column affine; rows stored through out_rowmap.
// q1 / q2: scratch activations of postnet_chans columns.
int pk_fft_run_postnet(pk_fft_core* h, const char* name, const std::vector<pk_fft_dense>& postnet, const float* before,
                       int odim, int chans, const pk_fft_timeline& tl, pk_dbuf& q1, pk_dbuf& q2, float* d_out,
                       const int* out_rowmap, const float* cscale, const float* cshift);
