// pk_rowgemm.h -- "row GEMM": y[M][N] = epilogue(LN?(x)[M][K] . W[K][N]) for the FEW rows (one per utterance of the
// batch) that an autoregressive decoder produces per step (tts.hip, taco2.hip).  The tile GEMM of gemm.hip needs
// 20-60 us for such a problem (4..32 workgroups, a three-slab pipeline to fill, a barrier per slab); here the weight
// matrix is the only real traffic and is streamed exactly once, sequentially:
//   weights are pre-tiled at finalize as [N / CW][K][CW] (CW = 16 or 64 columns per workgroup), so a workgroup's slab
//   is one contiguous K x CW x 4-byte range; a workgroup = CW output columns x up to 32 rows, 8 waves; a wave-wide load
//   covers 64 / CW consecutive k of the slab (256 bytes), the 8 waves the next 2 KB, 16 loads in flight per wave; a lane
//   owns one (column, K-part) pair and all 32 rows (accumulators in registers); the activations sit transposed in LDS
//   ([k][32 rows], broadcast reads); the 8 * 64 / CW K-parts are summed through LDS in a fixed order (deterministic).
//   CW = 16 everywhere: measured (MI355X, 32 rows) against CW = 64 for N >= 2048: K = 2560, N = 4096 took 68 us with 64
//   workgroups of 64 columns (42 MB at 0.6 TB/s: too few loads in flight), the narrow tiling gives 256 workgroups.
//   Arithmetic: plain fp32 FMA on the VALU (exact: no operand splitting, no block scaling) -- the matrix pipe would
//   buy nothing at 32 rows, the kernel is bound by the weight stream (4 B per 2 x 32 FLOP).
//   Optional prologue: LayerNorm over K (K <= 512) of every row, so that norm -> Linear is one launch.
//   Optional epilogue: bias, ReLU, dropout (the dropout stream of pk_synth.h), residual.
#pragma once
#include <vector>

#include "pk_common.h"

constexpr int PK_RG_KC = 512;     // K chunk staged in LDS
constexpr int PK_RG_ROWS = 32;

// columns per workgroup for a layer with N outputs
static inline int pk_rowgemm_cw(int N) {
    (void)N;
    return 16;
}
// [K][N] row-major -> tiles [ceil(N / cw)][K][cw], columns beyond N zero
void pk_rowgemm_pack(const float* Wkn, int K, int N, std::vector<float>& out);
// LSTM gate columns [i (H) | f (H) | g (H) | o (H)] -> blocks of 16: [i f g o] x 4 units each; perm[new] = old column
void pk_rowgemm_lstm_perm(int H, std::vector<int>& perm);

struct pk_rowgemm_args {
    const float* x = nullptr;   // [M][ldx], K columns used; ldx % 4 == 0, 16-byte aligned
    int ldx = 0;
    const float* Wt = nullptr;  // pk_rowgemm_pack of the [K][N] matrix (paddle Linear.weight is stored [in, out])
    const float* bias = nullptr;   // [N] or NULL
    const float* res = nullptr;    // [M][ldr] or NULL: added last
    int ldr = 0;
    float* y = nullptr;
    int ldy = 0;
    int M = 0, K = 0, N = 0;
    int act = 0;                   // PK_ACT_NONE / PK_ACT_RELU (pk_gemm.h)
    const float* ln_g = nullptr;   // LayerNorm(K) weight / bias applied to x first (needs K <= PK_RG_KC), or NULL
    const float* ln_b = nullptr;
    float ln_eps = 1e-5f;
    // LSTMCell epilogue (lstm_c != NULL): N = 4 * lstm_H gate columns, PERMUTED at pack time (pk_rowgemm_lstm_perm) so
    // that workgroup bx holds i | f | g | o of units 4 * bx .. 4 * bx + 3; the epilogue then finishes the cell
    // (c' = sigmoid(f) c + sigmoid(i) tanh(g), h' = sigmoid(o) tanh(c')), updates lstm_c [M][lstm_H] in place and writes h'
    // to two destinations (operand rows of the GEMMs that consume it); y is not written.  No act / dropout / res.
    float* lstm_c = nullptr;
    int lstm_H = 0;
    float* lstm_h1 = nullptr;
    int lstm_ld1 = 0;
    float* lstm_h2 = nullptr;
    int lstm_ld2 = 0;
    // dropout after the activation (before the residual): element index ((drop_base * drop_J + drop_j) * N + n) of the
    // utterance's stream (row m = utterance m), keep <=> word >= drop_thr, kept values * drop_scale
    int dropout = 0;
    unsigned long long drop_base = 0;
    int drop_J = 1, drop_j = 0;
    const unsigned long long* drop_seeds = nullptr;
    unsigned drop_thr = 0;
    float drop_scale = 1.f;
    // Stop-token head riding on the launch (stop_w != NULL; the autoregressive decoders' prob_out, transformer_tts.py:638-642):
    // a few more workgroups (a wave per row) compute, for every row m, p = sigmoid(LN?(x[m]) . stop_w + stop_bias) (the same LayerNorm prologue as
    // the GEMM when ln_g is set, K <= 1024), stores it at stop_probs[(stop_step - 1) * M + m] and applies the stop rule:
    // stop_len[m] == 0 (still running) and (p >= stop_thr or stop_step >= stop_maxlen[m]) and stop_step >= stop_minlen[m]
    // -> stop_len[m] = stop_step, ++*stop_ndone.  (The decoders launched a kernel of their own for this: 10 us per step of
    // a chain in which nothing else depends on it.)  M <= 32.
    // stop_kind 1 = Tacotron2's rule (models/tacotron2.py:515-528 with use_stop_token): the raw logit s is stored at
    // stop_probs[stop_step * M + m] (stop_step counted from 0), the utterance ends when sigmoid(s) > 0.5 or
    // stop_step + 1 >= stop_max_steps: stop_len[m] = stop_step + 1.  No LayerNorm, any K.
    const float* stop_w = nullptr;
    float stop_bias = 0.f, stop_thr = 0.5f;
    int stop_kind = 0, stop_max_steps = 0;
    int stop_step = 0;
    const int* stop_minlen = nullptr;
    const int* stop_maxlen = nullptr;
    float* stop_probs = nullptr;
    int* stop_len = nullptr;
    int* stop_ndone = nullptr;
};

int pk_rowgemm_launch(pk_ctx* ctx, const char* prof_name, const pk_rowgemm_args& a);
