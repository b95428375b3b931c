// pk_gemm.h -- fp32 MFMA implicit-conv GEMM used by every dense layer of the
// FastSpeech2 path (Linear, Conv1D k=1/3/5 in channels-last layout).
#pragma once
#include <cstdint>
#include <vector>

#include "pk_common.h"

constexpr int PK_GEMM_BM = 128;
constexpr int PK_GEMM_BN = 128;
constexpr int PK_GEMM_BK = 16;

constexpr int PK_GEMM_HBK = 32;   // K slab of the split-fp16 variant
constexpr int PK_GEMM_MAX_TAPS = 12;
enum { PK_GEMM_MATH_F32 = 0, PK_GEMM_MATH_F16X3 = 1, PK_GEMM_MATH_F16 = 2 };   // F16: WaveFlow only (pk_wf_set_math)

enum { PK_ACT_NONE = 0, PK_ACT_RELU = 1, PK_ACT_TANH = 2 };
enum { PK_EPI_STD = 0, PK_EPI_GATE = 1, PK_EPI_GATE_PROJ = 2 };
enum { PK_RES_AFTER_ACT = 0, PK_RES_AFTER_AFFINE = 1, PK_RES_BEFORE_ACT = 2 };

// C[r, n] = epilogue( sum_{tap, ci} A[r + tap - pad, ci] * W[tap*Cin + ci, n] )
//   epilogue: v += bias[n]; v = act(v); v += res[r, n]; v = rowvalid[r] ? v : 0;
//             v = v * cscale[n] + cshift[n]; store to C[out_rowmap ? out_rowmap[r] : r, n]
//   res_pos moves the residual: PK_RES_AFTER_AFFINE  v = rowvalid ? act(v + bias) * cscale + cshift + res : 0
//                               (x + BatchNorm(ReLU(conv(x))), SpeedySpeech's ResidualBlock)
//                               PK_RES_BEFORE_ACT    v = rowvalid ? act(v + bias + res) * cscale + cshift : 0
// A is row-major with leading dimension lda; rows r + tap - pad must be readable
// for every r in [0, ceil(M/128)*128) (buffers carry a margin).  Rows of the
// "row timeline" that belong to no utterance (gaps) are forced to zero via
// rowvalid so that the next k>1 convolution sees the reference's zero padding.
struct pk_gemm_args {
    const float* A = nullptr;
    int lda = 0;
    const float* Wp = nullptr;   // packed by pk_gemm_pack()
    const void* Wh = nullptr;    // packed by pk_gemm_pack_h3() (split-fp16 fragments); used when math == F16X3
    int math = PK_GEMM_MATH_F32; // F16X3 needs Wh and input channels that are multiples of 32, else falls back
    const float* bias = nullptr;
    const float* res = nullptr;
    int ldr = 0;
    int res_pos = PK_RES_AFTER_ACT;
    float* C = nullptr;
    int ldc = 0;
    const int* rowvalid = nullptr;   // >= 0 means valid (utterance id), < 0 gap
    const float* cscale = nullptr;
    const float* cshift = nullptr;
    const int* out_rowmap = nullptr;  // < 0: row not stored
    int M = 0, N = 0, Cin = 0, taps = 1, pad = 0;
    int act = PK_ACT_NONE;
    // ---- generalised K axis (WaveFlow's 2-D causal conv): K = ntaps*Cin (+ Cin2).
    // ntaps == 0 means "taps/pad" above (tap t reads row r + t - pad).  Otherwise tap t reads
    // A + tap_off[t] (floats) + r*lda and multiplies the weight slabs of packed tap tap_w[t]
    // (so a caller can skip taps whose input is known to be zero without repacking weights).
    int ntaps = 0;
    long tap_off[PK_GEMM_MAX_TAPS] = {0};
    int tap_w[PK_GEMM_MAX_TAPS] = {0};
    const float* A2 = nullptr;   // second operand block appended to K (e.g. the conditioning row)
    int lda2 = 0, Cin2 = 0;
    int w2_slab0 = 0;            // first weight slab of the A2 block
    int wslabs_total = 0;        // K slabs per N-block in the packed weight (set by the launcher when ntaps == 0)
    // ---- epilogue variants
    int epi = PK_EPI_STD;
    // PK_EPI_GATE: columns are packed so that the two N-subtiles of a wave hold (content, gate) of the
    //   same channel; stores tanh(content + b) * sigmoid(gate + b) to C[r, N/2 columns].
    // PK_EPI_GATE_PROJ (split-fp16 kernel only, N == 128, i.e. 64 gated channels): the gated tile z[128][64]
    //   never leaves the chip -- it is written to LDS and multiplied by the 64 x 128 matrix Wh2 (+ bias2) in
    //   the same launch; the STD epilogue (res / nsplit / C2 / acc2 / rowvalid) then applies to that product.
    //   WaveFlow: conv + gate (:228-262) followed by the res|skip 1x1 projection (:263-266) as one kernel.
    const void* Wh2 = nullptr;   // pk_gemm_pack_h3 of the [64][128] projection
    const float* bias2 = nullptr;
    // nsplit > 0: columns >= nsplit go to C2[r, n - nsplit] (+= if acc2) instead of C, without `res`.
    int nsplit = 0;
    float* C2 = nullptr;
    int ldc2 = 0;
    int acc2 = 0;
    // ---- block scaling of the split-fp16 kernel (pk_split.h).  a_amax[r] = max|A[r, 0..Cin)| indexed like the rows
    // of A (negative / beyond-M indices as far as the taps reach), a2_amax[r] likewise for A2.  NULL: the launcher
    // computes them with k_row_amax into the context's scratch (one extra read of A).  Producers that know their
    // row maxima (k_layernorm) pass them and save that pass.  tap_row[] is set by the launcher (tap_off / lda).
    const float* a_amax = nullptr;
    const float* a2_amax = nullptr;
    int tap_row[PK_GEMM_MAX_TAPS] = {0};
};

// max|A[r, 0..C)| for r0 <= r < r1 into amax[r] (amax indexed like the rows of A); on ctx->stream
int pk_row_amax_launch(pk_ctx* ctx, const float* A, long lda, int C, long r0, long r1, float* amax);

// Pack a [K][N] row-major matrix (K = taps*Cin, multiple of 16) into per-(N tile,
// K slab) LDS images.  Returns floats written: ceil(N/128) * (K/16) * 2048.
size_t pk_gemm_pack(const float* Wkn, int K, int N, std::vector<float>& out);
// Split-fp16 fragments for k_gemm_h3: K multiple of 32.  Returns halves written.
size_t pk_gemm_pack_h3(const float* Wkn, int K, int N, std::vector<uint16_t>& out);
// Conv1D weight [Cout][Cin][k] (paddle layout) -> [K = tap*Cin + ci][N = Cout] row-major.
void pk_conv_to_kn(const float* w, int Cout, int Cin, int k, std::vector<float>& out);

// Column permutation for PK_EPI_GATE: in[k][n] with n = half*Cz + c (half 0 = content, 1 = gate) ->
// out[k][nblk*128 + wn*64 + half*32 + j] with c = nblk*64 + wn*32 + j.  Cz must be a multiple of 64.
void pk_gemm_gate_permute(const float* Wkn, int K, int Cz, std::vector<float>& out);
void pk_gemm_gate_permute_bias(const float* b, int Cz, std::vector<float>& out);

int pk_gemm_launch(pk_ctx* ctx, const char* prof_name, const pk_gemm_args& a);
