// pk_gemm.h -- fp32 MFMA implicit-conv GEMM used by every dense layer of the
// FastSpeech2 path (Linear, Conv1D k=1/3/5 in channels-last layout).
#pragma once
#include <cstdint>
#include <vector>

#include "pk_common.h"

constexpr int PK_GEMM_BM = 128;
constexpr int PK_GEMM_BN = 128;
constexpr int PK_GEMM_BK = 16;

enum { PK_ACT_NONE = 0, PK_ACT_RELU = 1, PK_ACT_TANH = 2 };

// C[r, n] = epilogue( sum_{tap, ci} A[r + tap - pad, ci] * W[tap*Cin + ci, n] )
//   epilogue: v += bias[n]; v = act(v); v += res[r, n]; v = rowvalid[r] ? v : 0;
//             v = v * cscale[n] + cshift[n]; store to C[out_rowmap ? out_rowmap[r] : r, n]
// A is row-major with leading dimension lda; rows r + tap - pad must be readable
// for every r in [0, ceil(M/128)*128) (buffers carry a margin).  Rows of the
// "row timeline" that belong to no utterance (gaps) are forced to zero via
// rowvalid so that the next k>1 convolution sees the reference's zero padding.
struct pk_gemm_args {
    const float* A = nullptr;
    int lda = 0;
    const float* Wp = nullptr;   // packed by pk_gemm_pack()
    const float* bias = nullptr;
    const float* res = nullptr;
    int ldr = 0;
    float* C = nullptr;
    int ldc = 0;
    const int* rowvalid = nullptr;   // >= 0 means valid (utterance id), < 0 gap
    const float* cscale = nullptr;
    const float* cshift = nullptr;
    const int* out_rowmap = nullptr;  // < 0: row not stored
    int M = 0, N = 0, Cin = 0, taps = 1, pad = 0;
    int act = PK_ACT_NONE;
};

// Pack a [K][N] row-major matrix (K = taps*Cin, multiple of 16) into per-(N tile,
// K slab) LDS images.  Returns floats written: ceil(N/128) * (K/16) * 2048.
size_t pk_gemm_pack(const float* Wkn, int K, int N, std::vector<float>& out);
// Conv1D weight [Cout][Cin][k] (paddle layout) -> [K = tap*Cin + ci][N = Cout] row-major.
void pk_conv_to_kn(const float* w, int Cout, int Cin, int k, std::vector<float>& out);

int pk_gemm_launch(pk_ctx* ctx, const char* prof_name, const pk_gemm_args& a);
