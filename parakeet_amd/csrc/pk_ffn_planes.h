// pk_ffn_planes.h -- the position-wise conv layers of an FFT block (MultiLayeredConv1d: Conv1D(k) -> ReLU -> Conv1D(k),
// parakeet/modules/fastspeech2_transformer/multi_layer_conv.py:19-62) on the structure of the WaveFlow layer kernel
// (pk_wf_layer.h): activations stored as pre-split fp16 planes with one power-of-two scale per row, rows as the
// MFMA N dimension (a wave owns 32 rows and 128 or 256 output channels), weights as the A operand streamed through three
// 48 KB LDS slabs.  ffn_planes.hip.
//
// Planes of a row timeline with CH channels: [block = row / 32][octet = CH / 8][plane hi | lo][row 32][8 halves] -- the octets
// and channel order of pk_wf_layer.h with "position" = timeline row, the two planes of an octet apart (a half wave's operand
// load is 512 contiguous bytes); one block of margin in front of block 0 and behind the last one (the +-1 taps of the edge
// tiles).  Row maxima (the scale of a row is blk_scale_exp of its maximum, pk_split.h): fp32 bits per row, one element of
// margin on either side, kept zero.
#pragma once
#include <cstdint>
#include <vector>

#include "pk_common.h"

constexpr int FFNP_BLK = 32;    // rows per block
constexpr int FFNP_TAPS = 3;    // the only kernel size built
constexpr int FFNP_NQ1 = 8;     // first conv: a wave owns 8 x 32 output channels (long timelines) ...
constexpr int FFNP_NQ2 = 4;     // ... or 4 x 32 (short ones); second conv: 4 x 32
constexpr int FFNP_NQL = 6;     // Linear layer on planes (the fused q | k | v projection, N = 3 adim = 6 x 192): 6 x 32
constexpr int FFNP_NQ1_MIN_BLOCKS = 256;   // timelines from this many blocks on run the first conv with FFNP_NQ1 tiles per wave
constexpr int FFNP_MIN_BLOCKS = 0;         // timelines shorter than this stay on the tile GEMM.  0: the path does not depend on the
                                           // timeline's length, i.e. an utterance's result does not depend on its batch (bit for
                                           // bit: tests/test_fullsize_gpu.py); the price is the latency of short timelines (a wave
                                           // tile runs the whole k loop: the second conv's 288 k-steps take 80 us however few rows)

// The shapes the kernels are instantiated for (the k loop is unrolled completely): FastSpeech2's adim 384 / units 1536 with
// kernel size 3 (every released FastSpeech2 configuration); anything else stays on the tile GEMM.
static inline bool ffnp_supports(int A, int units, int taps1, int taps2) {
    return A == 384 && units == 1536 && taps1 == FFNP_TAPS && taps2 == FFNP_TAPS;
}
static inline size_t ffnp_plane_bytes(int nblk, int CH) { return (size_t)(nblk + 2) * CH * 128; }

// Pack kn [taps * Cin][N] (tap-major rows, gemm.hip pk_conv_to_kn) for column tiles of 32 nq channels:
//   [column tile][k-step = kq * taps + tap][part hi | lo][q nq][lane 64][8]
// each group of 32 output channels scaled by its own power of two 2^kw (pk_split.h); wscale [N / 32] = 2^-kw.  Appended to
// w16 at a 16-byte boundary; returns the offset in halves.
size_t ffnp_pack(const float* kn, int Cin, int N, int nq, std::vector<uint16_t>& w16, std::vector<float>& wscale,
                 int taps = FFNP_TAPS);
struct FfnpConv {
    const uint16_t* w;     // packed weights: first conv for FFNP_NQ1 tiles per wave, second conv for FFNP_NQ2
    const uint16_t* w4;    // first conv: the same weights packed for FFNP_NQ2 tiles per wave (short timelines), or NULL
    const uint16_t* w1;    // either conv: the same weights packed for ONE tile per wave (timelines of an utterance or two: a wave's k
                           // loop is a serial chain of TAPS * Cin / 16 steps x 3 NQ matrix instructions -- 78 us for the second conv at
                           // 4 tiles per wave however few rows there are -- and with 32 columns per wave four times as many CUs share it), or NULL
    const float* bias;     // [N] or NULL
    const float* wscale;   // [N / 32] 2^-kw of the packed weights' 32-channel groups
    int Cin, N;
    const void* in;        // input planes, block 0 (ffnp_linear_launch with ldin != 0: a row-major fp32 matrix)
    int ldin;              // 0, or floats per row of the fp32 input
    const unsigned* in_amax;   // row maxima of the input (fp32 bits), element 0 = row 0
    int nblk;              // blocks of the timeline (rows_alloc / 32)
    const int* row_utt;    // [32 nblk], < 0: gap row
    // first conv (out != NULL): relu(conv + bias) -> planes of N channels, each row scaled by the bound c1 max(in_amax of the
    // three rows it reads) + c0, which is also stored as the row's maximum; gap rows -> 0
    void* out;
    unsigned* out_amax;
    float c1, c0;
    // second conv (out == NULL): x[row][:] += conv + bias (fp32 row-major, ldx floats per row)
    float* x;
    int ldx;
    int one_max = 4096;    // one 32-column tile per wave while blocks x N / 32 <= this (the owner's "ffn_one_tile_max" option; needs w1)
    int variant = 0;       // tiling override (the owner's "ffnp_variant" option): first digit 8 / 4 = 256 / 128 columns per wave in
                           // the first conv, second digit = waves per workgroup of the second conv; 0 = by shape
};
int ffnp_conv_launch(pk_ctx* ctx, const char* prof_name, const FfnpConv& a);
// Linear layer (one tap).  ldin == 0: planes in, weights packed for FFNP_NQL tiles per wave, x[row][:] = in . W + bias.
// ldin != 0: fp32 rows in (in_amax = a magnitude bound per row, gap rows of `in` are not read), weights packed for FFNP_NQ2
// tiles per wave, x[row][:] += in . W + bias.
int ffnp_linear_launch(pk_ctx* ctx, const char* prof_name, const FfnpConv& a);

// The 256-channel convs of the variance predictors and of the postnet (round 4): 384 | 256 -> 256 channels, k = 3 | 5.
// act 1 = tanh, 2 = ReLU.  c.out != NULL: output as planes (tanh only; c1 = 0, c0 = 1), else fp32 rows c.x[row][ldx] (written;
// gap rows zero).  in_amax: taps / 2 zero elements of margin on either side.
bool ffnp_conv256_supports(int Cin, int N, int taps);
int ffnp_conv256_launch(pk_ctx* ctx, const char* prof_name, const FfnpConv& a, int taps, int act);

// LayerNorm (eps) of rows of C channels -> planes + row maxima; gap rows -> 0.  g == NULL: no normalisation (rows -> planes)
int ffnp_layernorm_launch(pk_ctx* ctx, const float* x, const float* g, const float* b, const int* row_utt, int nblk, int C,
                          float eps, void* out, unsigned* out_amax);
