// pk_wf_layer.h -- the fused WaveFlow residual-layer kernel (wf_layer.hip) and the kernels around it that work
// on its blocked feature layout.  Used by waveflow.hip for the 64-channel model under the split-fp16 math.
#pragma once
#include <cstdint>
#include <vector>

#include "pk_common.h"

constexpr int WFL_C = 64;          // residual channels the kernel is built for
constexpr int WFL_MP = 96;         // condition channels padded to a multiple of 16 (n_mels 80 -> 96)
constexpr int WFL_BLK = 32;        // positions per block of the blocked layout
// Blocked feature layout: a [positions][CH] tensor is stored as [pos / 32][CH][32] floats, i.e.
//   addr(p, ch) = (p >> 5) * (CH * 32) + ch * 32 + (p & 31)
// so that the 32 positions of a wave tile are contiguous per channel (coalesced MFMA B-operand loads and
// epilogue stores).  Buffers keep their margins: the base pointer is margin_positions * CH floats into the buffer.
static inline long wfl_off(long p, int ch, int CH) { return (p >> 5) * ((long)CH * 32) + (long)ch * 32 + (p & 31); }

constexpr int WFL_KS_TAP = WFL_C / 16;                 // 4 k-steps of 16 channels per conv tap
constexpr int WFL_KS_COND = WFL_MP / 16;               // 6
constexpr int WFL_KS1 = 9 * WFL_KS_TAP + WFL_KS_COND;  // 42 k-steps of the first contraction
constexpr int WFL_KS2 = WFL_C / 16;                    // 4 k-steps of the out projection
constexpr size_t WFL_KSTEP_HALVES = 2 * 4 * 64 * 8;    // one k-step of A fragments: [part 2][co-tile 4][lane 64][8]
constexpr size_t WFL_W1_HALVES = WFL_KS1 * WFL_KSTEP_HALVES;   // 172 032 halves = 336 KB
constexpr size_t WFL_W2_HALVES = WFL_KS2 * WFL_KSTEP_HALVES;   //  16 384 halves =  32 KB

struct WflWeights {          // one residual layer, as packed by wfl_pack()
    const uint16_t* w1;      // [WFL_KS1][part][co-tile][lane][8]: conv taps (kr*3 + kc) x 4 k-steps, then the condition block
    const uint16_t* w2;      // [WFL_KS2][part][out-tile][lane][8]: res (tiles 0, 1) | skip (tiles 2, 3)
    const float* b1;         // [128] conv bias + condition_proj bias: content 0..63, gate 64..127
    const float* b2s;        // [128] out_proj bias * 2^(14 + k2): res 0..63, skip 64..127
    int k1, k2res, k2skip;   // block-scale exponents of the three weight tensors (pk_split.h)
};

// Pack one layer.  conv [2C][C][3][3], conv bias [2C], condition_proj [2C][n_mels], its bias [2C], out_proj [2C][C],
// its bias [2C] (paddle layouts, weight norm folded).  Appends to w16 / f32 and returns the offsets.
struct WflPacked {
    size_t w1, w2;           // offsets (halves) into w16
    size_t b1, b2s;          // offsets (floats) into f32
    int k1, k2res, k2skip;
};
WflPacked wfl_pack(const float* conv, const float* conv_b, const float* cond, const float* cond_b, int n_mels,
                   const float* outp, const float* outp_b, std::vector<uint16_t>& w16, std::vector<float>& f32);

struct WflLaunch {
    WflWeights w;
    const float* in0;        // layer input ring, slot 0 (blocked [pos/32][64][32]); slot s at in0 + s * slot_stride
    long slot_stride;        // floats
    const unsigned* in_amax0;   // max|.| per 32-position block of slot 0; slot s at + s * amax_stride
    long amax_stride;
    int cur_slot;            // slot of the current row (residual input = centre tap of the last kernel row)
    float* out;              // next layer's input, slot of the current row, or NULL (last layer)
    unsigned* out_amax;
    float* skip;             // running skip sum (blocked), written (first) or accumulated
    int first;
    const float* cond;       // condition row (blocked [pos/32][96][32])
    const unsigned* cond_amax;
    int ntap;                // conv taps whose input row exists (3, 6 or 9)
    int tap_slot[9], tap_shift[9], tap_w[9];   // ring slot, position shift, weight tap index kr*3 + kc
    const int* pos_utt;      // [npos_alloc] utterance of a position, < 0: gap (outputs forced to 0)
    int npos_alloc;          // multiple of 32
    int active, tiles_per_wg;   // set by wfl_layer_launch: most waves that take a tile per round, tiles per workgroup
    int warm;                   // measurement switches: bit 0 touch the weight lines up front, bit 1 even rounds
};
int wfl_layer_launch(pk_ctx* ctx, const WflLaunch& a);

// max|cond| per 32-position block of every folded row: cond [rows][pos/32][96][32] -> amax [rows][pos/32]
int wfl_cond_amax_launch(pk_ctx* ctx, const float* cond, long row_stride, int rows, int nblk, long amax_row_stride,
                         unsigned* amax);
// k_wf_step on the blocked layout: params = output_proj(skip sum), x[i] = (z'[i] - b) * exp(-logs), then
// h0 = input_proj(x[i]) into layer 0's ring (blocked) with its block maxima
int wfl_step_launch(pk_ctx* ctx, const float* skip, const float* w_out, float b_logs, float b_b, const float* z_row,
                    float* x_row, const float* w_in, const float* b_in, float* h0_next, unsigned* h0_amax,
                    const int* pos_utt, int npos_alloc, int first);
