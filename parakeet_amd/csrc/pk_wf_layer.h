// pk_wf_layer.h -- the fused WaveFlow residual-layer kernel (wf_layer.hip) and the kernels around it that work on its
// feature storage.  Used by waveflow.hip for the 64- and 128-channel models under the split-fp16 math.
//
// Feature storage ("planes"): the layer inputs are MFMA B operands of nine conv taps each, i.e. every stored value is
// split into its fp16 (hi, lo) parts nine times if it is stored as fp32 (round 2: 12.5 k of a wave tile's 30 k cycles
// were VALU, two thirds of them operand splits).  They are therefore stored ALREADY SPLIT, by the kernel that produces
// them: per 32-position block one power-of-two scale 2^k (k = blk_scale_exp of the block's max|.|, pk_split.h, kept as
// the max's fp32 bits in a side array) and per value the pair hi = fp16_rne(x 2^k), lo = fp16_rne(x 2^k - hi) -- the
// same 4 bytes as the fp32 value, the same 22 significant bits the split-fp16 products used before.  A consumer brings
// the blocks its taps touch to one common scale with a power-of-two multiply of the packed halves (v_pk_mul_f16) and
// feeds them to the MFMA as they are.  Layout of a block of CH channels (bytes):
//   [octet = CH / 8][position 32][plane hi | lo][8 halves]      = CH * 128 bytes = CH * 32 "slots" of 4 bytes
// octet o = 2 * kq + hh holds the 8 channels that lane half hh supplies to k-step kq (16 channels) of a contraction:
//   channel(kq, hh, e) = 16 kq + 8 (e >> 2) + 4 hh + (e & 3),   e = 0..7
// which is also the set of channels lane half hh owns as accumulator rows (mfma_row) -- so a lane's B-operand vector
// of the centre tap is its residual input, and its 8 accumulator registers of a k-step are one stored vector.
// A lane's operand of one k-step is two 16-byte loads (hi, lo); across the 32 lanes of a half wave they are 1 KB
// contiguous.
// The skip path is folded: a flow's parameters are output_proj(sum_l skip_l) with skip_l = W2skip_l z_l + b_l, both linear,
// so each layer accumulates (W_out W2skip_l) z_l -- TWO numbers per position -- into prm[pos][2] instead of its C skip
// channels into a [pos][C] buffer (C = 64: 16 instead of 512 bytes of read-modify-write per position and layer, and the
// skip half of the out projection, 24 of a tile's 552 MFMAs, becomes 2 C FMAs per lane); the biases W_out b_l are
// constants and go into the flow's output_proj bias.
#pragma once
#include <cstdint>
#include <vector>

#include "pk_common.h"

constexpr int WFL_MP = 96;         // condition channels padded to a multiple of 16 (n_mels 80 -> 96)
constexpr int WFL_BLK = 32;        // positions per block
constexpr int WFL_KS_COND = WFL_MP / 16;   // 6 k-steps of the condition block

static inline bool wfl_supports(int C) { return C == 64 || C == 128; }
// The one-launch-per-row variant (option "persistent": slower than eight launches, never validated beyond small sizes) is a
// measurement configuration: the product refuses it, the profile build runs it only under PK_WF_MEASURE=1 -- the profile
// build's DEFAULT kernel choice is the product's.
bool wfl_measurement_configs_allowed();
// channel of element e of lane half hh in k-step kq (see above)
__host__ __device__ static inline int wfl_chan(int kq, int hh, int e) { return 16 * kq + 8 * (e >> 2) + 4 * hh + (e & 3); }
// byte offset of (position p, channel ch, plane) from the buffer base (position 0) in a planes buffer of CH channels
static inline long wfl_plane_off(long p, int ch, int plane, int CH) {
    const int kq = ch >> 4, w = ch & 15, hh = (w >> 2) & 1, e = 4 * (w >> 3) + (w & 3);
    return (p >> 5) * ((long)CH * 128) + (long)(2 * kq + hh) * 1024 + (p & 31) * 32 + plane * 16 + e * 2;
}

struct WflWeights {          // one residual layer, as packed by wfl_pack()
    const uint16_t* w1;      // [KS1][part 2][co-tile 2C/32][lane 64][8]: conv taps (kr*3 + kc) x C/16 k-steps, then the condition block,
                             // whose channel n_mels (a constant 1 in the planes) carries conv bias + condition_proj bias
    const uint16_t* w2;      // [k2 C/16][part 2][tile C/32][lane 64][8]: the res half of out_proj
    const float* b2r;        // [C] res half of the out_proj bias * 2^(14 + k2res)
    const float* wso;        // [2C] (W_out W2skip) * 2^-14 in the order a lane reads it: [hh 2][k2 C/16][e 8][logs | b]
    int k1, k2res;           // block-scale exponents of the two weight tensors (pk_split.h)
};

// Pack one layer.  conv [2C][C][3][3], conv bias [2C], condition_proj [2C][n_mels], its bias [2C], out_proj [2C][C],
// its bias [2C] (paddle layouts, weight norm folded).  Appends to w16 / f32 and returns the offsets.
struct WflPacked {
    size_t w1, w2;           // offsets (halves) into w16
    size_t b2r, wso;         // offsets (floats) into f32
    int k1, k2res;
    double cso[2];           // W_out . (skip half of the out_proj bias): this layer's constant share of (logs, b)
};
// w_out: the flow's output_proj weight [2][C] (logs row, b row)
WflPacked wfl_pack(int C, const float* conv, const float* conv_b, const float* cond, const float* cond_b, int n_mels,
                   const float* outp, const float* outp_b, const float* w_out, std::vector<uint16_t>& w16,
                   std::vector<float>& f32);

constexpr int WFL_MAX_LAYERS = 8;   // layers one launch can run (a flow of the released models has 8)

struct WflLayer {            // what differs between the residual layers of one row
    WflWeights w;
    const float* in0;        // layer input ring, slot 0 (planes, C channels; one float = one 4-byte slot); slot s at + s * slot_stride
    const unsigned* in_amax0;   // max|.| per 32-position block of slot 0 (fp32 bits); slot s at + s * amax_stride
    float* out;              // next layer's input, slot of the current row (planes), or NULL (last layer)
    unsigned* out_amax;
    int first;               // the flow's first layer: prm is written, not accumulated
    int dil;                 // width dilation 2^l: tap t is shifted by tap_col[t] * dil positions
};

// One launch = the layers [0, nl) of ONE row (nl == 1: a single layer, what rounds 2 / 3 launched 960 times per batch; nl == the
// flow's 8: the whole ResidualNet of the row behind grid barriers, pk_grid.h -- SURVEY K20: 120 launches per batch).
struct WflLaunch {
    int C;                   // 64 or 128
    int f16;                 // 1: fp16 operands, one MFMA per product (the reference's AMP precision); 0: three-term split
    long slot_stride;        // slots (= floats)
    long amax_stride;
    int cur_slot;            // slot of the current row (residual input = centre tap of the last kernel row)
    float* prm;              // [pos][2] running (logs, b) of this row without the constant terms, written (first) or accumulated
    const float* cond;       // condition row (planes, 96 channels)
    const unsigned* cond_amax;
    int ntap;                // conv taps whose input row exists (3, 6 or 9)
    int tap_slot[9], tap_col[9], tap_w[9];   // ring slot, kernel column - 1 (-1, 0, 1), weight tap index kr*3 + kc
    const int* pos_utt;      // [npos_alloc] utterance of a position, < 0: gap (outputs forced to 0)
    int npos_alloc;          // multiple of 32
    unsigned long long* trace;  // profiling only (PK_WF_ABLATE=16): s_memtime stamps of workgroup 5, [wave 8][round 2][24]
    int seq;                    // profiling only (PK_WF_ABLATE=128): the launch's number within the inference, for the verifier's records
    int waves;                  // 0 = the launcher chooses 8- or 12-wave workgroups (64 channels), 8 / 12 = forced
    int active, tiles_per_wg;   // set by wfl_layer_launch: most waves that take a tile per round, tiles per workgroup
    int nl;                     // layers in this launch
    unsigned* bar;              // nl > 1: the grid barrier's counter (zeroed by the launcher) ...
    int* err;                   // ... and its time-out flag
    // nl == the flow's layer count and step_* set: after the last layer the launch also finishes the row -- Flow._predict_row_
    // parameters + _inverse_transform_row + input_proj of the NEXT row's layer-0 input (what wfl_step_launch does as a kernel
    // of its own): x[i] = (z'[i] - b) exp(-logs), h0 = input_proj(x[i]) as planes
    const float* step_z;        // z' row of this row, or NULL: no fused step
    float* step_x;              // x row out
    const float* step_w_in;     // input_proj weight / bias [C]
    const float* step_b_in;
    float* step_h0;             // next row's layer-0 ring slot (planes) or NULL (last row of the flow)
    unsigned* step_h0_amax;
    float step_b_logs, step_b_b;   // the flow's folded biases
    const WflLayer* layers;     // DEVICE memory: the nl layer descriptors of this launch (not a by-value array: indexing kernel
                                // arguments with the layer counter makes the compiler hold every element in scalar registers)
    WflLayer l0;                // nl == 1: the layer's descriptor by value as well -- the one-layer kernels read it from the kernel
                                // argument segment, whose loads the compiler re-issues at will instead of holding (or spilling)
                                // their results across the slab loop (filled by wfl_layer_launch from layers[0]'s host image)
};
int wfl_layer_launch(pk_ctx* ctx, const WflLaunch& a);

// The folded condition rows, written by the upsampler as fp32 [rows][pos / 32][96][32], become planes IN PLACE (one wave
// per block: read the block, take its maximum, write it back split) with amax [rows][pos / 32].
// Channel n_mels of the planes is set to the constant 1: its weight column carries the layers' biases (wfl_pack).
int wfl_cond_planes_launch(pk_ctx* ctx, float* cond, long row_stride, int rows, int nblk, long amax_row_stride, unsigned* amax,
                           int n_mels);
// Flow._predict_row_parameters + _inverse_transform_row + input_proj of the new row: (logs, b) = prm[pos] + the folded
// biases, x[i] = (z'[i] - b) * exp(-logs), then h0 = input_proj(x[i]) into layer 0's ring (planes) with its block maxima
int wfl_step_launch(pk_ctx* ctx, int C, const float* prm, float b_logs, float b_b, const float* z_row,
                    float* x_row, const float* w_in, const float* b_in, float* h0_next, unsigned* h0_amax,
                    const int* pos_utt, int npos_alloc, int first);
