// ffn_planes.hip -- the two Conv1D(k = 3) layers of an FFT block's position-wise feed-forward (MultiLayeredConv1d,
// parakeet/modules/fastspeech2_transformer/multi_layer_conv.py:19-62: w_2(dropout(relu(w_1(x)))), both convs over time with
// zero padding at the utterance edges) and the LayerNorm in front of them (encoder_layer.py:108-112: residual = x; x =
// norm2(x); x = residual + feed_forward(x)) on pre-split fp16 planes (pk_ffn_planes.h).
//
// Why: the tile GEMM (gemm.hip k_gemm_h3) runs these two layers -- 72 % of FastSpeech2's flops -- at 26-32 % matrix-pipe
// utilisation: every activation element is split into its fp16 (hi, lo) parts once per column tile that reads it by the
// threads that stage it, both operands travel through LDS, and a 64 x 64 wave tile reads 2 LDS bytes per MFMA byte.  The
// WaveFlow layer kernel (wf_layer.hip) has the structure that avoids all three, and these layers fit it: rows of the
// timeline are the MFMA N dimension, a wave owns 32 rows x (256 | 128) output channels, its B operand (the activations of
// its own rows, shifted by the tap) comes straight from global memory as two 16-byte loads per k-step -- already split by
// the kernel that produced it --, only the weights go through LDS (three 48 KB slabs, requested two slabs ahead), and the
// k loop is unrolled completely so that the operand ring is registers.
//
//   k_ffn_ln_planes          LayerNorm of 32 rows -> planes + every row's maximum                                      (norm2)
//   k_ffn_planes<NQ, 24, 0>  relu(conv(planes) + b) -> planes of the hidden activations, scaled by a magnitude BOUND  (w_1)
//   k_ffn_planes<4, 96, 1>   x += conv(hidden planes) + b on the fp32 residual stream                                 (w_2)
// The hidden activations' scale cannot be their maximum (a row's 1536 channels are produced by several workgroups): it is
// the bound c1 max|norm2 output| + c0 of pk_fft_dense (pk_fft.h), as on the tile-GEMM path.
//
// k order: k-step = kq * 3 + tap (the three taps of 16 input channels are consecutive: their loads hit the same lines one
// row apart).  Scales are per ROW (its maximum as fp32 bits in a side array): a lane's three taps are three rows, brought
// to the largest of their three scales by a per-lane power-of-two multiply; the lane's accumulators -- all of them belong to
// its own row -- carry that scale to the epilogue.  Nothing a row's result depends on lies outside its utterance, so results
// do not depend on the batch an utterance is in.
//
// Measured (MI355X, 32 utterances: encoder 4 224 rows, decoder 20 544 rows; per launch, rocprofv3; DESIGN.md 4.3,
// profiles/r03_fs2_planes_timings.txt): decoder w_1 320 -> 190 us, w_2 244 -> 195 us, q|k|v 108 -> 64 us, attention-out 76 ->
// 35 us; encoder w_1 103 -> 52 us, w_2 48 -> 81 us; norm + bounds 24 -> 19 us.  The matrix pipe is 49-56 % busy in the
// decoder launches (tile GEMM: 29-35 %); a third of the wave cycles wait for global loads.
// Tried and measured without effect on that (variants since removed): 4-wave workgroups two per CU for w_2, 96 / 192
// columns per wave with each XCD working on its own column tiles (weights L2-resident: 10-25 % slower, every XCD then
// reads all activations), the weight slabs' LDS writes spread over the k-steps instead of bunched at the slab end, A
// fragments three column tiles ahead, non-temporal operand loads (10 % slower).  With the weight slabs neither loaded nor
// written (PK_FFNP_ABLATE=8 with PK_FFNP_VARIANT=84) w_2 takes 139 us of 205, loaded but not written 163, written but not
// loaded 171, with no operand traffic either 139: what is left of the gap to the matrix-pipe time (95 us at the 2.2 GHz
// the counters show) is LDS reads, barriers and issue.
#include "pk_ffn_planes.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "pk_split.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 pkh2 __attribute__((ext_vector_type(2)));

namespace {
// an octet's 1 KB of a block: [plane hi | lo][row 32][8 halves] -- a half wave's operand load of one k-step is 512 contiguous
// bytes per plane (pk_wf_layer.h interleaves the planes per row: [row][hi | lo][8], 16 of every 32 bytes per load)
constexpr int ROW_B = 16, LO_OFF = 512;
// (a weight slab is CPT 16-byte chunks per thread -- template parameter of the kernel, 6 by default: 6 KB per wave of the
// workgroup, 48 KB for 8 waves)

struct Args {
    FfnpConv c;
    int active;        // waves of a workgroup that take a tile (the rest only move weights)
    int nrg;           // row groups = ceil(nblk / active)
    int nct;           // column tiles
    long in_blk, out_blk;   // bytes per block of the input / output planes
};

__host__ __device__ inline int mfma_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }
__device__ __forceinline__ f32x16 mfma16(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// in-register split of 8 values (wf_layer.hip): hi = v_cvt_pkrtz (round toward zero), x - hi exactly by v_fma_mix_f32, lo = fp16_rne(x - hi)
__device__ __forceinline__ void split8(const float (&v)[8], f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const pkh2 h = __builtin_amdgcn_cvt_pkrtz(v[2 * p], v[2 * p + 1]);
        const unsigned hu = __builtin_bit_cast(unsigned, h);
        float l0, l1;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hu), "v"(v[2 * p]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hu), "v"(v[2 * p + 1]));
        hi[2 * p] = (_Float16)h[0];
        hi[2 * p + 1] = (_Float16)h[1];
        lo[2 * p] = (_Float16)l0;
        lo[2 * p + 1] = (_Float16)l1;
    }
}
// the stored pair of an activation: hi = fp16_rne(s x), lo = fp16_rne(s x - hi) (wf_layer.hip)
__device__ __forceinline__ void store_pair8(const float (&v)[8], float s, f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float t = v[e] * s;
        const _Float16 h = (_Float16)t;
        hi[e] = h;
        lo[e] = (_Float16)(t - (float)h);
    }
}
__device__ __forceinline__ int amax_exp(unsigned bits) {
    const int e = (int)(bits >> 23);
    return e < PK_EXP_MIN ? PK_EXP_MIN : (e > PK_EXP_MAX ? PK_EXP_MAX : e);
}
// the fp16 value 2^-d twice in a register (d >= 0)
__device__ __forceinline__ unsigned pow2_neg_h2(int d) {
    const float f = __uint_as_float((unsigned)(127 - min(d, 60)) << 23);
    return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(f, f));
}
__device__ __forceinline__ f16x8 h8_of(unsigned u) {
    const u32x4 v = {u, u, u, u};
    return __builtin_bit_cast(f16x8, v);
}
__device__ __forceinline__ f16x8 ld_h8(const char* p) { return *reinterpret_cast<const f16x8*>(p); }
__device__ __forceinline__ void st_h8(char* p, f16x8 v) { *reinterpret_cast<f16x8*>(p) = v; }

// NQ accumulator tiles (32 output channels each) per wave, KQ = Cin / 16, EPI 0: bias + ReLU -> planes, 1: += into fp32 rows.
// Grid: one workgroup per (row group of `active` blocks, column tile); blockIdx -> (row group, column tile) keeps the column
// tiles of a row group on one XCD (block b runs on XCD b % 8: observed, used for speed only), where its activations are read
// from memory once.
// W waves per workgroup: 8 (one workgroup per CU) or 4 (24 KB slabs, two workgroups per CU: half the rows per workgroup --
// finer scheduling granularity, barriers among four waves, and the two workgroups of a CU cover each other's barriers,
// prologues and epilogues; the weights travel to LDS twice per CU).
// ABL (profiling only, PK_FFNP_ABLATE, results are wrong when set): 1 = the operand ring is not refilled after the prologue,
// 4 = no epilogue loads / stores, 8 = the weight slabs are not reloaded after the prologue (barriers stay), 128 = weight
// slabs loaded but not written to LDS, 256 = written (stale registers) but not loaded; sums combine
// BF32 (one tap only): the B operand is a row-major fp32 matrix (in, ldin floats per row) with a magnitude bound per row in
// in_amax; it is scaled and split in registers, once per column tile that reads it (the attention output, whose producer
// holds a channel x 16 queries per lane -- the transpose of a planes vector)
// TAPS = 1 (Linear), 3 or 5 (round 4: the k = 5 convs of the postnet and of the pitch predictor); CPT = 16-byte chunks of a
// weight slab per thread -- a slab must be a whole number of k-steps AND divide the k loop: 6 for the shapes of rounds 1 - 3,
// 5 where TAPS * KQ = 80 (256 channels, k = 5).  ACT: activation of EPI 0 (0 ReLU, 1 tanh) and of EPI 2 (0 none, 1 tanh,
// 2 ReLU; with an activation EPI 2 also zeroes gap rows: its output is the zero-padded input of another conv).
template <int NQ, int KQ, int EPI, int W, int TAPS = FFNP_TAPS, int ABL = 0, bool BF32 = false, int CPT = 6, int ACT = 0>
__global__ __launch_bounds__(64 * W, 2) void k_ffn_planes(Args a) {
    constexpr int THREADS = 64 * W;
    constexpr int HT = TAPS / 2;              // taps reach HT rows to either side
    constexpr int SLAB_CH = CPT * THREADS;    // 16-byte chunks per slab buffer
    constexpr int KCH = 2 * NQ * 64;          // chunks per k-step of the packed weights
    constexpr int SLAB = SLAB_CH / KCH;       // k-steps per slab: 3 / 6
    constexpr int nks = TAPS * KQ;            // TAPS = 3: conv over rows r - 1, r, r + 1; 1: a Linear layer
    constexpr int G = nks / SLAB;             // slabs
    constexpr int RING = NQ >= 6 ? 6 : 9;     // operand ring depth in k-steps (wf_layer.hip Shape::RING)
    constexpr bool TIGHT = NQ == 8;
    static_assert(SLAB_CH % KCH == 0 && nks % SLAB == 0 && nks > RING && G >= 3, "shape");
    static_assert(!BF32 || (TAPS == 1 && !TIGHT), "fp32 operand: one tap, the untight loop");
    __shared__ __attribute__((aligned(16))) f16x8 wbuf[3][SLAB_CH];
    __shared__ __attribute__((aligned(16))) float lb[32 * NQ];   // this column tile's bias in lane order [hh][q][r]
    __shared__ float lbs[NQ];                                    // 2^-kw of its NQ 32-column groups (ffnp_pack)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, hh = lane >> 5;
    // blockIdx -> (row group, column tile): the column tiles of a row group on one XCD
    const int xcd = (int)blockIdx.x & 7, k = (int)blockIdx.x >> 3;
    const int rg = (k / a.nct) * 8 + xcd, ct = k % a.nct;
    if (rg >= a.nrg) return;   // (the whole workgroup)
    for (int i = tid; i < 32 * NQ; i += THREADS) {
        const int h2 = i / (16 * NQ), q = (i / 16) % NQ, r = i % 16;
        lb[i] = a.c.bias ? a.c.bias[ct * (32 * NQ) + 32 * q + mfma_row(r, h2)] : 0.f;
    }
    if (tid < NQ) lbs[tid] = a.c.wscale[ct * NQ + tid];
    const f16x8* wt = reinterpret_cast<const f16x8*>(a.c.w) + (long)ct * ((long)G * SLAB_CH) + tid;
    const int blk = rg * a.active + wave;
    const bool tile_ok = wave < a.active && blk < a.c.nblk;
    f16x8 wreg[CPT];   // one slab of weights on its way from global memory to LDS

    if (tile_ok) {
        const int p = blk * FFNP_BLK + j;
        const int rv = a.c.row_utt[p];
        // the maxima of the three blocks the taps read (lanes 0..2; the others repeat lane 0's)
        // the maxima of the rows this lane's taps read (its own row's output depends on nothing else: every row keeps its own
        // scale through the whole kernel, so a result does not depend on which utterances share the batch)
        unsigned am[TAPS];   // (TAPS = 1: the row's own)
#pragma unroll
        for (int t = 0; t < TAPS; ++t) am[t] = a.c.in_amax[p + t - HT];
        // this lane's operand of tap t = its 8 channels (octet 2 kq + hh) of row p + t - 1: byte offset from block blk - 1
        const char* inb = reinterpret_cast<const char*>(a.c.in) + ((long)blk - 1) * a.in_blk;
        unsigned off[TAPS];
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            const int q = j + t - HT;   // -HT .. 31 + HT
            off[t] = (unsigned)((q + 32) >> 5) * (unsigned)a.in_blk + (unsigned)((q & 31) * ROW_B + hh * 1024);
        }
        f16x8 rhi[RING], rlo[RING];
        auto load_b = [&](int ks) {
            const int kq = ks / TAPS, tap = ks % TAPS, slot = ks % RING;
            if (BF32) {   // channels 16 kq + 4 hh + (0..3) and + 8: wfl_chan(kq, hh, 0..7)
                const float* src = reinterpret_cast<const float*>(a.c.in) + (long)p * a.c.ldin + 16 * kq + 4 * hh;
                rhi[slot] = ld_h8(reinterpret_cast<const char*>(src));
                rlo[slot] = ld_h8(reinterpret_cast<const char*>(src + 8));
                return;
            }
            const char* src = inb + (off[tap] + (unsigned)(kq * 2048));
            rhi[slot] = ld_h8(src);
            rlo[slot] = ld_h8(src + LO_OFF);
        };
        {   // slabs 0 and 1 of the weights and the first ring in ONE round trip
            f16x8 wreg1[CPT];
#pragma unroll
            for (int c = 0; c < CPT; ++c) wreg[c] = wt[c * THREADS];
#pragma unroll
            for (int c = 0; c < CPT; ++c) wreg1[c] = wt[SLAB_CH + c * THREADS];
#pragma unroll
            for (int kk = 0; kk < RING; ++kk) load_b(kk);
            __builtin_amdgcn_sched_barrier(0);   // everything above is requested before anything below waits
#pragma unroll
            for (int c = 0; c < CPT; ++c) wbuf[0][c * THREADS + tid] = wreg[c];
#pragma unroll
            for (int c = 0; c < CPT; ++c) wbuf[1][c * THREADS + tid] = wreg1[c];
        }
        // common scale of the tile: the largest of the three block maxima; per tap and lane the power of two that brings the
        // block the lane reads to it
        int ex = amax_exp(am[0]);
#pragma unroll
        for (int t = 1; t < TAPS; ++t) ex = max(ex, amax_exp(am[t]));
        const int kx = PK_BLK_TOP + 127 - ex;
        const float sx = pow2f(kx);   // (fp32 operand: applied before the split)
        unsigned fu[TAPS];
#pragma unroll
        for (int t = 0; t < TAPS; ++t) fu[t] = pow2_neg_h2(ex - amax_exp(am[t]));
        f32x16 acc[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
        __syncthreads();
#pragma unroll
        for (int g = 0; g < G; ++g) {
            // (wf_layer.hip: the weights of slab g + 2 are requested first -- every load below is younger --, in two halves
            // where the accumulators take 128 registers)
            const int NW = g + 2 >= G ? 0 : CPT, HW = TIGHT ? NW / 2 : NW;
#pragma unroll
            for (int c = 0; c < CPT; ++c)
                if (c < HW && !(ABL & (8 | 256))) wreg[c] = wt[(long)(g + 2) * SLAB_CH + c * THREADS];
            __builtin_amdgcn_sched_barrier(0);
            unsigned wo = (g % 3) * SLAB_CH + lane;   // the slab's LDS base as one opaque register (wf_layer.hip)
            asm volatile("" : "+v"(wo));
            const f16x8* wl = &wbuf[0][0] + wo;
#pragma unroll
            for (int kk = 0; kk < SLAB; ++kk) {
                const int ks = SLAB * g + kk, slot = ks % RING;
                const f16x8 f = h8_of(fu[ks % TAPS]);   // (one tap: 2^0, the row's own scale)
                f16x8 bh, bl;
                if (TIGHT) {
                    rhi[slot] *= f;
                    rlo[slot] *= f;
                } else {
                    if (BF32) {
                        const f32x4 v0 = __builtin_bit_cast(f32x4, rhi[slot]), v1 = __builtin_bit_cast(f32x4, rlo[slot]);
                        float v[8];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = rv >= 0 ? v0[e] * sx : 0.f;        // (gap rows of that buffer are never written)
                            v[4 + e] = rv >= 0 ? v1[e] * sx : 0.f;
                        }
                        split8(v, bh, bl);
                    } else {
                        bh = TAPS == 1 ? rhi[slot] : rhi[slot] * f;
                        bl = TAPS == 1 ? rlo[slot] : rlo[slot] * f;
                    }
                    __builtin_amdgcn_sched_barrier(0);   // the slot's old value is dead before its refill is requested
                    if (!(ABL & 1) && ks + RING < nks) load_b(ks + RING);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const f16x8 ah = wl[kk * KCH + (0 * NQ + q) * 64];
                    acc[q] = mfma16(ah, TIGHT ? rhi[slot] : bh, acc[q]);
                    const f16x8 al = wl[kk * KCH + (1 * NQ + q) * 64];
                    acc[q] = mfma16(al, TIGHT ? rhi[slot] : bh, acc[q]);
                    acc[q] = mfma16(ah, TIGHT ? rlo[slot] : bl, acc[q]);
                }
                constexpr int AHEAD = NQ == 8 ? 1 : 2;   // A fragments this many column tiles ahead of their MFMAs
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * AHEAD, 0);
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    if (q + AHEAD < NQ) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                }
                if (TIGHT) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (!(ABL & 1) && ks + RING < nks) load_b(ks + RING);
                    if (kk == 0 && NW > 0 && !(ABL & 8)) {
#pragma unroll
                        for (int c = 0; c < CPT; ++c)
                            if (c < HW) wbuf[(g + 2) % 3][c * THREADS + tid] = wreg[c];
#pragma unroll
                        for (int c = 0; c < CPT; ++c)
                            if (c < NW - HW) wreg[c] = wt[(long)(g + 2) * SLAB_CH + (HW + c) * THREADS];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int c = 0; c < CPT; ++c)
                if (c < (TIGHT ? NW - HW : NW) && !(ABL & 8)) {
                    if (ABL & 128) {   // the loads stay (their results are "used"), the LDS writes go
                        unsigned keep = __builtin_bit_cast(u32x4, wreg[c])[0];
                        asm volatile("" : "+v"(keep));
                    } else {
                        wbuf[(g + 2) % 3][((TIGHT ? HW : 0) + c) * THREADS + tid] = wreg[c];
                    }
                }
            __syncthreads();   // everyone is done reading this slab's buffer and sees the next two
        }
        const float pinv = pow2f(-kx);   // x the weights' 2^-kw of the accumulator tile (lbs[q])
        const f32x4* lb4 = reinterpret_cast<const f32x4*>(lb) + hh * (NQ * 4);
        if (EPI == 0) {
            // relu(. + b) -> the hidden planes, scaled by the bound of this lane's row (gap rows: 0, their bound too)
            float m3 = __uint_as_float(am[0]);
#pragma unroll
            for (int t = 1; t < TAPS; ++t) m3 = fmaxf(m3, __uint_as_float(am[t]));
            const float hb = fmaf(m3, a.c.c1, a.c.c0);   // (tanh: c1 = 0, c0 = 1 -- |tanh| <= 1 is its own bound)
            const float so = pow2f(blk_scale_exp(__float_as_uint(hb)));
            char* dst = reinterpret_cast<char*>(a.c.out) + (long)blk * a.out_blk + (long)(ct * NQ * 2) * 2048 + j * ROW_B + hh * 1024;
            const bool row_ok = rv >= 0;
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    float v[8];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const f32x4 b4 = lb4[q * 4 + 2 * m + i];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float pre = fmaf(acc[q][8 * m + 4 * i + e], pinv * lbs[q], b4[e]);
                            const float t = ACT == 1 ? tanhf(pre) : fmaxf(pre, 0.f);
                            v[4 * i + e] = row_ok ? t : 0.f;
                        }
                    }
                    f16x8 oh, ol;
                    store_pair8(v, so, oh, ol);
                    if ((ABL & 4) && oh[0] != (_Float16)12345.f) continue;   // (never equal: keeps the arithmetic)
                    st_h8(dst + (2 * q + m) * 2048, oh);
                    st_h8(dst + (2 * q + m) * 2048 + LO_OFF, ol);
                }
            if (ct == 0 && hh == 0) a.c.out_amax[p] = row_ok ? __float_as_uint(hb) : 0u;
        } else if (EPI == 2) {
            // y = . + b (fp32 row-major): lane (j, hh) holds row p, channels 32 q + 8 i + 4 hh + (0..3) of the column tile
            float* yr = a.c.x + (long)p * a.c.ldx + ct * (32 * NQ) + 4 * hh;
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 b4 = lb4[q * 4 + i];
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float pre = fmaf(acc[q][4 * i + e], pinv * lbs[q], b4[e]);
                        o[e] = ACT == 0 ? pre : (rv >= 0 ? (ACT == 1 ? tanhf(pre) : fmaxf(pre, 0.f)) : 0.f);
                    }
                    *reinterpret_cast<f32x4*>(yr + 32 * q + 8 * i) = o;
                }
        } else {
            // x += . + b: lane (j, hh) holds row p, channels 32 q + 8 i + 4 hh + (0..3) of the column tile
            float* xr = a.c.x + (long)p * a.c.ldx + ct * (32 * NQ) + 4 * hh;
            f32x4 old[NQ][4];
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) old[q][i] = (ABL & 4) ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(xr + 32 * q + 8 * i);
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 b4 = lb4[q * 4 + i];
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = old[q][i][e] + fmaf(acc[q][4 * i + e], pinv * lbs[q], b4[e]);
                    if ((ABL & 4) && o[0] != 12345.f) continue;
                    *reinterpret_cast<f32x4*>(xr + 32 * q + 8 * i) = o;
                }
        }
    } else {
        // a wave without a tile only moves weights and keeps the barriers
#pragma unroll
        for (int c = 0; c < CPT; ++c) wbuf[0][c * THREADS + tid] = wt[c * THREADS];
#pragma unroll
        for (int c = 0; c < CPT; ++c) wbuf[1][c * THREADS + tid] = wt[SLAB_CH + c * THREADS];
        __syncthreads();
#pragma unroll 1
        for (int g = 0; g < G; ++g) {
            if (g + 2 < G) {
#pragma unroll
                for (int c = 0; c < CPT; ++c) wreg[c] = wt[(long)(g + 2) * SLAB_CH + c * THREADS];
#pragma unroll
                for (int c = 0; c < CPT; ++c) wbuf[(g + 2) % 3][c * THREADS + tid] = wreg[c];
            }
            __syncthreads();
        }
    }
}

// LayerNorm -> planes: one workgroup of four waves per block, a wave per row (eight rows each), lane o < C / 8 holds the 8
// channels of octet o.  Every row is stored with the scale of its own maximum.
__global__ __launch_bounds__(256) void k_ffn_ln_planes(const float* __restrict__ x, const float* __restrict__ g,
                                                      const float* __restrict__ b, const int* __restrict__ row_utt, int C,
                                                      float eps, char* __restrict__ out, unsigned* __restrict__ out_amax) {
    const int blk = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int noct = C >> 3;
    const bool on = lane < noct;
    const int kq = lane >> 1, hh = lane & 1;
    const int c0 = 16 * kq + 4 * hh;   // channels c0..c0+3 and c0+8..c0+11 (wfl_chan)
    // g == NULL: no normalisation -- the rows as they are become planes with the scale of their own maximum (the input of a
    // conv chain that starts from fp32 rows: encoder output -> predictors, first postnet layer -> the k = 5 convs)
    const bool ident = g == nullptr;
    f32x4 ga = {0.f, 0.f, 0.f, 0.f}, gb = ga, ba = ga, bb = ga;
    if (on && !ident) {
        ga = *reinterpret_cast<const f32x4*>(g + c0);
        gb = *reinterpret_cast<const f32x4*>(g + c0 + 8);
        ba = *reinterpret_cast<const f32x4*>(b + c0);
        bb = *reinterpret_cast<const f32x4*>(b + c0 + 8);
    }
    const float rc = 1.f / (float)C;
    char* dst = out + (long)blk * ((long)C * 128) + lane * 1024 + (wave * 8) * ROW_B;
    // all eight rows' loads first (one round trip), then row by row
    f32x4 xa[8], xb[8];
    float rmax[8];
    bool ok[8];
    // (unconditional: a branch on a loaded value serialises the round trips; lanes beyond the row and gap rows read memory that
    // exists -- the activation buffers end in PK_FFT_LEAD rows of margin -- and are masked afterwards)
    int ru[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) ru[i] = row_utt[blk * FFNP_BLK + wave * 8 + i];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = blk * FFNP_BLK + wave * 8 + i;
        xa[i] = *reinterpret_cast<const f32x4*>(x + (long)r * C + c0);
        xb[i] = *reinterpret_cast<const f32x4*>(x + (long)r * C + c0 + 8);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        ok[i] = ru[i] >= 0;   // wave-uniform
        if (!(ok[i] && on)) xa[i] = xb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = blk * FFNP_BLK + wave * 8 + i;
        float s = (xa[i][0] + xa[i][1]) + (xa[i][2] + xa[i][3]) + ((xb[i][0] + xb[i][1]) + (xb[i][2] + xb[i][3]));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s * rc;
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (!ident) {
                xa[i][e] -= mean;
                xb[i][e] -= mean;
            }
            q += xa[i][e] * xa[i][e] + xb[i][e] * xb[i][e];
        }
        if (!on) q = 0.f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
        const float inv = 1.0f / sqrtf(q * rc + eps);
        float am = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (ident) {   // (uniform) the row itself, untouched
                xa[i][e] = ok[i] && on ? xa[i][e] : 0.f;
                xb[i][e] = ok[i] && on ? xb[i][e] : 0.f;
            } else {
                xa[i][e] = ok[i] && on ? xa[i][e] * inv * ga[e] + ba[e] : 0.f;
                xb[i][e] = ok[i] && on ? xb[i][e] * inv * gb[e] + bb[e] : 0.f;
            }
            am = fmaxf(am, fmaxf(fabsf(xa[i][e]), fabsf(xb[i][e])));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o));
        rmax[i] = am;
    }
    // the stores of a lane's eight rows together: 8 x 16 bytes of one plane are one 128-byte line
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (on) {
            const float v[8] = {xa[i][0], xa[i][1], xa[i][2], xa[i][3], xb[i][0], xb[i][1], xb[i][2], xb[i][3]};
            f16x8 oh, ol;
            store_pair8(v, pow2f(blk_scale_exp(__float_as_uint(rmax[i]))), oh, ol);
            st_h8(dst + i * ROW_B, oh);
            st_h8(dst + i * ROW_B + LO_OFF, ol);
        }
    }
    if (lane < 8) {
        float m = rmax[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) m = lane == i ? rmax[i] : m;
        out_amax[blk * FFNP_BLK + wave * 8 + lane] = __float_as_uint(m);
    }
}

inline uint16_t f32_to_f16_rne(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x47800000u) return (uint16_t)(sign | (x > 0x7f800000u ? 0x7e00u : 0x7c00u));
    if (x < 0x38800000u) {
        if (x < 0x33000000u) return (uint16_t)sign;
        const int shift = 113 - (int)(x >> 23);
        const uint32_t m = (x & 0x7fffffu) | 0x800000u;
        const uint32_t half = 1u << (shift + 12), mask = (half << 1) - 1;
        uint32_t r = m >> (shift + 13);
        const uint32_t rem = m & mask;
        if (rem > half || (rem == half && (r & 1u))) ++r;
        return (uint16_t)(sign | r);
    }
    uint32_t r = x - 0x38000000u;
    const uint32_t rem = r & 0x1fffu;
    r >>= 13;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) ++r;
    return (uint16_t)(sign | r);
}
inline float f16_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu, m = h & 0x3ffu, x;
    if (e == 0) {
        if (m == 0) x = sign;
        else {
            e = 113;
            while (!(m & 0x400u)) { m <<= 1; --e; }
            x = sign | (e << 23) | ((m & 0x3ffu) << 13);
        }
    } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
    else x = sign | ((e + 112) << 23) | (m << 13);
    float f;
    memcpy(&f, &x, 4);
    return f;
}
}  // namespace

size_t ffnp_pack(const float* kn, int Cin, int N, int nq, std::vector<uint16_t>& w16, std::vector<float>& wscale, int taps) {
    const int KQ = Cin / 16, nks = taps * KQ, nct = N / (32 * nq);
    // one exponent per 32 output channels -- whatever the tile width, so that every packing of a layer holds the same numbers
    std::vector<int> kw(N / 32);
    wscale.resize(N / 32);
    for (int g = 0; g < N / 32; ++g) {
        float m = 0.f;
        for (size_t k = 0; k < (size_t)taps * Cin; ++k)
            for (int c = 0; c < 32; ++c) {
                const float v = std::fabs(kn[k * N + 32 * g + c]);
                if (std::isfinite(v) && v > m) m = v;
            }
        kw[g] = pk_weight_scale_exp(&m, 1);
        wscale[g] = std::ldexp(1.0f, -kw[g]);
    }
    w16.resize((w16.size() + 7) & ~(size_t)7);
    const size_t off = w16.size();
    w16.resize(off + (size_t)nct * nks * 2 * nq * 64 * 8, 0);
    uint16_t* dst = w16.data() + off;
    for (int ct = 0; ct < nct; ++ct)
        for (int ks = 0; ks < nks; ++ks) {
            const int kq = ks / taps, tap = ks % taps;
            uint16_t* base = dst + ((size_t)ct * nks + ks) * (2 * nq * 64 * 8);
            for (int q = 0; q < nq; ++q)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int i = lane & 31, hh = lane >> 5;
                        const int co = ct * 32 * nq + 32 * q + i;
                        const int ci = 16 * kq + 8 * (e >> 2) + 4 * hh + (e & 3);   // wfl_chan(kq, hh, e)
                        const float w = std::ldexp(kn[((size_t)tap * Cin + ci) * N + co], kw[co / 32]);
                        const uint16_t h = f32_to_f16_rne(w);
                        base[((size_t)(0 * nq + q) * 64 + lane) * 8 + e] = h;
                        base[((size_t)(1 * nq + q) * 64 + lane) * 8 + e] = f32_to_f16_rne(w - f16_to_f32(h));
                    }
        }
    return off;
}

// Timing ablations (PK_FFNP_ABLATE; results are WRONG): instantiated in the profile build only.  Returns 1 when none applies.
template <bool PROF, class Go>
static int ffnp_ablation(Go& go, bool shape_ok) {
    if constexpr (PROF) {
        static const int abl = pk_prof_env("PK_FFNP_ABLATE") ? atoi(pk_prof_env("PK_FFNP_ABLATE")) : 0;
        if (abl && shape_ok) {
            switch (abl) {
                case 1: return go(k_ffn_planes<4, 96, 1, 4, FFNP_TAPS, 1>);
                case 4: return go(k_ffn_planes<4, 96, 1, 4, FFNP_TAPS, 4>);
                case 8: return go(k_ffn_planes<4, 96, 1, 4, FFNP_TAPS, 8>);
                case 13: return go(k_ffn_planes<4, 96, 1, 4, FFNP_TAPS, 13>);
                case 128: return go(k_ffn_planes<4, 96, 1, 4, FFNP_TAPS, 128>);
                case 256: return go(k_ffn_planes<4, 96, 1, 4, FFNP_TAPS, 256>);
                default: PK_FAIL(PK_EINVAL, "PK_FFNP_ABLATE: 1, 4, 8, 13, 128 or 256 (with PK_FFNP_VARIANT=84)");
            }
        }
    }
    return 1;
}

int ffnp_conv_launch(pk_ctx* ctx, const char* prof_name, const FfnpConv& c) {
    const bool first = c.out != nullptr;
    if (!(first ? (c.Cin == 384 && c.N % (32 * FFNP_NQ1) == 0) : (c.Cin == 1536 && c.N % (32 * FFNP_NQ2) == 0)) || c.nblk <= 0 || !c.w || !c.wscale)
        PK_FAIL(PK_EINVAL, "ffnp_conv_launch: shape (Cin %d, N %d) not built", c.Cin, c.N);
    // First conv: 256 columns per wave in 8-wave workgroups (decoder-sized timelines: half the operand traffic per MFMA), or
    // 128 columns per wave in 4-wave workgroups, two per CU (short timelines: four times the workgroups).  Second conv: 128
    // columns per wave, 8-wave workgroups.  PK_FFNP_VARIANT (measurement switch): 88 / 44 force the first conv's kernel, the
    // second digit 4 runs the second conv in 4-wave workgroups.
    const int variant = c.variant;   // (the "ffnp_variant" option of the owning handle)
    // Short timelines (round 4; c.one_max, default 4 096 tiles): ONE 32-column tile per wave -- the k loop of a wave is a serial chain (TAPS * Cin / 16 steps x
    // 3 NQ matrix instructions: 110 k cycles for the second conv at NQ = 4, whatever the number of rows), and with few row blocks
    // the chip is empty anyway.  Same numbers as every other tiling (the weight scales are per 32 channels, a tile's k order is fixed).
    const bool one = c.w1 && variant == 0 && (long)c.nblk * (c.N / 32) <= c.one_max;
    const bool small = !one && first && c.w4 && (variant / 10 == 4 || (variant / 10 != 8 && c.nblk < FFNP_NQ1_MIN_BLOCKS));
    const int nq = one ? 1 : (first && !small ? FFNP_NQ1 : FFNP_NQ2);
    const int W = one ? 8 : (first ? (small ? 4 : 8) : (variant % 10 == 4 ? 4 : 8));
    Args a;
    a.c = c;
    if (small) a.c.w = c.w4;
    if (one) a.c.w = c.w1;
    a.nct = c.N / (32 * nq);
    a.in_blk = (long)c.Cin * 128;
    a.out_blk = (long)c.N * 128;
    // 8-wave workgroups: a short timeline does not fill the chip with 8-block row groups -- fewer working waves per
    // workgroup, more workgroups
    int active = W;
    if (W == 8)
        while (active > 2 && (long)pk_div_up(c.nblk, active) * a.nct < ctx->n_cu) active >>= 1;
    a.active = active;
    a.nrg = pk_div_up(c.nblk, active);
    const int grid = pk_div_up(a.nrg, 8) * 8 * a.nct;
    auto go = [&](auto kern) -> int {
        PK_LAUNCH(ctx, prof_name, kern, dim3(grid), dim3(64 * W), 0, a);
        return PK_OK;
    };
    if (int st = ffnp_ablation<PK_PROFILE_BUILD != 0>(go, !first && W == 4); st != 1) return st;
    if (one) return first ? go(k_ffn_planes<1, 24, 0, 8>) : go(k_ffn_planes<1, 96, 1, 8>);
    if (first) return small ? go(k_ffn_planes<FFNP_NQ2, 24, 0, 4>) : go(k_ffn_planes<FFNP_NQ1, 24, 0, 8>);
    return W == 8 ? go(k_ffn_planes<FFNP_NQ2, 96, 1, 8>) : go(k_ffn_planes<FFNP_NQ2, 96, 1, 4>);
}

int ffnp_linear_launch(pk_ctx* ctx, const char* prof_name, const FfnpConv& c) {
    const int nq = c.ldin ? FFNP_NQ2 : FFNP_NQL;
    if (c.Cin != 384 || c.N % (32 * nq) != 0 || c.nblk <= 0 || !c.w || !c.wscale || !c.x)
        PK_FAIL(PK_EINVAL, "ffnp_linear_launch: shape (Cin %d, N %d) not built", c.Cin, c.N);
    Args a;
    a.c = c;
    a.nct = c.N / (32 * nq);
    a.in_blk = (long)c.Cin * 128;
    a.out_blk = 0;
    int active = 8;
    while (active > 2 && (long)pk_div_up(c.nblk, active) * a.nct < ctx->n_cu) active >>= 1;
    a.active = active;
    a.nrg = pk_div_up(c.nblk, active);
    const int grid = pk_div_up(a.nrg, 8) * 8 * a.nct;
    auto go = [&](auto kern) -> int {
        PK_LAUNCH(ctx, prof_name, kern, dim3(grid), dim3(512), 0, a);
        return PK_OK;
    };
    if (c.ldin) return go(k_ffn_planes<FFNP_NQ2, 24, 1, 8, 1, 0, true>);   // fp32 operand, x += . + b
    return go(k_ffn_planes<FFNP_NQL, 24, 2, 8, 1>);
}

// The 256-channel conv layers of the variance predictors (k = 3 / 5, ReLU, fp32 rows out for the LayerNorm that follows) and of
// the postnet (k = 5, tanh; planes out, or fp32 rows out for the layer that leaves the planes path).  c.out != NULL: planes
// (EPI 0, c.c1 / c.c0 = the output bound: 0 / 1 for tanh); else c.x = fp32 rows [row][ldx] written (not accumulated: EPI 2).
// in_amax needs taps / 2 elements of zero margin on either side, the planes one block (as everywhere).
bool ffnp_conv256_supports(int Cin, int N, int taps) {
    return N == 256 && ((Cin == 384 && (taps == 3 || taps == 5)) || (Cin == 256 && (taps == 3 || taps == 5)));
}
int ffnp_conv256_launch(pk_ctx* ctx, const char* prof_name, const FfnpConv& c, int taps, int act) {
    if (!ffnp_conv256_supports(c.Cin, c.N, taps) || c.nblk <= 0 || !c.w || !c.wscale || (!c.out && !c.x) || act < 1 || act > 2)
        PK_FAIL(PK_EINVAL, "ffnp_conv256_launch: shape (Cin %d, N %d, k %d, act %d) not built", c.Cin, c.N, taps, act);
    if (c.out && act != 1) PK_FAIL(PK_EINVAL, "ffnp_conv256_launch: planes output is built for tanh");
    Args a;
    a.c = c;
    // (short timelines: one 32-column tile per wave, as ffnp_conv_launch)
    const bool one = c.w1 && c.variant == 0 && (long)c.nblk * (c.N / 32) <= c.one_max;
    if (one) a.c.w = c.w1;
    a.nct = one ? c.N / 32 : c.N / (32 * FFNP_NQ2);   // 2 column tiles of 128, or 8 of 32
    a.in_blk = (long)c.Cin * 128;
    a.out_blk = (long)c.N * 128;
    int active = 8;
    while (active > 2 && (long)pk_div_up(c.nblk, active) * a.nct < ctx->n_cu) active >>= 1;
    a.active = active;
    a.nrg = pk_div_up(c.nblk, active);
    const int grid = pk_div_up(a.nrg, 8) * 8 * a.nct;
    auto go = [&](auto kern) -> int {
        PK_LAUNCH(ctx, prof_name, kern, dim3(grid), dim3(512), 0, a);
        return PK_OK;
    };
    // <NQ, KQ, EPI, W, TAPS, ABL, BF32, CPT, ACT>: CPT such that a slab is a whole number of k-steps that divides TAPS * KQ
    if (one) {   // (a slab = 4 CPT k-steps; at least three slabs)
        if (c.out) {
            if (c.Cin == 256 && taps == 5) return go(k_ffn_planes<1, 16, 0, 8, 5, 0, false, 5, 1>);
            PK_FAIL(PK_EINVAL, "ffnp_conv256_launch: planes output: 256 -> 256, k = 5 only");
        }
        if (act == 1) {
            if (c.Cin == 256 && taps == 5) return go(k_ffn_planes<1, 16, 2, 8, 5, 0, false, 5, 1>);
            PK_FAIL(PK_EINVAL, "ffnp_conv256_launch: tanh rows output: 256 -> 256, k = 5 only");
        }
        if (c.Cin == 384) return taps == 3 ? go(k_ffn_planes<1, 24, 2, 8, 3, 0, false, 6, 2>) : go(k_ffn_planes<1, 24, 2, 8, 5, 0, false, 6, 2>);
        return taps == 3 ? go(k_ffn_planes<1, 16, 2, 8, 3, 0, false, 4, 2>) : go(k_ffn_planes<1, 16, 2, 8, 5, 0, false, 5, 2>);
    }
    if (c.out) {
        if (c.Cin == 256 && taps == 5) return go(k_ffn_planes<4, 16, 0, 8, 5, 0, false, 5, 1>);
        PK_FAIL(PK_EINVAL, "ffnp_conv256_launch: planes output: 256 -> 256, k = 5 only");
    }
    if (act == 1) {   // tanh -> fp32 rows: the postnet's last 256 -> 256 layer
        if (c.Cin == 256 && taps == 5) return go(k_ffn_planes<4, 16, 2, 8, 5, 0, false, 5, 1>);
        PK_FAIL(PK_EINVAL, "ffnp_conv256_launch: tanh rows output: 256 -> 256, k = 5 only");
    }
    if (c.Cin == 384) return taps == 3 ? go(k_ffn_planes<4, 24, 2, 8, 3, 0, false, 6, 2>) : go(k_ffn_planes<4, 24, 2, 8, 5, 0, false, 6, 2>);
    return taps == 3 ? go(k_ffn_planes<4, 16, 2, 8, 3, 0, false, 6, 2>) : go(k_ffn_planes<4, 16, 2, 8, 5, 0, false, 5, 2>);
}

int ffnp_layernorm_launch(pk_ctx* ctx, const float* x, const float* g, const float* b, const int* row_utt, int nblk, int C,
                          float eps, void* out, unsigned* out_amax) {
    if (C % 16 != 0 || C / 8 > 64) PK_FAIL(PK_EINVAL, "ffnp_layernorm_launch: %d channels", C);
    PK_LAUNCH(ctx, g ? "fs2_layernorm_planes" : "fs2_rows_to_planes", k_ffn_ln_planes, dim3(nblk), dim3(256), 0, x, g, b, row_utt, C, eps,
              reinterpret_cast<char*>(out), out_amax);
    return PK_OK;
}
