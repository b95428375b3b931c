// wf_layer.hip -- one WaveFlow residual layer for one autoregressive row as ONE kernel (64-channel model, split-fp16
// math): the (3,3) dilated causal conv over the 3-row input ring + condition_proj as one contraction (K = 9*64 + 96),
// gated tanh, the res|skip out projection, the residual into the next layer's ring and the skip accumulation.
//
// Reference: parakeet/models/waveflow.py ResidualBlock.add_input :248-283 (conv2d over the row buffer :268-274,
// condition_proj :275, gate :276-277, out_proj + chunk + residual :279-282), ResidualNet.add_input :368-392.
//
// Why not the shared GEMM (k_gemm_h3 with the PK_EPI_GATE_PROJ epilogue, which this replaces for C = 64): with
// N = 128 output columns a 64-row GEMM tile re-reads the whole 336 KB of split weights for 10.7 MFLOP of work, 1296
// times per launch -- the launch is bound by weight traffic out of the L2 and by LDS bandwidth (both operands pass
// through LDS), at 16 % of the matrix pipe.  Here the roles are swapped, as in the Parallel WaveGAN layer kernel:
//   * positions are the MFMA N dimension: a wave owns 32 positions and ALL 128 gate channels (4 accumulator tiles),
//     its B operand (the input features of those positions, per tap) comes straight from global memory / L2 into
//     registers -- the blocked [pos/32][ch][32] layout makes every load a coalesced 128-byte segment -- and is
//     split in registers;
//   * weights are the A operand: streamed through LDS in slabs of six k-steps (48 KB), double buffered, each slab
//     used by the 8 waves = 256 positions of the workgroup -- a slab is 72 MFMAs per wave, long enough to cover the
//     L2 latency of the next slab's loads (two-k-step slabs measured 1.9 us per slab against 0.64 us of matrix
//     work); the out-projection weights (32 KB) stay resident;
//   * the gated activations never leave the accumulator registers: register r of the first contraction IS the B
//     operand element of k-step r / 8 of the second (K order of W2 permuted at pack time), as in pwg.hip;
//   * results are stored in the same blocked layout (coalesced), with the block maxima the next layer's operand
//     scale needs (pk_split.h).
// Rows before the sequence start are skipped as taps (ntap = 3, 6, 9), never stored as zeros.
#include "pk_wf_layer.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "pk_split.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 pkh2 __attribute__((ext_vector_type(2)));

namespace {
constexpr int WAVES = 8;
constexpr int THREADS = WAVES * 64;
constexpr int WAVE_T = 32;
constexpr int KCH = 512;   // 16-byte chunks per k-step of A fragments (= threads: one chunk per thread per k-step)
constexpr int SLAB = 6;    // k-steps per weight slab: 18, 30 and 42 k-steps (3, 6, 9 taps + condition) are multiples
constexpr int BLK_C = WFL_C * WFL_BLK;      // 2048 floats per feature block
constexpr int BLK_M = WFL_MP * WFL_BLK;     // 3072 floats per condition block

__host__ __device__ inline int mfma_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

__device__ __forceinline__ f32x16 mfma16(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// hi = v_cvt_pkrtz (round toward zero, saturating); x - hi exactly by v_fma_mix_f32; lo = fp16_rne(x - hi)
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
__device__ __forceinline__ void split8s(const float (&v)[8], float s, f16x8& hi, f16x8& lo) {
    float t[8];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        f32x2 u = {v[2 * p], v[2 * p + 1]};
        u *= s;
        t[2 * p] = u[0];
        t[2 * p + 1] = u[1];
    }
    split8(t, hi, lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_max_step(float v) {
    const int t = __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
    return fmaxf(v, __int_as_float(t));
}
__device__ __forceinline__ float wave_max64(float v) {   // wave-uniform maximum (see pwg.hip)
    v = dpp_max_step<0xB1, 0xf>(v);
    v = dpp_max_step<0x4E, 0xf>(v);
    v = dpp_max_step<0x124, 0xf>(v);
    v = dpp_max_step<0x128, 0xf>(v);
    v = dpp_max_step<0x142, 0xa>(v);
    v = dpp_max_step<0x143, 0xc>(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// 2^14 * tanh(a / S) * sigmoid(b / S) from accumulators that hold S * (pre-activation): ca = -2 log2(e) / S, cb = ca / 2
__device__ __forceinline__ float gated_s(float a, float b, float ca, float cb) {
    const float ta = __builtin_amdgcn_fmed3f(a * ca, -28.853900817779268f, 28.853900817779268f);
    const float ea = __builtin_amdgcn_exp2f(ta);
    const float eb = __builtin_amdgcn_exp2f(b * cb);
    return fmaf(ea, -PK_UNIT_SCALE, PK_UNIT_SCALE) * __builtin_amdgcn_rcpf((1.f + ea) * (1.f + eb));
}

__global__ __launch_bounds__(THREADS, 2) void k_wf_layer(WflLaunch a) {
    __shared__ __attribute__((aligned(16))) f16x8 wbuf[2][SLAB * KCH];   // two slabs of six k-steps: 96 KB
    __shared__ __attribute__((aligned(16))) f16x8 w2l[WFL_KS2 * KCH]; // out projection, resident: 32 KB
    __shared__ float lb[256];                                         // b1 [128] | b2s [128]
    // per logical k-step (taps whose row exists, then the condition block): where its B operand lives and which
    // packed weight k-step multiplies it.  Table driven so that the loads below are straight-line code: with
    // branches around them hipcc's s_waitcnt insertion falls back to vmcnt(0) before every load (seen in the first
    // version of this kernel: 60 % of the wave cycles in SQ_WAIT_ANY).
    __shared__ long kt_off[WFL_KS1];     // element offset from in0 of (position 0, channel 0 of this k-step)
    __shared__ int kt_shift[WFL_KS1];    // position shift of the tap
    __shared__ int kt_blk[WFL_KS1];      // floats per 32-position block of the source (64 or 96 channels)
    __shared__ int kt_w[WFL_KS1];        // packed k-step of W1
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, hi = lane >> 5;
    {
        const f16x8* src = reinterpret_cast<const f16x8*>(a.w.w2);
        for (int i = tid; i < WFL_KS2 * KCH; i += THREADS) w2l[i] = src[i];
        if (tid < 128) lb[tid] = a.w.b1[tid];
        else if (tid < 256) lb[tid] = a.w.b2s[tid - 128];
    }
    __syncthreads();
    // Measurement switch (PK_WF_WARM bit 0, default off): touch one dword of every 128-byte line of W1 up front.
    // Every launch uses another layer's weights, so their lines are cold in this XCD's L2; measured: no gain (the
    // slab loads are issued a slab ahead, which covers the miss).
    float warm = 0.f;
    if (a.warm & 1) {
        const float* wl = reinterpret_cast<const float*>(a.w.w1);
        constexpr int W1_LINES = (int)(WFL_W1_HALVES * 2 / 128);   // 2688
        for (int i = tid; i < W1_LINES; i += THREADS) warm += wl[(long)i * 32];
    }
    const int ntap = a.ntap;
    const int nks_conv = WFL_KS_TAP * ntap;
    const int nks = nks_conv + WFL_KS_COND;
    const int nslab = nks / SLAB;  // 3, 5 or 7
    if (tid < nks) {
        const int ks = tid;
        if (ks < nks_conv) {
            const int t = ks >> 2;
            kt_off[ks] = (long)a.tap_slot[t] * a.slot_stride + (long)(16 * (ks & 3)) * WFL_BLK;
            kt_shift[ks] = a.tap_shift[t];
            kt_blk[ks] = BLK_C;
            kt_w[ks] = a.tap_w[t] * WFL_KS_TAP + (ks & 3);
        } else {
            kt_off[ks] = (a.cond - a.in0) + (long)(16 * (ks - nks_conv)) * WFL_BLK;
            kt_shift[ks] = 0;
            kt_blk[ks] = BLK_M;
            kt_w[ks] = 9 * WFL_KS_TAP + (ks - nks_conv);
        }
    }
    const f16x8* w1 = reinterpret_cast<const f16x8*>(a.w.w1) + tid;
    auto w_kstep = [&](int ks) -> const f16x8* { return w1 + (long)kt_w[ks] * KCH; };   // packed k-step of logical ks
    __syncthreads();   // tables visible
    const int ntiles = a.npos_alloc / WAVE_T;
    const float i_res = pow2f(-(PK_UNIT_EXP + a.w.k2res)), i_skip = pow2f(-(PK_UNIT_EXP + a.w.k2skip));
    // A workgroup owns tiles_per_wg consecutive wave tiles and works through them in rounds of at most `active`
    // tiles (one pass over the weights per round).  The waves of a round run in lockstep (they share the LDS weight
    // slabs), so a round takes as long as its busiest SIMD: measured 25 us with one working wave per SIMD, 35 us
    // with two.  Full rounds first (11 tiles = 8 + 3: 35 + 25 us) therefore beat even rounds (6 + 5: a SIMD with two
    // waves in both, 35 + 35 us) and rounds of four (3 x 25 us).  The other waves only move weights and keep the
    // barriers.
    const int t_begin = (int)blockIdx.x * a.tiles_per_wg, t_end = min(t_begin + a.tiles_per_wg, ntiles);
    const int nrounds = (t_end - t_begin + a.active - 1) / a.active;

    for (int base = t_begin, rnd = 0; base < t_end; ++rnd) {   // uniform over the workgroup
        const int nact = (a.warm & 2) ? (t_end - base + (nrounds - rnd) - 1) / (nrounds - rnd)   // even rounds (A/B)
                                      : min(a.active, t_end - base);                           // tiles of this round
        const int wt = base + wave;
        const bool tile_ok = wave < nact;
        base += nact;
        const int p0 = tile_ok ? wt * WAVE_T : 0;
        const int p = p0 + j;
        const bool lane_ok = tile_ok && a.pos_utt[p] >= 0;

        // ---- operand scale of this wave tile: the largest block maximum among the blocks its taps read
        int kx;
        {
            float m = 0.f;
            if (lane < 2 * ntap) {
                const int t = lane >> 1;
                const int blk = (p0 + a.tap_shift[t] + 31 * (lane & 1)) >> 5;
                m = __uint_as_float(a.in_amax0[(long)a.tap_slot[t] * a.amax_stride + blk]);
            } else if (lane == 2 * ntap) {
                m = __uint_as_float(a.cond_amax[p0 >> 5]);
            }
            m = wave_max64(m);
            kx = blk_scale_exp(__float_as_uint(m));
        }
        const int ks1 = kx + a.w.k1;
        const float sx = pow2f(kx), S1 = pow2f(ks1);
        const int cbb = __builtin_amdgcn_readfirstlane(__float_as_int(-1.4426950408889634f * pow2f(-ks1)));
        const float gcb = __int_as_float(cbb), gca = __int_as_float(cbb + (1 << 23));

        f32x16 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][r] = lb[32 * q + mfma_row(r, hi)] * S1;

        // ---- B operand of k-step ks: 8 channels per lane, 128-byte segments across the lanes of a half wave
        auto load_b = [&](int ks, float (&dst)[8]) {
            const int q = p + kt_shift[ks];
            const float* src = a.in0 + kt_off[ks] + (long)(q >> 5) * kt_blk[ks] + (q & 31) + (8 * hi) * WFL_BLK;
#pragma unroll
            for (int e = 0; e < 8; ++e) dst[e] = src[e * WFL_BLK];
        };
        f16x8 wreg[SLAB];      // next slab's weights on their way from global memory to the other LDS buffer
        // Working and idle waves run separate copies of the slab loop (same number of barriers): the working copy
        // has no branch inside, loads and LDS stores are unconditional (the slab after the last one is the last one
        // again, parked in the buffer nobody reads any more).
        if (tile_ok) {
            float ring[SLAB][8];   // B operands, one slab ahead: slot kk is refilled as soon as k-step kk has split it
#pragma unroll
            for (int kk = 0; kk < SLAB; ++kk) {
                wreg[kk] = *w_kstep(kk);
                load_b(kk, ring[kk]);
            }
#pragma unroll
            for (int kk = 0; kk < SLAB; ++kk) wbuf[0][kk * KCH + tid] = wreg[kk];
            __syncthreads();
            for (int s = 0; s < nslab; ++s) {
                const int sn = min(s + 1, nslab - 1);
#pragma unroll
                for (int kk = 0; kk < SLAB; ++kk) wreg[kk] = *w_kstep(SLAB * sn + kk);
                __builtin_amdgcn_sched_barrier(0);   // keep the weight loads up here: hipcc would sink them to their stores
                const f16x8* wl = wbuf[s & 1] + lane;
#pragma unroll
                for (int kk = 0; kk < SLAB; ++kk) {
                    f16x8 bh, bl;
                    split8s(ring[kk], sx, bh, bl);
                    load_b(SLAB * sn + kk, ring[kk]);
                    __builtin_amdgcn_sched_barrier(0);   // ... and these ahead of the k-step's MFMAs
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f16x8 ah = wl[kk * KCH + (0 * 4 + q) * 64];
                        const f16x8 al = wl[kk * KCH + (1 * 4 + q) * 64];
                        acc[q] = mfma16(ah, bh, acc[q]);
                        acc[q] = mfma16(al, bh, acc[q]);
                        acc[q] = mfma16(ah, bl, acc[q]);
                    }
                }
#pragma unroll
                for (int kk = 0; kk < SLAB; ++kk) wbuf[(s + 1) & 1][kk * KCH + tid] = wreg[kk];
                __syncthreads();   // everyone is done reading this slab's buffer and sees the other one
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < SLAB; ++kk) wreg[kk] = *w_kstep(kk);
#pragma unroll
            for (int kk = 0; kk < SLAB; ++kk) wbuf[0][kk * KCH + tid] = wreg[kk];
            __syncthreads();
            for (int s = 0; s < nslab; ++s) {
                const int sn = min(s + 1, nslab - 1);
#pragma unroll
                for (int kk = 0; kk < SLAB; ++kk) wreg[kk] = *w_kstep(SLAB * sn + kk);
#pragma unroll
                for (int kk = 0; kk < SLAB; ++kk) wbuf[(s + 1) & 1][kk * KCH + tid] = wreg[kk];
                __syncthreads();
            }
        }
        if (!tile_ok) continue;   // (no barrier below this point)
        // ---- gate: z * 2^14 in the accumulator registers -> split B operands of the out projection
        f16x8 zh[WFL_KS2], zl[WFL_KS2];
#pragma unroll
        for (int k2 = 0; k2 < WFL_KS2; ++k2) {
            const int zq = k2 >> 1, r0 = 8 * (k2 & 1);
            float zv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) zv[e] = gated_s(acc[zq][r0 + e], acc[zq + 2][r0 + e], gca, gcb);
            split8(zv, zh[k2], zl[k2]);
        }
        // ---- out projection in two passes (res, skip): 32 old values + 32 accumulators live at a time
        const long po = (long)(p >> 5) * BLK_C + (p & 31);
        const float* res_in = a.in0 + (long)a.cur_slot * a.slot_stride + po;
        float am = 0.f;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            float old[32];
            const float* osrc = pass == 0 ? res_in : a.skip + po;
            if (pass == 0 || !a.first) {
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int r = 0; r < 16; ++r) old[16 * q + r] = osrc[(32 * q + mfma_row(r, hi)) * WFL_BLK];
            } else {
#pragma unroll
                for (int e = 0; e < 32; ++e) old[e] = 0.f;
            }
            f32x16 acc2[2];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[q][r] = lb[128 + 64 * pass + 32 * q + mfma_row(r, hi)];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k2 = 0; k2 < WFL_KS2; ++k2)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const f16x8 ah = w2l[k2 * KCH + (0 * 4 + 2 * pass + q) * 64 + lane];
                    const f16x8 al = w2l[k2 * KCH + (1 * 4 + 2 * pass + q) * 64 + lane];
                    acc2[q] = mfma16(ah, zh[k2], acc2[q]);
                    acc2[q] = mfma16(al, zh[k2], acc2[q]);
                    acc2[q] = mfma16(ah, zl[k2], acc2[q]);
                }
            __builtin_amdgcn_sched_barrier(0);
            float* dst = pass == 0 ? a.out : a.skip;
            const float inv = pass == 0 ? i_res : i_skip;
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = fmaf(acc2[q][r], inv, old[16 * q + r]);   // res = x_in + res (:281); skips summed (:390)
                    if (!lane_ok) v = 0.f;                              // gap positions stay zero
                    if (pass == 0) am = fmaxf(am, fabsf(v));
                    if (dst) (dst + po)[(32 * q + mfma_row(r, hi)) * WFL_BLK] = v;
                }
        }
        am = wave_max64(am);
        if (lane == 0 && a.out_amax) a.out_amax[p0 >> 5] = __float_as_uint(am);
    }
    if (warm == 1.2345e-30f) a.skip[0] = warm;   // never true: keeps the warm-up loads
}

// max|cond| per block: one wave per (row, block) of 96 x 32 contiguous floats
__global__ __launch_bounds__(64) void k_wf_cond_amax(const float* __restrict__ cond, long row_stride, long amax_row_stride,
                                                     unsigned* __restrict__ amax) {
    const float* src = cond + (long)blockIdx.y * row_stride + (long)blockIdx.x * BLK_M;
    float m = 0.f;
    for (int i = threadIdx.x; i < BLK_M; i += 64) m = fmaxf(m, fabsf(src[i]));
    m = wave_max64(m);
    if (threadIdx.x == 0) amax[(long)blockIdx.y * amax_row_stride + blockIdx.x] = __float_as_uint(m);
}

// Flow._predict_row_parameters :496-501 + _inverse_transform_row :503-505 + input_proj of the new row (:497) on the
// blocked layout: one wave per 32 positions, lane (j, hi) = position j, channels 32*hi .. 32*hi + 31
__global__ __launch_bounds__(256) void k_wf_step_blk(const float* __restrict__ skip, const float* __restrict__ w_out,
                                                     float b_logs, float b_b, const float* __restrict__ z_row,
                                                     float* __restrict__ x_row, const float* __restrict__ w_in,
                                                     const float* __restrict__ b_in, float* __restrict__ h0_next,
                                                     unsigned* __restrict__ h0_amax, const int* __restrict__ pos_utt,
                                                     int npos_alloc, int first) {
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile * WAVE_T >= npos_alloc) return;
    const int lane = threadIdx.x & 63, j = lane & 31, hi = lane >> 5;
    const int p = tile * WAVE_T + j;
    const bool valid = pos_utt[p] >= 0;
    const long po = (long)tile * BLK_C + j + (long)(32 * hi) * WFL_BLK;
    float xn = 0.f;
    if (first) {
        xn = valid ? z_row[p] : 0.f;
    } else {
        float l = 0.f, bb = 0.f;
#pragma unroll 8
        for (int c = 0; c < 32; ++c) {
            const float v = skip[po + c * WFL_BLK];
            l = fmaf(w_out[32 * hi + c], v, l);
            bb = fmaf(w_out[WFL_C + 32 * hi + c], v, bb);
        }
        l += __shfl_xor(l, 32);
        bb += __shfl_xor(bb, 32);
        xn = valid ? (z_row[p] - (bb + b_b)) * expf(-(l + b_logs)) : 0.f;
    }
    if (hi == 0) x_row[p] = xn;
    if (h0_next) {
        float am = 0.f;
#pragma unroll 8
        for (int c = 0; c < 32; ++c) {
            const float v = valid ? fmaf(w_in[32 * hi + c], xn, b_in[32 * hi + c]) : 0.f;
            am = fmaxf(am, fabsf(v));
            h0_next[po + c * WFL_BLK] = v;
        }
        am = wave_max64(am);
        if (lane == 0) h0_amax[tile] = __float_as_uint(am);
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
inline void put_split(uint16_t* dst_hi, uint16_t* dst_lo, float w) {
    const uint16_t h = f32_to_f16_rne(w);
    *dst_hi = h;
    *dst_lo = f32_to_f16_rne(w - f16_to_f32(h));
}
}  // namespace

WflPacked wfl_pack(const float* conv, const float* conv_b, const float* cond, const float* cond_b, int n_mels,
                   const float* outp, const float* outp_b, std::vector<uint16_t>& w16, std::vector<float>& f32) {
    constexpr int C = WFL_C;
    WflPacked o;
    // one exponent for the first contraction (conv and condition weights share the accumulators), one each for the
    // res and the skip half of the out projection
    {
        float m = 0.f;
        for (size_t i = 0; i < (size_t)2 * C * C * 9; ++i) m = std::fmax(m, std::fabs(conv[i]));
        for (size_t i = 0; i < (size_t)2 * C * n_mels; ++i) m = std::fmax(m, std::fabs(cond[i]));
        o.k1 = pk_weight_scale_exp(&m, 1);
        o.k2res = pk_weight_scale_exp(outp, (size_t)C * C);
        o.k2skip = pk_weight_scale_exp(outp + (size_t)C * C, (size_t)C * C);
    }
    auto align8 = [&]() { w16.resize((w16.size() + 7) & ~(size_t)7); };
    align8();
    o.w1 = w16.size();
    w16.resize(o.w1 + WFL_W1_HALVES, 0);
    uint16_t* a1 = w16.data() + o.w1;
    for (int ks = 0; ks < WFL_KS1; ++ks)
        for (int q = 0; q < 4; ++q)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int i = lane & 31, hi = lane >> 5;
                    const int co = q < 2 ? 32 * q + i : C + 32 * (q - 2) + i;   // content | gate (chunk :276)
                    float w;
                    if (ks < 9 * WFL_KS_TAP) {
                        const int tap = ks / WFL_KS_TAP, kr = tap / 3, kc = tap % 3;
                        const int ci = 16 * (ks % WFL_KS_TAP) + 8 * hi + e;
                        w = conv[(((size_t)co * C + ci) * 3 + kr) * 3 + kc];
                    } else {
                        const int m = 16 * (ks - 9 * WFL_KS_TAP) + 8 * hi + e;
                        w = m < n_mels ? cond[(size_t)co * n_mels + m] : 0.f;
                    }
                    w = std::ldexp(w, o.k1);
                    put_split(a1 + ((((size_t)ks * 2 + 0) * 4 + q) * 64 + lane) * 8 + e,
                              a1 + ((((size_t)ks * 2 + 1) * 4 + q) * 64 + lane) * 8 + e, w);
                }
    align8();
    o.w2 = w16.size();
    w16.resize(o.w2 + WFL_W2_HALVES, 0);
    uint16_t* a2 = w16.data() + o.w2;
    for (int ks = 0; ks < WFL_KS2; ++ks)
        for (int q = 0; q < 4; ++q)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int i = lane & 31, hi = lane >> 5;
                    const int zc = 32 * (ks >> 1) + mfma_row(8 * (ks & 1) + e, hi);   // gated channel of this k-slot
                    const int row = q < 2 ? 32 * q + i : C + 32 * (q - 2) + i;        // res | skip (chunk :280)
                    const float w = std::ldexp(outp[(size_t)row * C + zc], q < 2 ? o.k2res : o.k2skip);
                    put_split(a2 + ((((size_t)ks * 2 + 0) * 4 + q) * 64 + lane) * 8 + e,
                              a2 + ((((size_t)ks * 2 + 1) * 4 + q) * 64 + lane) * 8 + e, w);
                }
    f32.resize((f32.size() + 3) & ~(size_t)3);
    o.b1 = f32.size();
    for (int c = 0; c < 2 * C; ++c) f32.push_back(conv_b[c] + cond_b[c]);   // :274-275
    o.b2s = f32.size();
    for (int c = 0; c < 2 * C; ++c) f32.push_back(std::ldexp(outp_b[c], PK_UNIT_EXP + (c < C ? o.k2res : o.k2skip)));
    return o;
}

int wfl_layer_launch(pk_ctx* ctx, const WflLaunch& a) {
    if (a.npos_alloc % WAVE_T != 0 || a.ntap % 3 != 0 || a.ntap < 3 || a.ntap > 9)
        PK_FAIL(PK_EINVAL, "wfl_layer_launch: bad shape (npos %d, taps %d)", a.npos_alloc, a.ntap);
    const int ntiles = a.npos_alloc / WAVE_T;
    WflLaunch b = a;
    static const int active_env = getenv("PK_WF_ACTIVE") ? atoi(getenv("PK_WF_ACTIVE")) : WAVES;   // measurement switch
    b.active = active_env >= 1 && active_env <= WAVES ? active_env : WAVES;
    static const int warm_env = getenv("PK_WF_WARM") ? atoi(getenv("PK_WF_WARM")) : 0;   // measurement switches
    b.warm = warm_env;
    b.tiles_per_wg = std::max(1, (ntiles + ctx->n_cu - 1) / ctx->n_cu);
    const int grid = (ntiles + b.tiles_per_wg - 1) / b.tiles_per_wg;
    PK_LAUNCH(ctx, "wf_layer", k_wf_layer, dim3(grid), dim3(THREADS), 0, b);
    return PK_OK;
}

int wfl_cond_amax_launch(pk_ctx* ctx, const float* cond, long row_stride, int rows, int nblk, long amax_row_stride,
                         unsigned* amax) {
    PK_LAUNCH(ctx, "wf_cond_amax", k_wf_cond_amax, dim3(nblk, rows), dim3(64), 0, cond, row_stride, amax_row_stride, amax);
    return PK_OK;
}

int wfl_step_launch(pk_ctx* ctx, const float* skip, const float* w_out, float b_logs, float b_b, const float* z_row,
                    float* x_row, const float* w_in, const float* b_in, float* h0_next, unsigned* h0_amax,
                    const int* pos_utt, int npos_alloc, int first) {
    PK_LAUNCH(ctx, "wf_step", k_wf_step_blk, dim3(pk_div_up(npos_alloc / WAVE_T, 4)), dim3(256), 0, skip, w_out, b_logs,
              b_b, z_row, x_row, w_in, b_in, h0_next, h0_amax, pos_utt, npos_alloc, first);
    return PK_OK;
}
