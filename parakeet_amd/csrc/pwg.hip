// pwg.hip -- Parallel WaveGAN generator on gfx950: kernels + pk_pwg_* entry points.
//
// Reference: parakeet/models/parallel_wavegan/parallel_wavegan.py
//   ConvInUpsampleNet.forward :201-216, UpsampleNet.forward :119-138,
//   ResidualBlock.forward :284-315, PWGGenerator.forward :445-472,
//   PWGGenerator.inference :498-520, PWGInference.forward :772-775.
//
// Data layout in HBM ("timeline" layout).  All per-sample activations are
// channel-major over ONE concatenated time axis of length Ttot that holds every
// utterance of the batch, separated (and framed) by zero gaps of GAP samples:
//
//      |GAP| utt 0 (S_0) |GAP| utt 1 (S_1) |GAP| ... |GAP|
//
// Storage is BLOCKED: block k holds samples [32k, 32k+32) of all 64 channels contiguously,
//     addr(ch, t) = (t >> 5) * 2048 + ch * 32 + (t & 31)            (floats)
// so the 64 x 32 patch a wave reads and writes per layer is ONE contiguous 8 KB range (a plain
// [ch][Ttot] layout makes it 64 separate 128-byte pieces 20 MB apart -- one DRAM page each).  GAP >= the largest
// dilation, so a dilated tap that leaves an utterance reads zeros -- exactly the
// zero padding nn.Conv1D applies at the utterance ends in the reference's
// one-utterance-per-call inference.  S_b is a multiple of the hop (256), work
// tiles are whole frames, so a tile is never partly valid and the gaps are never
// written after the per-call zeroing.  Ragged batches cost nothing extra.
//
// The residual stack is 93 % of the end-to-end FLOPs and is a
// chain of small dense contractions per sample; it runs on the exact-fp32 matrix
// pipe (v_mfma_f32_32x32x2_f32).
//
// Conditioning path.  UpsampleNet is linear and identical for every mel channel
// (one shared (1, 2s+1) FIR per stage, no bias, no activation in the LJSpeech
// recipe), so it commutes with the per-layer channel mix conv1x1_aux:
//     conv1x1_aux_l(Upsample(c0)) == Upsample(W_aux_l . c0)
// The engine therefore projects at FRAME rate (one GEMM: [frames x 80] x
// [80 x layers*128]) and applies the 4-stage upsampler as ONE composite,
// phase-dependent 5-tap filter over frames (table built at finalize from the
// stage FIRs, with exact zero-padding behaviour at utterance edges).  The
// [80][samples] conditioning tensor is never materialised and the per-sample K of
// the first contraction drops from 272 to 192.
#include <cmath>
#include <cstdlib>

#include "pk_gemm.h"
#include "pk_split.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// Round 6 (profiles/r06_pwg_nt_ab.txt): the layer kernel's skip stream (read once, written once per layer: 512 of the 1 024 B per sample) and its
// x-plane stores are non-temporal accesses -- they no longer displace the x lines the other two taps re-read from the XCD's L2, and the kernel
// boundary has less to write back.  One-box A/B, three interleaved repetitions, same waveform bit for bit: skip 1 333 -> 1 313 us per launch,
// both 1 314 (41.14 -> 40.45 ms per 30 layers, -1.7 %).  0 = plain accesses (the A/B).
// Round 6 (profiles/r06_pwg_tile_map.txt): which wave of an XCD takes which wave tile of the XCD's window of a sweep.  A tile's +-d taps are the centre
// taps of the tiles d / 32 further on, so a line is fetched from HBM once only if those tiles run while it is still in the XCD's L2 -- and the two
// waves of a SIMD run half a tile (8 us) apart, the time the L2 holds this stream.  FETCH_SIZE per launch by dilation showed it: with tile = 8 * workgroup
// + wave, the launches with d = 64 / 128 (partner = wave +-2 / +-4 of the same workgroup) fetched 1.5 x the bytes of d <= 4, those with d = 256 / 512
// (partner = the SAME wave of the next workgroups) 1.1 x.  1: the first waves of the SIMDs (0..3 of every workgroup) take the first half of the window,
// the second waves the second half -- fetched bytes -12 %, launch 1 316 -> 1 291 us, same waveform bit for bit.  0 = workgroup-major (the A/B),
// 2 = wave-major (as good), 3 = SIMD-major (the two waves of a SIMD on adjacent tiles: worse, which confirms the reading).
#ifndef PK_PWG_TILE_MAP
#define PK_PWG_TILE_MAP 1
#endif
#ifndef PK_PWG_NT_SKIP
#define PK_PWG_NT_SKIP 1
#endif
// measurement: the skip stream at another cache policy than `nt` -- 1: agent scope (sc1), 2: system scope (sc0 sc1), 3: workgroup scope (sc0)
#ifndef PK_PWG_SKIP_SCOPE
#define PK_PWG_SKIP_SCOPE 0
#endif
#ifndef PK_PWG_NT_EDGE   // k_pwg_first's plane stores and k_pwg_last_h3's skip loads as non-temporal accesses (profiles/r06_pwg_nt_ab.txt)
#define PK_PWG_NT_EDGE 0
#endif
#ifndef PK_PWG_AMAX_PROBE   // measurement: 1 = the AMAX instantiations without their atomics (what the reduction and the register pressure cost)
#define PK_PWG_AMAX_PROBE 0
#endif
#ifndef PK_PWG_NT_XOUT
#define PK_PWG_NT_XOUT 1
#endif
namespace {
__device__ __forceinline__ float pwg_skip_ld(const float* p) {
#if PK_PWG_SKIP_SCOPE == 1
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#elif PK_PWG_SKIP_SCOPE == 2
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#elif PK_PWG_SKIP_SCOPE == 3
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#elif PK_PWG_NT_SKIP
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
__device__ __forceinline__ void pwg_skip_st(float v, float* p) {
#if PK_PWG_SKIP_SCOPE == 1
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#elif PK_PWG_SKIP_SCOPE == 2
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#elif PK_PWG_SKIP_SCOPE == 3
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#elif PK_PWG_NT_SKIP
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
}  // namespace
#ifndef PK_PWG_GATE_SCALAR
#define PK_PWG_GATE_SCALAR 0   // 1: the layer kernel's gate on scalar fp32 instructions (round 6 A/B: packed fp32 beside matrix instructions, profiles/r06_pwg_packed_ab.txt)
#endif

namespace {

constexpr int R = 64;     // residual channels
constexpr int G = 128;    // gate channels
constexpr int SK = 64;    // skip channels
constexpr int AUX = 80;   // aux channels
constexpr int KTAP = 3;   // kernel size of the dilated conv
constexpr int KS_CONV = KTAP * R / 2;         // 96 k-steps (2 input channels per MFMA)
constexpr int KS1 = KS_CONV;                  // 96 (the aux 1x1 conv is applied at frame rate)
constexpr int KS2 = (G / 2) / 2;              // 32 k-steps over the 64 gated channels
constexpr int TILE = 256;                     // samples per workgroup tile (one frame at hop 256)
constexpr int WAVE_T = 32;                    // samples per wave (MFMA N)
constexpr int MAX_UP_TAPS = 17;
constexpr int UPW = 5;                        // frames touched by the composite upsampler (reach < 2 frames)
constexpr int UPW_PAD = 8;                    // table row stride (floats)
constexpr int N_EDGE_CLASS = 9;               // (min(frames before, 2), min(frames after, 2))
constexpr int P_LEAD = 8;                     // margin rows around the frame-rate projection

// blocked sample-timeline addressing (see the header comment)
constexpr int XBLK = 32;                       // samples per block
constexpr int XBLK_FLOATS = R * XBLK;          // 2048 floats = 8 KB
__host__ __device__ inline long xoff(long t) { return (t >> 5) * XBLK_FLOATS + (t & 31); }   // + ch * 32
typedef _Float16 pl_f16x8 __attribute__((ext_vector_type(8)));   // one vector of the planes layout (k_pwg_layer_b3<..., PL>)
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// Row of the 32x32 MFMA result held in accumulator register r of a lane whose
// (lane >> 5) is hi.  (cdna guide: row = (r&3) + 8*(r>>2) + 4*hi, col = lane&31)
__host__ __device__ inline int mfma_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// ---------------------------------------------------------------- small kernels

// Zero the GAP regions of a [rows][Ttot] buffer.  gap_start[g], g in [0, n_gaps).
__global__ void k_zero_gaps(float* buf, const int* gap_start, const int* gap_len, int rows, long Ttot) {
    int g = blockIdx.y;
    int row = blockIdx.z;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    (void)Ttot;
    if (i < gap_len[g] && row < rows) buf[xoff((long)gap_start[g] + i) + row * XBLK] = 0.f;
}

// conv_in, step 1: ZScore-normalise (PWGInference :773) and lay the mel frames out as a padded row timeline
// for the GEMM of step 2: utterance b owns rows [cuL[b] + 2*ctx*b, +frames_b + 2*ctx); the ctx rows on either
// side are Pad1D(ctx, 'replicate') of inference (:518) or, for forward(x, c), the context frames the caller's c
// already carries (:201-216).  Step 2 is Conv1D(aux -> aux, k = 2ctx+1, no bias) (:188-192,214) as one implicit
// GEMM (K = (2ctx+1) * 80) whose row map drops the padded rows.  prow_src[row] = source mel row.
__global__ void k_pwg_convin_prep(const float* __restrict__ mel, const float* __restrict__ mu,
                                  const float* __restrict__ sigma, int use_norm, const int* __restrict__ prow_src,
                                  int rows, float* __restrict__ out) {
    const int r = blockIdx.x;
    const int c = threadIdx.x;
    if (c >= AUX) return;
    float v = 0.f;
    if (r < rows) {
        v = mel[(long)prow_src[r] * AUX + c];
        if (use_norm) v = (v - mu[c]) / sigma[c];
    }
    out[(long)r * AUX + c] = v;
}

// ---------------------------------------------------------------- hop sizes other than 256 (GEN kernels)
// The kernels schedule 256-sample tiles of 8 wave tiles.  With hop = prod(upsample_scales) = 256 (LJSpeech) a tile
// IS a frame: utterances are whole numbers of tiles and the frame / phase of a sample are tile index / offset.
// For any other hop (baker, vctk: 300) a tile is a 256-sample chunk of its utterance: the last chunk is partly
// valid (samples at or beyond S_b = frames * hop are stored as zeros, so they stay the zero padding the reference's
// convolutions see), a 32-sample wave tile can straddle two frames (hop >= 32), and frame, phase and edge class are
// computed per lane from the tables below.
struct PwgGen {
    const int* tile_s0;    // [tiles] sample offset of the tile inside its utterance
    const int* tile_utt;   // [tiles] utterance of the tile
    const int* utt_S;      // [B] samples
    const int* utt_F;      // [B] frames
    const int* utt_row0;   // [B] row of the utterance's frame 0 in the frame-rate projection P
    const int* utt_off;    // [B] offset of the utterance in the packed noise / wav buffers
    const float* P0;       // P + this layer's column block, row 0 = frame 0 of utterance 0
    int hop;
    float inv_hop;
};
struct PwgGenCoord {
    int fa;       // frame of the wave tile's first (clamped) sample: P rows fa-2 .. fa+3 are staged
    int df;       // this lane's frame - fa (0 or 1)
    int cls;      // edge class of this lane's frame
    int phase;    // sample - frame * hop
    int row0;
    bool valid;   // sample < S_b
};
__device__ __forceinline__ int gen_div_hop(int s, int hop, float inv_hop) {   // floor(s / hop), 0 <= s < 2^24
    int q = (int)((float)s * inv_hop);
    const int r = s - q * hop;
    q += r < 0 ? -1 : (r >= hop ? 1 : 0);
    return q;
}
__device__ __forceinline__ PwgGenCoord gen_coord(const PwgGen& g, int tile, int sub, int j) {
    const int b = g.tile_utt[tile];
    const int S = g.utt_S[b], F = g.utt_F[b];
    const int s0 = g.tile_s0[tile] + sub * WAVE_T;
    PwgGenCoord c;
    c.valid = s0 + j < S;
    const int sc = min(s0 + j, S - 1), s0c = min(s0, S - 1);
    const int f = gen_div_hop(sc, g.hop, g.inv_hop);
    c.fa = gen_div_hop(s0c, g.hop, g.inv_hop);
    c.df = f - c.fa;
    c.phase = sc - f * g.hop;
    c.cls = min(f, 2) * 3 + min(F - 1 - f, 2);
    c.row0 = g.utt_row0[b];
    return c;
}

// first_conv: Conv1D(1 -> R, k=1, bias) (:401-402,464) from packed noise into the timeline.
// PL: x is written as pre-split fp16 planes (see k_pwg_layer_b3<..., PL>): per 32-sample block [octet 8][sample 32][hi 8 | lo 8]
// halves at the utterance's a-priori scale 2^tile_kx[tile] -- the same 8 KB per block, the same 4 bytes per value.
template <bool GEN, bool PL = false>
__global__ void k_pwg_first(const float* __restrict__ noise, const float* __restrict__ w,
                            const float* __restrict__ bias, const int* __restrict__ tile_t0, long Ttot,
                            float* __restrict__ x, unsigned* __restrict__ xe, PwgGen g, const int* __restrict__ tile_kx) {
    const int tile = blockIdx.x;
    const long t = (long)(tile_t0[tile] & ~255) + threadIdx.x;   // low bits: the tile's edge class (layer kernels)
    float n;
    bool valid = true;
    if constexpr (GEN) {
        const int b = g.tile_utt[tile];
        const int sidx = g.tile_s0[tile] + (int)threadIdx.x;
        valid = sidx < g.utt_S[b];
        n = valid ? noise[(long)g.utt_off[b] + sidx] : 0.f;
    } else {
        n = noise[(long)tile * TILE + threadIdx.x];
    }
    (void)Ttot;
    const long xo = xoff(t);
    float am = 0.f;
    if constexpr (!PL) {
#pragma unroll 8
        for (int c = 0; c < R; ++c) {
            const float v = valid ? fmaf(w[c], n, bias[c]) : 0.f;
            am = fmaxf(am, fabsf(v));
            x[xo + c * XBLK] = v;
        }
    }
    // max|x| per 32-sample block for the first layer's operand scale (see "block scaling")
    am = fmaxf(am, __shfl_xor(am, 16));
    am = fmaxf(am, __shfl_xor(am, 8));
    am = fmaxf(am, __shfl_xor(am, 4));
    am = fmaxf(am, __shfl_xor(am, 2));
    am = fmaxf(am, __shfl_xor(am, 1));
    if (!PL && (threadIdx.x & 31) == 0) xe[t >> 5] = __float_as_uint(am);
    if constexpr (PL) {
        const float so = pow2f(tile_kx[tile]);
        char* dst = reinterpret_cast<char*>(x) + (t >> 5) * (long)(XBLK_FLOATS * 4) + (t & 31) * 32;
#pragma unroll
        for (int o = 0; o < R / 8; ++o) {   // octet o = 2 cg + hh holds channels 16 cg + 8 (e >> 2) + 4 hh + (e & 3)
            pl_f16x8 oh, ol;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = 16 * (o >> 1) + 8 * (e >> 2) + 4 * (o & 1) + (e & 3);
                const float tv = (valid ? fmaf(w[c], n, bias[c]) : 0.f) * so;
                const _Float16 hv = (_Float16)tv;
                oh[e] = hv;
                ol[e] = (_Float16)(tv - (float)hv);
            }
            if (PK_PWG_NT_EDGE) {
                __builtin_nontemporal_store(oh, reinterpret_cast<pl_f16x8*>(dst + o * 1024));
                __builtin_nontemporal_store(ol, reinterpret_cast<pl_f16x8*>(dst + o * 1024 + 16));
            } else {
                *reinterpret_cast<pl_f16x8*>(dst + o * 1024) = oh;
                *reinterpret_cast<pl_f16x8*>(dst + o * 1024 + 16) = ol;
            }
        }
    }
}

// Planes path: max|noise| per utterance, then per work tile the scale exponents of x for every layer from the magnitude bound
// B_0 = wmax * max|noise| + bmax, B_(l+1) = (B_l + c_l) * sqrt(0.5) (a hair of slack for the engine's own rounding);
// tile_kx[l * ntiles + tile] = blk_scale_exp(B_l) (pk_split.h), l = 0 .. layers (the last row: the final x).
// (round 6: NMAX_PARTS workgroups per utterance, 16-byte loads, the partial maxima folded with an integer atomic max --
// non-negative floats order like their bit patterns; one workgroup per utterance with dword loads took 0.15 ms of the step)
constexpr int NMAX_PARTS = 16;
__global__ __launch_bounds__(256) void k_pwg_noise_max(const float* __restrict__ noise, const int* __restrict__ utt_off,
                                                       const int* __restrict__ utt_S, float* __restrict__ nmax) {
    __shared__ float red[4];
    const int b = blockIdx.x, S = utt_S[b];
    const float* p = noise + utt_off[b];
    const int per = ((S + NMAX_PARTS - 1) / NMAX_PARTS + 3) & ~3;          // samples per part, a multiple of 4
    const int lo = blockIdx.y * per, hi = min(lo + per, S);
    float m = 0.f;
    // head up to a 16-byte boundary, body in float4, tail
    int i0 = lo + (int)threadIdx.x;
    const int head = min(hi, lo + (int)((4 - ((reinterpret_cast<size_t>(p + lo) >> 2) & 3)) & 3));
    if (i0 < head) m = fabsf(p[i0]);
    const int nb = (hi - head) >> 2;
    const f32x4* q = reinterpret_cast<const f32x4*>(p + head);
    for (int i = threadIdx.x; i < nb; i += 256) {
        const f32x4 v = q[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    for (int i = head + 4 * nb + threadIdx.x; i < hi; i += 256) m = fmaxf(m, fabsf(p[i]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicMax(reinterpret_cast<unsigned*>(nmax) + b, __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
}
__global__ __launch_bounds__(256) void k_pwg_tile_scales(const int* __restrict__ tile_utt, int ntiles,
                                                         const float* __restrict__ nmax, float wmax, float bmax,
                                                         const float* __restrict__ cl, int layers, int* __restrict__ tile_kx) {
    const int tile = blockIdx.x * 256 + threadIdx.x;
    if (tile >= ntiles) return;
    float B = fmaf(wmax, nmax[tile_utt[tile]], bmax) * 1.001f;
    for (int l = 0; l <= layers; ++l) {
        tile_kx[(long)l * ntiles + tile] = blk_scale_exp(__float_as_uint(B));
        if (l < layers) B = (B + cl[l]) * (0.70710678118654752440f * 1.001f);
    }
}

// "scale_guard" (pk_pwg_set_option): the measured side of the a-priori bound.  max|x| per utterance of one layer's planes --
// a workgroup per 256-sample work tile decodes its 8 blocks ((hi + lo) / 2^k) and folds the tile's maximum into amax[utterance]
// (non-negative floats order like their bit patterns).  Round 6: one launch per guarded / sampled inference (the planes of
// k_pwg_first); the 30 layers' maxima come from the layer kernel's own epilogue on those calls (its AMAX instantiations, which
// only guarded / sampled calls launch: 30 x 0.28 ms of extra passes over x gone from every 16th call) -- the instantiation every
// other call runs is untouched.
__global__ __launch_bounds__(256) void k_pwg_planes_amax(const float* __restrict__ x, const int* __restrict__ tile_t0,
                                                         const int* __restrict__ tile_utt, const int* __restrict__ tile_kx,
                                                         unsigned* __restrict__ amax) {
    __shared__ float red[4];
    const int tile = blockIdx.x;
    const long blk0 = (long)(tile_t0[tile] & ~255) >> 5;
    const pl_f16x8* src = reinterpret_cast<const pl_f16x8*>(reinterpret_cast<const char*>(x) + blk0 * (long)(XBLK_FLOATS * 4));
    float m = 0.f;
    // a block: [octet 8][sample 32][hi 8 | lo 8] halves = 512 vectors of 8 halves, hi at even, lo at odd vector indices
#pragma unroll
    for (int i = 0; i < (TILE / XBLK) * 256 / 256; ++i) {
        const int pair = i * 256 + threadIdx.x;   // 0 .. 2047: (hi, lo) pairs of the tile
        const pl_f16x8 vh = src[2 * pair], vl = src[2 * pair + 1];
#pragma unroll
        for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf((float)vh[e] + (float)vl[e]));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * pow2f(-tile_kx[tile]);
        atomicMax(amax + tile_utt[tile], __float_as_uint(m));
    }
}

// The layer kernel's AMAX instantiations fold a tile's max|x_out| into parts[(utterance) * PWG_AMAX_PARTS + (workgroup & 31)] -- one address per
// utterance took +0.3 ms per launch in same-address atomics (163 840 tiles on 32 addresses); the workgroups that share a part sit on one XCD.
// This folds the parts of all layers into amax[1 + layer][utterance] at the end of the stack.
constexpr int PWG_AMAX_PARTS = 32;
__global__ __launch_bounds__(256) void k_pwg_amax_fold(const unsigned* __restrict__ parts, unsigned* __restrict__ amax, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned m = 0;
#pragma unroll
    for (int p = 0; p < PWG_AMAX_PARTS; ++p) m = max(m, parts[(size_t)i * PWG_AMAX_PARTS + p]);
    amax[i] = m;
}

// Test tap: the sample-rate aux contribution of one layer, aux[co][s] =
// sum_j T[class][phase][j] * P[frame + j - 2][layer*G + co], written channel-major for one utterance.
// (frame-indexed: row0 = P row of the utterance's frame 0, n_frames frames, hop threads per block)
__global__ void k_pwg_aux_debug(const float* __restrict__ P, int ldp, int col0, const float* __restrict__ uptab,
                                int row0, int n_frames, int hop, float* __restrict__ out) {
    const int f = blockIdx.x;   // frame within the utterance
    const int co = blockIdx.y;
    const int phase = threadIdx.x;
    const int cls = min(f, 2) * 3 + min(n_frames - 1 - f, 2);
    const float* w = uptab + ((long)cls * hop + phase) * UPW_PAD;
    float acc = 0.f;
    for (int jj = 0; jj < UPW; ++jj) acc = fmaf(w[jj], P[(long)(row0 + f + jj - 2) * ldp + col0 + co], acc);
    out[(long)co * n_frames * hop + (long)f * hop + phase] = acc;
}

// ---------------------------------------------------------------- residual block
struct PwgLayerArgs {
    const float* xin;     // [R][Ttot]
    float* xout;          // [R][Ttot]
    float* skip;          // [SK][Ttot]
    const float* w1;      // [KS1][64 lanes][4 co-tiles]  A fragments, dilated conv
    const float* w2;      // [KS2][64 lanes][4 out-tiles] A fragments, out / skip 1x1 convs
    const float* bias;    // [G + R + SK]: conv bias (gate), conv1x1_out bias, conv1x1_skip bias
    const float* P;       // frame-rate aux projection, row f = frame f, this layer's G columns at P + col
    const float* uptab;   // [N_EDGE_CLASS][TILE phases][UPW_PAD] composite upsampler weights
    const int* tile_t0;   // [ntiles] timeline offset of each 256-sample tile (= frame)
    const int* tile_cls;  // [ntiles] edge class of the frame
    long Ttot;
    int ldp;
    int ntiles;
    int dilation;
    int dbg;   // ablation switches for profiling only (PK_PWG_ABLATE env var); 0 in production
    // scaled split-fp16 path (k_pwg_layer_b3<., true>) only -- see "block scaling" below
    const unsigned* xe_in;   // [Ttot/32] bits of max|x| per 32-sample block of xin (0 in the gaps)
    unsigned* xe_out;        // the same for xout, written by this launch
    int k1;                  // W1 fragments hold conv.weight * 2^k1
    float i0, i1;            // sqrt(0.5) / (2^14 * 2^k2out), 1 / (2^14 * 2^k2skip): undo the stage-2 scales
    const int* tile_kx_in;   // PL: [ntiles] scale exponent of xin / of xout per tile (k_pwg_tile_scales)
    const int* tile_kx_out;
    const float* noise;      // NZ: the packed noise of the tiles (a wave tile wt covers noise[32 wt .. 32 wt + 31]) ...
    const float* nz_tab;     // ... and its image for the weight region of the LDS: one k-step of A fragments, then [64][w b] (see k_pwg_layer_b3's NZ)
    float nz_wmax, nz_bmax;  // NZ: max|w|, max|b| of first_conv (the B operand is built from wmax n and bmax)
    long noise_n;            // NZ: floats in noise (index clamp)
    unsigned* amax_out;      // PL + AMAX (guarded / sampled calls): [B][PWG_AMAX_PARTS] bits of max|xout| per utterance, folded in by the epilogue
    PwgGen gen;              // GEN kernels only (hop != 256)
};

// tanh(a) * sigmoid(b) (:309-310).  exp via v_exp_f32; |a| clamped where tanh is +-1 in fp32.
__device__ __forceinline__ float gated(float a, float b) {
    a = __builtin_amdgcn_fmed3f(a, -10.f, 10.f);
    const float ea = __expf(-2.f * a);
    const float eb = __expf(-b);
    // v_rcp_f32 (1 ulp) instead of an IEEE divide: the gate is VALU time that the other wave's MFMAs
    // only partly hide (ablation: -0.13 ms per layer launch with the gate removed)
    return (1.f - ea) * __builtin_amdgcn_rcpf((1.f + ea) * (1.f + eb));
}

// ---------------------------------------------------------------- block scaling of the split-fp16 operands
// (pk_split.h).  Here: weights one exponent per tensor; x one exponent per wave tile, from the per-block max|x|
// that the producer of x (k_pwg_first / the previous layer's epilogue) leaves in xe[]; z = tanh * sigmoid with the
// fixed 2^14, folded into the gate's last multiply.  tests/test_pwg_gpu.py::test_pwg_split_math_is_scale_invariant
constexpr float PK_Z_SCALE = PK_UNIT_SCALE;
// max over the 64 lanes of a wave, returned wave-uniform.  DPP only (no LDS-pipe round trips as with ds_bpermute):
// quad swaps, two rotations inside the 16-lane row (max is idempotent, so after them every lane holds its row's
// maximum), then row_bcast15 / row_bcast31 carry the running maximum into rows 1, 3 and 2, 3: row 3 holds the result.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_max_step(float v) {
    const int t = __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
    return fmaxf(v, __int_as_float(t));
}
__device__ __forceinline__ float wave_max64(float v) {
    v = dpp_max_step<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
    v = dpp_max_step<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
    v = dpp_max_step<0x124, 0xf>(v);   // row_ror:4
    v = dpp_max_step<0x128, 0xf>(v);   // row_ror:8
    v = dpp_max_step<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
    v = dpp_max_step<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// 2^14 * tanh(a / S) * sigmoid(b / S) from accumulators that hold S * (pre-activation): cb = -log2(e) / S, the
// clamp of gated() moves behind the multiply (|2 a log2 e| <= 20 log2 e).  Same instruction count as gated().
__device__ __forceinline__ float gated_s(float a, float b, float ca, float cb) {   // ca = 2 cb
    const float ta = __builtin_amdgcn_fmed3f(a * ca, -28.853900817779268f, 28.853900817779268f);
    const float ea = __builtin_amdgcn_exp2f(ta);
    const float eb = __builtin_amdgcn_exp2f(b * cb);
    return fmaf(ea, -PK_Z_SCALE, PK_Z_SCALE) * __builtin_amdgcn_rcpf((1.f + ea) * (1.f + eb));
}

// two gates at once on packed fp32 math (v_pk_mul / v_pk_add / v_pk_fma process a register pair per issue slot; the
// clamp and the transcendentals stay per element): 15 instructions per pair instead of ~20
__device__ __forceinline__ f32x2 gated_s2(f32x2 a, f32x2 b, float ca, float cb) {
    f32x2 ta = a * ca, tb = b * cb;
    ta[0] = __builtin_amdgcn_fmed3f(ta[0], -28.853900817779268f, 28.853900817779268f);
    ta[1] = __builtin_amdgcn_fmed3f(ta[1], -28.853900817779268f, 28.853900817779268f);
    f32x2 ea, eb, rc;
    ea[0] = __builtin_amdgcn_exp2f(ta[0]);
    ea[1] = __builtin_amdgcn_exp2f(ta[1]);
    eb[0] = __builtin_amdgcn_exp2f(tb[0]);
    eb[1] = __builtin_amdgcn_exp2f(tb[1]);
    const f32x2 num = ea * (-PK_Z_SCALE) + PK_Z_SCALE;
    const f32x2 den = (ea + 1.f) * (eb + 1.f);
    rc[0] = __builtin_amdgcn_rcpf(den[0]);
    rc[1] = __builtin_amdgcn_rcpf(den[1]);
    return num * rc;
}

constexpr int LDS_W1 = KS1 * 64 * 4;          // 24576 floats
constexpr int LDS_W2 = KS2 * 64 * 4;          //  8192
constexpr int LDS_BIAS = G + R + SK;          //   256
constexpr int LDS_PW = UPW * G;               //   640 per wave
constexpr int LDS_PW_GEN = (UPW + 1) * G;     //   768 per wave: one more row when a wave tile straddles two frames (GEN)
constexpr int LAYER_WAVES = 8;                // waves per workgroup (2 per SIMD; 12 = 3 per SIMD spills at 168 VGPRs and measured slower)
constexpr int LDS_TOTAL = LDS_W1 + LDS_W2 + LDS_BIAS + LAYER_WAVES * LDS_PW;   // 38144 floats = 152 576 B
constexpr int LDS_TOTAL_GEN = LDS_W1 + LDS_W2 + LDS_BIAS + LAYER_WAVES * LDS_PW_GEN;   // 39168 floats = 156 672 B

// One residual block for every tile of the batch.  Persistent workgroups of 8
// waves (2 per SIMD, no barrier after the weight load: the two waves of a SIMD
// drift apart so one's gate/epilogue VALU+memory phase hides under the other's
// MFMA phase).  Each wave owns 32 consecutive samples and ALL channels for them:
//   init     acc[4] = conv bias + composite-upsampled aux projection (5 frame taps)
//   stage 1  acc[4] (4 x 32 gate channels) += W1 (A, LDS) x dilated x taps (B, global/L2)
//   gate     z = tanh(acc[0..1]) * sigmoid(acc[2..3])   -- stays in the accumulator registers
//   stage 2  acc2[4] (out 0-31, out 32-63, skip 0-31, skip 32-63) += W2 (A, LDS) x z (B)
// The K order of stage 2 is permuted on the host so that accumulator register r of
// stage 1 IS the B operand of k-step r of stage 2 (no data movement between the GEMMs).
template <bool FIRST, bool GEN = false>
__global__ __launch_bounds__(LAYER_WAVES * 64, LAYER_WAVES / 4) void k_pwg_layer(PwgLayerArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[GEN ? LDS_TOTAL_GEN : LDS_TOTAL];
    float* lds_bias = lds + LDS_W1 + LDS_W2;
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(a.w1);
        f32x4* dst = reinterpret_cast<f32x4*>(lds);
        for (int i = threadIdx.x; i < KS1 * 64; i += LAYER_WAVES * 64) dst[i] = src[i];
        const f32x4* src2 = reinterpret_cast<const f32x4*>(a.w2);
        f32x4* dst2 = reinterpret_cast<f32x4*>(lds + LDS_W1);
        for (int i = threadIdx.x; i < KS2 * 64; i += LAYER_WAVES * 64) dst2[i] = src2[i];
        if (threadIdx.x < LDS_BIAS) lds_bias[threadIdx.x] = a.bias[threadIdx.x];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int j = lane & 31;
    const int hi = lane >> 5;
    const int d = a.dilation;
    const f32x4* lds_a = reinterpret_cast<const f32x4*>(lds) + lane;
    float* lds_p = lds + LDS_W1 + LDS_W2 + LDS_BIAS + wave * (GEN ? LDS_PW_GEN : LDS_PW);  // wave-private staging

    // Work unit = one wave-tile of 32 samples (8 per frame); waves are independent, so a workgroup
    // is just LAYER_WAVES of them sharing the LDS-resident weights.  Order: workgroup b is dispatched
    // to XCD b % 8 (observed, speed only); each XCD gets a contiguous run of wave-tiles per sweep so
    // the +-dilation taps of a tile are fetched by CUs that share its L2.
    const int per_xcd = gridDim.x >> 3;
    const int wg_slot = (gridDim.x & 7) == 0 ? (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3)
                                             : (int)blockIdx.x;
    const int my_slot = wg_slot * LAYER_WAVES + wave;
    const int stride_slots = (int)gridDim.x * LAYER_WAVES;
    const int n_wtiles = a.ntiles * (TILE / WAVE_T);
    constexpr int GRP = 8;
    constexpr int NGRP = KS1 / GRP;  // 12 groups of 8 k-steps: 4 groups of 8 channel pairs per tap
    // K order = (channel group, tap): the three reads of an x line (as tap -1, 0, +1 of three different
    // tiles) then happen at nearly the same progress point of those tiles, i.e. close in time -> the
    // line is still in the XCD's L2 for the 2nd and 3rd read (the L2 only holds ~3 us of this stream).
    auto group_ptr = [&](int g) -> const float* {   // wave-uniform
        const int cg = g / 3, tap = g - 3 * cg;
        (void)tap;   // the tap shift moves lanes across blocks: it lives in the lane offset
        return a.xin + (long)(2 * GRP * cg) * XBLK;
    };
    auto group_tap = [&](int g) -> int { return g % 3; };

    // Software pipeline across tiles: everything a tile needs before its first MFMA (aux projection
    // rows, upsampler weights, the first operand group) is requested while the previous tile is still
    // in its MFMA / epilogue phases.
    f32x4 preg[3];
    float uw[UPW];
    float bA[GRP], bB[GRP];
    int df_n = 0;          // GEN: this lane's frame minus the first staged frame (0 / 1) of the tile being prefetched
    bool valid_n = true;   // GEN: the lane's sample lies inside its utterance
    auto prefetch_head = [&](int wt) {
        const int tile = wt >> 3, phase = (wt & 7) * WAVE_T + j;
        const float* prow;
        long wsel;   // row of the upsampler table: edge class * hop + phase
        if constexpr (GEN) {
            const PwgGenCoord gc = gen_coord(a.gen, tile, wt & 7, j);
            prow = a.gen.P0 + (long)(gc.row0 + gc.fa - 2) * a.ldp;
            wsel = (long)gc.cls * a.gen.hop + gc.phase;
            df_n = gc.df;
            valid_n = gc.valid;
        } else {
            prow = a.P + (long)(tile - 2) * a.ldp;
        }
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int idx = lane + 64 * it;
            const int jj = idx >> 5, c4 = idx & 31;
            if (idx < (UPW + (GEN ? 1 : 0)) * (G / 4)) preg[it] = *reinterpret_cast<const f32x4*>(prow + (long)jj * a.ldp + 4 * c4);
        }
        if constexpr (!GEN) wsel = (long)a.tile_cls[tile] * TILE + phase;
        const float* wrow = a.uptab + wsel * UPW_PAD;
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wrow);
        uw[0] = w0[0]; uw[1] = w0[1]; uw[2] = w0[2]; uw[3] = w0[3];
        uw[4] = wrow[4];
    };
    auto load_group0 = [&](int wt) {
        const long tt = (long)(a.tile_t0[wt >> 3] & ~255) + (wt & 7) * WAVE_T + j;
        const unsigned vo = (unsigned)(xoff(tt - d) + hi * XBLK);   // group 0 = (cg 0, tap -1)
        const float* p = group_ptr(0);
#pragma unroll
        for (int s = 0; s < GRP; ++s) bA[s] = (p + (long)(2 * s) * XBLK)[vo];
    };
    if (my_slot < n_wtiles) {
        prefetch_head(my_slot);
        load_group0(my_slot);
    }

    for (int wt = my_slot; wt < n_wtiles; wt += stride_slots) {
        const int next_tile = wt + stride_slots < n_wtiles ? wt + stride_slots : wt;
        const int tile = wt >> 3;
        const int phase = (wt & 7) * WAVE_T + j;
        // addresses = wave-uniform row pointer (SGPR pair) + one 32-bit per-lane element offset,
        // so a load costs no address VGPRs: lane offset = block/sample offset of (t + tap shift) + its row part
        const long t = (long)(a.tile_t0[tile] & ~255) + phase;
        unsigned vo1t[3];                                             // B operand rows 2*cp + hi, per tap
#pragma unroll
        for (int tp = 0; tp < 3; ++tp) vo1t[tp] = (unsigned)(xoff(t + (long)(tp - 1) * d) + hi * XBLK);
        const unsigned vo4 = (unsigned)(xoff(t) + 4 * hi * XBLK);      // result rows mfma_row(r, hi)

        // aux projection rows f-2..f+2 (UPW x G floats) -> wave-private LDS (no barrier: same wave)
        // [wave-lds-exchange] every lane of the wave has finished reading the previous tile's rows out of lds_p
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int idx = lane + 64 * it;
            if (idx < (UPW + (GEN ? 1 : 0)) * (G / 4)) reinterpret_cast<f32x4*>(lds_p)[idx] = preg[it];
        }
        // [wave-lds-exchange] the rows are read below by OTHER lanes of the same wave (a wave's LDS instructions execute in
        // order for all its lanes, so the hardware needs no barrier here; tools/hipemu turns these markers into a rendezvous)
        const bool lane_valid = valid_n;                 // of THIS tile (prefetch_head(next) overwrites valid_n)
        const float* lds_pl = lds_p + (GEN ? df_n * G : 0);   // GEN: lanes in the tile's second frame read one row on
        // accumulators start from conv bias + upsampled aux projection
        f32x16 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int co0 = 32 * q + 8 * r4 + 4 * hi;
                f32x4 v = *reinterpret_cast<const f32x4*>(lds_bias + co0);
#pragma unroll
                for (int jj = 0; jj < UPW; ++jj) {
                    const f32x4 pv = *reinterpret_cast<const f32x4*>(lds_pl + jj * G + co0);
                    v[0] = fmaf(uw[jj], pv[0], v[0]);
                    v[1] = fmaf(uw[jj], pv[1], v[1]);
                    v[2] = fmaf(uw[jj], pv[2], v[2]);
                    v[3] = fmaf(uw[jj], pv[3], v[3]);
                }
                acc[q][4 * r4 + 0] = v[0];
                acc[q][4 * r4 + 1] = v[1];
                acc[q][4 * r4 + 2] = v[2];
                acc[q][4 * r4 + 3] = v[3];
            }

        // K loop of stage 1, two register sets in ping-pong: the B values of group g+1 are issued
        // BEFORE group g's 32 MFMAs (2048 matrix-pipe cycles) and first touched after them.  The
        // sched_barriers pin that order -- left alone, hipcc sinks the loads to just before their use
        // and every group eats a full memory round trip (measured: SQ_WAIT_ANY 54 % of wave cycles).
#pragma unroll 1
        for (int gg = 0; gg < NGRP / 2; ++gg) {
            {
                const float* p = group_ptr(2 * gg + 1);
                const unsigned vo1 = vo1t[group_tap(2 * gg + 1)];
#pragma unroll
                for (int s = 0; s < GRP; ++s) bB[s] = (p + (long)(2 * s) * XBLK)[vo1];
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                const f32x4* la = lds_a + (long)(2 * gg) * GRP * 64;
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int s = 0; s < GRP; ++s) {
                    const f32x4 af = la[s * 64];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q], bA[s], acc[q], 0, 0, 0);
                }
                __builtin_amdgcn_s_setprio(0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (2 * gg + 2 < NGRP) {
                const float* p = group_ptr(2 * gg + 2);
                const unsigned vo1 = vo1t[group_tap(2 * gg + 2)];
#pragma unroll
                for (int s = 0; s < GRP; ++s) bA[s] = (p + (long)(2 * s) * XBLK)[vo1];
            } else {
                load_group0(next_tile);      // the next tile's first operand group
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                const f32x4* la = lds_a + (long)(2 * gg + 1) * GRP * 64;
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int s = 0; s < GRP; ++s) {
                    const f32x4 af = la[s * 64];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q], bB[s], acc[q], 0, 0, 0);
                }
                __builtin_amdgcn_s_setprio(0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        prefetch_head(next_tile);            // next tile's aux rows + upsampler weights
        __builtin_amdgcn_sched_barrier(0);

        // The gated activation z = tanh(a) * sigmoid(b) (overwriting acc[0], acc[1]) is computed inside
        // pass 0 of stage 2, one k-step ahead of the MFMAs that consume it, so its VALU / transcendental
        // work runs under this wave's own matrix instructions.
        const bool do_gate = !(a.dbg & 1);
        // Stage 2 in two passes (out, then skip) so that only 32 old values + 32 accumulators are
        // live next to z: leaves registers for the next tile's prefetched head.  The old values of a
        // pass are requested before its 64 MFMAs (4096 matrix-pipe cycles) and consumed after them --
        // placed next to their stores they would be serialised (the pointers may alias for hipcc).
        const float rs = 0.70710678118654752440f;
        const f32x2* lds_w2 = reinterpret_cast<const f32x2*>(lds + LDS_W1) + 2 * lane;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            float old[32];
            const float* src = pass == 0 ? a.xin : a.skip;
            if ((pass == 0 || !FIRST) && !(a.dbg & 4)) {
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        old[16 * q + r] = (src + (long)(32 * q + mfma_row(r, 0)) * XBLK)[vo4];
            } else {
#pragma unroll
                for (int e = 0; e < 32; ++e) old[e] = 0.f;
            }
            __builtin_amdgcn_sched_barrier(0);
            f32x16 acc2[2];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const f32x4 bv =
                        *reinterpret_cast<const f32x4*>(lds_bias + G + 64 * pass + 32 * q + 8 * r4 + 4 * hi);
                    acc2[q][4 * r4 + 0] = bv[0];
                    acc2[q][4 * r4 + 1] = bv[1];
                    acc2[q][4 * r4 + 2] = bv[2];
                    acc2[q][4 * r4 + 3] = bv[3];
                }
            if (pass == 0 && do_gate) acc[0][0] = gated(acc[0][0], acc[2][0]);
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) {
                const f32x2 af = lds_w2[ks * 128 + pass];
                acc2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0], acc[ks >> 4][ks & 15], acc2[0], 0, 0, 0);
                if (pass == 0 && do_gate && ks + 1 < KS2) {
                    const int kn = ks + 1;   // gate of the next k-step, under the two MFMAs of this one
                    acc[kn >> 4][kn & 15] = gated(acc[kn >> 4][kn & 15], acc[(kn >> 4) + 2][kn & 15]);
                }
                acc2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1], acc[ks >> 4][ks & 15], acc2[1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            float* dst = pass == 0 ? a.xout : a.skip;
            if (a.dbg & 2) {
                // keep the results alive without the stores
                float keep = 0.f;
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int r = 0; r < 16; ++r) keep += acc2[q][r] + old[16 * q + r];
                if (keep == 1.2345e-30f) dst[vo4] = keep;
                continue;
            }
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v;
                    if (pass == 0) v = (acc2[q][r] + old[16 * q + r]) * rs;       // res = (out + x_in) * sqrt(0.5) (:314)
                    else v = FIRST ? acc2[q][r] : (old[16 * q + r] + acc2[q][r]);  // skips += skip (:468)
                    if (GEN && pass == 0 && !lane_valid) v = 0.f;   // beyond the utterance: stays zero padding
                    (dst + (long)(32 * q + mfma_row(r, 0)) * XBLK)[vo4] = v;
                }
        }
    }
}

// ---------------------------------------------------------------- residual block, split-bf16 matrix path
// Same data flow as k_pwg_layer, but every fp32 product a*b of the two contractions is evaluated as
//     a_hi*b_hi + a_lo*b_hi + a_hi*b_lo      (x = x_hi + x_lo + O(2^-17 x), hi/lo = bf16 roundings)
// on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; activations and weights stay fp32 in HBM, the
// split of the activations happens in registers (v_cvt_pk_bf16_f32), the weights are split once at
// finalize.  The dropped a_lo*b_lo term is ~2^-16 of a product; measured end-to-end effect on the
// 30-layer generator: relative max error 3e-6 vs fp64 (exact-fp32 path: 5e-7; tolerance 1e-4).
// 3 bf16 MFMAs (32 cycles, K = 16) replace 8 fp32 MFMAs (64 cycles, K = 2): 5.3x less matrix-pipe
// time, which makes the kernel HBM-bound (x in/out + skip read-modify-write).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <bool HALF> struct Split16 { typedef bf16x8 vec; typedef __bf16 elem; };
template <> struct Split16<true> { typedef f16x8 vec; typedef _Float16 elem; };
__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma16(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
constexpr int B3_KS1 = KTAP * R / 16;   // 12 k-steps of 16 channels
constexpr int B3_KS2 = (G / 2) / 16;    // 4
constexpr int B3_W1_BYTES = B3_KS1 * 2 * 4 * 64 * 16;   // [ks][part][co-tile][lane] x 8 bf16 = 98 304 B
constexpr int B3_W2_BYTES = B3_KS2 * 2 * 4 * 64 * 16;   // 32 768 B
constexpr int B3_RING = 4;              // operand groups in flight ahead of the MFMAs (6 spills at 256 VGPRs)

template <class V, class E, bool CLAMP>
__device__ __forceinline__ void split_x8(const float (&v)[8], V& hi, V& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        hi[e] = (E)v[e];
        lo[e] = (E)(v[e] - (float)hi[e]);
    }
}
// fp16 parts: the high part is rounded toward zero by v_cvt_pkrtz_f16_f32 (two elements per instruction;
// it saturates at +-65504 instead of overflowing to inf, so no clamp is needed), the remainder x - hi is formed
// exactly by v_fma_mix_f32 reading the packed fp16 high part in place (one instruction per element instead of
// v_cvt_f32_f16 + subtract: -1 % on the layer kernel, A/B on one box), the low part is its round-to-nearest
// fp16: |x - hi - lo| <= 2^-21 |x|.  2 VALU ops per element.
typedef __fp16 pkh2 __attribute__((ext_vector_type(2)));
template <>
__device__ __forceinline__ void split_x8<f16x8, _Float16, true>(const float (&v)[8], f16x8& hi, f16x8& lo) {
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

// split of s * x, s = the block's power of two (see "block scaling"): one v_pk_mul_f32 per pair on top of split_x8
__device__ __forceinline__ void split_x8s(const float (&v)[8], float s, f16x8& hi, f16x8& lo) {
    float t[8];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        f32x2 u = {v[2 * p], v[2 * p + 1]};
        u *= s;
        t[2 * p] = u[0];
        t[2 * p + 1] = u[1];
    }
    split_x8<f16x8, _Float16, true>(t, hi, lo);
}

// HALF = false: bf16 parts (8 significant bits each, fp32 range); HALF = true: fp16 parts (11 bits each,
// 22 bits per operand ~ fp32's 24) of block-scaled operands (see "block scaling" above: no subnormal parts, no
// dependence on the magnitude of weights or activations; |x| beyond 2^113 aside).
// ABL (profiling only, PK_PWG_ABLATE, results are wrong): 1 = no global loads / stores of x and skip (compute-only time);
// 2 = the x taps go to the MFMA as loaded, without the hi / lo split (what storing x pre-split would save); 4 / 8 (with 2):
// the taps loaded as two 16-byte vectors per group instead of eight dwords, in the two candidate planes layouts
// PL (HALF only): x lives in HBM as pre-split fp16 planes -- per 32-sample block [octet 8][sample 32][hi 8 | lo 8] halves (the
// same 8 KB) --, written so by the producer's epilogue, and an operand group is two 16-byte loads that go to the MFMAs as they
// are: no arithmetic between load and MFMA.  (The scale-and-split of the fp32 taps, 290 of the tile's 1 400 vector
// instructions, stood on the load -> split -> MFMA path of every k-step: 20 % of the kernel, PK_PWG_ABLATE=32.  A first
// planes version kept the per-block scales and rescaled each tap's vectors to the tile's common scale with v_pk_mul_f16:
// 1.39 instead of 1.41 ms -- any vector instruction on that path costs about the same.)  That needs ONE scale for everything
// a tile's taps read, known to the producer BEFORE it has seen its output: the scale of a layer's x is per utterance and
// a priori, from a magnitude bound B_l -- B_0 = max|w| max|noise| + max|b| (first_conv), B_(l+1) = (B_l + c_l) sqrt(0.5)
// with c_l = max_co (sum_k |W_out[co][k]| + |b_out[co]|) because |z| < 1 (k_pwg_tile_scales).  The bound overshoots the
// utterance's maximum by a small factor, i.e. the split's error floor moves from 2^-39 of a 32-sample block's maximum to
// about 2^-36 of the utterance's: elements more than 2^12 below it lose against fp32, by less than 2^-36 of it.
// NZ (round 6, "noise-fed first block"): layer 0 without its x planes.  first_conv is Conv1D(1 -> 64, k = 1), so x = w n + b inside an utterance and 0 in
// the gaps, and layer 0's dilated conv over x is a 3-tap conv on the scalar noise: sum_tap (u[co][tap] n[t + tap - 1] + v[co][tap]) over the taps inside the
// utterance, u = sum_c W1[co][c][tap] w[c], v = sum_c W1[co][c][tap] b[c] folded in fp64 by pk_pwg_finalize.  Stage 1's 12 k-steps become ONE (the six
// "channels" n(t-1), n(t), n(t+1) and the three in-utterance flags): 12 MFMAs instead of 144, no plane reads (768 B/sample), no k_pwg_first launch (1.34 GB
// written); the residual input w n + b is recomputed.  hop 256, planes path only.
template <bool FIRST, bool HALF, int ABL = 0, bool GEN = false, bool PL = false, bool AMAX = false, bool NZ = false>
__global__ __launch_bounds__(LAYER_WAVES * 64, LAYER_WAVES / 4) void k_pwg_layer_b3(PwgLayerArgs a) {
    static_assert(!NZ || (FIRST && PL && !GEN), "noise-fed: the first block of the planes path at hop 256");
    static_assert(!PL || (HALF && ABL == 0), "planes: the block-scaled split-fp16 path only");
    typedef typename Split16<HALF>::vec bf16x8;     // shadows the bf16 typedef inside this kernel
    typedef typename Split16<HALF>::elem elem16;
    __shared__ __attribute__((aligned(16))) float lds[GEN ? LDS_TOTAL_GEN : LDS_TOTAL];
    float* lds_bias = lds + LDS_W1 + LDS_W2;
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(NZ ? a.nz_tab : a.w1);
        f32x4* dst = reinterpret_cast<f32x4*>(lds);
        for (int i = threadIdx.x; i < (NZ ? (2048 + 128) * 4 : B3_W1_BYTES) / 16; i += LAYER_WAVES * 64) dst[i] = src[i];   // NZ: one k-step of fragments, then [64][w b]
        const f32x4* src2 = reinterpret_cast<const f32x4*>(a.w2);
        f32x4* dst2 = reinterpret_cast<f32x4*>(lds + LDS_W1);
        for (int i = threadIdx.x; i < B3_W2_BYTES / 16; i += LAYER_WAVES * 64) dst2[i] = src2[i];
        if (threadIdx.x < LDS_BIAS) lds_bias[threadIdx.x] = a.bias[threadIdx.x];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int j = lane & 31;
    const int hi = lane >> 5;
    const int d = a.dilation;
    const bf16x8* lds_a = reinterpret_cast<const bf16x8*>(lds) + lane;            // + ((ks*2+part)*4+q)*64
    const bf16x8* lds_a2 = reinterpret_cast<const bf16x8*>(lds + LDS_W1) + lane;
    float* lds_p = lds + LDS_W1 + LDS_W2 + LDS_BIAS + wave * (GEN ? LDS_PW_GEN : LDS_PW);

    const int per_xcd = gridDim.x >> 3;
    const int wg_slot = (gridDim.x & 7) == 0 ? (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3)
                                             : (int)blockIdx.x;
    int my_slot = wg_slot * LAYER_WAVES + wave;
#if PK_PWG_TILE_MAP
    // which wave of the XCD's window of 8 * per_xcd wave tiles takes which tile (PK_PWG_TILE_MAP above)
    if ((gridDim.x & 7) == 0) {
        const int wgl = (int)(blockIdx.x >> 3), win = per_xcd * LAYER_WAVES;
        const int local = PK_PWG_TILE_MAP == 1 ? (wave >> 2) * (win >> 1) + wgl * 4 + (wave & 3)      // first / second wave of the SIMDs: halves
                        : PK_PWG_TILE_MAP == 2 ? wave * per_xcd + wgl                                  // wave-major
                                               : (wave & 3) * (win >> 2) + wgl * 2 + (wave >> 2);      // SIMD-major
        my_slot = (int)(blockIdx.x & 7) * win + local;
    }
#endif
    const int stride_slots = (int)gridDim.x * LAYER_WAVES;
    const int n_wtiles = a.ntiles * (TILE / WAVE_T);

    // block scaling (HALF): lanes 0..4 fetch max|x| of the 32-sample blocks the three taps of a wave tile touch
    // (d < 32: the blocks before / at / after the tile; d >= 32, a multiple of 32: exactly the blocks at -d, 0, +d)
    int eoff = 0;
    if (HALF) {
        const int dc = (d + 31) >> 5, df = d >> 5;
        eoff = lane == 0 ? -dc : (lane == 1 ? -df : (lane == 3 ? df : (lane == 4 ? dc : 0)));
    }
    // (t0 = the tile's tile_t0 entry: timeline offset | edge class.  It is loaded ONE TILE AHEAD -- a load whose
    // result feeds an address makes hipcc wait with vmcnt(0), i.e. for every prefetched operand in flight behind it;
    // round 1 paid that twice per tile: tile_t0 at the top of the tile, tile_cls in prefetch_head)
    auto load_amax = [&](int t0, int wt) -> unsigned {
        const long tt0 = (long)(t0 & ~255) + (wt & 7) * WAVE_T;
        return (ABL & 1) ? 0x3f800000u : a.xe_in[(tt0 >> 5) + eoff];
    };
    auto tile_scale_exp = [&](unsigned ev) -> int {   // wave-uniform (SGPR) exponent k of the tile's x scale 2^k
        unsigned m = (unsigned)__builtin_amdgcn_readlane((int)ev, 0);
        const unsigned m1 = (unsigned)__builtin_amdgcn_readlane((int)ev, 1);
        const unsigned m2 = (unsigned)__builtin_amdgcn_readlane((int)ev, 2);
        const unsigned m3 = (unsigned)__builtin_amdgcn_readlane((int)ev, 3);
        const unsigned m4 = (unsigned)__builtin_amdgcn_readlane((int)ev, 4);
        m = m > m1 ? m : m1;
        m = m > m2 ? m : m2;
        m = m > m3 ? m : m3;
        m = m > m4 ? m : m4;
        return blk_scale_exp(m);
    };
    int kx = 0, kx_next = 0;   // x-scale exponents of the current / the next wave tile
    float xs_cur = 0.f;   // PL: sqrt(0.5) / (the scale of this tile's x_in), for the residual input
    int kxn_v = 0;        // PL: the next tile's x-scale exponent as loaded (one tile ahead, like t0n)

    // operand group g of a wave-tile = k-step g = (channel group cg = g/3, tap = g%3).  Element e of lane
    // (j, hi) is input channel 32*(cg>>1) + mfma_row(8*(cg&1) + e, hi) -- the SAME channel the lane owns as
    // output row in the epilogue, so the centre-tap operands double as the residual input x_in (no reload:
    // by the time of the epilogue those lines have left the L2 and would come from HBM again).
    auto group_row = [&](int g, int e) -> long {   // wave-uniform part of the channel row
        const int cg = g / 3;
        return (long)(32 * (cg >> 1) + mfma_row(8 * (cg & 1) + e, 0));
    };
    // lane offsets of a wave-tile for the three taps (the shift moves lanes across 32-sample blocks)
    auto lane_off = [&](int t0, int wt, int tap) -> unsigned {
        const long tt = (long)(t0 & ~255) + (wt & 7) * WAVE_T + j + (long)(tap - 1) * d;
        return (unsigned)(xoff(tt) + 4 * hi * XBLK);
    };
    // PL: byte offset of the lane's (hi, lo) vectors of k-group 0 in the planes (+ 2048 per k-group)
    auto lane_off_pl = [&](int t0, int wt, int tap) -> unsigned {
        const long tt = (long)(t0 & ~255) + (wt & 7) * WAVE_T + j + (long)(tap - 1) * d;
        return (unsigned)((tt >> 5) * (long)(XBLK_FLOATS * 4) + (tt & 31) * 32 + hi * 1024);
    };
    float ring[B3_RING][8];   // PL: the raw (hi | lo) vectors of a group, four registers each
    auto load_group = [&](float (&dst)[8], int gt, unsigned vo, unsigned pvo) {
        if constexpr (PL) {
            const f32x4* src = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(a.xin) + (pvo + (unsigned)((gt / 3) * 2048)));
            const f32x4 h4 = src[0], l4 = src[1];
            dst[0] = h4[0]; dst[1] = h4[1]; dst[2] = h4[2]; dst[3] = h4[3];
            dst[4] = l4[0]; dst[5] = l4[1]; dst[6] = l4[2]; dst[7] = l4[3];
        } else if constexpr ((ABL & 12) != 0) {
            // profiling only (with ABL & 2): the group as two 16-byte loads of the planes' shape from the fp32 buffer -- 4: hi and lo
            // of a sample adjacent (32 bytes per sample), 8: the planes apart (16 bytes per sample, lo 512 bytes on)
            const unsigned bo = (vo >> 11) * 8192u + (unsigned)(hi * 1024 + (gt / 3) * 2048) + (vo & 31) * ((ABL & 4) ? 32u : 16u);
            const f32x4* src = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(a.xin) + bo);
            const f32x4 h4 = src[0], l4 = src[(ABL & 4) ? 1 : 32];
            dst[0] = h4[0]; dst[1] = h4[1]; dst[2] = h4[2]; dst[3] = h4[3];
            dst[4] = l4[0]; dst[5] = l4[1]; dst[6] = l4[2]; dst[7] = l4[3];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) dst[e] = (ABL & 1) ? (float)(lane + e + gt) * 1e-3f : (a.xin + group_row(gt, e) * XBLK)[vo];
        }
    };
    auto planes_operand = [&](const float (&rr)[8], bf16x8& oh, bf16x8& ol) {   // PL: the vectors as loaded
        if constexpr (PL) {
            oh = __builtin_bit_cast(pl_f16x8, f32x4{rr[0], rr[1], rr[2], rr[3]});
            ol = __builtin_bit_cast(pl_f16x8, f32x4{rr[4], rr[5], rr[6], rr[7]});
        }
    };
    bf16x8 ph, pl;     // split operands of the k-step about to run (produced one step ahead, under the MFMAs)
    f32x4 preg[3];
    float uw[UPW];
    int df_n = 0;          // GEN: this lane's frame minus the first staged frame (0 / 1) of the tile being prefetched
    bool valid_n = true;   // GEN: the lane's sample lies inside its utterance
    auto prefetch_head = [&](int wt, int cls) {   // cls: edge class of the tile (hop 256), known a tile ahead
        const int tile = wt >> 3, phase = (wt & 7) * WAVE_T + j;
        const float* prow;
        long wsel;   // row of the upsampler table: edge class * hop + phase
        if constexpr (GEN) {
            const PwgGenCoord gc = gen_coord(a.gen, tile, wt & 7, j);
            prow = a.gen.P0 + (long)(gc.row0 + gc.fa - 2) * a.ldp;
            wsel = (long)gc.cls * a.gen.hop + gc.phase;
            df_n = gc.df;
            valid_n = gc.valid;
        } else {
            prow = a.P + (long)(tile - 2) * a.ldp;
        }
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int idx = lane + 64 * it;
            const int jj = idx >> 5, c4 = idx & 31;
            if (idx < (UPW + (GEN ? 1 : 0)) * (G / 4)) preg[it] = *reinterpret_cast<const f32x4*>(prow + (long)jj * a.ldp + 4 * c4);
        }
        if constexpr (!GEN) wsel = (long)cls * TILE + phase;
        const float* wrow = a.uptab + wsel * UPW_PAD;
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wrow);
        uw[0] = w0[0]; uw[1] = w0[1]; uw[2] = w0[2]; uw[3] = w0[3];
        uw[4] = wrow[4];
    };
    // NZ: the lane's noise at t - 1, t, t + 1 of the NEXT tile (requested a tile ahead, like t0n), and the edge class of the current tile
    float nzn[3] = {0.f, 0.f, 0.f};
    int cls_cur = 0;
    auto nz_load = [&](int wt_) {
        const long i0 = (long)wt_ * WAVE_T + j;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            long i = i0 + k - 1;
            i = i < 0 ? 0 : (i >= a.noise_n ? a.noise_n - 1 : i);   // (the clamped values belong to taps outside the utterance: masked where used)
            nzn[k] = a.noise[i];
        }
    };
    unsigned vo8n[3] = {0, 0, 0};   // lane offsets of the three taps of the NEXT tile (this tile's at the loop top)
    unsigned pvo8n[3] = {0, 0, 0};  // PL: the same in the planes
    if (my_slot < n_wtiles) {
        const int t0c = a.tile_t0[my_slot >> 3];
#pragma unroll
        for (int tp = 0; tp < 3; ++tp) {
            vo8n[tp] = lane_off(t0c, my_slot, tp);
            if constexpr (PL) pvo8n[tp] = lane_off_pl(t0c, my_slot, tp);
        }
        prefetch_head(my_slot, __builtin_amdgcn_readfirstlane(t0c & 255));
        if constexpr (NZ) {
            cls_cur = __builtin_amdgcn_readfirstlane(t0c & 255);
            kx = __builtin_amdgcn_readfirstlane(a.tile_kx_in[my_slot >> 3]);
            nz_load(my_slot);
        }
#pragma unroll
        for (int g = 0; g < (NZ ? 0 : B3_RING); ++g) load_group(ring[g], g, vo8n[g % 3], pvo8n[g % 3]);
        if constexpr (NZ) {
        } else if constexpr (PL) {
            kx = __builtin_amdgcn_readfirstlane(a.tile_kx_in[my_slot >> 3]);
            planes_operand(ring[0], ph, pl);
        } else if constexpr (HALF) {
            kx = tile_scale_exp(load_amax(t0c, my_slot));
            if constexpr ((ABL & 2) != 0) {   // what pre-split storage of x would leave: no arithmetic between the load and the MFMA
                ph = __builtin_bit_cast(bf16x8, f32x4{ring[0][0], ring[0][1], ring[0][2], ring[0][3]});
                pl = __builtin_bit_cast(bf16x8, f32x4{ring[0][4], ring[0][5], ring[0][6], ring[0][7]});
            } else
            split_x8s(ring[0], pow2f(kx), ph, pl);
        } else {
            split_x8<bf16x8, elem16, HALF>(ring[0], ph, pl);
        }
    }
    // W1 fragments of the next (k-step, co-tile) in issue order, read one co-tile ahead of their MFMAs
    bf16x8 c_ah, c_al;
    if constexpr (!NZ) {
        c_ah = lds_a[0];
        c_al = lds_a[4 * 64];
    }

    // Invariant at the top of k-step g: (ph, pl) = split operands of group g; ring slots (g+1..g+RING-1) % RING
    // hold groups g+1..g+RING-1 (of this tile, continuing into the next one); slot g % RING is free.
    int amax_utt = -1;        // PL + AMAX: the utterance of the running maximum (wave-uniform) ...
    unsigned amax_run = 0u;   // ... and its bits
    for (int wt = my_slot; wt < n_wtiles; wt += stride_slots) {
        const int next_wt = wt + stride_slots < n_wtiles ? wt + stride_slots : wt;
        unsigned vo8[3], pvo8[3];
#pragma unroll
        for (int tp = 0; tp < 3; ++tp) {
            vo8[tp] = vo8n[tp];
            pvo8[tp] = pvo8n[tp];
        }
        const int t0n = a.tile_t0[next_wt >> 3];   // requested here, first used at k-step T0_USE of stage 1
        int ko_v = 0;   // PL: scale exponent of this tile's x_out
        int utt_v = 0;  // PL + AMAX: the tile's utterance
        if constexpr (PL) {
            kxn_v = a.tile_kx_in[next_wt >> 3];
            ko_v = a.tile_kx_out[wt >> 3];
            if constexpr (AMAX) utt_v = a.gen.tile_utt[wt >> 3];   // requested here like ko_v: a load in the epilogue would wait for every prefetched operand
            xs_cur = __uint_as_float(0x3f3504f3u - ((unsigned)kx << 23));   // sqrt(0.5) * 2^-kx
        }
        int cls_next = 0;
        const unsigned vo4 = vo8[1];   // centre tap: operand rows and result rows share the lane offset
        float x_old[32];
        float nz_cur[3] = {0.f, 0.f, 0.f};
        if constexpr (NZ) {
#pragma unroll
            for (int k = 0; k < 3; ++k) nz_cur[k] = nzn[k];
            nz_load(next_wt);
        }
        // [wave-lds-exchange] every lane of the wave has finished reading the previous tile's rows out of lds_p
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int idx = lane + 64 * it;
            if (idx < (UPW + (GEN ? 1 : 0)) * (G / 4)) reinterpret_cast<f32x4*>(lds_p)[idx] = preg[it];
        }
        // [wave-lds-exchange] the rows are read below by OTHER lanes of the same wave (a wave's LDS instructions execute in
        // order for all its lanes, so the hardware needs no barrier here; tools/hipemu turns these markers into a rendezvous)
        const bool lane_valid = valid_n;                 // of THIS tile (prefetch_head(next) overwrites valid_n)
        const float* lds_pl = lds_p + (GEN ? df_n * G : 0);   // GEN: lanes in the tile's second frame read one row on
        // HALF: the stage-1 accumulators hold S1 * (pre-activation), S1 = 2^(kx + k1) = x scale * W1 scale
        unsigned ev_next = 0;
        float sx = 1.f, gca = 0.f, gcb = 0.f;
        if constexpr (HALF) {
            const int ks1 = kx + a.k1;
            sx = pow2f(kx);
            const float S1 = pow2f(ks1);
#pragma unroll
            for (int jj = 0; jj < UPW; ++jj) uw[jj] *= S1;
            const int cbb = __builtin_amdgcn_readfirstlane(__float_as_int(-1.4426950408889634f * pow2f(-ks1)));
            gcb = __int_as_float(cbb);
            gca = __int_as_float(cbb + (1 << 23));   // 2 * gcb
        }
        const float S1b = HALF ? pow2f(kx + a.k1) : 1.f;
        f32x16 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int co0 = 32 * q + 8 * r4 + 4 * hi;
                f32x4 v = *reinterpret_cast<const f32x4*>(lds_bias + co0);
                if constexpr (HALF) v *= S1b;
#pragma unroll
                for (int jj = 0; jj < UPW; ++jj) {
                    const f32x4 pv = *reinterpret_cast<const f32x4*>(lds_pl + jj * G + co0);
                    v[0] = fmaf(uw[jj], pv[0], v[0]);
                    v[1] = fmaf(uw[jj], pv[1], v[1]);
                    v[2] = fmaf(uw[jj], pv[2], v[2]);
                    v[3] = fmaf(uw[jj], pv[3], v[3]);
                }
                acc[q][4 * r4 + 0] = v[0];
                acc[q][4 * r4 + 1] = v[1];
                acc[q][4 * r4 + 2] = v[2];
                acc[q][4 * r4 + 3] = v[3];
            }

        // stage 1: 12 k-steps.  Step g: refill the free ring slot with group g+RING, then run the 12 MFMAs of
        // group g while the VALU splits group g+1 (sched_group_barrier interleaves them: a 32x32x16 MFMA
        // occupies the matrix pipe for 8 issue slots, the split fits in the gaps).
        constexpr int T0_USE = 3;   // the next tile's offsets are formed here: before the ring reaches into it (g = 8)
        if constexpr (NZ) {
            // ONE k-step: the B operand's 16 "channels" are wmax n(t-1), wmax n(t), wmax n(t+1), bmax m(t-1), bmax m(t), bmax m(t+1), 0 ... (m = 1 inside the
            // utterance; |wmax n|, bmax <= the bound the tile's x scale was made for), at the tile's x scale like any plane; the A fragments hold
            // u / wmax, v / bmax * 2^k1 (a.k1 = their own exponent for this launch): the accumulators come out at S1 = 2^(kx + k1) like the init above
            const bool first_s = (cls_cur / 3 == 0) && (wt & 7) == 0 && j == 0;               // t - 1 lies before the utterance
            const bool last_s = (cls_cur % 3 == 0) && (wt & 7) == 7 && j == WAVE_T - 1;       // t + 1 lies behind it
            float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // lane half 0: channels 0..3 (e 0..3), half 1: channels 4..7
            bv[0] = hi ? a.nz_bmax : (first_s ? 0.f : a.nz_wmax * nz_cur[0]);
            bv[1] = hi ? (last_s ? 0.f : a.nz_bmax) : a.nz_wmax * nz_cur[1];
            bv[2] = hi ? 0.f : (last_s ? 0.f : a.nz_wmax * nz_cur[2]);
            bv[3] = hi ? 0.f : (first_s ? 0.f : a.nz_bmax);
            bf16x8 bh, bl;
            split_x8s(bv, sx, bh, bl);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bf16x8 ah = lds_a[(0 * 4 + q) * 64], al = lds_a[(1 * 4 + q) * 64];
                acc[q] = mfma16(ah, bh, acc[q]);
                acc[q] = mfma16(al, bh, acc[q]);
                acc[q] = mfma16(ah, bl, acc[q]);
            }
            // x_in sqrt(1/2) = (w n + b) sqrt(1/2) at this lane's output rows, from the [64][w b] table behind the fragments
            const f32x2* wb = reinterpret_cast<const f32x2*>(lds + 2048);
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const f32x2 t = wb[32 * q + mfma_row(r, hi)];
                    x_old[16 * q + r] = fmaf(t[0], nz_cur[1], t[1]) * 0.70710678118654752440f;
                }
            // what stage 1 does on the side for the next tile
#pragma unroll
            for (int tp = 0; tp < 3; ++tp) {
                vo8n[tp] = lane_off(t0n, next_wt, tp);
                pvo8n[tp] = lane_off_pl(t0n, next_wt, tp);
            }
            cls_next = __builtin_amdgcn_readfirstlane(t0n & 255);
            kx_next = __builtin_amdgcn_readfirstlane(kxn_v);
        }
#pragma unroll
        for (int g = 0; g < (NZ ? 0 : B3_KS1); ++g) {
            if (g == T0_USE) {
#pragma unroll
                for (int tp = 0; tp < 3; ++tp) {
                    vo8n[tp] = lane_off(t0n, next_wt, tp);
                    if constexpr (PL) pvo8n[tp] = lane_off_pl(t0n, next_wt, tp);
                }
                cls_next = __builtin_amdgcn_readfirstlane(t0n & 255);
                if constexpr (HALF && !PL) ev_next = load_amax(t0n, next_wt);
            }
            {
                const int gn = g + B3_RING;
                const int gt = gn < B3_KS1 ? gn : gn - B3_KS1;
                load_group(ring[g % B3_RING], gt, gn < B3_KS1 ? vo8[gt % 3] : vo8n[gt % 3], gn < B3_KS1 ? pvo8[gt % 3] : pvo8n[gt % 3]);
            }
            __builtin_amdgcn_sched_barrier(0);
            bf16x8 nh, nl;
            {
                const int g1 = (g + 1) % B3_KS1;
                if constexpr (PL) {
                    if (g == B3_KS1 - 2) kx_next = __builtin_amdgcn_readfirstlane(kxn_v);
                    planes_operand(ring[(g + 1) % B3_RING], nh, nl);
                } else if constexpr (HALF) {
                    if (g == B3_KS1 - 2) kx_next = tile_scale_exp(ev_next);
                    // group 0 of the NEXT tile is split under the last k-step of this one: its own scale
                    if constexpr ((ABL & 2) != 0) {
                        const float(&rr)[8] = ring[(g + 1) % B3_RING];
                        nh = __builtin_bit_cast(bf16x8, f32x4{rr[0], rr[1], rr[2], rr[3]});
                        nl = __builtin_bit_cast(bf16x8, f32x4{rr[4], rr[5], rr[6], rr[7]});
                    } else
                    split_x8s(ring[(g + 1) % B3_RING], g + 1 < B3_KS1 ? sx : pow2f(kx_next), nh, nl);
                } else {
                    split_x8<bf16x8, elem16, HALF>(ring[(g + 1) % B3_RING], nh, nl);
                }
                if (g1 % 3 == 1) {   // centre tap: these fp32 values are x_in at this lane's output rows
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        x_old[16 * ((g1 / 3) >> 1) + 8 * ((g1 / 3) & 1) + e] = ring[(g + 1) % B3_RING][e];
                }
            }
            bf16x8 ah = c_ah, al = c_al;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[q] = mfma16(ah, ph, acc[q]);
                const int gq = (g * 4 + q + 1) % (B3_KS1 * 4);
                const bf16x8 nah = lds_a[(((gq >> 2) * 2 + 0) * 4 + (gq & 3)) * 64];
                const bf16x8 nal = lds_a[(((gq >> 2) * 2 + 1) * 4 + (gq & 3)) * 64];
                acc[q] = mfma16(al, ph, acc[q]);
                acc[q] = mfma16(ah, pl, acc[q]);
                ah = nah;
                al = nal;
            }
            c_ah = ah;
            c_al = al;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                if (q >= 2) __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (q >= 2) __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (q >= 2) __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            ph = nh;
            pl = nl;
        }
        prefetch_head(next_wt, cls_next);
        float sk_old[32];   // skip accumulator: requested half-way through pass 0, used at the end of pass 1
        __builtin_amdgcn_sched_barrier(0);

        // stage 2 (two passes: out, skip).  The gate of k-step ks+1 (VALU + transcendental) is cut in pieces
        // placed after the MFMAs of k-step ks in pass 0; its split parts are kept for pass 1.
        const float rs = 0.70710678118654752440f;
        bf16x8 zh[B3_KS2], zl[B3_KS2];
        float zv[8];
        auto gate_piece = [&](int ks, int slot) {
            if (ks >= B3_KS2) return;
            const int zq = ks >> 1, r0 = 8 * (ks & 1);
            if (slot < 4) {
                if constexpr (HALF && PK_PWG_GATE_SCALAR) {   // (the A/B of round 6: the gate without packed fp32 instructions)
                    zv[2 * slot] = gated_s(acc[zq][r0 + 2 * slot], acc[zq + 2][r0 + 2 * slot], gca, gcb);
                    zv[2 * slot + 1] = gated_s(acc[zq][r0 + 2 * slot + 1], acc[zq + 2][r0 + 2 * slot + 1], gca, gcb);
                } else if constexpr (HALF) {   // z * 2^14 from the scaled accumulators
                    const f32x2 av = {acc[zq][r0 + 2 * slot], acc[zq][r0 + 2 * slot + 1]};
                    const f32x2 bv = {acc[zq + 2][r0 + 2 * slot], acc[zq + 2][r0 + 2 * slot + 1]};
                    const f32x2 z2 = gated_s2(av, bv, gca, gcb);
                    zv[2 * slot] = z2[0];
                    zv[2 * slot + 1] = z2[1];
                } else {
                    zv[2 * slot] = gated(acc[zq][r0 + 2 * slot], acc[zq + 2][r0 + 2 * slot]);
                    zv[2 * slot + 1] = gated(acc[zq][r0 + 2 * slot + 1], acc[zq + 2][r0 + 2 * slot + 1]);
                }
            } else if (slot == 4) {
                split_x8<bf16x8, elem16, HALF>(zv, zh[ks], zl[ks]);
            }
        };
#pragma unroll
        for (int slot = 0; slot < 5; ++slot) gate_piece(0, slot);
        bf16x8 f_ah = lds_a2[0], f_al = lds_a2[4 * 64];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            f32x16 acc2[2];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const f32x4 bv =
                        *reinterpret_cast<const f32x4*>(lds_bias + G + 64 * pass + 32 * q + 8 * r4 + 4 * hi);
                    acc2[q][4 * r4 + 0] = bv[0];
                    acc2[q][4 * r4 + 1] = bv[1];
                    acc2[q][4 * r4 + 2] = bv[2];
                    acc2[q][4 * r4 + 3] = bv[3];
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int idx = 0; idx < 2 * B3_KS2; ++idx) {
                const int ks = idx >> 1, q = idx & 1;
                if (!FIRST && pass == 0 && idx == B3_KS2) {   // half of acc[] is dead by now: registers are free
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            sk_old[16 * qq + r] = (ABL & 1) ? 0.5f
                                                            : pwg_skip_ld((a.skip + (long)(32 * qq + mfma_row(r, 0)) * XBLK) + vo4);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // W2 fragments of the next (pass, ks, q) in issue order
                const int nidx = (pass * 2 * B3_KS2 + idx + 1) % (4 * B3_KS2);
                const int npass = nidx / (2 * B3_KS2), nks = (nidx >> 1) % B3_KS2, nq = nidx & 1;
                acc2[q] = mfma16(f_ah, zh[ks], acc2[q]);
                const bf16x8 n_ah = lds_a2[((nks * 2 + 0) * 4 + 2 * npass + nq) * 64];
                const bf16x8 n_al = lds_a2[((nks * 2 + 1) * 4 + 2 * npass + nq) * 64];
                if (pass == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    gate_piece(ks + 1, 3 * q + 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                acc2[q] = mfma16(f_al, zh[ks], acc2[q]);
                if (pass == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    gate_piece(ks + 1, 3 * q + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                acc2[q] = mfma16(f_ah, zl[ks], acc2[q]);
                if (pass == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    gate_piece(ks + 1, 3 * q + 2);
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                }
                f_ah = n_ah;
                f_al = n_al;
            }
            __builtin_amdgcn_sched_barrier(0);
            float* dst = pass == 0 ? a.xout : a.skip;
            float am = 0.f;
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v;
                    if constexpr (HALF) {   // acc2 = 2^14 * 2^k2 * (W2 z + b); i0 carries the sqrt(0.5) of :314
                        if (pass == 0) {
                            if constexpr (NZ) {   // x_old = (w n + b) sqrt(1/2), recomputed above
                                v = fmaf(acc2[q][r], a.i0, x_old[16 * q + r]);
                            } else if constexpr (PL) {   // x_in = (hi + lo) / (its block's scale); x_old holds the raw vectors of the centre taps
                                const int cg = 2 * q + (r >> 3), e = r & 7;
                                const pl_f16x8 xh = __builtin_bit_cast(pl_f16x8, f32x4{x_old[8 * cg], x_old[8 * cg + 1], x_old[8 * cg + 2], x_old[8 * cg + 3]});
                                const pl_f16x8 xl = __builtin_bit_cast(pl_f16x8, f32x4{x_old[8 * cg + 4], x_old[8 * cg + 5], x_old[8 * cg + 6], x_old[8 * cg + 7]});
                                // (one multiply and two v_fma_mix_f32: the halves are read in place)
                                v = fmaf((float)xh[e], xs_cur, fmaf((float)xl[e], xs_cur, acc2[q][r] * a.i0));
                            } else {
                                v = fmaf(acc2[q][r], a.i0, x_old[16 * q + r] * rs);
                            }
                        } else v = FIRST ? acc2[q][r] * a.i1 : fmaf(acc2[q][r], a.i1, sk_old[16 * q + r]);
                        if (GEN && pass == 0 && !lane_valid) v = 0.f;   // beyond the utterance: stays zero padding
                        if (pass == 0) am = fmaxf(am, fabsf(v));
                    } else {
                        if (pass == 0) v = (acc2[q][r] + x_old[16 * q + r]) * rs;
                        else v = FIRST ? acc2[q][r] : (sk_old[16 * q + r] + acc2[q][r]);
                        if (GEN && pass == 0 && !lane_valid) v = 0.f;
                    }
                    if constexpr (PL) {
                        if (pass == 0) acc2[q][r] = v;   // stored below, once the block's maximum (its scale) is known
                        else pwg_skip_st(v, (dst + (long)(32 * q + mfma_row(r, 0)) * XBLK) + vo4);
                    } else {
                        if (!(ABL & 1) || a.Ttot < 0) (dst + (long)(32 * q + mfma_row(r, 0)) * XBLK)[vo4] = v;
                    }
                }
            if constexpr (HALF) {
                if (pass == 0) {   // max|x_out| of this 64 x 32 block for the next layer's operand scale
                    if constexpr (!PL) {
                        am = wave_max64(am);
                        if (lane == 0 && (!(ABL & 1) || a.Ttot < 0)) a.xe_out[(vo4 - 4 * hi * XBLK) >> 11] = __float_as_uint(am);
                    }
                    if constexpr (PL && AMAX) {   // scale guard: the tile's max|x_out| towards its utterance's slot (guarded / sampled calls only)
                        // a wave's tiles walk the timeline, ~2.5 in a row within one utterance: the running maximum stays in scalar registers and goes
                        // out when the utterance changes (an atomic is a vmcnt entry the next tile's operand waits queue up behind: +86 us per launch
                        // with one per tile)
                        const unsigned m = (unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(wave_max64(am)));
                        const int u = __builtin_amdgcn_readfirstlane(utt_v);
                        if (u != amax_utt) {
                            if (amax_utt >= 0 && lane == 0 && !PK_PWG_AMAX_PROBE)
                                atomicMax(a.amax_out + amax_utt * PWG_AMAX_PARTS + (int)(blockIdx.x & (PWG_AMAX_PARTS - 1)), amax_run);
                            amax_utt = u;
                            amax_run = m;
                        } else {
                            amax_run = amax_run > m ? amax_run : m;
                        }
                    }
                    if constexpr (PL) {   // x_out as planes at its utterance's a-priori scale
                        const float so = pow2f(__builtin_amdgcn_readfirstlane(ko_v));
                        char* pd = reinterpret_cast<char*>(a.xout) + pvo8[1];
#pragma unroll
                        for (int cg = 0; cg < 4; ++cg) {
                            // the same scale-and-split the consumer used to do per tap (v_pk_mul_f32, v_cvt_pkrtz_f16_f32,
                            // v_fma_mix_f32 x 2, v_cvt_f16_f32 per pair), once per value
                            float t8[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) t8[e] = acc2[cg >> 1][8 * (cg & 1) + e];
                            pl_f16x8 oh, ol;
                            split_x8s(t8, so, oh, ol);
                            if (PK_PWG_NT_XOUT) {
                                __builtin_nontemporal_store(oh, reinterpret_cast<pl_f16x8*>(pd + cg * 2048));
                                __builtin_nontemporal_store(ol, reinterpret_cast<pl_f16x8*>(pd + cg * 2048 + 16));
                            } else {
                                *reinterpret_cast<pl_f16x8*>(pd + cg * 2048) = oh;
                                *reinterpret_cast<pl_f16x8*>(pd + cg * 2048 + 16) = ol;
                            }
                        }
                    }
                }
            }
        }
        kx = kx_next;
        if constexpr (NZ) cls_cur = cls_next;
    }
    if constexpr (PL && AMAX) {
        if (amax_utt >= 0 && lane == 0 && !PK_PWG_AMAX_PROBE)
            atomicMax(a.amax_out + amax_utt * PWG_AMAX_PARTS + (int)(blockIdx.x & (PWG_AMAX_PARTS - 1)), amax_run);
    }
}

// last_conv_layers: ReLU -> Conv1D(SK->SK,1) -> ReLU -> Conv1D(SK->1,1) (:429-440,471) on
// skips * sqrt(1/layers) (:469).  One workgroup per tile, 8 waves x 32 samples.
struct PwgLastArgs {
    const float* skip;   // [SK][Ttot]
    const float* w1;     // [SK/2 k-steps][64 lanes][2 co-tiles] A fragments
    const float* b1;     // [SK]
    const float* w2;     // [SK]
    float b2;
    float scale;
    const int* tile_t0;
    long Ttot;
    float* wav;          // packed (ntiles*TILE)
    int kw;              // k_pwg_last_h3: the W fragments hold last_conv_layers.1.weight * 2^kw
    PwgGen gen;          // GEN kernels only (hop != 256): wav is packed per utterance, S_b samples each
};
// packed output index of sample (tile, offset) and whether it belongs to the utterance
template <bool GEN>
__device__ __forceinline__ long last_out_index(const PwgLastArgs& a, int tile, int off, bool& valid) {
    if constexpr (GEN) {
        const int b = a.gen.tile_utt[tile];
        const int sidx = a.gen.tile_s0[tile] + off;
        valid = sidx < a.gen.utt_S[b];
        return (long)a.gen.utt_off[b] + sidx;
    } else {
        valid = true;
        return (long)tile * TILE + off;
    }
}

template <bool GEN>
__global__ __launch_bounds__(512) void k_pwg_last(PwgLastArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int j = lane & 31;
    const int hi = lane >> 5;
    const int tile = blockIdx.x;
    const long t = (long)(a.tile_t0[tile] & ~255) + wave * WAVE_T + j;
    const float* sb = a.skip + xoff(t) + hi * XBLK;
    const f32x2* w1 = reinterpret_cast<const f32x2*>(a.w1) + lane;
    f32x16 acc[2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = a.b1[32 * q + mfma_row(r, hi)];
#pragma unroll 8
    for (int cp = 0; cp < SK / 2; ++cp) {
        const float bv = fmaxf(sb[(long)(2 * cp) * XBLK] * a.scale, 0.f);
        const f32x2 af = w1[cp * 64];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0], bv, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1], bv, acc[1], 0, 0, 0);
    }
    float part = 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            part = fmaf(a.w2[32 * q + mfma_row(r, hi)], fmaxf(acc[q][r], 0.f), part);
    part += __shfl_xor(part, 32);
    bool valid;
    const long oi = last_out_index<GEN>(a, tile, wave * WAVE_T + j, valid);
    if (hi == 0 && valid) a.wav[oi] = part + a.b2;
}

// Split-fp16 variant (default math): the 64 -> 64 conv as 3-term split-fp16 MFMA sums (24 x 32-cycle MFMAs
// per wave instead of 64 x 64-cycle ones), all 32 skip values of a lane requested before the first MFMA,
// W fragments ([ks 4][part 2][co-tile 2][lane][8 halves] = 16 KB) staged in LDS once per workgroup.  The
// operand channel of element e of k-step ks is 32*(ks>>1) + mfma_row(8*(ks&1) + e, hi), as in the layer kernel.
template <bool GEN>
__global__ __launch_bounds__(512) void k_pwg_last_h3(PwgLastArgs a) {
    __shared__ __attribute__((aligned(16))) f16x8 wl[4 * 2 * 2 * 64];
    {
        const f16x8* src = reinterpret_cast<const f16x8*>(a.w1);
        for (int i = threadIdx.x; i < 4 * 2 * 2 * 64; i += 512) wl[i] = src[i];
    }
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int j = lane & 31;
    const int hi = lane >> 5;
    const int tile = blockIdx.x;
    const long t = (long)(a.tile_t0[tile] & ~255) + wave * WAVE_T + j;
    const float* sb = a.skip + xoff(t) + 4 * hi * XBLK;
    float sv[4][8];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e)
            sv[ks][e] = PK_PWG_NT_EDGE ? __builtin_nontemporal_load(sb + (long)(32 * (ks >> 1) + mfma_row(8 * (ks & 1) + e, 0)) * XBLK)
                                       : sb[(long)(32 * (ks >> 1) + mfma_row(8 * (ks & 1) + e, 0)) * XBLK];
    f32x16 acc[2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = a.b1[32 * q + mfma_row(r, hi)];
    __syncthreads();
    // block scaling: the wave has its whole 64 x 32 operand in registers, so the block maximum is formed here
    bool lane_valid;
    const long out_idx = last_out_index<GEN>(a, tile, wave * WAVE_T + j, lane_valid);
    float am = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sv[ks][e] = lane_valid ? fmaxf(sv[ks][e] * a.scale, 0.f) : 0.f;   // ReLU(skips * sqrt(1/layers)) (:469-471)
            am = fmaxf(am, sv[ks][e]);
        }
    am = wave_max64(am);
    const int kx = blk_scale_exp((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(am)));
    const float sx = pow2f(kx), S = pow2f(kx + a.kw), Sinv = pow2f(-(kx + a.kw));
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] *= S;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        f16x8 bh, bl;
        split_x8s(sv[ks], sx, bh, bl);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const f16x8 ah = wl[((ks * 2 + 0) * 2 + q) * 64 + lane];
            const f16x8 al = wl[((ks * 2 + 1) * 2 + q) * 64 + lane];
            acc[q] = mfma16(ah, bh, acc[q]);
            acc[q] = mfma16(al, bh, acc[q]);
            acc[q] = mfma16(ah, bl, acc[q]);
        }
    }
    float part = 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            part = fmaf(a.w2[32 * q + mfma_row(r, hi)], fmaxf(acc[q][r], 0.f), part);   // ReLU commutes with S > 0
    part += __shfl_xor(part, 32);
    if (hi == 0 && lane_valid) a.wav[out_idx] = fmaf(part, Sinv, a.b2);
}

}  // namespace

// ================================================================== host side
struct pk_pwg {
    pk_ctx* ctx = nullptr;
    pk_pwg_cfg cfg;
    pk_param_map params;
    bool finalized = false;
    int hop = 256;
    int max_dilation = 512;
    int gap = 512;
    bool use_norm = false;
    std::vector<float> h_mu, h_sigma;
    // device weights
    pk_dbuf d_first_w, d_first_b, d_convin_wT, d_uptab, d_mu, d_sigma;
    pk_dbuf d_w1, d_w2, d_bias;     // all layers, concatenated
    pk_dbuf d_w1b, d_w2b, d_w1h, d_w2h;   // split (hi, lo) A fragments for k_pwg_layer_b3: bf16 / fp16 parts
    int math = PK_PWG_MATH_F16X3;   // default: fp32-equivalent error (5e-7), 1.9x faster than the fp32 matrix pipe
    pk_dbuf d_waux;                 // packed GEMM weight [AUX] x [layers*G]
    pk_dbuf d_l1, d_l1b, d_l2, d_l1h;   // d_l1h: split-fp16 fragments of last_conv_layers.1
    float l2_bias = 0.f;
    // block-scaled split-fp16 path: per-layer weight exponents (fragments hold w * 2^k), the bias image whose
    // stage-2 entries carry 2^14 * 2^k2, and max|x| per 32-sample block of the two x buffers
    std::vector<int> k1, k2o, k2s;
    int kw_last = 0;
    pk_dbuf d_bias_h, ws_xe0, ws_xe1;
    // workspace
    pk_dbuf ws_mel, ws_noise, ws_wav, ws_c0, ws_cin, ws_P, ws_x0, ws_x1, ws_skip, ws_dbg;
    pk_dbuf ws_tab;   // int tables
    // last call layout (for debug reads)
    std::vector<int> last_frames, last_toff, last_cuL, last_cuC;
    float first_wmax = 0.f, first_bmax = 0.f;   // planes path: bound of first_conv
    std::vector<float> first_w_host, first_b_host;
    pk_dbuf d_nz;              // k_pwg_layer_b3's NZ image: one k-step of A fragments + [64][w b] (split-fp16 math, hop 256)
    int nz_k1 = 0;             // ... and the exponent its fragments were scaled with
    bool noise_fed = true;     // option "noise_fed_first": layer 0 from the noise (NZ); 0 = k_pwg_first + the ordinary first block
    pk_dbuf d_cl, ws_nmax, ws_tkx;              // ... growth constants per layer, max|noise| per utterance, [layers + 1][tiles] scale exponents
    int last_ntiles = 0;
    long last_Ttot = 0;
    int last_x_final = 0;
    int last_ldp = 0;
    size_t last_o_cls = 0;
    int dbg = 0;
    bool planes_on = true;     // x as pre-split fp16 planes under the split-fp16 math (k_pwg_layer_b3<..., PL>); PK_PWG_PLANES=0: off
    bool last_planes = false;  // ... and whether the last run used them (debug tap 1 decodes)
    // "scale_guard": 0 off, 1 the first inference after finalize, 2 every inference.  A guarded inference on the planes path
    // measures max|x| per utterance and layer (k_pwg_planes_amax) and compares with the a-priori bound of k_pwg_tile_scales
    int scale_guard = 1;
    bool guard_done = false;   // mode 1: a guarded inference has run since finalize
    // mode 1 also RE-SAMPLES: every guard_every-th inference on the planes path measures the same way (AMAX layer kernels + one k_pwg_planes_amax launch), but its
    // verdict is DEFERRED -- maxima copied to pinned host memory behind an event, judged at the start of a later call -- so a
    // pipelined caller never stalls after the first call.  A deferred verdict above the limit moves the handle to the
    // fp32-x path from the next call on (the sampled call itself stays as computed: at the limit of 2^10 the planes still
    // carry 26 bits of the actual maximum, see GUARD_MAX_LOG2).
    int guard_every = 16;
    int calls_since_sample = 0;
    int samples_dropped = 0;           // samples not taken (more utterances than the pinned buffer holds) or not judged (event error)
    bool sample_pending = false;
    hipEvent_t ev_sample = nullptr;
    float* host_sample = nullptr;      // pinned: [layers + 1][B] maxima, then B noise maxima
    size_t host_sample_cap = 0;
    int sample_B = 0;
    std::vector<int> sample_frames;
    int deferred_fallbacks = 0;        // verdicts that arrived after their call (reported by pk_pwg_scale_overshoot's fell_back = 2)
    bool fell_back = false;    // the bound overshot by more than 2^GUARD_MAX_LOG2: the handle left the planes path
    std::vector<float> overshoot_log2;   // [layers + 1] of the last guarded inference (max over utterances)
    std::vector<float> cl_host;          // c_l of the bound, as uploaded to d_cl
    pk_dbuf ws_amax;
    unsigned long long seed = 0, rng_offset = 0;   // internal noise stream (noise == NULL)
    long chunk_samples = 1L << 40;                  // residual-stack chunk (env PK_PWG_CHUNK_SAMPLES); default: one chunk
};

// Timing ablations of the layer kernel (PK_PWG_ABLATE -> pk_pwg::dbg; results are WRONG): instantiated in the profile build only
template <bool PROF>
static int pwg_ablation_launch(pk_ctx* ctx, int dbg, int grid, dim3 blk, const PwgLayerArgs& a) {
    if constexpr (PROF) {
        if (dbg == 1) PK_LAUNCH(ctx, "pwg_layer_h3", (k_pwg_layer_b3<false, true, 1>), dim3(grid), blk, 0, a);
        else if (dbg == 32) PK_LAUNCH(ctx, "pwg_layer_h3", (k_pwg_layer_b3<false, true, 2>), dim3(grid), blk, 0, a);
        else if (dbg == 96) PK_LAUNCH(ctx, "pwg_layer_h3", (k_pwg_layer_b3<false, true, 6>), dim3(grid), blk, 0, a);
        else if (dbg == 160) PK_LAUNCH(ctx, "pwg_layer_h3", (k_pwg_layer_b3<false, true, 10>), dim3(grid), blk, 0, a);
        else PK_LAUNCH(ctx, "pwg_layer_h3", (k_pwg_layer_b3<false, true>), dim3(grid), blk, 0, a);
        return PK_OK;
    } else {
        PK_FAIL(PK_ESTATE, "ablation switches exist in the profile build only");
    }
}

extern "C" int pk_pwg_create(pk_ctx* ctx, const pk_pwg_cfg* cfg, pk_pwg** out) {
    if (!ctx || !cfg || !out) PK_FAIL(PK_EINVAL, "pk_pwg_create: NULL argument");
    *out = nullptr;
    if (cfg->stacks <= 0 || cfg->layers <= 0 || cfg->layers % cfg->stacks != 0)
        PK_FAIL(PK_ESHAPE, "PWGGenerator: layers (%d) must be a positive multiple of stacks (%d)",
                cfg->layers, cfg->stacks);  // assert layers % stacks == 0 (:398)
    if (cfg->use_causal_conv) PK_FAIL(PK_EUNSUPPORTED, "PWGGenerator: use_causal_conv=True is not implemented");
    if (cfg->in_channels != 1 || cfg->out_channels != 1 || cfg->kernel_size != KTAP ||
        cfg->residual_channels != R || cfg->gate_channels != G || cfg->skip_channels != SK ||
        cfg->aux_channels != AUX)
        PK_FAIL(PK_EUNSUPPORTED,
                "PWGGenerator: kernels are built for in/out 1, kernel 3, residual 64, gate 128, "
                "skip 64, aux 80 (got %d/%d, k%d, %d, %d, %d, %d)",
                cfg->in_channels, cfg->out_channels, cfg->kernel_size, cfg->residual_channels,
                cfg->gate_channels, cfg->skip_channels, cfg->aux_channels);
    if (cfg->n_upsample < 1 || cfg->n_upsample > 8) PK_FAIL(PK_EINVAL, "PWGGenerator: 1..8 upsample scales");
    int hop = 1;
    double reach = 0.0;  // of the composite upsampler, in frames
    for (int i = 0; i < cfg->n_upsample; ++i) {
        int s = cfg->upsample_scales[i];
        if (s < 1 || 2 * s + 1 > MAX_UP_TAPS) PK_FAIL(PK_EUNSUPPORTED, "upsample scale %d unsupported", s);
        hop *= s;
        reach += (double)s / hop;
    }
    // hop == 256 (LJSpeech): a work tile is a frame; any other hop >= 32 (baker / vctk: [4,5,3,5] = 300) runs the
    // GEN kernels, which compute frame and phase per sample
    if (hop < WAVE_T || hop > 1024)
        PK_FAIL(PK_EUNSUPPORTED, "prod(upsample_scales) = %d: hop sizes from %d to 1024 are supported", hop, WAVE_T);
    if (reach >= 2.0) PK_FAIL(PK_EUNSUPPORTED, "upsample scales reach %.2f frames (>= 2)", reach);
    if (cfg->aux_context_window < 0 || cfg->aux_context_window > 8)
        PK_FAIL(PK_EINVAL, "aux_context_window out of range");
    int lps = cfg->layers / cfg->stacks;
    if (lps > 12) PK_FAIL(PK_EUNSUPPORTED, "dilation 2^%d too large", lps - 1);
    pk_pwg* h = new pk_pwg();
    h->ctx = ctx;
    h->cfg = *cfg;
    h->hop = hop;
    h->max_dilation = 1 << (lps - 1);
    h->gap = ((h->max_dilation + TILE - 1) / TILE) * TILE;
    if (h->gap < TILE) h->gap = TILE;
    if (const char* e = pk_prof_env("PK_PWG_ABLATE")) h->dbg = atoi(e);   // profile build only: results are wrong when set
    if (const char* e = pk_prof_env("PK_PWG_PLANES")) h->planes_on = e[0] != '0';   // PK_PWG_PLANES=0: x as fp32 with per-block scales (round 2)
    if (const char* e = pk_prof_env("PK_PWG_CHUNK_SAMPLES")) h->chunk_samples = std::max(1L, atol(e));
    if (const char* e = pk_prof_env("PK_PWG_MATH"))
        h->math = strcmp(e, "bf16x3") == 0 ? PK_PWG_MATH_BF16X3 : (strcmp(e, "f16x3") == 0 ? PK_PWG_MATH_F16X3 : PK_PWG_MATH_F32);
    *out = h;
    return PK_OK;
}

extern "C" int pk_pwg_set_param(pk_pwg* h, const char* name, const float* data, const int64_t* shape,
                                int32_t ndim) {
    if (!h) PK_FAIL(PK_EINVAL, "pk_pwg_set_param: handle is NULL");
    h->finalized = false;
    return pk_store_param(h->params, name, data, shape, ndim);
}

extern "C" int pk_pwg_set_normalizer(pk_pwg* h, const float* mu, const float* sigma, int32_t n) {
    if (!h) PK_FAIL(PK_EINVAL, "pk_pwg_set_normalizer: handle is NULL");
    if (!mu && !sigma) {
        h->use_norm = false;
        return PK_OK;
    }
    if (!mu || !sigma || n != AUX) PK_FAIL(PK_ESHAPE, "normalizer needs mu and sigma of %d elements", AUX);
    h->h_mu.assign(mu, mu + n);
    h->h_sigma.assign(sigma, sigma + n);
    h->use_norm = true;
    PK_DEVICE(h->ctx->device);
    PK_TRY(pk_upload(h->ctx, h->d_mu, h->h_mu.data(), n * sizeof(float)));
    PK_TRY(pk_upload(h->ctx, h->d_sigma, h->h_sigma.data(), n * sizeof(float)));
    return PK_OK;
}

extern "C" int pk_pwg_set_math(pk_pwg* h, int32_t mode) {
    if (!h) PK_FAIL(PK_EINVAL, "pk_pwg_set_math: handle is NULL");
    if (mode != PK_PWG_MATH_F32 && mode != PK_PWG_MATH_BF16X3 && mode != PK_PWG_MATH_F16X3)
        PK_FAIL(PK_EINVAL, "pk_pwg_set_math: unknown mode %d", mode);
    h->math = mode;
    return PK_OK;
}

extern "C" int pk_pwg_set_chunk_samples(pk_pwg* h, int64_t samples) {
    if (!h) PK_FAIL(PK_EINVAL, "pk_pwg_set_chunk_samples: handle is NULL");
    if (samples <= 0) PK_FAIL(PK_EINVAL, "pk_pwg_set_chunk_samples: must be positive");
    h->chunk_samples = samples;
    return PK_OK;
}

// The verdict of a guarded / sampled inference: the bound (the recursion of k_pwg_tile_scales, on the host) against the measured
// maxima am[l][b]; fills overshoot_log2 and returns the worst log2(bound / max|x|).
static float pwg_guard_verdict(pk_pwg* h, const float* am, const float* nmax, int B, const int* frames) {
    const int layers = (int)h->cl_host.size();
    h->overshoot_log2.assign(layers + 1, 0.f);
    float worst = 0.f;
    for (int b = 0; b < B; ++b) {
        if (frames[b] <= 0) continue;
        float bound = std::fma(h->first_wmax, nmax[b], h->first_bmax) * 1.001f;
        for (int l = 0; l <= layers; ++l) {
            const float m = am[(size_t)l * B + b];
            // (an all-zero stream has no precision to lose)
            const float o = m > 0.f ? std::log2(bound / m) : 0.f;
            h->overshoot_log2[l] = std::max(h->overshoot_log2[l], o);
            worst = std::max(worst, o);
            if (l < layers) bound = (bound + h->cl_host[l]) * (0.70710678118654752440f * 1.001f);
        }
    }
    return worst;
}

constexpr float PWG_GUARD_MAX_LOG2 = 10.f;
constexpr int PWG_SAMPLE_MAX_B = 4096;   // utterances a deferred sample can hold (pinned buffer of pk_pwg_finalize)

// A deferred sample whose copies have landed is judged here (start of every inference, and pk_pwg_scale_overshoot).
static void pwg_poll_sample(pk_pwg* h, bool wait) {
    if (!h->sample_pending) return;
    const hipError_t st = wait ? hipEventSynchronize(h->ev_sample) : hipEventQuery(h->ev_sample);
    if (st == hipErrorNotReady) {
        (void)hipGetLastError();   // not an error: the copies have not landed yet
        return;
    }
    h->sample_pending = false;
    if (st != hipSuccess) {        // the event failed (device error, reset): whatever is in host_sample is not a measurement
        (void)hipGetLastError();
        ++h->samples_dropped;
        return;
    }
    const int layers = (int)h->cl_host.size(), B = h->sample_B;
    const float worst = pwg_guard_verdict(h, h->host_sample, h->host_sample + (size_t)(layers + 1) * B, B, h->sample_frames.data());
    if (worst > PWG_GUARD_MAX_LOG2 && h->planes_on) {
        h->planes_on = false;
        h->fell_back = true;
        ++h->deferred_fallbacks;
    }
}

extern "C" int pk_pwg_set_option(pk_pwg* h, const char* key, int64_t value) {
    if (!h || !key) PK_FAIL(PK_EINVAL, "pk_pwg_set_option: NULL argument");
    if (strcmp(key, "planes") == 0) {
        h->planes_on = value != 0;
        if (h->planes_on) h->fell_back = false;
    } else if (strcmp(key, "scale_guard") == 0) {
        if (value < 0 || value > 2) PK_FAIL(PK_EINVAL, "pk_pwg_set_option: scale_guard %lld (0, 1, 2)", (long long)value);
        h->scale_guard = (int)value;
    } else if (strcmp(key, "noise_fed_first") == 0) {
        if (value < 0 || value > 1) PK_FAIL(PK_EINVAL, "pk_pwg_set_option: noise_fed_first %lld (0, 1)", (long long)value);
        h->noise_fed = value != 0;
    } else if (strcmp(key, "scale_guard_every") == 0) {
        if (value < 0 || value > (1 << 30)) PK_FAIL(PK_EINVAL, "pk_pwg_set_option: scale_guard_every %lld", (long long)value);
        h->guard_every = (int)value;
    } else PK_FAIL(PK_EINVAL, "pk_pwg_set_option: unknown option '%s'", key);
    return PK_OK;
}

extern "C" int pk_pwg_scale_overshoot(pk_pwg* h, float* log2_overshoot, int32_t n, int32_t* fell_back) {
    if (!h || !log2_overshoot) PK_FAIL(PK_EINVAL, "pk_pwg_scale_overshoot: NULL argument");
    {
        pk_device_guard _dg(h->ctx->device);
        pwg_poll_sample(h, true);    // a sample still in flight is waited for: the report is of the LAST guarded or sampled call
    }
    if (h->overshoot_log2.empty()) PK_FAIL(PK_ESTATE, "pk_pwg_scale_overshoot: no guarded inference has run (option \"scale_guard\")");
    if (n != (int32_t)h->overshoot_log2.size())
        PK_FAIL(PK_ESHAPE, "pk_pwg_scale_overshoot: expected %d floats (layers + 1)", (int)h->overshoot_log2.size());
    memcpy(log2_overshoot, h->overshoot_log2.data(), (size_t)n * sizeof(float));
    if (fell_back) *fell_back = h->fell_back ? (h->deferred_fallbacks > 0 ? 2 : 1) : 0;
    return PK_OK;
}

extern "C" int pk_pwg_set_seed(pk_pwg* h, uint64_t seed) {
    if (!h) PK_FAIL(PK_EINVAL, "pk_pwg_set_seed: handle is NULL");
    h->seed = seed;
    h->rng_offset = 0;
    return PK_OK;
}

static inline uint16_t f32_to_bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0;
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static inline float bf16_to_f32(uint16_t hbits) {
    const uint32_t u = (uint32_t)hbits << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline uint16_t f32_to_f16_rne(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x47800000u) return (uint16_t)(sign | (x > 0x7f800000u ? 0x7e00u : 0x7c00u));   // overflow / nan
    if (x < 0x38800000u) {                                   // subnormal half (or zero)
        if (x < 0x33000000u) return (uint16_t)sign;
        const int shift = 113 - (int)(x >> 23);              // 1..24
        uint32_t m = (x & 0x7fffffu) | 0x800000u;
        const uint32_t half = 1u << (shift + 12), mask = (half << 1) - 1;
        uint32_t r = m >> (shift + 13);
        const uint32_t rem = m & mask;
        if (rem > half || (rem == half && (r & 1u))) ++r;
        return (uint16_t)(sign | r);
    }
    uint32_t r = x - 0x38000000u;                            // rebias exponent
    const uint32_t rem = r & 0x1fffu;
    r >>= 13;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) ++r;
    return (uint16_t)(sign | r);
}
static inline float f16_to_f32(uint16_t h) {
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
static inline void split16_host(float w, bool half, uint16_t& hi, uint16_t& lo) {
    if (half) {
        hi = f32_to_f16_rne(w);
        lo = f32_to_f16_rne(w - f16_to_f32(hi));
    } else {
        hi = f32_to_bf16_rne(w);
        lo = f32_to_bf16_rne(w - bf16_to_f32(hi));
    }
}

// UpsampleNet on a host vector (one channel): [stretch by s, FIR(2s+1) with zero padding s] per stage.
static std::vector<double> upsample_sim(std::vector<double> x, const pk_pwg_cfg& c,
                                        const std::vector<std::vector<double>>& firs) {
    for (int i = 0; i < c.n_upsample; ++i) {
        const int s = c.upsample_scales[i];
        const long n = (long)x.size() * s;
        std::vector<double> y(n, 0.0);
        for (long t = 0; t < n; ++t) {
            double acc = 0.0;
            for (int j = 0; j <= 2 * s; ++j) {
                const long u = t + j - s;
                if (u >= 0 && u < n) acc += firs[i][j] * x[u / s];
            }
            y[t] = acc;
        }
        x.swap(y);
    }
    return x;
}

extern "C" int pk_pwg_finalize(pk_pwg* h) {
    if (!h) PK_FAIL(PK_EINVAL, "pk_pwg_finalize: handle is NULL");
    pk_ctx* ctx = h->ctx;
    PK_DEVICE(ctx->device);
    const pk_pwg_cfg& c = h->cfg;
    std::vector<float> w, b;
    // first_conv
    PK_TRY(pk_get_weight(h->params, "first_conv", {R, 1, 1}, w));
    PK_TRY(pk_get_vector(h->params, "first_conv.bias", R, b));
    PK_TRY(pk_upload(ctx, h->d_first_w, w.data(), R * sizeof(float)));
    PK_TRY(pk_upload(ctx, h->d_first_b, b.data(), R * sizeof(float)));
    h->first_w_host = w;   // (the noise-fed first block folds them into layer 0's conv: nz_tab below)
    h->first_b_host = b;
    h->first_wmax = h->first_bmax = 0.f;   // |first_conv(n)| <= wmax |n| + bmax (the planes path's scale bound)
    for (int i = 0; i < R; ++i) {
        h->first_wmax = std::max(h->first_wmax, std::fabs(w[i]));
        h->first_bmax = std::max(h->first_bmax, std::fabs(b[i]));
    }
    // conv_in -> implicit-GEMM weight [K = tap*AUX + ci][N = AUX]
    const int kin = 2 * c.aux_context_window + 1;
    if (kin > PK_GEMM_MAX_TAPS) PK_FAIL(PK_EUNSUPPORTED, "PWG: aux_context_window %d too wide", c.aux_context_window);
    PK_TRY(pk_get_weight(h->params, "upsample_net.conv_in", {AUX, AUX, kin}, w));
    {
        std::vector<float> kn, packed;
        pk_conv_to_kn(w.data(), AUX, AUX, kin, kn);
        pk_gemm_pack(kn.data(), AUX * kin, AUX, packed);
        PK_TRY(pk_upload(ctx, h->d_convin_wT, packed.data(), packed.size() * sizeof(float)));
    }
    // composite upsampler table from the stage FIRs up_layers.{2i+1}.weight (1,1,1,2s+1):
    // class (a, b) = (min(frames before, 2), min(frames after, 2)); a canonical utterance with exactly
    // a frames before and b after reproduces the zero-padding behaviour at that distance from the edges
    // (the composite reach is < 2 frames), impulse responses give the weights.
    {
        std::vector<std::vector<double>> firs(c.n_upsample);
        for (int i = 0; i < c.n_upsample; ++i) {
            int taps = 2 * c.upsample_scales[i] + 1;
            PK_TRY(pk_get_weight(h->params, "upsample_net.upsample.up_layers." + std::to_string(2 * i + 1),
                                 {1, 1, 1, taps}, w));
            firs[i].assign(w.begin(), w.begin() + taps);
        }
        const int hop = h->hop;   // phases per frame
        std::vector<float> tab((size_t)N_EDGE_CLASS * hop * UPW_PAD, 0.f);
        for (int a = 0; a <= 2; ++a)
            for (int bb = 0; bb <= 2; ++bb) {
                const int Lc = a + bb + 1, fc = a, cls = a * 3 + bb;
                for (int fi = 0; fi < Lc; ++fi) {
                    std::vector<double> imp(Lc, 0.0);
                    imp[fi] = 1.0;
                    std::vector<double> y = upsample_sim(imp, c, firs);
                    const int jj = fi - fc + 2;  // tap index of frame fi in the window fc-2..fc+2
                    for (int p = 0; p < hop; ++p)
                        tab[((size_t)cls * hop + p) * UPW_PAD + jj] = (float)y[(size_t)fc * hop + p];
                }
            }
        PK_TRY(pk_upload(ctx, h->d_uptab, tab.data(), tab.size() * sizeof(float)));
    }
    // residual blocks -> MFMA A-fragment layouts; aux 1x1 convs -> one frame-rate GEMM weight
    {
        const size_t n1 = (size_t)KS1 * 64 * 4, n2 = (size_t)KS2 * 64 * 4, nb = G + R + SK;
        std::vector<float> W1(n1 * c.layers), W2(n2 * c.layers), B(nb * c.layers);
        std::vector<float> Wa((size_t)AUX * c.layers * G);   // [K = aux ch][N = layer*G + co]
        std::vector<float> wc, wa, wo, ws, bc, bo, bs;
        std::vector<float> cl_h(c.layers, 0.f);
        for (int l = 0; l < c.layers; ++l) {
            const std::string p = "conv_layers." + std::to_string(l);
            PK_TRY(pk_get_weight(h->params, p + ".conv", {G, R, KTAP}, wc));
            PK_TRY(pk_get_weight(h->params, p + ".conv1x1_aux", {G, AUX, 1}, wa));
            PK_TRY(pk_get_weight(h->params, p + ".conv1x1_out", {R, G / 2, 1}, wo));
            PK_TRY(pk_get_weight(h->params, p + ".conv1x1_skip", {SK, G / 2, 1}, ws));
            PK_TRY(pk_get_vector(h->params, p + ".conv.bias", G, bc));
            PK_TRY(pk_get_vector(h->params, p + ".conv1x1_out.bias", R, bo));
            PK_TRY(pk_get_vector(h->params, p + ".conv1x1_skip.bias", SK, bs));
            {   // |conv1x1_out(z) + b| <= c_l for |z| < 1: the growth of the planes path's magnitude bound per layer
                double m = 0.0;
                for (int i = 0; i < R; ++i) {
                    double acc = std::fabs((double)bo[i]);
                    for (int k = 0; k < G / 2; ++k) acc += std::fabs((double)wo[(size_t)i * (G / 2) + k]);
                    m = std::max(m, acc);
                }
                cl_h[l] = (float)(m * (1.0 + 1e-6));
            }
            float* a1 = W1.data() + n1 * l;
            for (int ks = 0; ks < KS1; ++ks)
                for (int lane = 0; lane < 64; ++lane) {
                    const int i = lane & 31, hi = lane >> 5;
                    const int g = ks / 8, cg = g / 3, tap = g % 3;      // kernel's group order: (channel group, tap)
                    const int ci = 2 * (8 * cg + ks % 8) + hi;
                    for (int q = 0; q < 4; ++q)
                        a1[((size_t)ks * 64 + lane) * 4 + q] = wc[((size_t)(32 * q + i) * R + ci) * KTAP + tap];
                }
            for (int ca = 0; ca < AUX; ++ca)
                for (int co = 0; co < G; ++co)
                    Wa[(size_t)ca * c.layers * G + (size_t)l * G + co] = wa[(size_t)co * AUX + ca];
            float* a2 = W2.data() + n2 * l;
            for (int zq = 0; zq < 2; ++zq)
                for (int r = 0; r < 16; ++r)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int i = lane & 31, hi = lane >> 5;
                        const int zc = 32 * zq + mfma_row(r, hi);  // gated channel fed by this k-slot
                        for (int q = 0; q < 4; ++q) {
                            const int row = 32 * (q & 1) + i;
                            const float v = (q < 2) ? wo[(size_t)row * (G / 2) + zc] : ws[(size_t)row * (G / 2) + zc];
                            a2[((size_t)(zq * 16 + r) * 64 + lane) * 4 + q] = v;
                        }
                    }
            float* bb = B.data() + nb * l;
            for (int i = 0; i < G; ++i) bb[i] = bc[i];
            for (int i = 0; i < R; ++i) bb[G + i] = bo[i];
            for (int i = 0; i < SK; ++i) bb[G + R + i] = bs[i];
        }
        // split 16-bit fragments: W1b [ks][part][co-tile][lane][8], W2b [ks][part][out-tile][lane][8];
        // variant 0 = bf16 parts, variant 1 = fp16 parts
        for (int variant = 0; variant < 2; ++variant) {
            const bool half = variant == 1;
            const size_t n1b = (size_t)B3_W1_BYTES / 2, n2b = (size_t)B3_W2_BYTES / 2;
            std::vector<uint16_t> W1b(n1b * c.layers), W2b(n2b * c.layers);
            if (half) {
                h->k1.assign(c.layers, 0);
                h->k2o.assign(c.layers, 0);
                h->k2s.assign(c.layers, 0);
            }
            for (int l = 0; l < c.layers; ++l) {
                const std::string p = "conv_layers." + std::to_string(l);
                PK_TRY(pk_get_weight(h->params, p + ".conv", {G, R, KTAP}, wc));
                PK_TRY(pk_get_weight(h->params, p + ".conv1x1_out", {R, G / 2, 1}, wo));
                PK_TRY(pk_get_weight(h->params, p + ".conv1x1_skip", {SK, G / 2, 1}, ws));
                if (half && l == 0 && G == 128 && R == 64 && KTAP == 3) {   // the noise-fed first block's LDS image (k_pwg_layer_b3, NZ): folded in fp64
                    const double wmax = h->first_wmax, bmax = h->first_bmax;
                    std::vector<float> wn((size_t)G * 16, 0.f);   // [co][16 "channels"]: u / wmax (3 taps), v / bmax (3 taps), zeros
                    for (int co = 0; co < G; ++co)
                        for (int tap = 0; tap < KTAP; ++tap) {
                            double u = 0.0, v = 0.0;
                            for (int ci = 0; ci < R; ++ci) {
                                const double wv = wc[((size_t)co * R + ci) * KTAP + tap];
                                u += wv * (double)h->first_w_host[ci];
                                v += wv * (double)h->first_b_host[ci];
                            }
                            wn[(size_t)co * 16 + tap] = wmax > 0.0 ? (float)(u / wmax) : 0.f;
                            wn[(size_t)co * 16 + 3 + tap] = bmax > 0.0 ? (float)(v / bmax) : 0.f;
                        }
                    h->nz_k1 = pk_weight_scale_exp(wn.data(), wn.size());
                    std::vector<float> img(2048 + 128, 0.f);
                    uint16_t* fr = reinterpret_cast<uint16_t*>(img.data());
                    for (int q = 0; q < 4; ++q)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 8; ++e) {
                                const int i = lane & 31, hi = lane >> 5;
                                const int ci = mfma_row(e, hi);   // the kernel's operand order within a k-step
                                uint16_t bh, bl;
                                split16_host(std::ldexp(wn[(size_t)(32 * q + i) * 16 + ci], h->nz_k1), true, bh, bl);
                                fr[(((size_t)0 * 4 + q) * 64 + lane) * 8 + e] = bh;
                                fr[(((size_t)1 * 4 + q) * 64 + lane) * 8 + e] = bl;
                            }
                    for (int co = 0; co < R; ++co) {
                        img[2048 + 2 * co] = h->first_w_host[co];
                        img[2048 + 2 * co + 1] = h->first_b_host[co];
                    }
                    PK_TRY(pk_upload(ctx, h->d_nz, img.data(), img.size() * sizeof(float)));
                }
                if (half) {   // fp16 parts are taken of w * 2^k (exact), k per tensor: no subnormal parts
                    h->k1[l] = pk_weight_scale_exp(wc.data(), wc.size());
                    h->k2o[l] = pk_weight_scale_exp(wo.data(), wo.size());
                    h->k2s[l] = pk_weight_scale_exp(ws.data(), ws.size());
                    for (auto& v : wc) v = std::ldexp(v, h->k1[l]);
                    for (auto& v : wo) v = std::ldexp(v, h->k2o[l]);
                    for (auto& v : ws) v = std::ldexp(v, h->k2s[l]);
                }
                uint16_t* a1 = W1b.data() + n1b * l;
                for (int ks = 0; ks < B3_KS1; ++ks)
                    for (int q = 0; q < 4; ++q)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 8; ++e) {
                                const int i = lane & 31, hi = lane >> 5;
                                const int tap = ks % 3, cg = ks / 3;
                                const int ci = 32 * (cg >> 1) + mfma_row(8 * (cg & 1) + e, hi);   // kernel's operand order
                                uint16_t bh, bl;
                                split16_host(wc[((size_t)(32 * q + i) * R + ci) * KTAP + tap], half, bh, bl);
                                a1[((((size_t)ks * 2 + 0) * 4 + q) * 64 + lane) * 8 + e] = bh;
                                a1[((((size_t)ks * 2 + 1) * 4 + q) * 64 + lane) * 8 + e] = bl;
                            }
                uint16_t* a2 = W2b.data() + n2b * l;
                for (int ks = 0; ks < B3_KS2; ++ks)
                    for (int q = 0; q < 4; ++q)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 8; ++e) {
                                const int i = lane & 31, hi = lane >> 5;
                                const int zc = 32 * (ks >> 1) + mfma_row(8 * (ks & 1) + e, hi);
                                const int row = 32 * (q & 1) + i;
                                const float w = (q < 2) ? wo[(size_t)row * (G / 2) + zc] : ws[(size_t)row * (G / 2) + zc];
                                uint16_t bh, bl;
                                split16_host(w, half, bh, bl);
                                a2[((((size_t)ks * 2 + 0) * 4 + q) * 64 + lane) * 8 + e] = bh;
                                a2[((((size_t)ks * 2 + 1) * 4 + q) * 64 + lane) * 8 + e] = bl;
                            }
            }
            PK_TRY(pk_upload(ctx, half ? h->d_w1h : h->d_w1b, W1b.data(), W1b.size() * sizeof(uint16_t)));
            PK_TRY(pk_upload(ctx, half ? h->d_w2h : h->d_w2b, W2b.data(), W2b.size() * sizeof(uint16_t)));
        }
        PK_TRY(pk_upload(ctx, h->d_w1, W1.data(), W1.size() * sizeof(float)));
        PK_TRY(pk_upload(ctx, h->d_w2, W2.data(), W2.size() * sizeof(float)));
        PK_TRY(pk_upload(ctx, h->d_bias, B.data(), B.size() * sizeof(float)));
        PK_TRY(pk_upload(ctx, h->d_cl, cl_h.data(), cl_h.size() * sizeof(float)));
        h->cl_host = cl_h;
        h->guard_done = false;   // new weights: the next inference is guarded again (scale_guard 1)
        h->sample_pending = false;
        h->calls_since_sample = 0;
        h->deferred_fallbacks = 0;
        {   // bias image of the scaled path: stage-2 accumulators start from 2^14 * 2^k2 * bias
            std::vector<float> Bh(B);
            for (int l = 0; l < c.layers; ++l) {
                float* bb = Bh.data() + nb * l;
                for (int i = 0; i < R; ++i) bb[G + i] = std::ldexp(bb[G + i], 14 + h->k2o[l]);
                for (int i = 0; i < SK; ++i) bb[G + R + i] = std::ldexp(bb[G + R + i], 14 + h->k2s[l]);
            }
            PK_TRY(pk_upload(ctx, h->d_bias_h, Bh.data(), Bh.size() * sizeof(float)));
        }
        std::vector<float> packed;
        pk_gemm_pack(Wa.data(), AUX, c.layers * G, packed);
        PK_TRY(pk_upload(ctx, h->d_waux, packed.data(), packed.size() * sizeof(float)));
    }
    // last layers
    {
        std::vector<float> w1, b1, w2, b2;
        PK_TRY(pk_get_weight(h->params, "last_conv_layers.1", {SK, SK, 1}, w1));
        PK_TRY(pk_get_vector(h->params, "last_conv_layers.1.bias", SK, b1));
        PK_TRY(pk_get_weight(h->params, "last_conv_layers.3", {1, SK, 1}, w2));
        PK_TRY(pk_get_vector(h->params, "last_conv_layers.3.bias", 1, b2));
        std::vector<float> A((size_t)(SK / 2) * 64 * 2);
        for (int cp = 0; cp < SK / 2; ++cp)
            for (int lane = 0; lane < 64; ++lane) {
                const int i = lane & 31, hi = lane >> 5;
                for (int q = 0; q < 2; ++q)
                    A[((size_t)cp * 64 + lane) * 2 + q] = w1[(size_t)(32 * q + i) * SK + 2 * cp + hi];
            }
        PK_TRY(pk_upload(ctx, h->d_l1, A.data(), A.size() * sizeof(float)));
        {
            std::vector<uint16_t> Ah((size_t)4 * 2 * 2 * 64 * 8);
            h->kw_last = pk_weight_scale_exp(w1.data(), w1.size());
            for (auto& v : w1) v = std::ldexp(v, h->kw_last);   // (w1 is not used below this block)
            for (int ks = 0; ks < 4; ++ks)
                for (int q = 0; q < 2; ++q)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int i = lane & 31, hi = lane >> 5;
                            const int ci = 32 * (ks >> 1) + mfma_row(8 * (ks & 1) + e, hi);
                            uint16_t bh, bl;
                            split16_host(w1[(size_t)(32 * q + i) * SK + ci], true, bh, bl);
                            Ah[((((size_t)ks * 2 + 0) * 2 + q) * 64 + lane) * 8 + e] = bh;
                            Ah[((((size_t)ks * 2 + 1) * 2 + q) * 64 + lane) * 8 + e] = bl;
                        }
            PK_TRY(pk_upload(ctx, h->d_l1h, Ah.data(), Ah.size() * sizeof(uint16_t)));
        }
        PK_TRY(pk_upload(ctx, h->d_l1b, b1.data(), SK * sizeof(float)));
        PK_TRY(pk_upload(ctx, h->d_l2, w2.data(), SK * sizeof(float)));
        h->l2_bias = b2[0];
    }
    // The deferred scale-guard sample's landing place and its event are created HERE (hipHostMalloc and
    // hipEventCreate synchronise the device and are illegal under stream capture -- they used to run inside the 16th
    // pk_pwg_infer).  Sized for PWG_SAMPLE_MAX_B utterances (0.5 MB pinned); a sampled call with more utterances than that is
    // not sampled (samples_skipped counts them; "scale_guard" 2 guards every call regardless of size).
    if (!h->host_sample) {
        const size_t need = (size_t)(c.layers + 2) * PWG_SAMPLE_MAX_B * sizeof(float);
        PK_HIP(hipHostMalloc(reinterpret_cast<void**>(&h->host_sample), need, hipHostMallocDefault));
        h->host_sample_cap = need;
    }
    if (!h->ev_sample) PK_HIP(hipEventCreateWithFlags(&h->ev_sample, hipEventDisableTiming));
    h->finalized = true;
    return PK_OK;
}

extern "C" int pk_pwg_infer(pk_pwg* h, const float* mel, const int32_t* frames, int32_t B,
                            const float* noise, float* wav, int32_t flags) {
    if (!h || !mel || !frames || !wav) PK_FAIL(PK_EINVAL, "pk_pwg_infer: NULL argument");
    if (!h->finalized) PK_FAIL(PK_ESTATE, "pk_pwg_infer: call pk_pwg_finalize first");
    if (B <= 0) PK_FAIL(PK_EINVAL, "pk_pwg_infer: batch size must be positive");
    pk_ctx* ctx = h->ctx;
    PK_DEVICE(ctx->device);
    pwg_poll_sample(h, false);   // the verdict of an earlier sampled call, if its copies have landed (never waits)
    const pk_pwg_cfg& c = h->cfg;
    const int hop = h->hop, gap = h->gap;
    // ---- layout.  Utterances start on 256-sample boundaries (work tiles are 256-sample chunks of an utterance;
    // with hop == 256 a tile is a frame), the zeroed gap in front of each runs from the end of the previous
    // utterance's last 32-sample block: at least `gap` samples, longer by the alignment when hop != 256.
    const bool gen = hop != TILE;
    std::vector<int> cuL(B + 1, 0), cuC(B + 1, 0), toff(B), gap_start(B + 1), gap_len(B + 1), utt_S(B), utt_off(B);
    long t = 0, packed = 0;
    for (int b = 0; b < B; ++b) {
        if (frames[b] <= 0) PK_FAIL(PK_EINVAL, "pk_pwg_infer: utterance %d has %d frames", b, frames[b]);
        const long S_b = (long)frames[b] * hop;
        if (S_b >= (1L << 24)) PK_FAIL(PK_EUNSUPPORTED, "pk_pwg_infer: utterance %d is longer than 2^24 samples", b);
        cuL[b + 1] = cuL[b] + frames[b];
        cuC[b + 1] = cuC[b] + (int)((S_b + TILE - 1) / TILE);
        gap_start[b] = (int)t;
        t = (t + TILE - 1) / TILE * TILE + gap;
        gap_len[b] = (int)(t - gap_start[b]);
        toff[b] = (int)t;
        utt_S[b] = (int)S_b;
        utt_off[b] = (int)packed;
        packed += S_b;
        t = (t + S_b + XBLK - 1) / XBLK * XBLK;     // the last block is written whole (zeros beyond S_b)
    }
    gap_start[B] = (int)t;
    t = (t + TILE - 1) / TILE * TILE + gap;
    gap_len[B] = (int)(t - gap_start[B]);
    const long Ttot = t;
    const int sumL = cuL[B];          // frames
    const int sumC = cuC[B];          // 256-sample work tiles (== sumL when hop == 256)
    const long sumS = (long)sumL * hop;
    if (packed >= (1L << 31)) PK_FAIL(PK_EUNSUPPORTED, "pk_pwg_infer: %ld samples do not fit one call", sumS);
    // kernels address with 32-bit per-lane byte offsets of up to 5 rows of the timeline
    if (Ttot * R >= (1L << 32)) PK_FAIL(PK_EUNSUPPORTED, "pk_pwg_infer: %ld samples do not fit one call", sumS);
    h->last_frames.assign(frames, frames + B);
    h->last_toff = toff;
    h->last_cuL = cuL;
    h->last_cuC = cuC;
    h->last_ntiles = sumC;
    h->last_Ttot = Ttot;

    // int tables: [cuL (B+1)] [gap_start (B+1)] [frame_utt (sumL)] [tile_t0 (sumL)] [tile_cls (sumL)]
    std::vector<int> tab;
    auto push = [&](const std::vector<int>& v) {
        size_t o = tab.size();
        tab.insert(tab.end(), v.begin(), v.end());
        return o;
    };
    const size_t o_cul = push(cuL);
    const size_t o_gap = push(gap_start), o_gaplen = push(gap_len);
    // per work tile: timeline offset, edge class of the frame (hop == 256), sample offset in / index of the utterance
    std::vector<int> tile_t0(sumC), tile_cls(sumC, 0), tile_s0(sumC), tile_utt(sumC);
    for (int b = 0; b < B; ++b)
        for (int k = 0; k < cuC[b + 1] - cuC[b]; ++k) {
            const int i = cuC[b] + k;
            tile_t0[i] = toff[b] + k * TILE;     // a multiple of 256: the low bits carry the edge class (below)
            tile_s0[i] = k * TILE;
            tile_utt[i] = b;
            if (!gen) {
                const int before = k < 2 ? k : 2, after = (frames[b] - 1 - k) < 2 ? (frames[b] - 1 - k) : 2;
                tile_cls[i] = before * 3 + after;
                tile_t0[i] |= tile_cls[i];       // one load gives the split layer kernel both, a tile ahead
            }
        }
    const size_t o_tile = push(tile_t0), o_cls = push(tile_cls), o_ts0 = push(tile_s0), o_tutt = push(tile_utt);
    const size_t o_uS = push(utt_S), o_uF = push(std::vector<int>(frames, frames + B)), o_uoff = push(utt_off);
    h->last_o_cls = o_cls;
    // conv_in's padded row timeline (k_pwg_convin_prep): source mel row and destination c0 row (-1: padding)
    const int cw = c.aux_context_window;
    const int rows_p = sumL + 2 * cw * B;
    const int rows_p_alloc = ((rows_p + PK_GEMM_BM - 1) / PK_GEMM_BM) * PK_GEMM_BM;
    std::vector<int> prow_src(rows_p_alloc, 0), prow_out(rows_p_alloc, -1);
    for (int b = 0; b < B; ++b) {
        const int r0 = cuL[b] + 2 * cw * b;
        for (int jr = 0; jr < frames[b] + 2 * cw; ++jr) {
            int f = jr - cw;
            if (flags & PK_PWG_C_HAS_CONTEXT) {
                prow_src[r0 + jr] = r0 + jr;   // c already carries the context frames
            } else {
                const int fc = f < 0 ? 0 : (f >= frames[b] ? frames[b] - 1 : f);
                prow_src[r0 + jr] = cuL[b] + fc;
            }
            if (f >= 0 && f < frames[b]) prow_out[r0 + jr] = cuL[b] + f;
        }
    }
    const size_t o_psrc = push(prow_src), o_pout = push(prow_out);
    PK_TRY(h->ws_tab.reserve(tab.size() * sizeof(int)));
    PK_HIP(hipMemcpyAsync(h->ws_tab.p, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    PK_HIP(hipStreamSynchronize(ctx->stream));  // tab is a stack vector
    const int* d_tab = h->ws_tab.as<int>();

    // ---- workspaces
    const float* d_mel = mel;
    const float* d_noise = noise;
    float* d_wav = wav;
    if (flags & PK_HOST_IO) {
        const size_t mel_rows = (size_t)sumL + ((flags & PK_PWG_C_HAS_CONTEXT) ? (size_t)2 * c.aux_context_window * B : 0);
        PK_TRY(h->ws_mel.reserve(mel_rows * AUX * 4));
        PK_TRY(h->ws_wav.reserve((size_t)sumS * 4));
        PK_HIP(hipMemcpyAsync(h->ws_mel.p, mel, mel_rows * AUX * 4, hipMemcpyHostToDevice, ctx->stream));
        d_mel = h->ws_mel.as<float>();
        d_wav = h->ws_wav.as<float>();
        if (noise) {
            PK_TRY(h->ws_noise.reserve((size_t)sumS * 4));
            PK_HIP(hipMemcpyAsync(h->ws_noise.p, noise, (size_t)sumS * 4, hipMemcpyHostToDevice, ctx->stream));
            d_noise = h->ws_noise.as<float>();
        }
    }
    if (!noise) {   // x = randn(...) (:515-516) drawn by the engine: next range of the handle's stream
        PK_TRY(h->ws_noise.reserve((size_t)sumS * 4));
        PK_TRY(pk_randn_device(ctx, h->ws_noise.as<float>(), sumS, h->seed, h->rng_offset));
        h->rng_offset += ((unsigned long long)sumS + 3) / 4 * 4;
        d_noise = h->ws_noise.as<float>();
    }
    const int rows_alloc = ((sumL + PK_GEMM_BM - 1) / PK_GEMM_BM) * PK_GEMM_BM;
    const int ldp = c.layers * G;
    h->last_ldp = ldp;
    PK_TRY(h->ws_c0.reserve((size_t)(rows_alloc + 2 * P_LEAD) * AUX * 4));
    PK_TRY(h->ws_P.reserve((size_t)(rows_alloc + 2 * P_LEAD) * ldp * 4));
    PK_TRY(h->ws_x0.reserve((size_t)R * Ttot * 4));
    PK_TRY(h->ws_x1.reserve((size_t)R * Ttot * 4));
    PK_TRY(h->ws_skip.reserve((size_t)SK * Ttot * 4));
    const size_t n_blk = (size_t)(Ttot / XBLK) + 1;
    PK_TRY(h->ws_xe0.reserve(n_blk * 4));
    PK_TRY(h->ws_xe1.reserve(n_blk * 4));
    float* c0 = h->ws_c0.as<float>() + (size_t)P_LEAD * AUX;
    float* P = h->ws_P.as<float>() + (size_t)P_LEAD * ldp;

    // ---- zero the gaps of both ping-pong buffers and the margins of P (read with zero weights)
    PwgGen gtab;
    gtab.tile_s0 = d_tab + o_ts0;
    gtab.tile_utt = d_tab + o_tutt;
    gtab.utt_S = d_tab + o_uS;
    gtab.utt_F = d_tab + o_uF;
    gtab.utt_row0 = d_tab + o_cul;
    gtab.utt_off = d_tab + o_uoff;
    gtab.P0 = nullptr;
    gtab.hop = hop;
    gtab.inv_hop = 1.0f / (float)hop;
    {
        dim3 grid(pk_div_up(gap + TILE, 256), B + 1, R);
        PK_LAUNCH(ctx, "pwg_zero_gaps", k_zero_gaps, grid, dim3(256), 0, h->ws_x0.as<float>(), d_tab + o_gap,
                  d_tab + o_gaplen, R, Ttot);
        PK_LAUNCH(ctx, "pwg_zero_gaps", k_zero_gaps, grid, dim3(256), 0, h->ws_x1.as<float>(), d_tab + o_gap,
                  d_tab + o_gaplen, R, Ttot);
        // max|x| per block: 0 in the gaps, the producers of x fill the rest
        PK_HIP(hipMemsetAsync(h->ws_xe0.p, 0, n_blk * 4, ctx->stream));
        PK_HIP(hipMemsetAsync(h->ws_xe1.p, 0, n_blk * 4, ctx->stream));
        PK_HIP(hipMemsetAsync(h->ws_P.p, 0, (size_t)P_LEAD * ldp * 4, ctx->stream));
        PK_HIP(hipMemsetAsync(P + (size_t)sumL * ldp, 0, (size_t)(rows_alloc - sumL + P_LEAD) * ldp * 4, ctx->stream));
    }
    // ---- conditioning at frame rate: conv_in, then all layers' aux 1x1 convs as one GEMM
    {
        const int kin = 2 * cw + 1;
        PK_TRY(h->ws_cin.reserve((size_t)(rows_p_alloc + 2 * P_LEAD) * AUX * 4));
        float* cin = h->ws_cin.as<float>() + (size_t)P_LEAD * AUX;
        PK_LAUNCH(ctx, "pwg_convin_prep", k_pwg_convin_prep, dim3(rows_p_alloc), dim3(128), 0, d_mel,
                  h->d_mu.as<float>(), h->d_sigma.as<float>(), (h->use_norm && (flags & PK_APPLY_NORMALIZER)) ? 1 : 0, d_tab + o_psrc, rows_p, cin);
        {
            pk_gemm_args g;
            g.A = cin;
            g.lda = AUX;
            g.Wp = h->d_convin_wT.as<float>();
            g.C = c0;
            g.ldc = AUX;
            g.M = rows_p;
            g.N = AUX;
            g.Cin = AUX;
            g.taps = kin;
            g.pad = cw;
            g.out_rowmap = d_tab + o_pout;
            PK_TRY(pk_gemm_launch(ctx, "pwg_convin_gemm", g));
        }
        pk_gemm_args g;
        g.A = c0;
        g.lda = AUX;
        g.Wp = h->d_waux.as<float>();
        g.C = P;
        g.ldc = ldp;
        g.M = sumL;
        g.N = ldp;
        g.Cin = AUX;
        g.taps = 1;
        g.pad = 0;
        PK_TRY(pk_gemm_launch(ctx, "pwg_aux_gemm", g));
    }
    // ---- first conv
    // (planes: lane offsets into x are 32-bit byte offsets)
    // (profile build: PK_PWG_ABLATE=1024 runs every layer on the FIRST kernel -- skip written, never read: the skip sum and the
    // waveform are wrong, x and the gates are not -- to measure what 256 of the 1 024 B per sample and layer cost)
    const bool all_first = PK_PROFILE_BUILD != 0 && h->dbg == 1024;
    bool planes = h->planes_on && h->math == PK_PWG_MATH_F16X3 && (h->dbg == 0 || all_first) && (size_t)R * Ttot * 4 < ((size_t)1 << 32);
    // a guarded inference (option "scale_guard") measures max|x| per utterance and layer next to the a-priori bound; should
    // the bound overshoot by more than 2^GUARD_MAX_LOG2 the stack is run again on the fp32-x path (second pass of this loop:
    // the noise, the conditioning P and the zeroed gaps are all still in place), which the handle then keeps
    constexpr float GUARD_MAX_LOG2 = PWG_GUARD_MAX_LOG2;
    const bool guard = planes && (h->scale_guard == 2 || (h->scale_guard == 1 && !h->guard_done));
    // mode 1 after the first call: every guard_every-th inference measures too, verdict deferred (see the handle's comment)
    bool sample = false;
    if (planes && !guard && h->scale_guard == 1 && h->guard_every > 0 && !h->sample_pending &&
        ++h->calls_since_sample >= h->guard_every) {
        if (B <= PWG_SAMPLE_MAX_B && h->host_sample && h->ev_sample) sample = true;
        else ++h->samples_dropped;     // larger than the pinned buffer of pk_pwg_finalize: nothing is allocated inside a call
        h->calls_since_sample = 0;
    }
    unsigned* amax = nullptr;
    if (guard || sample) {
        // [layers + 1][B] maxima, then the layer kernels' parts [layers][B][PWG_AMAX_PARTS] (k_pwg_amax_fold)
        const size_t n_amax = (size_t)(c.layers + 1) * B + (size_t)c.layers * B * PWG_AMAX_PARTS;
        PK_TRY(h->ws_amax.reserve(n_amax * sizeof(unsigned)));
        amax = h->ws_amax.as<unsigned>();
        PK_HIP(hipMemsetAsync(amax, 0, n_amax * sizeof(unsigned), ctx->stream));
    }
    for (int attempt = 0;; ++attempt) {
    bool amax_parts_used = false;   // AMAX layer kernels ran: their parts are folded into amax[] behind the stack
    h->last_planes = planes;
    int* tkx = nullptr;
    if (planes) {
        PK_TRY(h->ws_nmax.reserve((size_t)B * 4));
        PK_TRY(h->ws_tkx.reserve((size_t)(c.layers + 1) * sumC * 4));
        tkx = h->ws_tkx.as<int>();
        PK_HIP(hipMemsetAsync(h->ws_nmax.p, 0, (size_t)B * 4, ctx->stream));
        PK_LAUNCH(ctx, "pwg_noise_max", k_pwg_noise_max, dim3(B, NMAX_PARTS), dim3(256), 0, d_noise, d_tab + o_uoff, d_tab + o_uS,
                  h->ws_nmax.as<float>());
        PK_LAUNCH(ctx, "pwg_tile_scales", k_pwg_tile_scales, dim3(pk_div_up(sumC, 256)), dim3(256), 0, d_tab + o_tutt, sumC,
                  h->ws_nmax.as<float>(), h->first_wmax, h->first_bmax, h->d_cl.as<float>(), c.layers, tkx);
    }
    // noise-fed first block (k_pwg_layer_b3's NZ): no first_conv launch, no x planes for layer 0 -- hop 256 on the planes path
    const bool nz = planes && !gen && !all_first && h->noise_fed && h->d_nz.p != nullptr;
    if (nz) {
    } else if (planes && gen)
        PK_LAUNCH(ctx, "pwg_first", (k_pwg_first<true, true>), dim3(sumC), dim3(TILE), 0, d_noise, h->d_first_w.as<float>(),
                  h->d_first_b.as<float>(), d_tab + o_tile, Ttot, h->ws_x0.as<float>(), h->ws_xe0.as<unsigned>(), gtab, tkx);
    else if (planes)
        PK_LAUNCH(ctx, "pwg_first", (k_pwg_first<false, true>), dim3(sumC), dim3(TILE), 0, d_noise, h->d_first_w.as<float>(),
                  h->d_first_b.as<float>(), d_tab + o_tile, Ttot, h->ws_x0.as<float>(), h->ws_xe0.as<unsigned>(), gtab, tkx);
    else if (gen)
        PK_LAUNCH(ctx, "pwg_first", k_pwg_first<true>, dim3(sumC), dim3(TILE), 0, d_noise, h->d_first_w.as<float>(),
                  h->d_first_b.as<float>(), d_tab + o_tile, Ttot, h->ws_x0.as<float>(), h->ws_xe0.as<unsigned>(), gtab, tkx);
    else
        PK_LAUNCH(ctx, "pwg_first", k_pwg_first<false>, dim3(sumC), dim3(TILE), 0, d_noise, h->d_first_w.as<float>(),
                  h->d_first_b.as<float>(), d_tab + o_tile, Ttot, h->ws_x0.as<float>(), h->ws_xe0.as<unsigned>(), gtab, tkx);
    if (guard && planes && !nz)   // (noise-fed: x_0 is never stored, amax[0] stays 0 = "nothing to lose")
        PK_LAUNCH(ctx, "pwg_planes_amax", k_pwg_planes_amax, dim3(sumC), dim3(256), 0, h->ws_x0.as<float>(), d_tab + o_tile,
                  d_tab + o_tutt, tkx, amax);
    // ---- residual stack.  Optionally the batch is cut into chunks of whole utterances whose x ping-pong +
    // skip buffers (3 x 256 B per sample) fit the 256 MB Infinity Cache, all layers running over one chunk
    // before the next.  A pure load/store kernel with this access pattern gains from that (tools/micro/
    // stream_pattern: 7.3 TB/s for a 252 MB working set vs 4.9 TB/s streaming the batch), the layer kernel
    // does not (measured: 2-utterance chunks 43.6 ms, 4-utterance 42.5 ms, one chunk 41.8 ms per 30 layers:
    // it is bound by its own MFMA + VALU issue, not by HBM), so the default is one chunk.
    {
        const int lps = c.layers / c.stacks;
        const int grid = ctx->n_cu;   // persistent: one workgroup per CU (LDS-resident weights)
        std::vector<int> chunk_first;   // first utterance of every chunk
        {
            long acc = 0;
            for (int b = 0; b < B; ++b) {
                const long s_b = (long)frames[b] * hop;
                if (b == 0 || acc + s_b > h->chunk_samples) {
                    chunk_first.push_back(b);
                    acc = 0;
                }
                acc += s_b;
            }
            chunk_first.push_back(B);
        }
        for (size_t ck = 0; ck + 1 < chunk_first.size(); ++ck) {
        const int tile0 = cuC[chunk_first[ck]], ntile = cuC[chunk_first[ck + 1]] - tile0;
        const int row0 = cuL[chunk_first[ck]];   // first P row of the chunk (== tile0 when hop == 256)
        for (int l = 0; l < c.layers; ++l) {
            PwgLayerArgs a;
            a.amax_out = nullptr;
            a.noise = nullptr;
            a.nz_tab = nullptr;
            a.noise_n = 0;
            a.nz_wmax = a.nz_bmax = 0.f;
            a.xin = (l & 1) ? h->ws_x1.as<float>() : h->ws_x0.as<float>();
            a.xout = (l & 1) ? h->ws_x0.as<float>() : h->ws_x1.as<float>();
            a.skip = h->ws_skip.as<float>();
            a.w1 = h->d_w1.as<float>() + (size_t)l * KS1 * 64 * 4;
            a.w2 = h->d_w2.as<float>() + (size_t)l * KS2 * 64 * 4;
            a.bias = h->d_bias.as<float>() + (size_t)l * (G + R + SK);
            a.P = P + (size_t)l * G + (size_t)row0 * ldp;
            a.gen = gtab;
            a.gen.tile_s0 += tile0;
            a.gen.tile_utt += tile0;
            a.gen.P0 = P + (size_t)l * G;
            a.uptab = h->d_uptab.as<float>();
            a.tile_t0 = d_tab + o_tile + tile0;
            a.tile_cls = d_tab + o_cls + tile0;
            a.Ttot = Ttot;
            a.ldp = ldp;
            a.ntiles = ntile;
            a.dilation = 1 << (l % lps);
            a.dbg = h->dbg;
            a.xe_in = (l & 1) ? h->ws_xe1.as<unsigned>() : h->ws_xe0.as<unsigned>();
            a.xe_out = (l & 1) ? h->ws_xe0.as<unsigned>() : h->ws_xe1.as<unsigned>();
            a.k1 = 0;
            a.i0 = a.i1 = 1.f;
            a.tile_kx_in = planes ? tkx + (size_t)l * sumC + tile0 : nullptr;
            a.tile_kx_out = planes ? tkx + (size_t)(l + 1) * sumC + tile0 : nullptr;
            if (h->math == PK_PWG_MATH_BF16X3 || h->math == PK_PWG_MATH_F16X3) {
                const bool half = h->math == PK_PWG_MATH_F16X3;
                if (half) {
                    a.bias = h->d_bias_h.as<float>() + (size_t)l * (G + R + SK);
                    a.k1 = h->k1[l];
                    a.i0 = (float)std::ldexp(0.70710678118654752440, -(14 + h->k2o[l]));
                    a.i1 = (float)std::ldexp(1.0, -(14 + h->k2s[l]));
                }
                a.w1 = reinterpret_cast<const float*>((half ? h->d_w1h : h->d_w1b).as<char>() + (size_t)l * B3_W1_BYTES);
                a.w2 = reinterpret_cast<const float*>((half ? h->d_w2h : h->d_w2b).as<char>() + (size_t)l * B3_W2_BYTES);
                const dim3 blk(LAYER_WAVES * 64);
                if (nz && l == 0) {
                    a.noise = d_noise + (size_t)tile0 * TILE;
                    a.noise_n = (long)sumS - (long)tile0 * TILE;
                    a.nz_tab = h->d_nz.as<float>();
                    a.nz_wmax = h->first_wmax;
                    a.nz_bmax = h->first_bmax;
                    a.k1 = h->nz_k1;   // the accumulators of this launch: 2^(kx + nz_k1) * (pre-activation)
                }
                if (planes && (guard || sample) && !all_first) {   // ... and the scale guard's maxima from the epilogue (AMAX instantiations)
                    a.amax_out = amax + (size_t)(c.layers + 1) * B + (size_t)l * B * PWG_AMAX_PARTS;
                    amax_parts_used = true;
                    if (nz && l == 0) {
                        PK_LAUNCH(ctx, "pwg_layer_h3", (k_pwg_layer_b3<true, true, 0, false, true, true, true>), dim3(grid), blk, 0, a);
                    } else if (gen) {
                        if (l == 0) PK_LAUNCH(ctx, "pwg_layer_h3", (k_pwg_layer_b3<true, true, 0, true, true, true>), dim3(grid), blk, 0, a);
                        else PK_LAUNCH(ctx, "pwg_layer_h3", (k_pwg_layer_b3<false, true, 0, true, true, true>), dim3(grid), blk, 0, a);
                    } else {
                        if (l == 0) PK_LAUNCH(ctx, "pwg_layer_h3", (k_pwg_layer_b3<true, true, 0, false, true, true>), dim3(grid), blk, 0, a);
                        else PK_LAUNCH(ctx, "pwg_layer_h3", (k_pwg_layer_b3<false, true, 0, false, true, true>), dim3(grid), blk, 0, a);
                    }
                } else if (planes) {   // (half) x as pre-split planes
                    if (gen) {
                        if (l == 0) PK_LAUNCH(ctx, "pwg_layer_h3", (k_pwg_layer_b3<true, true, 0, true, true>), dim3(grid), blk, 0, a);
                        else PK_LAUNCH(ctx, "pwg_layer_h3", (k_pwg_layer_b3<false, true, 0, true, true>), dim3(grid), blk, 0, a);
                    } else if (nz && l == 0) {
                        PK_LAUNCH(ctx, "pwg_layer_h3", (k_pwg_layer_b3<true, true, 0, false, true, false, true>), dim3(grid), blk, 0, a);
                    } else {
                        if (l == 0 || all_first) PK_LAUNCH(ctx, "pwg_layer_h3", (k_pwg_layer_b3<true, true, 0, false, true>), dim3(grid), blk, 0, a);
                        else PK_LAUNCH(ctx, "pwg_layer_h3", (k_pwg_layer_b3<false, true, 0, false, true>), dim3(grid), blk, 0, a);
                    }
                } else if (gen) {   // hop != 256: frame / phase per sample (GEN kernels)
                    if (half) {
                        if (l == 0) PK_LAUNCH(ctx, "pwg_layer_h3", (k_pwg_layer_b3<true, true, 0, true>), dim3(grid), blk, 0, a);
                        else PK_LAUNCH(ctx, "pwg_layer_h3", (k_pwg_layer_b3<false, true, 0, true>), dim3(grid), blk, 0, a);
                    } else {
                        if (l == 0) PK_LAUNCH(ctx, "pwg_layer_b3", (k_pwg_layer_b3<true, false, 0, true>), dim3(grid), blk, 0, a);
                        else PK_LAUNCH(ctx, "pwg_layer_b3", (k_pwg_layer_b3<false, false, 0, true>), dim3(grid), blk, 0, a);
                    }
                } else if (half) {
                    if (l == 0) PK_LAUNCH(ctx, "pwg_layer_h3", (k_pwg_layer_b3<true, true>), dim3(grid), blk, 0, a);
                    else if (h->dbg != 0) PK_TRY(pwg_ablation_launch<PK_PROFILE_BUILD != 0>(ctx, h->dbg, grid, blk, a));
                    else PK_LAUNCH(ctx, "pwg_layer_h3", (k_pwg_layer_b3<false, true>), dim3(grid), blk, 0, a);
                } else {
                    if (l == 0) PK_LAUNCH(ctx, "pwg_layer_b3", (k_pwg_layer_b3<true, false>), dim3(grid), blk, 0, a);
                    else PK_LAUNCH(ctx, "pwg_layer_b3", (k_pwg_layer_b3<false, false>), dim3(grid), blk, 0, a);
                }
            } else if (gen) {
                if (l == 0) PK_LAUNCH(ctx, "pwg_layer", (k_pwg_layer<true, true>), dim3(grid), dim3(LAYER_WAVES * 64), 0, a);
                else PK_LAUNCH(ctx, "pwg_layer", (k_pwg_layer<false, true>), dim3(grid), dim3(LAYER_WAVES * 64), 0, a);
            } else if (l == 0)
                PK_LAUNCH(ctx, "pwg_layer", (k_pwg_layer<true, false>), dim3(grid), dim3(LAYER_WAVES * 64), 0, a);
            else
                PK_LAUNCH(ctx, "pwg_layer", (k_pwg_layer<false, false>), dim3(grid), dim3(LAYER_WAVES * 64), 0, a);
            if ((guard || sample) && planes && all_first)   // (the all-FIRST measurement configuration keeps the separate pass)
                PK_LAUNCH(ctx, "pwg_planes_amax", k_pwg_planes_amax, dim3(ntile), dim3(256), 0, a.xout, a.tile_t0, a.gen.tile_utt,
                          a.tile_kx_out, amax + (size_t)(l + 1) * B);
        }
        }
        h->last_x_final = c.layers & 1;
        if (amax_parts_used)
            PK_LAUNCH(ctx, "pwg_amax_fold", k_pwg_amax_fold, dim3(pk_div_up(c.layers * B, 256)), dim3(256), 0,
                      amax + (size_t)(c.layers + 1) * B, amax + B, c.layers * B);
    }
    if (sample) {   // deferred verdict: the maxima travel to pinned memory behind an event; pwg_poll_sample judges them later
        const size_t n_am = (size_t)(c.layers + 1) * B;   // (fits: `sample` is only set for B <= PWG_SAMPLE_MAX_B; buffer and event: pk_pwg_finalize)
        PK_HIP(hipMemcpyAsync(h->host_sample, amax, n_am * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        PK_HIP(hipMemcpyAsync(h->host_sample + n_am, h->ws_nmax.p, (size_t)B * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        PK_HIP(hipEventRecord(h->ev_sample, ctx->stream));
        h->sample_B = B;
        h->sample_frames.assign(frames, frames + B);
        h->sample_pending = true;
    }
    if (!(guard && attempt == 0)) break;
    {   // the verdict of the guard, inside the call: one stream synchronisation and two small blocking copies -- this is the
        // stall of the FIRST inference after finalize (and of every inference under scale_guard 2); not legal under stream capture
        std::vector<float> am((size_t)(c.layers + 1) * B), nmax(B);
        PK_HIP(hipStreamSynchronize(ctx->stream));
        PK_HIP(hipMemcpy(am.data(), amax, am.size() * sizeof(float), hipMemcpyDeviceToHost));
        PK_HIP(hipMemcpy(nmax.data(), h->ws_nmax.p, (size_t)B * sizeof(float), hipMemcpyDeviceToHost));
        const float worst = pwg_guard_verdict(h, am.data(), nmax.data(), B, frames);
        h->guard_done = true;
        if (worst <= GUARD_MAX_LOG2) break;
        h->planes_on = false;
        h->fell_back = true;
        planes = false;
    }
    }
    // ---- last layers
    {
        PwgLastArgs a;
        a.skip = h->ws_skip.as<float>();
        a.w1 = h->d_l1.as<float>();
        a.b1 = h->d_l1b.as<float>();
        a.w2 = h->d_l2.as<float>();
        a.b2 = h->l2_bias;
        a.scale = (float)std::sqrt(1.0 / c.layers);
        a.tile_t0 = d_tab + o_tile;
        a.Ttot = Ttot;
        a.wav = d_wav;
        a.kw = h->kw_last;
        a.gen = gtab;
        if (h->math == PK_PWG_MATH_F32) {
            if (gen) PK_LAUNCH(ctx, "pwg_last", k_pwg_last<true>, dim3(sumC), dim3(512), 0, a);
            else PK_LAUNCH(ctx, "pwg_last", k_pwg_last<false>, dim3(sumC), dim3(512), 0, a);
        } else {
            a.w1 = reinterpret_cast<const float*>(h->d_l1h.as<char>());
            if (gen) PK_LAUNCH(ctx, "pwg_last_h3", k_pwg_last_h3<true>, dim3(sumC), dim3(512), 0, a);
            else PK_LAUNCH(ctx, "pwg_last_h3", k_pwg_last_h3<false>, dim3(sumC), dim3(512), 0, a);
        }
    }
    if (flags & PK_HOST_IO) {
        PK_HIP(hipMemcpyAsync(wav, d_wav, (size_t)sumS * 4, hipMemcpyDeviceToHost, ctx->stream));
        PK_HIP(hipStreamSynchronize(ctx->stream));
    }
    return PK_OK;
}

extern "C" int pk_pwg_debug_read(pk_pwg* h, int32_t what, int32_t b, float* host_out, int64_t n_floats) {
    if (!h || !host_out) PK_FAIL(PK_EINVAL, "pk_pwg_debug_read: NULL argument");
    if (h->last_Ttot == 0) PK_FAIL(PK_ESTATE, "pk_pwg_debug_read: no inference has run");
    if (b < 0 || b >= (int)h->last_frames.size()) PK_FAIL(PK_EINVAL, "pk_pwg_debug_read: utterance out of range");
    pk_ctx* ctx = h->ctx;
    PK_DEVICE(ctx->device);
    const long S = (long)h->last_frames[b] * h->hop;
    if (what == 0) {
        // sample-rate aux contribution of layer 0: conv1x1_aux(upsample_net(c)) (G, S_b), recomputed
        if (n_floats != (int64_t)G * S) PK_FAIL(PK_ESHAPE, "pk_pwg_debug_read: expected %ld floats", (long)G * S);
        PK_TRY(h->ws_dbg.reserve((size_t)G * S * 4));
        const float* P = h->ws_P.as<float>() + (size_t)P_LEAD * h->last_ldp;
        dim3 grid(h->last_frames[b], G);
        PK_LAUNCH(ctx, "pwg_aux_debug", k_pwg_aux_debug, grid, dim3(h->hop), 0, P, h->last_ldp, 0,
                  h->d_uptab.as<float>(), h->last_cuL[b], h->last_frames[b], h->hop, h->ws_dbg.as<float>());
        PK_HIP(hipStreamSynchronize(ctx->stream));
        PK_HIP(hipMemcpy(host_out, h->ws_dbg.p, (size_t)G * S * 4, hipMemcpyDeviceToHost));
        return PK_OK;
    }
    if (what == 3 && h->last_planes) PK_FAIL(PK_EUNSUPPORTED, "pk_pwg_debug_read: the planes path keeps no block maxima");
    if (what == 3) {
        // max|x| per 32-sample block of the final residual stream, as the last layer's epilogue left it for a next
        // layer's operand scale (block-scaled split-fp16 path only)
        const long nblk = (S + XBLK - 1) / XBLK;
        if (n_floats != nblk) PK_FAIL(PK_ESHAPE, "pk_pwg_debug_read: expected %ld floats", nblk);
        const pk_dbuf& xe = h->last_x_final ? h->ws_xe1 : h->ws_xe0;
        PK_HIP(hipStreamSynchronize(ctx->stream));
        PK_HIP(hipMemcpy(host_out, xe.as<float>() + h->last_toff[b] / XBLK, (size_t)nblk * 4, hipMemcpyDeviceToHost));
        return PK_OK;
    }
    const float* src;
    int rows;
    switch (what) {
        case 1: src = h->last_x_final ? h->ws_x1.as<float>() : h->ws_x0.as<float>(); rows = R; break;
        case 2: src = h->ws_skip.as<float>(); rows = SK; break;
        default: PK_FAIL(PK_EINVAL, "pk_pwg_debug_read: unknown tap %d", what);
    }
    if (n_floats != (int64_t)rows * S)
        PK_FAIL(PK_ESHAPE, "pk_pwg_debug_read: expected %ld floats, got %lld", rows * S, (long long)n_floats);
    PK_HIP(hipStreamSynchronize(ctx->stream));
    if (what == 1 && h->last_planes) {
        // x as planes: whole blocks to the host, decoded there -- (hi + lo) / 2^k of the block's maximum (xe)
        const long nblk = (S + XBLK - 1) / XBLK, blk0 = h->last_toff[b] / XBLK;
        std::vector<uint16_t> raw((size_t)nblk * XBLK_FLOATS * 2);
        int kfin = 0;   // scale exponent of the utterance's final x: the last row of the tile table, any tile of the utterance
        PK_HIP(hipMemcpy(raw.data(), reinterpret_cast<const char*>(src) + (size_t)blk0 * XBLK_FLOATS * 4, raw.size() * 2, hipMemcpyDeviceToHost));
        PK_HIP(hipMemcpy(&kfin, h->ws_tkx.as<int>() + (size_t)h->cfg.layers * h->last_ntiles + h->last_cuC[b], sizeof(int), hipMemcpyDeviceToHost));
        for (long bi = 0; bi < nblk; ++bi) {
            const double inv = std::ldexp(1.0, -kfin);
            for (int ch = 0; ch < R; ++ch)
                for (int sidx = 0; sidx < XBLK && bi * XBLK + sidx < S; ++sidx) {
                    const int cg = ch >> 4, w16 = ch & 15, hh = (w16 >> 2) & 1, e = 4 * (w16 >> 3) + (w16 & 3);
                    const uint16_t* v = raw.data() + (size_t)bi * XBLK_FLOATS * 2 + (size_t)(2 * cg + hh) * 512 + sidx * 16;
                    host_out[(size_t)ch * S + bi * XBLK + sidx] = (float)(((double)f16_to_f32(v[e]) + (double)f16_to_f32(v[8 + e])) * inv);
                }
        }
        return PK_OK;
    }
    // blocked layout: channel ch of this utterance = S/32 pieces of 32 floats, one per block (+ a shorter last
    // piece when S is not a multiple of 32, hop != 256)
    for (int ch = 0; ch < rows; ++ch) {
        if (S / XBLK > 0)
            PK_HIP(hipMemcpy2D(host_out + (size_t)ch * S, XBLK * sizeof(float),
                               src + xoff(h->last_toff[b]) + (size_t)ch * XBLK, XBLK_FLOATS * sizeof(float),
                               XBLK * sizeof(float), S / XBLK, hipMemcpyDeviceToHost));
        if (S % XBLK)
            PK_HIP(hipMemcpy(host_out + (size_t)ch * S + S / XBLK * XBLK,
                             src + xoff(h->last_toff[b] + S / XBLK * XBLK) + (size_t)ch * XBLK,
                             (size_t)(S % XBLK) * sizeof(float), hipMemcpyDeviceToHost));
    }
    return PK_OK;
}

extern "C" void pk_pwg_destroy(pk_pwg* h) {
    if (!h) return;
    pk_device_guard _dg(h->ctx->device);
    (void)hipStreamSynchronize(h->ctx->stream);
    pk_dbuf* bufs[] = {&h->d_first_w, &h->d_first_b, &h->d_convin_wT, &h->d_uptab, &h->d_mu, &h->d_sigma,
                       &h->d_w1, &h->d_w2, &h->d_bias, &h->d_w1b, &h->d_w2b, &h->d_w1h, &h->d_w2h, &h->d_waux, &h->d_nz, &h->d_l1, &h->d_l1h, &h->d_l1b, &h->d_l2,
                       &h->d_bias_h, &h->ws_xe0, &h->ws_xe1,
                       &h->ws_mel, &h->ws_cin, &h->ws_noise, &h->ws_wav, &h->ws_c0, &h->ws_P,
                       &h->ws_x0, &h->ws_x1, &h->ws_skip, &h->ws_dbg, &h->ws_tab, &h->d_cl, &h->ws_nmax, &h->ws_tkx, &h->ws_amax};
    for (auto* b : bufs) b->release();
    if (h->host_sample) (void)hipHostFree(h->host_sample);
    if (h->ev_sample) (void)hipEventDestroy(h->ev_sample);
    delete h;
}
