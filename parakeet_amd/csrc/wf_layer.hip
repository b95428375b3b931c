// wf_layer.hip -- one WaveFlow residual layer for one autoregressive row as ONE kernel (64- or 128-channel model,
// split-fp16 math): the (3,3) dilated causal conv over the 3-row input ring + condition_proj as one contraction
// (K = 9 C + 96), gated tanh, the res half of the out projection with the residual into the next layer's ring, and this
// layer's share of the flow's (logs, b) -- the skip half of the out projection folded with output_proj (pk_wf_layer.h).
//
// Reference: parakeet/models/waveflow.py ResidualBlock.add_input :248-283 (conv2d over the row buffer :268-274,
// condition_proj :275, gate :276-277, out_proj + chunk + residual :279-282), ResidualNet.add_input :368-392.
//
// Structure (as the Parallel WaveGAN layer kernel): positions are the MFMA N dimension -- a wave owns 32 positions and
// ALL 2C gate channels (2C/32 accumulator tiles); weights are the A operand, streamed through LDS in 48 KB slabs,
// double buffered, each slab used by the 8 waves = 256 positions of the workgroup; the gated activations never leave
// the accumulator registers (register r of the first contraction IS the B operand element of k-step r / 8 of the
// second: K order of W2 permuted at pack time); the out-projection weights follow the conv weights through the same
// slab buffers (32 KB resident in round 2; at 128 channels they are 128 KB).
//
// Round 3: the layer inputs are stored as pre-split fp16 planes (pk_wf_layer.h).  In round 2 a wave tile spent 12.5 k
// of its 30 k cycles in the VALU (rocprofv3: SQ_ACTIVE_INST_VALU 8.1 M quad-cycles per launch against 43 M matrix
// cycles / 4 SIMDs), most of it the hi / lo split of every operand -- nine times per stored value, once per tap that
// reads it -- and the scalar loads that fed it (8 dword loads per k-step: the vmcnt counter holds 63 loads, i.e. less
// than 8 k-steps of prefetch).  Now the producer's epilogue splits each value once; a k-step's operand is two 16-byte
// loads and one v_pk_mul_f16 per register (the power of two that brings the tap's block to the tile's common scale).
// The operand ring is two weight slabs deep (k-step k's registers are refilled with k-step k + 2 slabs' operand as
// soon as k has been consumed; the slab loop is unrolled by two so that ring slots are compile-time registers) and
// the epilogue's old values (residual input, running skip sum) are requested before the gate, a few hundred VALU
// instructions ahead of their use, not right before the 24 MFMAs of their pass.
// Rows before the sequence start are skipped as taps (ntap = 3, 6, 9), never stored as zeros.
#include "pk_wf_layer.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "pk_grid.h"
#include "pk_split.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 pkh2 __attribute__((ext_vector_type(2)));

#ifndef PK_WF_BIG16
#define PK_WF_BIG16 1     // the register diet of the 128-channel kernel with fp16 operands too (0: its round-4 form, 8 spilled registers).
#endif                    // Before the linear weight addresses the diet cost this vector-bound kernel 6 %; with them it gains 1 %
#ifndef PK_WF_LINW
#define PK_WF_LINW 1      // the weight chunks' source addresses as a linear function of the thread index (0: through the kt_w table, as
#endif                    // rounds 2 - 4 did; kept for the A/B)
#ifndef PK_WF_LATE_REFILL
#define PK_WF_LATE_REFILL 0   // experiment of HISTORY 9.9 (1: the operand ring's slot refilled one k-step later -- made the three-waves-per-SIMD kernels fail in EVERY run)
#endif
// THE OP_SEL RULE (round 6, HISTORY 10) -- round 5's "cause (ii)", found.  The folded skip path sums `pl += w0 z, pb += w1 z` per
// gated channel pair; hipcc's SLP vectoriser packs (pl, pb) into a register pair and emits, per z pair,
//     v_pk_fma_f32 v[2:3], v[w0:w1], v[z:z+1], v[2:3] op_sel_hi:[1,0,1]      {pl, pb} += {w0, w1} * z.lo   (high half from a LOW register: fine)
//     v_pk_fma_f32 v[2:3], v[w2:w3], v[z:z+1], v[2:3] op_sel:[0,1,0]         {pl, pb} += {w2, w3} * z.hi   (LOW half from a HIGH register)
// On the MI355X the second form now and then DROPS ITS PRODUCT in the low half for lanes 48 - 63 (the last quarter of the wave's pass through
// the vector ALU) -- the result is the addend alone -- when another wave of the SIMD is executing matrix instructions: the logs sum of the
// 16 positions those lanes hand over in the half-wave exchange lost one term, an error of 1e-3, and nothing else in the tile was touched (the
// stored planes and every b sum bit-identical to the 8-wave kernel's: 33 000 replays of one launch, tools/r06_wf_replay_call.sh).  7 - 25 % of
// the calls of BASELINE config 5's shape in 12-wave workgroups, 1 - 5 % at 128 channels, every call in an instantiation that happened to
// overlap its epilogue with more matrix work.  Reproduced in isolation (tools/micro/mfma_chain_hazard.hip `b`: 1 in 3e7 per instruction with
// two or three waves per SIMD, never with one; wait states anywhere, other registers, full waits do not help; the same sums by scalar
// v_fma_f32, or by packed FMAs WITHOUT op_sel, never fail in 4e9).  LLVM knows no such hazard, no table lists one.  The rule: no packed fp32
// instruction whose LOW half reads a HIGH source register (`op_sel:[..1..]`) in code that runs beside matrix instructions.
// 1: pl and pb pass through an asm statement as two scalar operands behind the sums -- the vectoriser then keeps them scalar (v_fmac_f32; same
// values bit for bit; 0 wrong of 33 000 replays, 0 of 410 whole calls).  0: as compiled before (the A/B).  tools/pk_opsel_lint.py checks every
// kernel of the library for the instruction form (tests/test_isa_rules_cpu.py: none is left).
#ifndef PK_WF_SCALAR_SUMS
#define PK_WF_SCALAR_SUMS 1
#endif
#ifndef PK_WF_AHEAD128
#define PK_WF_AHEAD128 1   // A fragments of the 128-channel kernel this many co-tiles ahead (round 5: 2; 1 = rounds 3 - 4)
#endif
#ifndef PK_WF_RING128
#define PK_WF_RING128 6   // operand ring of the 128-channel kernel in k-steps (two slabs).  Round 5: 3 ... 6 compile to the same
#endif                    // register use once nothing is hoisted into the slab loop (LEAN below): the spills were never the ring

namespace {
constexpr int WAVE_T = 32;
constexpr int BLK_M_BYTES = WFL_MP * 128;  // bytes per 32-position block of the condition planes

// W = waves per workgroup: 8 (two per SIMD, 256 registers each), or 12 at 64 channels (three per SIMD, 168 registers: the 11
// wave tiles a workgroup owns at the benchmark's shape -- 2 592 tiles on 256 CUs -- run as ONE round instead of 8 + 3)
// W = 6 (round 4, 64 channels): TWO workgroups per CU, six waves and three 24 KB slabs each (3 k-steps per slab) -- the same three
// waves per SIMD and 168 registers as W = 12, but the two workgroups are independent: one's slab barriers, prologue and epilogue
// lie under the other's MFMAs (the twelve waves of ONE workgroup move in lockstep; the counters of the 12-wave kernel show the
// matrix pipe 30 % busy with half the wave cycles waiting).  Price: the weights travel from L2 to LDS twice per CU.
// F16 (round 5): with fp16 operands only the hi parts of the weights are multiplied -- only they travel to LDS (HIW): half the
// weight bytes per launch from L2 and into LDS.  64 channels: 24 KB slabs of the same six k-steps; 128 channels: the 48 KB slab
// holds six k-steps instead of three, half as many slabs and barriers.  PK_WF_HIW=0: both parts as in rounds 2 - 4 (the A/B).
#ifndef PK_WF_HIW
#define PK_WF_HIW 1
#endif
template <int CT, int W = 8, bool F16 = false>
struct Shape {
    static constexpr bool HIW = F16 && PK_WF_HIW;
    static constexpr int SLAB_BYTES = (W == 6 || (HIW && CT == 2)) ? 24 * 1024 : 48 * 1024;   // one weight slab in LDS; three of them
    static constexpr int SLAB_CH = SLAB_BYTES / 16;                     // 16-byte chunks per slab buffer
    static constexpr int THREADS = W * 64;
    static constexpr int C = 32 * CT;
    static constexpr int KS_TAP = C / 16;                 // k-steps per conv tap: 4 / 8
    static constexpr int KS1 = 9 * KS_TAP + WFL_KS_COND;  // 42 / 78
    static constexpr int NQ = 2 * CT;                     // accumulator tiles of the first contraction: 4 / 8
    static constexpr int KCH1 = 2 * NQ * 64;              // chunks per k-step of W1 in memory (hi | lo): 512 / 1024
    static constexpr int KCHL = (HIW ? 1 : 2) * NQ * 64;  // ... of them brought to LDS
    static constexpr int SLAB = SLAB_CH / KCHL;           // k-steps per main slab: 6 / 3 (HIW: 6 / 6)
    static constexpr int CPT1 = SLAB * KCHL / THREADS;    // chunks per thread per main slab: 6 (4 with 12 waves; HIW: 3 / 2 / 6)
    static constexpr int KS2 = C / 16;                    // k-steps of the out projection (res half): 4 / 8
    static constexpr int KCH2 = 2 * CT * 64;              // chunks per k-step of W2: 256 / 512
    static constexpr int SLAB2 = 4;                       // k-steps per W2 slab: 16 / 32 KB
    static constexpr int NS2 = KS2 / SLAB2;               // W2 slabs: 1 / 2
    static constexpr int CPT2 = (SLAB2 * KCH2 + THREADS - 1) / THREADS;   // chunks per thread per W2 slab: 2 / 4 (12 waves: 2, the
                                                          // second one clamped to the slab's last chunk for waves 4 - 11)
    static constexpr int BLK_BYTES = C * 128;             // bytes per 32-position block of the feature planes
    static constexpr int RING = CT == 2 ? (W != 8 ? 6 : 9) : PK_WF_RING128;   // operand ring depth in k-steps: 9 / 6 (64 channels: 12
                                                          // would leave the A fragments two register quads -- every LDS read
                                                          // latency exposed; 12 waves: 6, what 168 registers hold)
    static_assert(SLAB * KCHL % THREADS == 0, "a main slab is a whole number of chunks per thread");
    static_assert(SLAB2 * KCH2 <= SLAB_CH && CPT2 <= CPT1, "an out-projection slab fits a slab buffer and the staging registers");
};

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
// The stored pair of a layer input (producer side, once per value): hi = fp16_rne(s x), lo = fp16_rne(s x - hi).  Round to
// nearest, not toward zero as in the in-register splits above: |lo| is at most half an ulp of hi (one more bit for the pair),
// and hi alone IS the correctly rounded fp16 of the value -- the fp16-operand mode reads only the hi plane.  (The block scale
// keeps |s x| below 2^14, so the conversion cannot overflow.)
__device__ __forceinline__ void store_pair8(const float (&v)[8], float s, f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float t = v[e] * s;
        const _Float16 h = (_Float16)t;
        hi[e] = h;
        lo[e] = (_Float16)(t - (float)h);
    }
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
// two gates at once on packed fp32 math (v_pk_mul / v_pk_add / v_pk_fma take a register pair per issue slot; the clamp and
// the transcendentals stay per element), as in pwg.hip
__device__ __forceinline__ f32x2 gated_s2(f32x2 a, f32x2 b, float ca, float cb) {
    f32x2 ta = a * ca, tb = b * cb;
    ta[0] = __builtin_amdgcn_fmed3f(ta[0], -28.853900817779268f, 28.853900817779268f);
    ta[1] = __builtin_amdgcn_fmed3f(ta[1], -28.853900817779268f, 28.853900817779268f);
    f32x2 ea, eb, rc;
    ea[0] = __builtin_amdgcn_exp2f(ta[0]);
    ea[1] = __builtin_amdgcn_exp2f(ta[1]);
    eb[0] = __builtin_amdgcn_exp2f(tb[0]);
    eb[1] = __builtin_amdgcn_exp2f(tb[1]);
    const f32x2 num = ea * (-PK_UNIT_SCALE) + PK_UNIT_SCALE;
    const f32x2 den = (ea + 1.f) * (eb + 1.f);
    rc[0] = __builtin_amdgcn_rcpf(den[0]);
    rc[1] = __builtin_amdgcn_rcpf(den[1]);
    return num * rc;
}
// biased exponent of a block maximum, clamped as blk_scale_exp clamps it (pk_split.h)
__device__ __forceinline__ int amax_exp(unsigned bits) {
    const int e = (int)(bits >> 23);
    return e < PK_EXP_MIN ? PK_EXP_MIN : (e > PK_EXP_MAX ? PK_EXP_MAX : e);
}
// eight copies of the fp16 value 2^-d (d >= 0; subnormal / zero beyond 2^-14: what the rescaled values would be anyway)
__device__ __forceinline__ f16x8 pow2_neg_h8(int d) {
    const float f = __uint_as_float((unsigned)(127 - min(d, 60)) << 23);
    const unsigned u = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(f, f));
    const u32x4 v = {u, u, u, u};
    return __builtin_bit_cast(f16x8, v);
}
// v of lane ^ 32
__device__ __forceinline__ float xor32(float v, int lane, bool own_lane) {
#ifndef PK_HIPEMU
    if (own_lane) return __int_as_float(__builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __float_as_int(v)));
#endif
    return __shfl_xor(v, 32);
}
__device__ __forceinline__ f16x8 ld_h8(const char* p) { return *reinterpret_cast<const f16x8*>(p); }
__device__ __forceinline__ void st_h8(char* p, f16x8 v) { *reinterpret_cast<f16x8*>(p) = v; }
// Round 6: the 64-channel kernels store the next layer's planes with non-temporal stores -- 21 MB per launch that the kernel
// boundary then does not have to write back from L2 (one-box A/B, profiles/r06_wf_ab.txt: 64 channels -1.5 % in both maths;
// 128 channels -0.4 % / +0.6 %: left as plain stores).  PK_WF_NT_STORE=0: plain stores everywhere (the A/B).
#ifndef PK_WF_NT_STORE
#define PK_WF_NT_STORE 1
#endif
template <bool NT>
__device__ __forceinline__ void st_h8_out(char* p, f16x8 v) {
    if constexpr (NT && PK_WF_NT_STORE) __builtin_nontemporal_store(v, reinterpret_cast<f16x8*>(p));
    else *reinterpret_cast<f16x8*>(p) = v;
}

// An opaque zero (ON) or a literal one: values derived from it cannot be hoisted out of the place it is renewed.  The 12-wave
// kernel (168 registers) uses it to keep round- and epilogue-only address arithmetic from living in registers through the slab
// loop -- hoisted out of the round loop they were spilled at the top and reloaded in the hot loop, where a scratch load's wait
// drains the operand prefetch (vmcnt counts in order).  The 8-wave kernels are compiled exactly as before.
template <bool ON>
__device__ __forceinline__ int opaque_zero() {
    int z = 0;
    if constexpr (ON) asm volatile("" : "+s"(z));
    return z;
}

// NT = ntap / 3 (1, 2, 3 rows of the ring exist): a template parameter so that the slab loop unrolls completely.  With a
// run-time loop the operand ring is carried around the back edge, hipcc's register allocator does not keep the refilled
// slots in place, and the copies it inserts at the loop end wait for every load in flight (vmcnt(0) once per slab pair).
//
// Weights: the stream of a (layer, NT) is nslab conv slabs followed by NS2 out-projection slabs, through THREE LDS buffers.
// Slab g + 2 is requested at the start of slab g, before any of slab g's operand loads, and written to LDS at its end:
// vmcnt counts loads in order, so waiting for a weight load also waits for every older load -- with the weights
// requested first, "older" is only the operands the next slab needs anyway, and the operands of slab g + 2 (requested
// during slab g) stay in flight across the wait.  (With two buffers the weights of slab g + 1 would be requested during
// slab g - 1's operand prefetch and their wait would drain it: measured in the ISA as vmcnt(6) two k-steps after a load
// that is needed twelve k-steps later.)
// ABL (profiling only, PK_WF_ABLATE, results are wrong when set): 1 = the operand ring is not refilled after the prologue
// (no operand traffic), 4 = no epilogue loads / stores, 8 = the weight slabs are not reloaded after the prologue (barriers
// stay); sums combine.  16 = s_memtime stamps of workgroup 5 (results stay right), 64 = the prologue in which every
// thread brings its share of slabs 0 and 1 before the barrier (results stay right; 80 = both)
//
// F16 = the reference's own inference precision for this model (examples/waveflow/synthesize.py:40 runs under
// paddle.amp.auto_cast: fp16 conv operands, fp32 accumulation): every product is ONE fp16 MFMA of the operands rounded to
// nearest -- weights: their stored hi part (rounded to nearest at pack time); activations: the hi plane alone (store_pair8
// rounds it to nearest: half the operand bytes, no arithmetic on the way to the MFMA); gate outputs: one conversion.  The
// layer inputs stay the same 22-bit pairs, so only the products lose precision, not the residual stream.  A third of the matrix
// work; not the default.
// MULTI: the launch may hold several layers of the row (a.nl > 1: grid barriers between them, pk_grid.h).  A separate
// instantiation because the loop around the layer body is not free: it lengthens live ranges (the 128-channel kernel went from
// 28 to 63 spilled registers, 88 -> 128 us per launch with fp16 operands) -- the one-layer kernels are compiled without it.
template <int CT, int NT, int ABL = 0, bool F16 = false, int W = 8, bool MULTI = false>
__global__ __launch_bounds__(W * 64, W == 8 ? 2 : 3) void k_wf_layer_p(WflLaunch a) {
    typedef Shape<CT, W, F16> S;
    constexpr int C = S::C, NQ = S::NQ, SLAB = S::SLAB, THREADS = S::THREADS;
    constexpr int RING = (W != 8 && F16) ? 9 : S::RING;   // (fp16 operands: a ring slot is one vector, nine fit the 168 registers)
    static_assert(W == 8 || ((W == 12 || W == 6) && CT == 2 && (ABL == 0 || ABL == 16 || ABL == 64 || ABL == 80 || ABL == 128)), "12- / 6-wave workgroups: the 64-channel model (ablations: the trace, the old prologue, the slab verifier)");
    constexpr int SLAB_CH = S::SLAB_CH;
    // LEAN: the kernels that live at their register limit (three waves per SIMD: 168; 128 channels: 128 of the 256 are
    // accumulators) derive round-, slab- and epilogue-only coordinates from opaque zeros so that nothing thread-invariant is
    // hoisted to the kernel's top, spilled there and reloaded inside the slab loop (round 4 for the 12-wave kernel; round 5 for
    // the 128-channel one: 29 - 31 spilled registers and 2 665 scratch instructions in the unrolled slab loop -> 0;
    // fp16 operands: 8 -> 0).  The 8-wave 64-channel kernels are compiled exactly as before.
    // (fp16 operands at 128 channels -- 8 spilled registers, bound by vector issue: with the weight addresses still read from
    // the kt_w table the recomputed coordinates cost more than the spills did, 92.7 -> 98.5 us per launch; with the linear
    // addresses 94.5 -> 93.3: profiles/r05_wf_ab.txt.)
    constexpr bool BIG = CT == 4 && (!F16 || PK_WF_BIG16);   // the 128-channel default-math kernel: the round-5 register diet
    constexpr bool LEAN = W != 8 || BIG;
    static_assert(!MULTI || ABL == 0, "multi-layer launches: no ablations");
    constexpr int ntap = 3 * NT;
    constexpr int nks_conv = S::KS_TAP * ntap;
    constexpr int nks = nks_conv + WFL_KS_COND;
    constexpr int nslab = nks / SLAB;    // C = 64: 3, 5, 7;  C = 128: 10, 18, 26
    constexpr int G = nslab + S::NS2;    // slabs of the weight stream
    __shared__ __attribute__((aligned(16))) f16x8 wbuf[3][SLAB_CH];   // three weight slabs: 144 KB
    __shared__ __attribute__((aligned(16))) float lb[(CT == 2 ? 5 : 3) * C];   // b2r [C] | wso [2C] | fused step: input_proj w [C] | b [C]
    // per source (the taps whose row exists, then the condition block): where its B operand lives (run-time: which ring slot
    // a tap reads depends on the row); per logical k-step: the packed weight k-step that multiplies it
    __shared__ long tp_off[ntap + 1];     // byte offset from in0 of (position 0, octet 0) of the source
    __shared__ long tp_am[ntap + 1];      // element offset from in_amax0 of block 0 of the source's block maxima
    __shared__ int tp_shift[ntap + 1];    // position shift of the tap
    __shared__ int tp_blk[ntap + 1];      // bytes per 32-position block of the source (C or 96 channels)
    __shared__ unsigned kt_w[nks];        // byte offset of the packed k-step of W1
    // per layer of the launch: what the slab loop
    // needs stays in scalar registers; what only a tile's start or epilogue needs -- and the fused step's arguments -- is parked
    // in LDS (the kernel has no scalar registers to spare: kernel arguments referenced inside the loops are loaded up front and
    // held, 20 of them per layer, and the spills go to vector registers the 12-wave and 128-channel kernels do not have)
    struct Cold {
        float* out;
        unsigned* out_amax;
        int first, k1, k2res, has_out;
        const float* step_z;
        float* step_x;
        const float* step_w_in;
        const float* step_b_in;
        float* step_h0;
        unsigned* step_h0_amax;
        float step_b_logs, step_b_b;
    };
    __shared__ Cold cold;
    // ---- the layers of this launch: nl == 1, or the whole residual stack of the row with a grid barrier between two layers
    // (layer l + 1 reads, 2^(l+1) positions to either side, what OTHER workgroups wrote in layer l).  EVERYTHING of a layer sits
    // inside this loop, thread coordinates included (derived from an opaque zero renewed per layer): with the loop around a
    // straight-line body the compiler hoists each thread-invariant address out of it and keeps it in a register through all
    // layers (+ 40 vector registers, spills in the 12-wave and 128-channel kernels).
#pragma unroll 1
    for (int li = 0; li < (MULTI ? a.nl : 1); ++li) {
    if (MULTI && li > 0) pk_grid_barrier(a.bar, (unsigned)li * gridDim.x, a.err);   // (every wave of this workgroup is past its epilogue too)
    const int oz = opaque_zero<MULTI>();
    const int tid = (int)threadIdx.x + oz, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, hh = lane >> 5;
    const int ntiles = a.npos_alloc / WAVE_T;
    const f16x8* w1 = nullptr;
    const f16x8* w2 = nullptr;
    const char* in0b = nullptr;
    const unsigned* in_amax0 = nullptr;
    // A workgroup owns tiles_per_wg consecutive wave tiles and works through them in rounds of at most `active` tiles
    // (one pass over the weights per round).  The waves of a round run in lockstep (they share the LDS weight slabs);
    // waves without a tile only move weights and keep the barriers.
    const int t_begin = (int)blockIdx.x * a.tiles_per_wg, t_end = min(t_begin + a.tiles_per_wg, ntiles);

    int rnd = 0;
    auto stamp = [&](int i) {   // ABL & 16: where does a round's time go (s_memtime of lane 0 of every wave of workgroup 5)
        if ((ABL & 16) && blockIdx.x == 5 && lane == 0 && rnd < 2) a.trace[(wave * 2 + rnd) * 24 + i] = __builtin_amdgcn_s_memtime();
    };
    // chunk c of this thread in slab g of the weight stream (its slot in the LDS slab buffer is c * THREADS + tid)
    // (tz: an opaque zero, renewed per slab -- the tables are the same in every round and every slab, and with the slab
    // loop unrolled the compiler would otherwise read all of them up front and hold them in registers)
    // (addresses = a scalar base + a 32-bit byte offset: one add per chunk instead of a 64-bit multiply-add chain -- 54 chunks
    // per tile)
    // Round 5: the taps a launch skips are the ring's OLDEST rows, i.e. the first 9 - ntap weight taps (wfl_layer_launch checks
    // tap_w[t] == 9 - ntap + t), so logical k-step ks is packed k-step ks + KOFF and chunk f of slab g sits (SLAB g + KOFF) KCH1 + f
    // chunks into W1: a constant plus the thread index.  Through the kt_w table every chunk cost an LDS read, a signed division
    // by KCH1 and a remainder -- ~8 vector instructions x 40 chunks per tile in kernels that are bound by vector issue
    // (profiles/r05_wf_layer_isa_hist.txt).
    constexpr int KOFF = (9 - ntap) * S::KS_TAP;
    auto w_srcf = [&](int g, int f, int tz) -> const f16x8* {   // chunk f of slab g
        if (PK_WF_LINW && g < nslab)
        {
            const unsigned fl = (unsigned)(f + (LEAN ? 0 : tz));   // (tz: the opaque zero of the slab keeps the request where it is written)
            // chunk fl of the LDS slab = chunk fl % KCHL of k-step fl / KCHL (HIW: the k-step's hi block; else the whole k-step: fl itself)
            const unsigned src = S::HIW ? (fl / S::KCHL) * S::KCH1 + fl % S::KCHL : fl;
            return reinterpret_cast<const f16x8*>(reinterpret_cast<const char*>(w1) + ((unsigned)((SLAB * g + KOFF) * S::KCH1) + src) * 16u);
        }
        if (g < nslab)
            return reinterpret_cast<const f16x8*>(reinterpret_cast<const char*>(w1) +
                                                  (kt_w[SLAB * g + f / S::KCHL + tz] + (unsigned)((f % S::KCHL) * 16)));
        return reinterpret_cast<const f16x8*>(reinterpret_cast<const char*>(w2) +
                                              (unsigned)(((g - nslab) * (S::SLAB2 * S::KCH2) + min(f, S::SLAB2 * S::KCH2 - 1)) * 16));
    };
    auto w_src = [&](int g, int c, int tz) -> const f16x8* { return w_srcf(g, c * THREADS + tid + (LEAN ? tz : 0), tz); };
    // The prologue of a round brings the first weight slabs.  Round 4 (s_memtime trace of the 12-wave kernel,
    // profiles/r04_wf_layer_trace_12_waves.txt): with every thread bringing its share of slabs 0 and 1, the prologue barrier
    // passed at 12.2 k of a launch's 80 k cycles -- the first wave of a SIMD had its data at 5 k, the second at 8 k, the third at
    // 10.4 k (the whole chip starts at once: 264 KB per CU behind an idle memory system, served oldest wave first), and nobody
    // could start before the youngest wave's weight chunks had arrived.  Raising the priority of the weight requests
    // (tools/r04_wf_prio_call.sh) moved the barrier to 10.6 k: -1 % / 0 / +0.4 % -- the requests were not the problem, the
    // barrier's wait for the last arrival was.  Now the first NA waves (role A: the oldest wave of every SIMD) bring ALL of
    // slab 0 and store it before the barrier; the others (role B) bring slab 1 -- and slab 2 where they are twice as many --
    // pass the barrier without waiting for anything and store behind it (slab 1 is needed behind the barrier that ends
    // slab 0; buffer 2 is free): the old waves start their matrix work when THEIR data is there.  Every thread runs the
    // same CL loads into the same registers, the role only selects addresses (no control flow around the loads: with the
    // roles as branches the register allocator spilled the chunks behind their loads or merged the operand ring through
    // memory).  ABL & 64: the old prologue, for the A/B.
    constexpr int NA = W == 6 ? 2 : 4;
    constexpr int NCH = S::CPT1 * THREADS;       // chunks of a main slab
    constexpr int CL = NCH / (NA * 64);          // prologue chunks per thread (12)
    constexpr bool PRO2 = (W - NA) == 2 * NA;    // role B brings slabs 1 and 2 (half of CL each)
    static_assert(CL % 2 == 0 && CL * NA * 64 == NCH && (PRO2 ? CL / 2 : CL) * (W - NA) * 64 == NCH, "the prologue's split of the slabs over the waves");
    static_assert(nslab >= 3, "slabs 0 - 2 are conv slabs");
    constexpr bool NEWPRO = (ABL & 64) == 0 && CT == 2;   // (128 channels: the 12 chunks in flight cost 25 more spilled registers)
    const bool role_a = __builtin_amdgcn_readfirstlane(wave) < NA;
    // prologue chunk c of this thread: (slab, chunk in the slab)
    auto pro_g = [&](int c) { return role_a ? 0 : (PRO2 && c >= CL / 2 ? 2 : 1); };
    auto pro_f = [&](int c) {
        return role_a ? c * (NA * 64) + tid : ((PRO2 ? c % (CL / 2) : c) * ((W - NA) * 64) + (tid - NA * 64));
    };
    auto pro_src = [&](int c, int tz) -> const f16x8* {
        const unsigned f = (unsigned)(pro_f(c) + (W != 8 ? tz : 0));
        if (PK_WF_LINW)
            return reinterpret_cast<const f16x8*>(reinterpret_cast<const char*>(w1) + ((unsigned)((SLAB * pro_g(c) + KOFF) * S::KCH1) +
                                                                                        (S::HIW ? (f / S::KCHL) * S::KCH1 + f % S::KCHL : f)) * 16u);
        return reinterpret_cast<const f16x8*>(reinterpret_cast<const char*>(w1) +
                                              (kt_w[SLAB * pro_g(c) + (int)(f / S::KCHL) + tz] + (f % S::KCHL) * 16u));
    };
    f16x8 wreg[S::CPT1];   // one slab of weights on its way from global memory to LDS
    auto w_load = [&](int g, int tz) {
        if (g >= G) return;
#pragma unroll
        for (int c = 0; c < S::CPT1; ++c)
            if (c < (g < nslab ? S::CPT1 : S::CPT2)) wreg[c] = *w_src(g, c, tz);
    };
    // this thread's slot 0 of weight buffer `buf`.  128 channels: from an opaque scalar base -- with constant bases the compiler
    // keeps one address register per (buffer, chunk) beyond the 64 KB reach of a ds_write offset, sixteen of them, from the
    // kernel's top to its end
    // (the 64-channel kernels keep the indexed form they were tuned with: same addresses, another register allocation)
    constexpr bool WST = BIG;
    auto wst = [&](int buf) -> f16x8* {
        unsigned o = (unsigned)buf * SLAB_CH;
        asm volatile("" : "+s"(o));
        return &wbuf[0][0] + (o + (unsigned)tid);
    };
    auto w_store = [&](int g) {
        if (g >= G) return;
        if constexpr (WST) {
            f16x8* const dst = wst(g % 3);
#pragma unroll
            for (int c = 0; c < S::CPT1; ++c)
                if (c < (g < nslab ? S::CPT1 : S::CPT2)) dst[c * THREADS] = wreg[c];
        } else {
#pragma unroll
            for (int c = 0; c < S::CPT1; ++c)
                if (c < (g < nslab ? S::CPT1 : S::CPT2)) wbuf[g % 3][c * THREADS + tid] = wreg[c];
        }
    };

    // (MULTI: device memory, wave-uniform address -- scalar loads, here only; one layer: the kernel argument segment)
    const WflLayer& L = MULTI ? a.layers[li] : a.l0;
    if (MULTI && tid == 0) {
        cold.out = L.out;
        cold.out_amax = L.out_amax;
        cold.has_out = L.out != nullptr;
        cold.first = L.first;
        cold.k1 = L.w.k1;
        cold.k2res = L.w.k2res;
        const bool st = li + 1 == a.nl && a.step_z != nullptr;
        cold.step_z = st ? a.step_z : nullptr;
        cold.step_x = a.step_x;
        cold.step_w_in = a.step_w_in;
        cold.step_b_in = a.step_b_in;
        cold.step_h0 = a.step_h0;
        cold.step_h0_amax = a.step_h0_amax;
        cold.step_b_logs = a.step_b_logs;
        cold.step_b_b = a.step_b_b;
    }
    for (int i = tid; i < 3 * C; i += THREADS) lb[i] = i < C ? L.w.b2r[i] : L.w.wso[i - C];
    // the fused step's input_proj weights: through LDS with the other tables.  (Round 5: read from memory where they are used --
    // 2 x 32 scalar-indexed loads inside `lane_ok ? ... : 0` -- they compiled to 32 predicated blocks, each a load pair and an
    // s_waitcnt vmcnt(0): 32 memory round trips in a row at the end of every row's last layer.)
    if constexpr (CT == 2) {
        if (a.step_z != nullptr && a.step_h0 != nullptr && (!MULTI || li + 1 == a.nl))
            for (int i = tid; i < 2 * C; i += THREADS) lb[3 * C + i] = i < C ? a.step_w_in[i] : a.step_b_in[i - C];
    }
    if (tid < nks) {
        const int ks = tid;
        kt_w[tid] = (unsigned)(ks < nks_conv ? a.tap_w[ks / S::KS_TAP] * S::KS_TAP + ks % S::KS_TAP : 9 * S::KS_TAP + (ks - nks_conv)) *
                    (unsigned)(S::KCH1 * 16);
    }
    if (tid < ntap) {
        tp_off[tid] = ((long)a.tap_slot[tid] * a.slot_stride) * 4;
        tp_am[tid] = (long)a.tap_slot[tid] * a.amax_stride;
        tp_shift[tid] = a.tap_col[tid] * L.dil;
        tp_blk[tid] = S::BLK_BYTES;
    } else if (tid == ntap) {
        tp_off[tid] = reinterpret_cast<const char*>(a.cond) - reinterpret_cast<const char*>(L.in0);
        tp_am[tid] = a.cond_amax - L.in_amax0;
        tp_shift[tid] = 0;
        tp_blk[tid] = BLK_M_BYTES;
    }
    w1 = reinterpret_cast<const f16x8*>(L.w.w1);
    w2 = reinterpret_cast<const f16x8*>(L.w.w2);
    in0b = reinterpret_cast<const char*>(L.in0);
    in_amax0 = L.in_amax0;
    __syncthreads();   // tables visible
    for (int base = t_begin; base < t_end;) {   // uniform over the workgroup
        const int nact = min(a.active, t_end - base);
        const int wt = base + (LEAN ? __builtin_amdgcn_readfirstlane(wave) : wave);   // (LEAN: the tile index as a scalar -- what derives from it
                                                                                       // alone, e.g. the block index of out_amax, needs no vector register)
        const bool tile_ok = wave < nact;
        base += nact;
        const int p0 = tile_ok ? wt * WAVE_T : 0;
        const int p = p0 + j;
        const int p_utt = a.pos_utt[p];   // (compared in the epilogue: a comparison here would wait for the load before anything else is requested)
        stamp(0);
        // the bias reads below are the same in every round: without this the compiler keeps them in registers across the
        // round loop
        int lz = 0;
        asm volatile("" : "+s"(lz));
        const float* lbr = lb + lz;

        // Working and idle waves run separate copies of the whole round (same number of barriers): the working copy
        // has no branch inside, and nothing of it is live in the idle copy.
        if (tile_ok) {
            // ---- common scale of this wave tile: the largest block maximum among the blocks its taps read.  Only the loads
            // here: the reduction (which waits for them) comes after the prologue has requested the weights and the operands
            // -- one memory round trip for the whole prologue instead of three in a row (s_memtime trace, round 3: 14 k of a
            // round's 77 k cycles were the prologue)
            // (one unconditional load per lane -- lanes beyond the 2 ntap + 1 sources repeat lane 0's: loads inside divergent
            // branches are waited for where the branches join)
            unsigned m_raw;
            {
                const int lane_r = lane + (LEAN ? lz : 0);
                const int li = lane_r <= 2 * ntap ? lane_r : 0, t = li >> 1;   // t = ntap: the condition block (shift 0)
                const int blk = (p0 + tp_shift[t + lz] + 31 * (li & 1)) >> 5;   // (the LDS tables, not the kernel arguments: those
                m_raw = (in_amax0 + tp_am[t + lz])[blk];                          //  indexed per lane would be loads from memory)
            }

            // ---- B operand of k-step ks: the lane's 8 channels (octet 2 kq + hh) of position p + shift, hi and lo vectors;
            // with the first k-step of a tap also the maximum of the block that position lies in (for the rescale to the
            // tile's common scale; a tap's k-steps share it: at most 4 taps are in flight)
            f16x8 rhi[RING], rlo[RING];
            unsigned ram[4];
            auto tap_of = [&](int ks) { return ks < nks_conv ? ks / S::KS_TAP : ntap; };
            // the source of the k-steps being requested: its base as a scalar pair, this lane's byte offset in it (a tap's
            // k-steps differ by a constant: one address add per k-step instead of a 64-bit multiply-add chain)
            const char* cur_base = in0b;
            unsigned cur_off = 0;
            auto load_b = [&](int ksu, int tz) {   // ksu may run past the end: the last k-step again
                const int ks = ksu < nks ? ksu : nks - 1, slot = ksu % RING;
                const int tap = tap_of(ks), kq = ks - (tap < ntap ? tap * S::KS_TAP : nks_conv);
                if (ksu < nks && kq == 0) {
                    // q can be negative (a tap of the first tiles reaches into the buffer's leading margin): the offset is
                    // taken from 8 blocks before the source's position 0, so that it is a non-negative 32-bit number
                    const int q = p + tp_shift[tap + tz];
                    const int blkb = PK_WF_LINW ? (tap < ntap ? S::BLK_BYTES : BLK_M_BYTES) : tp_blk[tap + tz];   // (a constant per source kind: a shift, not a 64-bit multiply)
                    cur_off = (unsigned)(((q >> 5) + 8) * blkb + (q & 31) * 32 + hh * 1024);
                    const long bo = tp_off[tap + tz] - 8L * blkb;
                    const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)bo), bhi = __builtin_amdgcn_readfirstlane((unsigned)(bo >> 32));
                    cur_base = in0b + (long)(((unsigned long)bhi << 32) | blo);
                    ram[tap % 4] = (in_amax0 + tp_am[tap + tz])[q >> 5];
                }
                const char* src = cur_base + (cur_off + (unsigned)(kq * 2048));
                rhi[slot] = ld_h8(src);
                if (!F16) rlo[slot] = ld_h8(src + 16);   // fp16-operand mode: the hi plane is the rounded value
            };
            const long pblk = (long)(p >> 5);
            const int pin = p & 31;

            // the first weight slabs and the first RING k-steps of the operands in ONE round trip
            f16x8 wpro[NEWPRO ? CL : 1];
            if constexpr (!NEWPRO) {
                f16x8 wreg1[S::CPT1];
                w_load(0, lz);
#pragma unroll
                for (int c = 0; c < S::CPT1; ++c) wreg1[c] = *w_src(1, c, lz);
#pragma unroll
                for (int kk = 0; kk < RING; ++kk) load_b(kk, lz);   // nks >= 18 > RING
                __builtin_amdgcn_sched_barrier(0);   // everything above is requested before anything below waits
                w_store(0);
                if constexpr (WST) {
                    f16x8* const dst1 = wst(1);
#pragma unroll
                    for (int c = 0; c < S::CPT1; ++c) dst1[c * THREADS] = wreg1[c];
                } else {
#pragma unroll
                    for (int c = 0; c < S::CPT1; ++c) wbuf[1][c * THREADS + tid] = wreg1[c];
                }
            } else {
                // (half of the chunks, the operands, the other half: role B's second half is slab 2, which can wait; role A
                // needs its operands before it can start anyway)
#pragma unroll
                for (int c = 0; c < CL / 2; ++c) wpro[c] = *pro_src(c, lz);
#pragma unroll
                for (int kk = 0; kk < RING; ++kk) load_b(kk, lz);   // nks >= 18 > RING
#pragma unroll
                for (int c = CL / 2; c < CL; ++c) wpro[c] = *pro_src(c, lz);
                __builtin_amdgcn_sched_barrier(0);   // everything above is requested before anything below waits
                if (role_a) {
#pragma unroll
                    for (int c = 0; c < CL; ++c) wbuf[0][c * (NA * 64) + tid] = wpro[c];
                }
            }
            // its (clamped) biased exponent; the tile's operands are scaled by 2^kx, kx = 13 + 127 - ex.  (New prologue: behind the
            // barrier -- it waits for the block maxima, and a role-B wave must reach the barrier without waiting for any load:
            // its first answer arrives when the older waves' requests have been served, 10 k cycles into the launch.)
            auto tile_exp = [&]() { return __builtin_amdgcn_readfirstlane(amax_exp(__float_as_uint(wave_max64(__uint_as_float(m_raw))))); };   // (all sources are maxima: the repeats change nothing)
            int ex_ = 0;
            if constexpr (!NEWPRO) ex_ = tile_exp();
            stamp(1);
            __syncthreads();
            stamp(2);
            if constexpr (NEWPRO) {
                if (!role_a) {
#pragma unroll
                    for (int c = 0; c < CL; ++c) wbuf[PRO2 && c >= CL / 2 ? 2 : 1][(PRO2 ? c % (CL / 2) : c) * ((W - NA) * 64) + (tid - NA * 64)] = wpro[c];
                }
                ex_ = tile_exp();
            }
            const int ex = ex_;
            const int kx = PK_BLK_TOP + 127 - ex;
            const int ks1 = kx + (MULTI ? (&cold)[lz].k1 : a.l0.w.k1);
            const int cbb = __builtin_amdgcn_readfirstlane(__float_as_int(-1.4426950408889634f * pow2f(-ks1)));
            const float gcb = __int_as_float(cbb), gca = __int_as_float(cbb + (1 << 23));
            f16x8 f;          // rescale factor of the tap being consumed
            // the accumulators start at zero: the bias of the first contraction is the weight column of condition channel
            // n_mels, which k_wf_cond_planes sets to 1 (the 80 mel channels are padded to 96 anyway) -- 2C LDS reads and
            // multiplies per tile less
            f32x16 acc[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
#pragma unroll
            for (int g = 0; g < nslab; ++g) {
                int tz = 0;
                asm volatile("" : "+s"(tz));
                // 128 channels (128 accumulator registers): the slab's weights travel in two halves of 12 registers, the
                // second requested when the first has gone to LDS after the first k-step, and a ring slot is rescaled in
                // place and refilled after its k-step's MFMAs.  (A k-step is 24 MFMAs there: the loads a weight wait drains
                // early are still three k-steps = 2 300 matrix cycles old.)
                constexpr bool TIGHT = CT == 4 || W != 8;   // (12 waves: 168 registers)
                // PERK (128 channels, round 5): the slab's weights travel in SLAB parts, one per k-step -- part kk + 1 is requested
                // when part kk has gone to LDS after k-step kk's MFMAs: two chunks (8 registers) on the way instead of three (12),
                // every part one k-step (24 MFMAs per wave) old when it is stored, as the halves were
                constexpr bool PERK = BIG;
                const int NW = (g + 2 >= G || (NEWPRO && PRO2 && g == 0)) ? 0 : (g + 2 < nslab ? S::CPT1 : S::CPT2), HW = PERK ? NW / SLAB : (TIGHT ? NW / 2 : NW);   // constants once unrolled (slab 2: with the prologue where role B brings it)
                auto part_lo = [&](int i) { return i * NW / SLAB; };   // PERK: chunks [part_lo(i), part_lo(i + 1)) are part i
#pragma unroll
                for (int c = 0; c < S::CPT1; ++c)
                    if (c < HW && !(ABL & 8)) wreg[c] = *w_src(g + 2, c, tz);   // first: every load below is younger
                __builtin_amdgcn_sched_barrier(0);   // keep the weight loads up here: hipcc would sink them to their stores
                // (the slab's LDS base as ONE opaque register: from a constant base the third buffer's fragments lie beyond the
                // 64 KB reach of a ds_read offset, and the compiler keeps a separate address register for each of them)
                // (12 waves: the opaque part is the scalar base -- "constant | lane" would be hoisted out of the round loop, spilled,
                // and its reload in the middle of the slab loop waits for the weight loads just requested)
                unsigned wo = (g % 3) * SLAB_CH;
                if constexpr (LEAN) {
                    asm volatile("" : "+s"(wo));
                    wo += lane;
                } else {
                    wo += lane;
                    asm volatile("" : "+v"(wo));
                }
                const f16x8* wl = &wbuf[0][0] + wo;
#pragma unroll
                for (int kk = 0; kk < SLAB; ++kk) {
                    const int ks = SLAB * g + kk, slot = ks % RING;
                    if (ks == nks_conv || (ks < nks_conv && ks % S::KS_TAP == 0)) f = pow2_neg_h8(ex - amax_exp(ram[tap_of(ks) % 4]));
                    f16x8 bh, bl;
                    if (TIGHT) {
                        if (F16) rhi[slot] *= f;
                        else {
                            rhi[slot] *= f;
                            rlo[slot] *= f;
                        }
                    } else {
                        bh = rhi[slot] * f;
                        if (!F16) bl = rlo[slot] * f;
                        __builtin_amdgcn_sched_barrier(0);   // the slot's old value is dead before its refill is requested
                        if (!(ABL & 1) && ks + RING < nks) load_b(ks + RING, tz);   // (a compile-time condition once unrolled)
                    }
                    __builtin_amdgcn_sched_barrier(0);   // ... and the loads ahead of the k-step's MFMAs
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const f16x8 ah = wl[kk * S::KCHL + (0 * NQ + q) * 64];
                        acc[q] = mfma16(ah, TIGHT ? rhi[slot] : bh, acc[q]);
                        if (!F16) {
                            const f16x8 al = wl[kk * S::KCHL + (1 * NQ + q) * 64];
                            acc[q] = mfma16(al, TIGHT ? rhi[slot] : bh, acc[q]);
                            acc[q] = mfma16(ah, TIGHT ? rlo[slot] : bl, acc[q]);
                        }
                    }
                    // A fragments AHEAD co-tiles ahead of their MFMAs (not all NQ of them: registers).  Two at 64 channels:
                    // with one, every co-tile's three MFMAs (96 cycles) had to cover a whole LDS read latency, and the trace
                    // showed about 200 cycles per k-step and wave that nothing covered
                    constexpr int AHEAD = ((CT == 2 || PK_WF_AHEAD128 == 2) && !(ABL & 32) && !(W != 8 && !F16)) ? 2 : 1, PER = F16 ? 1 : 2, MM = F16 ? 1 : 3;   // (12 waves, split
                    // math: one -- registers; the other two waves of the SIMD cover the LDS latency)
                    __builtin_amdgcn_sched_group_barrier(0x100, PER * (AHEAD + 0), 0);
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        if (q + AHEAD < NQ) __builtin_amdgcn_sched_group_barrier(0x100, PER, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, MM, 0);
                    }
                    if (TIGHT) {
                        __builtin_amdgcn_sched_barrier(0);
                        // (round 5, HISTORY 9.9: the slot consumed ONE k-step ago -- PK_WF_LATE_REFILL=1 -- not the one this k-step's
                        // matrix instructions may still be reading when the refill's data arrives)
                        if (ks - PK_WF_LATE_REFILL >= 0 && ks - PK_WF_LATE_REFILL + RING < nks) load_b(ks - PK_WF_LATE_REFILL + RING, tz);
                        if (PERK && NW > 0) {
                            const int lo = part_lo(kk), n = part_lo(kk + 1) - lo, n1 = kk + 1 < SLAB ? part_lo(kk + 2) - part_lo(kk + 1) : 0;
                            f16x8* const dst = wst((g + 2) % 3);
#pragma unroll
                            for (int c = 0; c < S::CPT1; ++c)
                                if (c < n) dst[(lo + c) * THREADS] = wreg[c];
#pragma unroll
                            for (int c = 0; c < S::CPT1; ++c)
                                if (c < n1) wreg[c] = *w_src(g + 2, lo + n + c, tz);
                        } else if (kk == 0 && NW > 0) {
#pragma unroll
                            for (int c = 0; c < S::CPT1; ++c)
                                if (c < HW) wbuf[(g + 2) % 3][c * THREADS + tid] = wreg[c];
#pragma unroll
                            for (int c = 0; c < S::CPT1; ++c)
                                if (c < NW - HW) wreg[c] = *w_src(g + 2, HW + c, tz);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
                for (int c = 0; c < S::CPT1; ++c)
                    if (!PERK && c < (TIGHT ? NW - HW : NW) && !(ABL & 8)) wbuf[(g + 2) % 3][((TIGHT ? HW : 0) + c) * THREADS + tid] = wreg[c];
                if (g < 8) stamp(3 + 2 * g);
                __syncthreads();   // everyone is done reading this slab's buffer and sees the next two
                if (g < 8) stamp(4 + 2 * g);
            }
            // the ring is dead: request the epilogue's old values now, a whole gate ahead of their use
            f16x8 xin_hi[S::KS2], xin_lo[S::KS2];   // residual input: this lane's centre-tap vectors of the current row
            unsigned cur_am;
            float2 prm_old = {0.f, 0.f};
            // (12 waves: the epilogue's lane coordinates are derived HERE, from an opaque zero -- not hoisted, not spilled)
            const int ez = opaque_zero<LEAN>();
            // (LEAN: the lane index itself recomputed -- v_mbcnt of an all-ones mask on top of the opaque zero -- so that not even
            // it has to live, or be spilled, through the slab loop)
            const int lane_e = LEAN ? (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)ez)) : lane + ez;
            const int hh_e = LEAN ? lane_e >> 5 : hh;
            const int p_e = LEAN ? p0 + (lane_e & 31) : p;
            const long pblk_e = LEAN ? (long)(p_e >> 5) : pblk;
            const int pin_e = LEAN ? p_e & 31 : pin;
            // MULTI: the parked copies (read here, with the other old values: wave-uniform, made scalar where they branch);
            // one layer: the kernel arguments themselves
            Cold cd;
            if constexpr (MULTI) {
                cd = (&cold)[ez];
                cd.has_out = __builtin_amdgcn_readfirstlane(cd.has_out);
                cd.first = __builtin_amdgcn_readfirstlane(cd.first);
            } else {
                cd.out = a.l0.out;
                cd.out_amax = a.l0.out_amax;
                cd.has_out = a.l0.out != nullptr;
                cd.first = a.l0.first;
                cd.k2res = a.l0.w.k2res;
                cd.step_z = a.step_z;
                cd.step_x = a.step_x;
                cd.step_w_in = a.step_w_in;
                cd.step_b_in = a.step_b_in;
                cd.step_h0 = a.step_h0;
                cd.step_h0_amax = a.step_h0_amax;
                cd.step_b_logs = a.step_b_logs;
                cd.step_b_b = a.step_b_b;
            }
            const bool has_out = cd.has_out != 0;
            const bool l_first = cd.first != 0;
            const float i_res = pow2f(-(PK_UNIT_EXP + cd.k2res));
            {
                const char* cur = in0b + ((long)a.cur_slot * a.slot_stride) * 4 + pblk_e * S::BLK_BYTES + pin_e * 32 + hh_e * 1024;
#pragma unroll
                for (int kq = 0; kq < S::KS2; ++kq) {
                    xin_hi[kq] = (ABL & 4) ? rhi[kq] : ld_h8(cur + kq * 2048);
                    xin_lo[kq] = (ABL & 4) ? rlo[kq] : ld_h8(cur + kq * 2048 + 16);
                }
                cur_am = in_amax0[(long)a.cur_slot * a.amax_stride + (p0 >> 5)];
                if (!l_first && !(ABL & 4)) prm_old = reinterpret_cast<const float2*>(a.prm)[p_e];
            }
            const int p_utt_e = LEAN ? a.pos_utt[p_e] : p_utt;   // (12 waves: requested with the old values, not held since the round's start)
            // ---- gate: z * 2^14 in the accumulator registers -> split B operands of the out projection; on the way this
            // lane's part of the folded skip path: (logs, b) += sum over its C/2 channels of wso[.][channel] * z
            __builtin_amdgcn_sched_barrier(0);   // the old-value loads stay ahead of the gate
            stamp(19);
            f16x8 zh[S::KS2], zl[S::KS2];
            float pl = 0.f, pb = 0.f;
            const f32x4* wso = reinterpret_cast<const f32x4*>(lbr + C) + hh_e * (S::KS2 * 4);
#pragma unroll
            for (int k2 = 0; k2 < S::KS2; ++k2) {
                const int zq = k2 >> 1, r0 = 8 * (k2 & 1);
                float zv[8];
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) {
                    const f32x2 av = {acc[zq][r0 + 2 * e2], acc[zq][r0 + 2 * e2 + 1]};
                    const f32x2 bv = {acc[zq + CT][r0 + 2 * e2], acc[zq + CT][r0 + 2 * e2 + 1]};
                    const f32x2 z2 = gated_s2(av, bv, gca, gcb);
                    zv[2 * e2] = z2[0];
                    zv[2 * e2 + 1] = z2[1];
                }
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) {   // [k2][e][logs | b]: two channels per 16-byte read
                    const f32x4 w = wso[k2 * 4 + e2];
                    pl = fmaf(w[0], zv[2 * e2], pl);
                    pb = fmaf(w[1], zv[2 * e2], pb);
                    pl = fmaf(w[2], zv[2 * e2 + 1], pl);
                    pb = fmaf(w[3], zv[2 * e2 + 1], pb);
                }
                if (F16) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) zh[k2][e] = (_Float16)zv[e];
                } else {
                    split8(zv, zh[k2], zl[k2]);
                }
            }
            const bool lane_ok = p_utt_e >= 0;
            // the other half wave holds the other C/2 channels of the same position.  (LEAN: the exchange addressed from the
            // recomputed lane index -- __shfl_xor derives its own from v_mbcnt, which the compiler merges with the kernel's first
            // and keeps, or spills, through the slab loop)
#if PK_WF_SCALAR_SUMS
            asm volatile("" : "+v"(pl), "+v"(pb));   // THE OP_SEL RULE (top of the file): pl and pb as two scalars -- no packed FMA with op_sel in the sums above
#endif
            pl += xor32(pl, lane_e, LEAN);
            pb += xor32(pb, lane_e, LEAN);
            float2 prm_new = {prm_old.x + pl, prm_old.y + pb};   // skips summed (:390), then output_proj (:499-500)
            if (!lane_ok) prm_new = float2{0.f, 0.f};
            if (hh_e == 0 && !(ABL & 4)) reinterpret_cast<float2*>(a.prm)[p_e] = prm_new;
            stamp(20);
            // ---- out projection, res half (-> next layer's input planes; the last layer has none): its weights follow the
            // conv weights through the slab buffers (NS2 slabs of SLAB2 k-steps)
            f32x16 acc2[CT];
#pragma unroll
            for (int t = 0; t < CT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[t][r] = lbr[32 * t + mfma_row(r, hh_e)];
#pragma unroll
            for (int h2 = 0; h2 < S::NS2; ++h2) {
                const int g = nslab + h2;   // slab of the weight stream
                unsigned wo = (g % 3) * SLAB_CH + lane_e;
                asm volatile("" : "+v"(wo));
                const f16x8* buf = &wbuf[0][0] + wo;
                if (S::NS2 >= 2) w_load(g + 2, 0);
                if (has_out) {
#pragma unroll
                    for (int kk = 0; kk < S::SLAB2; ++kk) {
                        const int k2 = h2 * S::SLAB2 + kk;
#pragma unroll
                        for (int t = 0; t < CT; ++t) {
                            const f16x8 ah = buf[kk * S::KCH2 + (0 * CT + t) * 64];
                            acc2[t] = mfma16(ah, zh[k2], acc2[t]);
                            if (!F16) {
                                const f16x8 al = buf[kk * S::KCH2 + (1 * CT + t) * 64];
                                acc2[t] = mfma16(al, zh[k2], acc2[t]);
                                acc2[t] = mfma16(ah, zl[k2], acc2[t]);
                            }
                        }
                    }
                }
                if (S::NS2 >= 2) {
                    w_store(g + 2);
                    __syncthreads();   // slab g consumed by every wave, the next visible
                }
            }
            stamp(21);
            // res = x_in + res (:281) -> the next layer's input of this row, as planes with this block's scale
            if (has_out) {
                const float xs = pow2f(-(PK_BLK_TOP + 127 - amax_exp(cur_am)));   // the stored input is x * 2^k
                float v[CT][16];
                float am = 0.f;
                // x_in first, all of it, and only then the accumulators (round 5 put this order and the s_nop below in against what it took for
                // a short MFMA -> VALU distance; round 6 showed that distance to be safe down to 8 slots and the wrong tiles to be the op_sel
                // defect of the sums above -- HISTORY 10.3; the order costs nothing and stays).
                // Gap positions are zero in the planes: their scale factors are zero (bit-identical to a select per element, 64 of them).
                const float xs_l = lane_ok ? xs : 0.f, i_res_l = lane_ok ? i_res : 0.f;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < CT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int kq = 2 * t + (r >> 3), e = r & 7;
                        // x_in = (hi + lo) * 2^-k: one multiply and one v_fma_mix (the halves are sources)
                        v[t][r] = fmaf((float)xin_hi[kq][e], xs_l, (float)xin_lo[kq][e] * xs_l);
                        asm volatile("" : "+v"(v[t][r]));   // (computed HERE: pure arithmetic would sink to its use behind the barrier)
                    }
                __builtin_amdgcn_sched_barrier(0);
#ifndef PK_HIPEMU
                asm volatile("s_nop 15");   // (belt and braces: sixteen more slots, 64 cycles of a 25 000-cycle tile)
#endif
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < CT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float o = fmaf(acc2[t][r], i_res_l, v[t][r]);   // res + x_in (gap positions: 0 * finite + 0)
                        am = fmaxf(am, fabsf(o));
                        v[t][r] = o;
                    }
                am = wave_max64(am);
                const float so = pow2f(blk_scale_exp(__float_as_uint(am)));
                char* dst = reinterpret_cast<char*>(cd.out) + pblk_e * S::BLK_BYTES + pin_e * 32 + hh_e * 1024;
#pragma unroll
                for (int kq = 0; kq < S::KS2; ++kq) {
                    float t8[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) t8[e] = v[kq >> 1][8 * (kq & 1) + e];
                    f16x8 oh, ol;
                    store_pair8(t8, so, oh, ol);
                    if ((ABL & 4) && oh[0] != (_Float16)12345.f) continue;   // (never equal: keeps the arithmetic)
                    st_h8_out<CT == 2>(dst + kq * 2048, oh);
                    st_h8_out<CT == 2>(dst + kq * 2048 + 16, ol);
                }
                if (lane_e == 0) cd.out_amax[p0 >> 5] = __float_as_uint(am);
            }
            stamp(22);
            // ---- the row's last layer done (and the launch asked for it): finish the row here instead of in a kernel of its own
            // (k_wf_step_p below, operation for operation): (logs, b) = prm + the folded biases, x = (z' - b) exp(-logs)
            // (Flow._predict_row_parameters :496-501, _inverse_transform_row :503-505), h0 = input_proj(x) -> the next row's
            // layer-0 input as planes.  Its ring slot is the one layer 0 of THIS row read as its oldest row: every workgroup
            // is past that layer (grid barriers / earlier launches).
            // (64 channels only: at 128 the kernel has no registers for it -- 10 more spilled --, and the step is 1 % of the batch)
            const float* const step_z = CT == 2 ? cd.step_z : nullptr;
            if (CT == 2 && (MULTI ? __builtin_amdgcn_readfirstlane(step_z != nullptr) : step_z != nullptr)) {
                const float xn = lane_ok ? (step_z[p_e] - (prm_new.y + cd.step_b_b)) * expf(-(prm_new.x + cd.step_b_logs)) : 0.f;
                if (hh_e == 0) cd.step_x[p_e] = xn;
                float* const step_h0 = cd.step_h0;
                if (MULTI ? __builtin_amdgcn_readfirstlane(step_h0 != nullptr) : step_h0 != nullptr) {
                    float v[S::KS2][8];
                    float am = 0.f;
                    // wfl_chan(kq, hh, e) = 16 kq + 8 (e >> 2) + 4 hh + (e & 3): two aligned quads per k-step
                    const float* const lw = lbr + 3 * C + 4 * hh_e;
#pragma unroll
                    for (int kq = 0; kq < S::KS2; ++kq)
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2) {
                            const f32x4 w4 = *reinterpret_cast<const f32x4*>(lw + 16 * kq + 8 * h2);
                            const f32x4 b4 = *reinterpret_cast<const f32x4*>(lw + C + 16 * kq + 8 * h2);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float t = fmaf(w4[e], xn, b4[e]);
                                v[kq][4 * h2 + e] = lane_ok ? t : 0.f;
                                am = fmaxf(am, fabsf(v[kq][4 * h2 + e]));
                            }
                        }
                    am = wave_max64(am);
                    const float so = pow2f(blk_scale_exp(__float_as_uint(am)));
                    char* dst = reinterpret_cast<char*>(step_h0) + pblk_e * S::BLK_BYTES + pin_e * 32 + hh_e * 1024;
#pragma unroll
                    for (int kq = 0; kq < S::KS2; ++kq) {
                        f16x8 oh, ol;
                        store_pair8(v[kq], so, oh, ol);
                        st_h8(dst + kq * 2048, oh);
                        st_h8(dst + kq * 2048 + 16, ol);
                    }
                    if (lane_e == 0) cd.step_h0_amax[p0 >> 5] = __float_as_uint(am);
                }
            }
        } else {
            if constexpr (!NEWPRO) {
                f16x8 wreg1[S::CPT1];
                w_load(0, lz);
#pragma unroll
                for (int c = 0; c < S::CPT1; ++c) wreg1[c] = *w_src(1, c, lz);
                w_store(0);
                if constexpr (WST) {
                    f16x8* const dst1 = wst(1);
#pragma unroll
                    for (int c = 0; c < S::CPT1; ++c) dst1[c * THREADS] = wreg1[c];
                } else {
#pragma unroll
                    for (int c = 0; c < S::CPT1; ++c) wbuf[1][c * THREADS + tid] = wreg1[c];
                }
                __syncthreads();
            } else {
                f16x8 wpro[CL];
#pragma unroll
                for (int c = 0; c < CL; ++c) wpro[c] = *pro_src(c, lz);
                if (role_a) {
#pragma unroll
                    for (int c = 0; c < CL; ++c) wbuf[0][c * (NA * 64) + tid] = wpro[c];
                }
                __syncthreads();
                if (!role_a) {
#pragma unroll
                    for (int c = 0; c < CL; ++c) wbuf[PRO2 && c >= CL / 2 ? 2 : 1][(PRO2 ? c % (CL / 2) : c) * ((W - NA) * 64) + (tid - NA * 64)] = wpro[c];
                }
            }
            // ABL & 128 (round 6, results stay right): an IDLE wave checks what the working waves are reading.  During slab g every
            // wave of the workgroup is between the barriers that end slab g - 1 and slab g: buffer g % 3 holds slab g, complete and
            // stable (stored two slabs ago, or by the prologue) -- the idle wave compares all of it with the weights in memory and
            // records every 16-byte chunk that differs: (launch, workgroup, slab, chunk) -> which thread stored it, in which half, and
            // what was there instead.  The working waves' instruction stream is untouched (this is the idle copy of the round).
            auto verify_slab = [&](int g) {
                if constexpr ((ABL & 128) != 0) {
                    if (a.trace == nullptr) return;
                    const int n = g < nslab ? S::CPT1 * THREADS : S::SLAB2 * S::KCH2;
                    for (int i = lane; i < n; i += 64) {
                        const u32x4 got = *reinterpret_cast<const u32x4*>(&wbuf[g % 3][i]);
                        const u32x4 want = *reinterpret_cast<const u32x4*>(w_srcf(g, i, 0));
                        if (got[0] != want[0] || got[1] != want[1] || got[2] != want[2] || got[3] != want[3]) {
                            const unsigned long long k = atomicAdd(a.trace, 1ull);
                            if (k < 200) {
                                unsigned long long* r = a.trace + 1 + 6 * k;
                                r[0] = ((unsigned long long)(unsigned)a.seq << 32) | ((unsigned long long)blockIdx.x << 16) | ((unsigned long long)g << 12) | (unsigned long long)i;
                                r[1] = ((unsigned long long)wave << 32) | (unsigned)nact;
                                r[2] = ((unsigned long long)got[1] << 32) | got[0];
                                r[3] = ((unsigned long long)got[3] << 32) | got[2];
                                r[4] = ((unsigned long long)want[1] << 32) | want[0];
                                r[5] = ((unsigned long long)want[3] << 32) | want[2];
                            }
                        }
                    }
                }
            };
#pragma unroll
            for (int g = 0; g < (S::NS2 >= 2 ? G : nslab); ++g) {
                int tz = 0;
                asm volatile("" : "+s"(tz));
                if (!(NEWPRO && PRO2 && g == 0)) {
                    w_load(g + 2, tz);
                    w_store(g + 2);
                }
                verify_slab(g);
                __syncthreads();
            }
            if (S::NS2 < 2) verify_slab(nslab);   // the out-projection slab: read by the working waves until the round's last barrier
        }
        stamp(23);
        if (S::NS2 < 2) __syncthreads();   // one slab for both passes: consumed before the next round overwrites its buffer
        ++rnd;
    }
    }
}

// The folded condition rows as planes, in place: one wave per (row, block) of 96 x 32 floats stored [channel][32]
__global__ __launch_bounds__(64) void k_wf_cond_planes(float* __restrict__ cond, long row_stride, long amax_row_stride,
                                                      unsigned* __restrict__ amax, int one_ch) {
    float* blk = cond + (long)blockIdx.y * row_stride + (long)blockIdx.x * (WFL_MP * WFL_BLK);
    const int lane = threadIdx.x, j = lane & 31, hh = lane >> 5;
    float v[WFL_KS_COND][8];
    float m = 0.f;
#pragma unroll
    for (int kq = 0; kq < WFL_KS_COND; ++kq)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ch = wfl_chan(kq, hh, e);
            v[kq][e] = ch == one_ch ? 1.f : blk[ch * WFL_BLK + j];   // the constant channel that carries the layers' biases
            m = fmaxf(m, fabsf(v[kq][e]));
        }
    m = wave_max64(m);   // every lane has read its values: the block can be overwritten
    const float s = pow2f(blk_scale_exp(__float_as_uint(m)));
    char* dst = reinterpret_cast<char*>(blk) + j * 32 + hh * 1024;
#pragma unroll
    for (int kq = 0; kq < WFL_KS_COND; ++kq) {
        f16x8 oh, ol;
        store_pair8(v[kq], s, oh, ol);
        st_h8(dst + kq * 2048, oh);
        st_h8(dst + kq * 2048 + 16, ol);
    }
    if (lane == 0) amax[(long)blockIdx.y * amax_row_stride + blockIdx.x] = __float_as_uint(m);
}

// Flow._predict_row_parameters :496-501 + _inverse_transform_row :503-505 + input_proj of the new row (:497): one wave
// per 32 positions, lane (j, hh) = position j and the channels of the octets 2 kq + hh (the ones it stores).  (logs, b) of a
// position = what the layer kernels accumulated in prm + the folded biases.
template <int CT>
__global__ __launch_bounds__(256) void k_wf_step_p(const float* __restrict__ prm, float b_logs, float b_b,
                                                   const float* __restrict__ z_row, float* __restrict__ x_row,
                                                   const float* __restrict__ w_in, const float* __restrict__ b_in,
                                                   float* __restrict__ h0_next, unsigned* __restrict__ h0_amax,
                                                   const int* __restrict__ pos_utt, int npos_alloc, int first) {
    constexpr int C = 32 * CT, KS = C / 16;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile * WAVE_T >= npos_alloc) return;
    const int lane = threadIdx.x & 63, j = lane & 31, hh = lane >> 5;
    const int p = tile * WAVE_T + j;
    const bool valid = pos_utt[p] >= 0;
    float xn = 0.f;
    if (first) {
        xn = valid ? z_row[p] : 0.f;
    } else {
        const float2 lb2 = reinterpret_cast<const float2*>(prm)[p];
        xn = valid ? (z_row[p] - (lb2.y + b_b)) * expf(-(lb2.x + b_logs)) : 0.f;
    }
    if (hh == 0) x_row[p] = xn;
    if (h0_next) {
        float v[KS][8];
        float am = 0.f;
#pragma unroll
        for (int kq = 0; kq < KS; ++kq)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = wfl_chan(kq, hh, e);
                v[kq][e] = valid ? fmaf(w_in[c], xn, b_in[c]) : 0.f;
                am = fmaxf(am, fabsf(v[kq][e]));
            }
        am = wave_max64(am);
        const float s = pow2f(blk_scale_exp(__float_as_uint(am)));
        char* dst = reinterpret_cast<char*>(h0_next) + (long)tile * (C * 128) + j * 32 + hh * 1024;
#pragma unroll
        for (int kq = 0; kq < KS; ++kq) {
            f16x8 oh, ol;
            store_pair8(v[kq], s, oh, ol);
            st_h8(dst + kq * 2048, oh);
            st_h8(dst + kq * 2048 + 16, ol);
        }
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

WflPacked wfl_pack(int C, const float* conv, const float* conv_b, const float* cond, const float* cond_b, int n_mels,
                   const float* outp, const float* outp_b, const float* w_out, std::vector<uint16_t>& w16,
                   std::vector<float>& f32) {
    const int CT = C / 32, NQ = 2 * CT, KS_TAP = C / 16, KS1 = 9 * KS_TAP + WFL_KS_COND, KS2 = C / 16;
    WflPacked o;
    // one exponent for the first contraction (conv and condition weights share the accumulators), one for the res half
    // of the out projection
    {
        float m = 0.f;
        for (size_t i = 0; i < (size_t)2 * C * C * 9; ++i) m = std::fmax(m, std::fabs(conv[i]));
        for (size_t i = 0; i < (size_t)2 * C * n_mels; ++i) m = std::fmax(m, std::fabs(cond[i]));
        for (int i = 0; i < 2 * C; ++i) m = std::fmax(m, std::fabs(conv_b[i] + cond_b[i]));   // the bias column (below)
        o.k1 = pk_weight_scale_exp(&m, 1);
        o.k2res = pk_weight_scale_exp(outp, (size_t)C * C);
    }
    auto align8 = [&]() { w16.resize((w16.size() + 7) & ~(size_t)7); };
    align8();
    o.w1 = w16.size();
    w16.resize(o.w1 + (size_t)KS1 * 2 * NQ * 64 * 8, 0);
    uint16_t* a1 = w16.data() + o.w1;
    for (int ks = 0; ks < KS1; ++ks)
        for (int q = 0; q < NQ; ++q)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int i = lane & 31, hh = lane >> 5;
                    const int co = q < CT ? 32 * q + i : C + 32 * (q - CT) + i;   // content | gate (chunk :276)
                    float w;
                    if (ks < 9 * KS_TAP) {
                        const int tap = ks / KS_TAP, kr = tap / 3, kc = tap % 3;
                        const int ci = wfl_chan(ks % KS_TAP, hh, e);
                        w = conv[(((size_t)co * C + ci) * 3 + kr) * 3 + kc];
                    } else {
                        const int m = wfl_chan(ks - 9 * KS_TAP, hh, e);
                        // channel n_mels is the constant 1 of k_wf_cond_planes: its weight is the bias (:274-275)
                        w = m < n_mels ? cond[(size_t)co * n_mels + m] : (m == n_mels ? conv_b[co] + cond_b[co] : 0.f);
                    }
                    w = std::ldexp(w, o.k1);
                    put_split(a1 + ((((size_t)ks * 2 + 0) * NQ + q) * 64 + lane) * 8 + e,
                              a1 + ((((size_t)ks * 2 + 1) * NQ + q) * 64 + lane) * 8 + e, w);
                }
    align8();
    o.w2 = w16.size();
    w16.resize(o.w2 + (size_t)KS2 * 2 * CT * 64 * 8, 0);
    uint16_t* a2 = w16.data() + o.w2;
    for (int ks = 0; ks < KS2; ++ks)
        for (int t = 0; t < CT; ++t)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int i = lane & 31, hh = lane >> 5;
                    const int zc = 32 * (ks >> 1) + mfma_row(8 * (ks & 1) + e, hh);   // gated channel of this k-slot
                    const int row = 32 * t + i;                                       // res half (chunk :280)
                    const float w = std::ldexp(outp[(size_t)row * C + zc], o.k2res);
                    put_split(a2 + ((((size_t)ks * 2 + 0) * CT + t) * 64 + lane) * 8 + e,
                              a2 + ((((size_t)ks * 2 + 1) * CT + t) * 64 + lane) * 8 + e, w);
                }
    f32.resize((f32.size() + 3) & ~(size_t)3);
    o.b2r = f32.size();
    for (int c = 0; c < C; ++c) f32.push_back(std::ldexp(outp_b[c], PK_UNIT_EXP + o.k2res));
    // the skip half folded with the flow's output_proj: wso[which][zc] = sum_s w_out[which][s] * outp[C + s][zc], scaled
    // by 2^-14 (the gate leaves z * 2^14), in the order lane half hh reads it: [hh][k2][e][logs | b]
    o.wso = f32.size();
    for (int hh = 0; hh < 2; ++hh)
        for (int ks = 0; ks < KS2; ++ks)
            for (int e = 0; e < 8; ++e) {
                const int zc = 32 * (ks >> 1) + mfma_row(8 * (ks & 1) + e, hh);
                for (int which = 0; which < 2; ++which) {
                    double acc = 0.0;
                    for (int sc = 0; sc < C; ++sc) acc += (double)w_out[(size_t)which * C + sc] * (double)outp[(size_t)(C + sc) * C + zc];
                    f32.push_back((float)std::ldexp(acc, -PK_UNIT_EXP));
                }
            }
    for (int which = 0; which < 2; ++which) {
        double acc = 0.0;
        for (int sc = 0; sc < C; ++sc) acc += (double)w_out[(size_t)which * C + sc] * (double)outp_b[C + sc];
        o.cso[which] = acc;
    }
    return o;
}

// Timing ablations and the s_memtime trace (PK_WF_ABLATE; results are WRONG): instantiated in the profile build only
// (parakeet_amd/build.py build(profile=True)).  Returns 1 when no ablation applies.
template <bool PROF, class Go>
static int wfl_ablation(Go& go, bool shape_ok, bool trace, bool w12, bool f16, bool c128) {
    if constexpr (PROF) {
        static const int abl = pk_prof_env("PK_WF_ABLATE") ? atoi(pk_prof_env("PK_WF_ABLATE")) : 0;
        if (abl && shape_ok) {
            const bool plain8 = !w12 && !f16 && !c128;   // (the timing ablations exist for the 8-wave kernel in the default math)
            switch (abl) {
                case 1: if (plain8) return go(k_wf_layer_p<2, 3, 1>); break;
                case 4: if (plain8) return go(k_wf_layer_p<2, 3, 4>); break;
                case 8: if (plain8) return go(k_wf_layer_p<2, 3, 8>); break;
                case 9: if (plain8) return go(k_wf_layer_p<2, 3, 9>); break;
                case 13: if (plain8) return go(k_wf_layer_p<2, 3, 13>); break;
                case 16: if (trace && !c128) return w12 ? (f16 ? go(k_wf_layer_p<2, 3, 16, true, 12>) : go(k_wf_layer_p<2, 3, 16, false, 12>)) : go(k_wf_layer_p<2, 3, 16>); break;
                case 80: if (trace && !c128) return w12 ? go(k_wf_layer_p<2, 3, 80, false, 12>) : go(k_wf_layer_p<2, 3, 80>); break;
                case 64: if (c128) return f16 ? go(k_wf_layer_p<4, 3, 64, true>) : go(k_wf_layer_p<4, 3, 64>);
                         return w12 ? (f16 ? go(k_wf_layer_p<2, 3, 64, true, 12>) : go(k_wf_layer_p<2, 3, 64, false, 12>))
                                    : (f16 ? go(k_wf_layer_p<2, 3, 64, true>) : go(k_wf_layer_p<2, 3, 64>));
                case 32: if (plain8) return go(k_wf_layer_p<2, 3, 32>); break;   // A fragments one co-tile ahead (A/B of the LDS prefetch depth)
                case 128: if (w12 && !f16 && !c128) return go(k_wf_layer_p<2, 3, 128, false, 12>); break;   // the idle wave verifies the LDS slabs (round 6)
                default: PK_FAIL(PK_EINVAL, "PK_WF_ABLATE: 1, 4, 8, 9, 13, 16, 32, 64, 80 or 128");
            }
        }
    }
    return 1;
}

bool wfl_measurement_configs_allowed() {
    static const bool on = PK_PROFILE_BUILD && pk_prof_env("PK_WF_MEASURE") && atoi(pk_prof_env("PK_WF_MEASURE")) != 0;
    return on;
}

int wfl_layer_launch(pk_ctx* ctx, const WflLaunch& a) {
    if (!wfl_supports(a.C) || a.npos_alloc % WAVE_T != 0 || a.ntap % 3 != 0 || a.ntap < 3 || a.ntap > 9 || a.nl < 1 || a.nl > WFL_MAX_LAYERS)
        PK_FAIL(PK_EINVAL, "wfl_layer_launch: bad shape (C %d, npos %d, taps %d, layers %d)", a.C, a.npos_alloc, a.ntap, a.nl);
    if (a.nl > 1 && (!pk_grid_available() || !a.bar)) PK_FAIL(PK_ESTATE, "wfl_layer_launch: several layers per launch need the grid barrier");
    for (int t = 0; t < a.ntap; ++t)   // the kernel addresses W1 linearly: the taps present are the LAST ntap of the nine, in order
        if (a.tap_w[t] != 9 - a.ntap + t) PK_FAIL(PK_EINVAL, "wfl_layer_launch: tap %d carries weight tap %d, expected %d", t, a.tap_w[t], 9 - a.ntap + t);
    // operand offsets inside a source are 32-bit (k_wf_layer_p: scalar base + unsigned offset): 16 blocks of margin included
    if (((long)a.npos_alloc / WAVE_T + 16) * (long)std::max(a.C * 128, BLK_M_BYTES) >= (1L << 32))
        PK_FAIL(PK_EUNSUPPORTED, "wfl_layer_launch: %d positions per row exceed the 32-bit operand offsets; split the batch", a.npos_alloc);
    const int ntiles = a.npos_alloc / WAVE_T;
    WflLaunch b = a;
    b.tiles_per_wg = std::max(1, (ntiles + ctx->n_cu - 1) / ctx->n_cu);
    int grid = (ntiles + b.tiles_per_wg - 1) / b.tiles_per_wg;
    // 64 channels: 12-wave workgroups (three waves per SIMD, 168 registers) where they save a round -- the tiles of a
    // workgroup in ceil(t / 12) rounds instead of ceil(t / 8) (the benchmark's 8 x 640 frames: 11 tiles per workgroup, one round
    // instead of 8 + 3 with the second 3/8 full).  a.waves = 6 / 8 / 12 forces one (option "layer_waves" of pk_wf_set_option).
    // Round 5 had taken three waves per SIMD away from the default math (and two working waves per SIMD from the 128-channel
    // model): wrong tiles in 7 - 25 % of the calls.  Round 6 found the instruction pair -- the half-wave exchange of the folded
    // skip sums came out of packed FMAs with op_sel, which drop a product now and then beside other waves' matrix instructions (the op_sel
    // rule at the top of this file) --
    // and with the sums kept scalar every configuration is back: 0 wrong calls of 410 (profiles/r06_wf_fix_check.txt).
    const bool w6 = a.C == 64 && a.waves == 6 && a.nl == 1;   // two 6-wave workgroups per CU (see Shape)
    const bool w12 = !w6 && a.C == 64 && (a.waves == 12 || a.waves == 6 ||
                                          (a.waves != 8 && (b.tiles_per_wg + 11) / 12 < (b.tiles_per_wg + 7) / 8));
    const int W = w6 ? 6 : (w12 ? 12 : 8);
    if (w6) {
        b.tiles_per_wg = std::max(1, (ntiles + 2 * ctx->n_cu - 1) / (2 * ctx->n_cu));
        grid = (ntiles + b.tiles_per_wg - 1) / b.tiles_per_wg;
    }
    static const int active_env = pk_prof_env("PK_WF_ACTIVE") ? atoi(pk_prof_env("PK_WF_ACTIVE")) : 0;   // measurement switch
    const int active_default = W;   // (round 5 ran the 128-channel default math with 4: see above)
    b.active = active_env >= 1 && active_env <= W ? active_env : active_default;
    // several layers: the workgroups wait for one another (pk_grid.h) -- at most one per CU by construction (LDS), launched
    // cooperatively so that a grid that cannot be co-resident is an error, not a hang
    if (a.nl > 1) PK_HIP(hipMemsetAsync(a.bar, 0, sizeof(unsigned), ctx->stream));
    auto go = [&](auto kern) -> int {
        if (a.nl > 1) PK_LAUNCH_COOP(ctx, "wf_row", kern, dim3(grid), dim3(W * 64), b);
        else PK_LAUNCH(ctx, "wf_layer", kern, dim3(grid), dim3(W * 64), 0, b);
        return PK_OK;
    };
    const int nt = a.ntap / 3;
    if (a.nl > 1) {   // the multi-layer instantiations (option "persistent")
        if (w12) {
            if (a.f16) return nt == 1 ? go(k_wf_layer_p<2, 1, 0, true, 12, true>) : (nt == 2 ? go(k_wf_layer_p<2, 2, 0, true, 12, true>) : go(k_wf_layer_p<2, 3, 0, true, 12, true>));
            return nt == 1 ? go(k_wf_layer_p<2, 1, 0, false, 12, true>) : (nt == 2 ? go(k_wf_layer_p<2, 2, 0, false, 12, true>) : go(k_wf_layer_p<2, 3, 0, false, 12, true>));
        }
        if (a.f16) {
            if (a.C == 64)
                return nt == 1 ? go(k_wf_layer_p<2, 1, 0, true, 8, true>) : (nt == 2 ? go(k_wf_layer_p<2, 2, 0, true, 8, true>) : go(k_wf_layer_p<2, 3, 0, true, 8, true>));
            return nt == 1 ? go(k_wf_layer_p<4, 1, 0, true, 8, true>) : (nt == 2 ? go(k_wf_layer_p<4, 2, 0, true, 8, true>) : go(k_wf_layer_p<4, 3, 0, true, 8, true>));
        }
        if (a.C == 64) return nt == 1 ? go(k_wf_layer_p<2, 1, 0, false, 8, true>) : (nt == 2 ? go(k_wf_layer_p<2, 2, 0, false, 8, true>) : go(k_wf_layer_p<2, 3, 0, false, 8, true>));
        return nt == 1 ? go(k_wf_layer_p<4, 1, 0, false, 8, true>) : (nt == 2 ? go(k_wf_layer_p<4, 2, 0, false, 8, true>) : go(k_wf_layer_p<4, 3, 0, false, 8, true>));
    }
    // (profile build: the s_memtime trace exists for the 8- and the 12-wave kernel, the timing ablations for the 8-wave one)
    if (int st = wfl_ablation<PK_PROFILE_BUILD != 0>(go, nt == 3 && !w6, b.trace != nullptr, w12, a.f16 != 0, a.C == 128); st != 1) return st;
    if (w6) {
        if (a.f16) return nt == 1 ? go(k_wf_layer_p<2, 1, 0, true, 6>) : (nt == 2 ? go(k_wf_layer_p<2, 2, 0, true, 6>) : go(k_wf_layer_p<2, 3, 0, true, 6>));
        return nt == 1 ? go(k_wf_layer_p<2, 1, 0, false, 6>) : (nt == 2 ? go(k_wf_layer_p<2, 2, 0, false, 6>) : go(k_wf_layer_p<2, 3, 0, false, 6>));
    }
    if (w12) {
        if (a.f16) return nt == 1 ? go(k_wf_layer_p<2, 1, 0, true, 12>) : (nt == 2 ? go(k_wf_layer_p<2, 2, 0, true, 12>) : go(k_wf_layer_p<2, 3, 0, true, 12>));
        return nt == 1 ? go(k_wf_layer_p<2, 1, 0, false, 12>) : (nt == 2 ? go(k_wf_layer_p<2, 2, 0, false, 12>) : go(k_wf_layer_p<2, 3, 0, false, 12>));
    }
    if (a.f16) {
        if (a.C == 64)
            return nt == 1 ? go(k_wf_layer_p<2, 1, 0, true>) : (nt == 2 ? go(k_wf_layer_p<2, 2, 0, true>) : go(k_wf_layer_p<2, 3, 0, true>));
        return nt == 1 ? go(k_wf_layer_p<4, 1, 0, true>) : (nt == 2 ? go(k_wf_layer_p<4, 2, 0, true>) : go(k_wf_layer_p<4, 3, 0, true>));
    }
    if (a.C == 64) return nt == 1 ? go(k_wf_layer_p<2, 1>) : (nt == 2 ? go(k_wf_layer_p<2, 2>) : go(k_wf_layer_p<2, 3>));
    return nt == 1 ? go(k_wf_layer_p<4, 1>) : (nt == 2 ? go(k_wf_layer_p<4, 2>) : go(k_wf_layer_p<4, 3>));
}

int wfl_cond_planes_launch(pk_ctx* ctx, float* cond, long row_stride, int rows, int nblk, long amax_row_stride,
                           unsigned* amax, int n_mels) {
    if (n_mels >= WFL_MP) PK_FAIL(PK_EUNSUPPORTED, "the fused WaveFlow layer kernel needs a free condition channel (n_mels < %d)", WFL_MP);
    PK_LAUNCH(ctx, "wf_cond_planes", k_wf_cond_planes, dim3(nblk, rows), dim3(64), 0, cond, row_stride, amax_row_stride, amax,
              n_mels);
    return PK_OK;
}

int wfl_step_launch(pk_ctx* ctx, int C, const float* prm, float b_logs, float b_b, const float* z_row,
                    float* x_row, const float* w_in, const float* b_in, float* h0_next, unsigned* h0_amax,
                    const int* pos_utt, int npos_alloc, int first) {
    if (!wfl_supports(C)) PK_FAIL(PK_EINVAL, "wfl_step_launch: %d channels", C);
    const dim3 grid(pk_div_up(npos_alloc / WAVE_T, 4));
    if (C == 64)
        PK_LAUNCH(ctx, "wf_step", k_wf_step_p<2>, grid, dim3(256), 0, prm, b_logs, b_b, z_row, x_row, w_in, b_in, h0_next,
                  h0_amax, pos_utt, npos_alloc, first);
    else
        PK_LAUNCH(ctx, "wf_step", k_wf_step_p<4>, grid, dim3(256), 0, prm, b_logs, b_b, z_row, x_row, w_in, b_in, h0_next,
                  h0_amax, pos_utt, npos_alloc, first);
    return PK_OK;
}
