// gemm.hip -- exact-fp32 MFMA implicit-conv GEMM (v_mfma_f32_32x32x2_f32).
//
// Block tile 128(M) x 128(N) x 16(K), 4 waves as 2(M) x 2(N), each wave a 64x64
// sub-tile = 2x2 MFMA tiles of 32x32 (64 accumulator registers).  Activations are
// channels-last [rows][C]; a k-tap Conv1D is the same GEMM with the A rows
// shifted by (tap - pad) -- no im2col, no NCL<->NLC transposes (the reference
// does four per FFN, fastspeech2_transformer/multi_layer_conv.py:75-77).
//
// LDS images (double buffered, 2 x 16 KB):
//   As[kk][wm][i]{mt}  float2 per (k, wave-row, lane-row): one ds_read_b64 gives a
//   lane its A operand for both of its M tiles; Bs[kk][wn][j]{nt} likewise.  The
//   weight image is pre-packed on the host, so its staging is a straight copy.
#include <algorithm>
#include <cstring>
#include <string>

#include <type_traits>

#include "pk_gemm.h"
#include "pk_split.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {
constexpr int BM = PK_GEMM_BM, BN = PK_GEMM_BN, BK = PK_GEMM_BK;

__device__ __forceinline__ int mfma_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// Shared epilogue: acc[mt][nt] is the wave's (32*MT)x64 sub-tile (rows m0 + wm*32*MT + mt*32 + mfma_row(r, hi),
// columns nblk*128 + wn*64 + nt*32 + i).
template <int MT>
__device__ __forceinline__ void gemm_epilogue(const pk_gemm_args& a, f32x16 (&acc)[MT][2], int m0, int nblk, int wm,
                                              int wn, int i, int hi) {
    // epilogue
    if (a.epi == PK_EPI_GATE) {
        // acc[mt][0] = content, acc[mt][1] = gate of output channel nblk*64 + wn*32 + i
        const int col0 = nblk * BN + wn * 64 + i;
        const int n_out = nblk * 64 + wn * 32 + i;
        if (n_out >= a.N / 2) return;
        const float b0 = a.bias ? a.bias[col0] : 0.f, b1 = a.bias ? a.bias[col0 + 32] : 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * (32 * MT) + mt * 32 + mfma_row(r, hi);
                if (m >= a.M) continue;
                float ca = acc[mt][0][r] + b0;
                const float cb = acc[mt][1][r] + b1;
                ca = fminf(fmaxf(ca, -10.f), 10.f);
                const float ea = __expf(-2.f * ca), eb = __expf(-cb);
                float v = (1.f - ea) / ((1.f + ea) * (1.f + eb));
                if (a.rowvalid && a.rowvalid[m] < 0) v = 0.f;
                a.C[(long)m * a.ldc + n_out] = v;
            }
        return;
    }
    const int n_base = nblk * BN + wn * 64 + i;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int n = n_base + nt * 32;
        if (n >= a.N) continue;
        const float bias = a.bias ? a.bias[n] : 0.f;
        const float cs = a.cscale ? a.cscale[n] : 1.f;
        const float ch = a.cshift ? a.cshift[n] : 0.f;
        const bool to2 = a.nsplit > 0 && n >= a.nsplit;
        // Phase 1: every load of the epilogue (row validity, residual / running sum, row map) is issued before
        // the first store AND none of them depends on another loaded value (row indices are clamped instead of
        // branching on validity).  Otherwise each lane pays one serialised memory round trip per element:
        // load rowvalid -> wait -> branch -> load residual -> wait -> store, 32 times (measured on WaveFlow's
        // res|skip projection: 64 us of a 64 us kernel).
        // (done in batches of EB rows: EB values per lane in flight; 8 keeps k_gemm<1> at 3 waves per SIMD)
        constexpr int EB = 8;
        const int m_last = a.M - 1;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int rb = 0; rb < 16; rb += EB) {
                float oldv[EB];
                int valid[EB];    // 1 = store the value, 0 = store zero (gap row), -1 = no store
                int mo[EB];
#pragma unroll
                for (int q = 0; q < EB; ++q) {
                    const int m = min(m0 + wm * (32 * MT) + mt * 32 + mfma_row(rb + q, hi), m_last);
                    valid[q] = a.rowvalid ? a.rowvalid[m] : 0;          // raw value for now
                    mo[q] = a.out_rowmap && !to2 ? a.out_rowmap[m] : m;
                    if (to2) oldv[q] = a.acc2 ? a.C2[(long)m * a.ldc2 + (n - a.nsplit)] : 0.f;
                    else oldv[q] = a.res ? a.res[(long)m * a.ldr + n] : 0.f;
                }
#pragma unroll
                for (int q = 0; q < EB; ++q) {
                    const int m = m0 + wm * (32 * MT) + mt * 32 + mfma_row(rb + q, hi);
                    const bool gap = valid[q] < 0;
                    valid[q] = (m > m_last || mo[q] < 0) ? -1 : (gap ? 0 : 1);
                    if (gap && to2) oldv[q] = 0.f;
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < EB; ++q) {
                    if (valid[q] < 0) continue;
                    const int r = rb + q;
                    const int m = m0 + wm * (32 * MT) + mt * 32 + mfma_row(r, hi);
                    float v = acc[mt][nt][r] + bias;
                    if (a.res_pos == PK_RES_BEFORE_ACT && !to2) v += oldv[q];
                    if (a.act == PK_ACT_RELU) v = fmaxf(v, 0.f);
                    else if (a.act == PK_ACT_TANH) v = tanhf(v);
                    if (to2) {
                        v = valid[q] ? v + oldv[q] : 0.f;
                        a.C2[(long)m * a.ldc2 + (n - a.nsplit)] = v;
                        continue;
                    }
                    if (a.res_pos == PK_RES_AFTER_ACT) {
                        v += oldv[q];
                        if (!valid[q]) v = 0.f;
                        if (a.cscale) v = v * cs + ch;
                    } else {
                        if (a.cscale) v = v * cs + ch;
                        if (a.res_pos == PK_RES_AFTER_AFFINE) v += oldv[q];
                        if (!valid[q]) v = 0.f;
                    }
                    a.C[(long)mo[q] * a.ldc + n] = v;
                }
            }
    }
}

// SPI = K slabs (of 16) consumed per barrier: 2 halves the barrier / LDS-turnaround count on long K
// (64 KB of LDS, two blocks per CU); 1 keeps small problems at 32 KB.
template <int SPI>
__global__ __launch_bounds__(256, 3) void k_gemm(pk_gemm_args a) {
    __shared__ __attribute__((aligned(16))) float As[2][SPI][BK * BM];
    __shared__ __attribute__((aligned(16))) float Bs[2][SPI][BK * BN];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int i = lane & 31, hi = lane >> 5;
    const int m0 = blockIdx.x * BM;
    const int nblk = blockIdx.y;
    const int slabs_per_tap = a.Cin / BK;
    const int nmain = a.ntaps * slabs_per_tap;
    const int nslabs = nmain + a.Cin2 / BK;

    // A loader: thread -> (row, 8 consecutive k)
    const int lrow = tid >> 1, lhalf = tid & 1;
    const float* arow = a.A + (long)(m0 + lrow) * a.lda + lhalf * 8;
    const float* arow2 = a.A2 + (long)(m0 + lrow) * a.lda2 + lhalf * 8;
    const int a_lds = (((lhalf * 8) * 2 + (lrow >> 6)) * 32 + (lrow & 31)) * 2 + ((lrow >> 5) & 1);
    const float* wsrc = a.Wp + (long)nblk * a.wslabs_total * (BK * BN) + tid * 8;

    f32x4 ra0[SPI], ra1[SPI], rb0[SPI], rb1[SPI];
    auto load_slab = [&](int s, int u) {
        if (s >= nslabs) return;
        const float* p;
        int wslab;
        if (s < nmain) {
            const int tap = s / slabs_per_tap, sl = s - tap * slabs_per_tap;
            p = arow + a.tap_off[tap] + sl * BK;
            wslab = a.tap_w[tap] * slabs_per_tap + sl;
        } else {
            p = arow2 + (s - nmain) * BK;
            wslab = a.w2_slab0 + (s - nmain);
        }
        ra0[u] = *reinterpret_cast<const f32x4*>(p);
        ra1[u] = *reinterpret_cast<const f32x4*>(p + 4);
        const float* q = wsrc + (long)wslab * (BK * BN);
        rb0[u] = *reinterpret_cast<const f32x4*>(q);
        rb1[u] = *reinterpret_cast<const f32x4*>(q + 4);
    };
    auto store_slab = [&](int buf, int u) {
        float* d = As[buf][u] + a_lds;
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e * 128] = ra0[u][e];
#pragma unroll
        for (int e = 0; e < 4; ++e) d[(4 + e) * 128] = ra1[u][e];
        f32x4* b = reinterpret_cast<f32x4*>(Bs[buf][u] + tid * 8);
        b[0] = rb0[u];
        b[1] = rb1[u];
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

#pragma unroll
    for (int u = 0; u < SPI; ++u) load_slab(u, u);
#pragma unroll
    for (int u = 0; u < SPI; ++u) store_slab(0, u);
    __syncthreads();
    for (int s = 0; s < nslabs; s += SPI) {
        const int buf = (s / SPI) & 1;
#pragma unroll
        for (int u = 0; u < SPI; ++u) load_slab(s + SPI + u, u);
#pragma unroll
        for (int u = 0; u < SPI; ++u) {
            if (s + u < nslabs) {
                const f32x2* fa = reinterpret_cast<const f32x2*>(As[buf][u]) + (hi * 2 + wm) * 32 + i;
                const f32x2* fb = reinterpret_cast<const f32x2*>(Bs[buf][u]) + (hi * 2 + wn) * 32 + i;
#pragma unroll
                for (int ks = 0; ks < BK / 2; ++ks) {
                    const f32x2 av = fa[ks * 128];
                    const f32x2 bv = fb[ks * 128];
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bv[0], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bv[1], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bv[0], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bv[1], acc[1][1], 0, 0, 0);
                }
            }
        }
        if (s + SPI < nslabs) {
#pragma unroll
            for (int u = 0; u < SPI; ++u)
                if (s + SPI + u < nslabs) store_slab(buf ^ 1, u);
        }
        __syncthreads();
    }

    gemm_epilogue<2>(a, acc, m0, nblk, wm, wn, i, hi);
}

// ---------------------------------------------------------------- split-fp16 variant
// Same tiling and epilogue; every fp32 product is a_hi*b_hi + a_lo*b_hi + a_hi*b_lo with fp16 parts on
// v_mfma_f32_32x32x16_f16, fp32 accumulation (error of the result = exact-fp32 class, see pwg.hip).
// K slab = 32.  A operand = activations (rows), B operand = weights (cols).
// Block scaling (pk_split.h): the weight fragments of 128-column block nb hold W * 2^kw[nb] (kw[] = header of the
// packed blob); an activation row is split as 2^kx * a with kx from a_amax[] = max|A[r, :]| over the rows its taps
// read (and a2_amax[] for the appended operand), supplied by the producer of A or computed by k_row_amax; the
// accumulators are multiplied by 2^-(kx + kw) before the epilogue.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 pkh2 __attribute__((ext_vector_type(2)));
constexpr int HBK = PK_GEMM_HBK;           // 32
constexpr int H_B_BYTES = 2 * 2 * 4 * 64 * 16;   // 16 KB: [ks 2][part 2][nt 4][lane 64] x 16 B
constexpr int H_DEPTH = 3;                 // slabs in flight between global memory and LDS

// Both operands live in LDS as ready MFMA fragments of fp16 (hi, lo) parts:
//   Af[buf][ks 2][part 2][mt 4][lane 64] x 16 B  (activations: split ONCE per element by the thread that loaded
//                                                it, when it moves its registers to LDS -- MFMA and VALU do not
//                                                overlap on this part, so every VALU instruction kept out of
//                                                the four waves' inner loops is matrix time won)
//   Bs[buf][ks 2][part 2][nt 4][lane 64] x 16 B  (weights: pre-split at finalize)
// so the inner loop is ds_read_b128 + MFMA only.
__device__ __forceinline__ void gemm_split8(const float (&v)[8], f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        // hi = v_cvt_pkrtz (round toward zero, saturating at +-65504); x - hi exactly by v_fma_mix_f32 on the packed
        // high part; lo = fp16_rne(x - hi): |x - hi - lo| <= 2^-21 |x|
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

// max|A[r, 0..C)| for rows r0 <= r < r1 of a row-major matrix (amax is indexed like A: row r -> amax[r], rows
// before the base are valid memory on both).  LPR lanes share a row (16 for C <= 64, 32 for C <= 128, else 64), so
// narrow matrices keep all lanes busy: a wave covers 64 / LPR rows with 16-byte loads.
__global__ __launch_bounds__(256) void k_row_amax(const float* __restrict__ A, long lda, int C, long r0, long r1,
                                                  float* __restrict__ amax, int lpr) {
    const int lane = threadIdx.x & 63;
    const int rpw = 64 / lpr;                       // rows per wave
    const int sub = lane & (lpr - 1);
    const long r = r0 + ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * rpw + lane / lpr;
    float m = 0.f;
    if (r < r1) {
        const float* row = A + r * lda;
        if ((C & 3) == 0 && (lda & 3) == 0) {
            for (int c = sub * 4; c < C; c += lpr * 4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(row + c);
                m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
            }
        } else {
            for (int c = sub; c < C; c += lpr) m = fmaxf(m, fabsf(row[c]));
        }
    }
    for (int o = lpr >> 1; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (sub == 0 && r < r1) amax[r] = m;
}

// split of 2^k * x (x scaled exactly, then as gemm_split8)
__device__ __forceinline__ void gemm_split8s(const float (&v)[8], float s, f16x8& hi, f16x8& lo) {
    float t[8];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        f32x2 u = {v[2 * p], v[2 * p + 1]};
        u *= s;
        t[2 * p] = u[0];
        t[2 * p + 1] = u[1];
    }
    gemm_split8(t, hi, lo);
}

// MT = 32-row MFMA tiles per wave along M: 2 -> 128-row workgroup tiles (2 workgroups per CU), 1 -> 64-row tiles
// (48 KB of LDS, <= 170 VGPRs: 3 workgroups per CU) for grids that would otherwise end in a nearly empty round.
template <int MT>
__global__ __launch_bounds__(256, MT == 2 ? 2 : 3) void k_gemm_h3(pk_gemm_args a) {
    constexpr int TM = 64 * MT;                 // rows per workgroup
    constexpr int AG = MT;                      // 8-float groups of the A slab per thread
    // 65 slots per 64-lane fragment block: the pad staggers the blocks over the banks, so the two k-steps that
    // neighbouring loader threads write no longer collide (PMC: 20 % of the LDS cycles were bank conflicts)
    __shared__ __attribute__((aligned(16))) f16x8 Af[2][2 * 2 * 2 * MT * 65];
    __shared__ __attribute__((aligned(16))) f16x8 Bs[2][H_B_BYTES / 16];
    __shared__ float rinv[TM];                  // per output row: 2^-(kx + kw), applied to the accumulators
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int i = lane & 31, hi = lane >> 5;
    const int m0 = blockIdx.x * TM;
    const int nblk = blockIdx.y;
    const int slabs_per_tap = a.Cin / HBK;
    const int nmain = a.ntaps * slabs_per_tap;
    const int nslabs = nmain + a.Cin2 / HBK;

    // thread -> (row, AG consecutive 8-float groups): MT = 2: 16 floats = one k-step; MT = 1: 8 floats = half a k-step
    const int lrow = (tid * AG) >> 2, lgrp = (tid * AG) & 3;
    const float* arow = a.A + (long)(m0 + lrow) * a.lda + lgrp * 8;
    const float* arow2 = a.A2 + (long)(m0 + lrow) * a.lda2 + lgrp * 8;
    // packed weights: header of ceil(nblks / 4) x 16 B with the exponent of every 128-column block, then fragments
    const int hdr16 = ((int)gridDim.y + 3) >> 2;
    const int kw = reinterpret_cast<const int*>(a.Wh)[nblk];
    const f16x8* wsrc = reinterpret_cast<const f16x8*>(a.Wh) + hdr16 + (long)nblk * a.wslabs_total * (H_B_BYTES / 16) + tid;
    // this thread's A row: its scale 2^kx from the largest magnitude among the rows its taps read
    float sx;
    {
        const long m = m0 + lrow;
        float am = 0.f;
        for (int t = 0; t < a.ntaps; ++t) am = fmaxf(am, a.a_amax[m + a.tap_row[t]]);
        if (a.Cin2 > 0) am = fmaxf(am, a.a2_amax[m]);
        const int kx = blk_scale_exp(__float_as_uint(am));
        sx = pow2f(kx);
        if (lgrp == 0) rinv[lrow] = pow2f(-(kx + kw));   // visible after the first barrier of the K loop
    }
    // fragment slot of group g: (ks = g / 2, part, mt = lrow / 32, lane = lrow % 32 + 32 * (g % 2))
    auto a_slot = [&](int g, int part) {
        return (((g >> 1) * 2 + part) * (2 * MT) + (lrow >> 5)) * 65 + (lrow & 31) + 32 * (g & 1);
    };

    // Global -> register staging ring, H_DEPTH slabs ahead of the MFMAs (one slab of compute is ~0.4 us,
    // far less than the load latency under load), then registers -> LDS double buffer.
    f32x4 ra[H_DEPTH][2 * AG];
    f16x8 rb[H_DEPTH][4];
    auto load_slab = [&](int s, auto SET) {
        constexpr int set = decltype(SET)::value;
        const float* p;
        int wslab;
        if (s < nmain) {
            const int tap = s / slabs_per_tap, sl = s - tap * slabs_per_tap;
            p = arow + a.tap_off[tap] + sl * HBK;
            wslab = a.tap_w[tap] * slabs_per_tap + sl;
        } else {
            p = arow2 + (s - nmain) * HBK;
            wslab = a.w2_slab0 + (s - nmain);
        }
#pragma unroll
        for (int c = 0; c < 2 * AG; ++c) ra[set][c] = *reinterpret_cast<const f32x4*>(p + 4 * c);
        const f16x8* q = wsrc + (long)wslab * (H_B_BYTES / 16);
#pragma unroll
        for (int c = 0; c < 4; ++c) rb[set][c] = q[c * 256];
    };
    auto store_slab = [&](int buf, auto SET) {
        constexpr int set = decltype(SET)::value;
#pragma unroll
        for (int g = 0; g < AG; ++g) {
            const f32x4 v0 = ra[set][2 * g], v1 = ra[set][2 * g + 1];
            const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            f16x8 fh, fl;
            gemm_split8s(v, sx, fh, fl);
            Af[buf][a_slot(lgrp + g, 0)] = fh;
            Af[buf][a_slot(lgrp + g, 1)] = fl;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) Bs[buf][tid + c * 256] = rb[set][c];
    };

    f32x16 acc[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // one K slab of MFMAs from LDS buffer `buf`
    auto mma_slab = [&](int buf) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8 ah[MT], al[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                ah[mt] = Af[buf][((ks * 2 + 0) * (2 * MT) + wm * MT + mt) * 65 + lane];
                al[mt] = Af[buf][((ks * 2 + 1) * (2 * MT) + wm * MT + mt) * 65 + lane];
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const f16x8 bh = Bs[buf][((ks * 2 + 0) * 4 + wn * 2 + nt) * 64 + lane];
                const f16x8 bl = Bs[buf][((ks * 2 + 1) * 4 + wn * 2 + nt) * 64 + lane];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl, acc[mt][nt], 0, 0, 0);
                }
            }
        }
    };

    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;
    typedef std::integral_constant<int, 2> S2;
    static_assert(H_DEPTH == 3, "the slab loop below is unrolled for a staging ring of 3");
    // Loads and LDS stores are unconditional (slab indices clamp to the last slab, the surplus store goes to
    // the buffer nobody reads any more): with branches around them the compiler's s_waitcnt insertion has
    // to assume the shortest path and degrades every wait to vmcnt(0), i.e. no prefetch at all.
    const int last = nslabs - 1;
    load_slab(0, S0{});
    load_slab(1 < last ? 1 : last, S1{});
    load_slab(2 < last ? 2 : last, S2{});
    store_slab(0, S0{});
    __syncthreads();
    // one slab: request slab s+DEPTH into the set slab s came from, run slab s, move slab s+1 to LDS
    auto step = [&](int s, auto SET, auto NEXT) {
        const int buf = s & 1;
        load_slab(s + H_DEPTH < last ? s + H_DEPTH : last, SET);
        __builtin_amdgcn_sched_barrier(0);
        mma_slab(buf);
        __builtin_amdgcn_sched_barrier(0);
        store_slab(buf ^ 1, NEXT);
        __syncthreads();
    };
    int s = 0;
    for (; s + H_DEPTH <= nslabs; s += H_DEPTH) {
        step(s, S0{}, S1{});
        step(s + 1, S1{}, S2{});
        step(s + 2, S2{}, S0{});
    }
    if (s < nslabs) {
        step(s, S0{}, S1{});
        if (s + 1 < nslabs) step(s + 1, S1{}, S2{});
    }
    // undo the block scales: row r of the tile was accumulated as 2^(kx_r + kw) * (A W)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float ri = rinv[wm * (32 * MT) + mt * 32 + mfma_row(r, hi)];
            acc[mt][0][r] *= ri;
            acc[mt][1][r] *= ri;
        }
    if (a.epi == PK_EPI_GATE_PROJ) {
        // ---- stage 2: z = tanh(content + b) * sigmoid(gate + b) -> LDS fragments, then out = z . W2 (K = 64, N = 128)
        {
            const int col0 = wn * 64 + i;
            const float b0 = a.bias ? a.bias[col0] : 0.f, b1 = a.bias ? a.bias[col0 + 32] : 0.f;
            // z channel wn*32 + i = k-slab wn, k = i: k-step i/16, k half (i/8)&1, element i%8
            _Float16* zf = reinterpret_cast<_Float16*>(Af[wn]);
            const int e_off = (((i >> 4) * 2 + 0) * (2 * MT) * 65 + 32 * ((i >> 3) & 1)) * 8 + (i & 7);   // part 0, mt 0, row 0
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * (32 * MT) + mt * 32 + mfma_row(r, hi);
                    float ca = acc[mt][0][r] + b0;
                    const float cb = acc[mt][1][r] + b1;
                    ca = fminf(fmaxf(ca, -10.f), 10.f);
                    const float ea = __expf(-2.f * ca), eb = __expf(-cb);
                    float v = (1.f - ea) / ((1.f + ea) * (1.f + eb));
                    const int m = m0 + row;
                    if (m >= a.M || (a.rowvalid && a.rowvalid[m] < 0)) v = 0.f;
                    v *= PK_UNIT_SCALE;                              // |z| < 1: fixed block scale 2^14
                    const _Float16 vh = (_Float16)v;
                    const _Float16 vl = (_Float16)(v - (float)vh);
                    const int o = e_off + ((row >> 5) * 65 + (row & 31)) * 8;
                    zf[o] = vh;
                    zf[o + (2 * MT) * 65 * 8] = vl;                  // part 1
                }
            const f16x8* w2 = reinterpret_cast<const f16x8*>(a.Wh2) + 1 + tid;   // + header (one 128-column block)
#pragma unroll
            for (int c = 0; c < 8; ++c) Bs[c >> 2][tid + (c & 3) * 256] = w2[c * 256];
        }
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
        mma_slab(0);
        mma_slab(1);
        {
            const float s2 = pow2f(-(PK_UNIT_EXP + reinterpret_cast<const int*>(a.Wh2)[0]));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mt][nt][r] *= s2;
        }
        pk_gemm_args b = a;
        b.epi = PK_EPI_STD;
        b.bias = a.bias2;
        b.act = PK_ACT_NONE;
        gemm_epilogue<MT>(b, acc, m0, nblk, wm, wn, i, hi);
        return;
    }
    gemm_epilogue<MT>(a, acc, m0, nblk, wm, wn, i, hi);
}
}  // namespace

size_t pk_gemm_pack(const float* Wkn, int K, int N, std::vector<float>& out) {
    const int nblks = (N + BN - 1) / BN, nslabs = K / BK;
    out.assign((size_t)nblks * nslabs * BK * BN, 0.f);
    for (int nb = 0; nb < nblks; ++nb)
        for (int s = 0; s < nslabs; ++s) {
            float* img = out.data() + ((size_t)nb * nslabs + s) * (BK * BN);
            for (int kk = 0; kk < BK; ++kk)
                for (int wn = 0; wn < 2; ++wn)
                    for (int j = 0; j < 32; ++j)
                        for (int nt = 0; nt < 2; ++nt) {
                            const int n = nb * BN + wn * 64 + nt * 32 + j;
                            const int k = s * BK + kk;
                            img[((kk * 2 + wn) * 32 + j) * 2 + nt] = (n < N) ? Wkn[(size_t)k * N + n] : 0.f;
                        }
        }
    return out.size();
}

static inline uint16_t gemm_f32_to_f16(float f) {
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
static inline float gemm_f16_to_f32(uint16_t h) {
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

size_t pk_gemm_pack_h3(const float* Wkn, int K, int N, std::vector<uint16_t>& out) {
    const int nblks = (N + BN - 1) / BN, nslabs = K / PK_GEMM_HBK;
    const size_t per = H_B_BYTES / 2;   // halves per (n-block, slab)
    // header: one int per 128-column block = exponent kw of its block scale (pk_split.h), padded to 16 bytes
    const size_t hdr = (size_t)((nblks + 3) / 4) * 8;   // halves
    out.assign(hdr + (size_t)nblks * nslabs * per, 0);
    std::vector<int> kws(nblks, 0);
    for (int nb = 0; nb < nblks; ++nb) {
        float m = 0.f;
        for (int k = 0; k < K; ++k)
            for (int n = nb * BN; n < N && n < (nb + 1) * BN; ++n) {
                const float v = std::fabs(Wkn[(size_t)k * N + n]);
                if (std::isfinite(v) && v > m) m = v;
            }
        kws[nb] = pk_weight_scale_exp(&m, 1);
    }
    memcpy(out.data(), kws.data(), nblks * sizeof(int));
    for (int nb = 0; nb < nblks; ++nb)
        for (int s = 0; s < nslabs; ++s) {
            uint16_t* img = out.data() + hdr + ((size_t)nb * nslabs + s) * per;
            for (int ks = 0; ks < 2; ++ks)
                for (int nt = 0; nt < 4; ++nt)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int j = lane & 31, kb = lane >> 5;
                            const int k = s * PK_GEMM_HBK + ks * 16 + kb * 8 + e;
                            const int n = nb * BN + nt * 32 + j;
                            const float w = n < N ? std::ldexp(Wkn[(size_t)k * N + n], kws[nb]) : 0.f;
                            const uint16_t h = gemm_f32_to_f16(w);
                            const uint16_t l = gemm_f32_to_f16(w - gemm_f16_to_f32(h));
                            img[((((size_t)ks * 2 + 0) * 4 + nt) * 64 + lane) * 8 + e] = h;
                            img[((((size_t)ks * 2 + 1) * 4 + nt) * 64 + lane) * 8 + e] = l;
                        }
        }
    return out.size();
}

void pk_conv_to_kn(const float* w, int Cout, int Cin, int k, std::vector<float>& out) {
    out.resize((size_t)k * Cin * Cout);
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci)
            for (int tap = 0; tap < k; ++tap)
                out[((size_t)tap * Cin + ci) * Cout + co] = w[((size_t)co * Cin + ci) * k + tap];
}

void pk_gemm_gate_permute(const float* Wkn, int K, int Cz, std::vector<float>& out) {
    const int N = 2 * Cz;
    out.resize((size_t)K * N);
    for (int k = 0; k < K; ++k)
        for (int nb = 0; nb < Cz / 64; ++nb)
            for (int wn = 0; wn < 2; ++wn)
                for (int half = 0; half < 2; ++half)
                    for (int j = 0; j < 32; ++j)
                        out[(size_t)k * N + nb * 128 + wn * 64 + half * 32 + j] =
                            Wkn[(size_t)k * N + half * Cz + nb * 64 + wn * 32 + j];
}

void pk_gemm_gate_permute_bias(const float* b, int Cz, std::vector<float>& out) {
    pk_gemm_gate_permute(b, 1, Cz, out);
}

int pk_row_amax_launch(pk_ctx* ctx, const float* A, long lda, int C, long r0, long r1, float* amax) {
    if (r1 <= r0) return PK_OK;
    const int lpr = C <= 64 ? 16 : (C <= 128 ? 32 : 64);
    PK_LAUNCH(ctx, "row_amax", k_row_amax, dim3(pk_div_up(r1 - r0, 4 * (64 / lpr))), dim3(256), 0, A, lda, C, r0, r1,
              amax, lpr);
    return PK_OK;
}

int pk_gemm_launch(pk_ctx* ctx, const char* prof_name, const pk_gemm_args& in) {
    pk_gemm_args a = in;
    if (a.Cin % BK != 0 || a.Cin2 % BK != 0)
        PK_FAIL(PK_EUNSUPPORTED, "GEMM: input channels (%d, %d) must be multiples of %d", a.Cin, a.Cin2, BK);
    if (a.M <= 0 || a.N <= 0) PK_FAIL(PK_EINVAL, "GEMM: empty problem");
    // short-K problems are epilogue / memory bound: the fp32 kernel's higher occupancy wins there
    // (WaveFlow out_proj, K = 64: 64 us vs 92 us); from K = 128 the split kernel with its 64-row tiles is ahead
    // (WaveFlow C = 128 out_proj 140 -> 104 us, SpeedySpeech Linear layers)
    const int k_total = (a.ntaps ? a.ntaps : a.taps) * a.Cin + a.Cin2;
    static const int h3_min_k = pk_prof_env("PK_GEMM_H3_MINK") ? atoi(pk_prof_env("PK_GEMM_H3_MINK")) : 128;
    const bool h3 = a.math == PK_GEMM_MATH_F16X3 && a.Wh && a.Cin % PK_GEMM_HBK == 0 && a.Cin2 % PK_GEMM_HBK == 0 &&
                    k_total >= h3_min_k;
    const int bk = h3 ? PK_GEMM_HBK : BK;
    if (a.ntaps == 0) {
        if (a.taps > PK_GEMM_MAX_TAPS) PK_FAIL(PK_EUNSUPPORTED, "GEMM: more than %d taps", PK_GEMM_MAX_TAPS);
        a.ntaps = a.taps;
        for (int t = 0; t < a.taps; ++t) {
            a.tap_off[t] = (long)(t - a.pad) * a.lda;
            a.tap_w[t] = t;
        }
        a.wslabs_total = a.taps * a.Cin / bk + a.Cin2 / bk;
    } else if (h3) {
        // callers give slab bookkeeping in units of 16; the split kernel uses slabs of 32
        a.w2_slab0 /= 2;
        a.wslabs_total /= 2;
    }
    if (a.wslabs_total <= 0) PK_FAIL(PK_EINVAL, "GEMM: wslabs_total not set");
    if (!a.A2) { a.A2 = a.A; a.lda2 = 0; }
    if (a.epi == PK_EPI_GATE && (a.N % 128 != 0)) PK_FAIL(PK_EUNSUPPORTED, "gated GEMM needs N %% 128 == 0");
    if (a.epi == PK_EPI_GATE_PROJ && (!h3 || a.N != 128 || !a.Wh2))
        PK_FAIL(PK_EUNSUPPORTED, "fused gate + projection needs the split-fp16 kernel, N == 128 and Wh2");
    dim3 grid((a.M + BM - 1) / BM, (a.N + BN - 1) / BN);
    const int nslabs = a.ntaps * (a.Cin / BK) + a.Cin2 / BK;
    // k_gemm<2> (two slabs per barrier, 64 KB LDS) measured slower on every FS2 / WaveFlow shape
    // (ffn1 0.49 vs 0.42 ms, WaveFlow conv 160 vs 140 us): occupancy beats fewer barriers here.
    (void)nslabs;
    if (h3) {
        // block scaling: row maxima of the A operand(s) over every row a tile can read
        const long rows_pad = (long)grid.x * BM;
        long lo = 0, hi_row = 0;
        for (int t = 0; t < a.ntaps; ++t) {
            if (a.lda <= 0 || a.tap_off[t] % a.lda != 0)
                PK_FAIL(PK_EUNSUPPORTED, "split-fp16 GEMM: tap offset %ld is not a whole number of rows (lda %d)",
                        a.tap_off[t], a.lda);
            a.tap_row[t] = (int)(a.tap_off[t] / a.lda);
            lo = std::min<long>(lo, a.tap_row[t]);
            hi_row = std::max<long>(hi_row, a.tap_row[t]);
        }
        if (!a.a_amax) {
            pk_ctx_scratch* sc = pk_ctx_get_scratch(ctx);
            PK_TRY(sc->row_amax.reserve((size_t)(rows_pad + hi_row - lo) * sizeof(float)));
            float* am = sc->row_amax.as<float>() - lo;
            PK_TRY(pk_row_amax_launch(ctx, a.A, a.lda, a.Cin, lo, rows_pad + hi_row, am));
            a.a_amax = am;
        }
        if (a.Cin2 > 0 && !a.a2_amax) {
            pk_ctx_scratch* sc = pk_ctx_get_scratch(ctx);
            PK_TRY(sc->row_amax2.reserve((size_t)rows_pad * sizeof(float)));
            PK_TRY(pk_row_amax_launch(ctx, a.A2, a.lda2, a.Cin2, 0, rows_pad, sc->row_amax2.as<float>()));
            a.a2_amax = sc->row_amax2.as<float>();
        }
        const std::string nm = std::string(prof_name) + "_h3";
        // 128-row tiles at 2 workgroups per CU, or 64-row tiles at 3 per CU: take the 64-row grid when the
        // 128-row one would leave the machine idle in its last round (fewer rounds-equivalents of work)
        const long slots2 = 2L * ctx->n_cu, slots3 = 3L * ctx->n_cu;
        const long wg2 = (long)grid.x * grid.y, wg1 = (long)((a.M + 63) / 64) * grid.y;
        const double t2 = (double)((wg2 + slots2 - 1) / slots2);            // rounds of full-size work
        static const double small_factor = pk_prof_env("PK_GEMM_SMALL_FACTOR") ? atof(pk_prof_env("PK_GEMM_SMALL_FACTOR")) : 0.7;
        const double t1 = small_factor * (double)((wg1 + slots3 - 1) / slots3);   // half-size tiles, 3 per CU share the pipes
        static const int force = pk_prof_env("PK_GEMM_TILE") ? atoi(pk_prof_env("PK_GEMM_TILE")) : 0;   // 64 / 128: measurement override
        if (force == 64 || (force != 128 && t1 < t2)) {
            dim3 g1((a.M + 63) / 64, grid.y);
            PK_LAUNCH(ctx, nm.c_str(), k_gemm_h3<1>, g1, dim3(256), 0, a);
        } else {
            PK_LAUNCH(ctx, nm.c_str(), k_gemm_h3<2>, grid, dim3(256), 0, a);
        }
        return PK_OK;
    }
    if (!a.Wp) PK_FAIL(PK_EINVAL, "GEMM: fp32 weights missing");
    PK_LAUNCH(ctx, prof_name, k_gemm<1>, grid, dim3(256), 0, a);
    return PK_OK;
}
