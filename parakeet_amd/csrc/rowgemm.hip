// rowgemm.hip -- see pk_rowgemm.h.
#include "pk_rowgemm.h"

#include "pk_gemm.h"
#include "pk_philox.h"

void pk_rowgemm_pack(const float* Wkn, int K, int N, std::vector<float>& out) {
    // [N / 16][KP / 4][16 columns][4 consecutive k], KP = K rounded up to 16, zeros beyond K and N: the float4 a lane of
    // k_rowgemm reads holds W[4 q .. 4 q + 3][column] -- the B operands of four matrix instructions
    const int cw = pk_rowgemm_cw(N), nb = (N + cw - 1) / cw, KP = (K + 15) / 16 * 16;
    out.assign((size_t)nb * KP * cw, 0.f);
    for (int b = 0; b < nb; ++b)
        for (int k = 0; k < K; ++k)
            for (int c = 0; c < cw && b * cw + c < N; ++c)
                out[(((size_t)b * (KP / 4) + k / 4) * cw + c) * 4 + (k & 3)] = Wkn[(size_t)k * N + b * cw + c];
}

void pk_rowgemm_lstm_perm(int H, std::vector<int>& perm) {
    perm.resize((size_t)4 * H);
    for (int u = 0; u < H; ++u)
        for (int g = 0; g < 4; ++g) perm[(size_t)(u / 4) * 16 + g * 4 + (u % 4)] = g * H + u;
}

namespace {
constexpr int KC = PK_RG_KC, ROWS = PK_RG_ROWS;
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float rg_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

// One workgroup = 16 output columns x up to 32 rows, 8 waves; the K range is dealt out in steps of 16 k, wave w takes the steps
// w, w + 8, ...  Latency is everything here (32 rows: a launch is 10 - 40 MFLOP), and round 4's rocprof / HIP-event figures put
// the previous kernel (activations transposed into LDS, one fp32 FMA per LDS-broadcast operand) at 10 - 20 us per launch with
// 3 us of that between two launches: its inner loop was bound by the LDS (a ds_read_b128 per four FMAs, 8 waves on one LDS:
// 8 k cycles per 512 k), twice with K = 1024, ..., and the tile went through LDS before anything could start.  Now the rows
// never touch LDS: v_mfma_f32_16x16x4_f32 (fp32 operands, fp32 accumulation: no splitting, no block scales) takes A = 16 rows x
// 4 k and B = 4 k x 16 columns one value per lane -- lane (r = lane % 16, j = lane / 16) reads the float4 x[row r][16 t + 4 j ..]
// of both row halves and the float4 W[16 t + 4 j ..][column r] of the packed slab, and component c of the three is one
// instruction's operands (the k of lane group j in instruction c is 16 t + 4 j + c for A and B alike).  Everything a wave
// needs for 4 steps (12 float4 per lane, + LayerNorm weight and bias) is requested at once, the next 4 steps before the
// arithmetic of these; the 8 waves' partial tiles are summed through 16 KB of LDS in wave order (deterministic).
//   Optional prologue: LayerNorm over K <= 512: mean and squared deviations of each lane's 16 values of its two rows, summed
//   over the wave's four lane groups by shuffles and over the waves through LDS (two passes over the registers), applied there.
//   Epilogues: bias, ReLU, dropout, residual; or the LSTM cell (see pk_rowgemm.h).
// The stop-token head (pk_rowgemm.h): one wave per row, 16 values per lane (K <= 1024); the arithmetic of the kernel the
// decoders used to launch for it.
__device__ __forceinline__ void rg_stop_block(const pk_rowgemm_args& a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nb = (a.N + 15) / 16;   // the GEMM's workgroups come first; stop workgroup i owns rows 8 i .. 8 i + 7, a wave each
    const int b = ((int)blockIdx.x - nb) * 8 + wave;
    if (b < a.M) {
        const float* z = a.x + (long)b * a.ldx;
        float s = 0.f;
        if (a.ln_g) {
            float v[16];
            float t = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int c = lane + 64 * e;
                v[e] = c < a.K ? z[c] : 0.f;
                t += v[e];
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
            const float mean = t / (float)a.K;
            float q = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float d = lane + 64 * e < a.K ? v[e] - mean : 0.f;
                q += d * d;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
            const float rstd = 1.0f / sqrtf(q / (float)a.K + a.ln_eps);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int c = lane + 64 * e;
                if (c < a.K) s = fmaf((v[e] - mean) * rstd * a.ln_g[c] + a.ln_b[c], a.stop_w[c], s);
            }
        } else {
            for (int c = lane; c < a.K; c += 64) s = fmaf(z[c], a.stop_w[c], s);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0 && a.stop_kind == 1) {
            s += a.stop_bias;
            a.stop_probs[(long)a.stop_step * a.M + b] = s;
            const bool end = 1.f / (1.f + expf(-s)) > 0.5f || a.stop_step + 1 >= a.stop_max_steps;
            if (end && a.stop_len[b] == 0) {
                a.stop_len[b] = a.stop_step + 1;
                atomicAdd(a.stop_ndone, 1);
            }
        } else if (lane == 0) {
            const float p = 1.f / (1.f + expf(-(s + a.stop_bias)));
            a.stop_probs[(long)(a.stop_step - 1) * a.M + b] = p;
            if (a.stop_len[b] == 0 && (p >= a.stop_thr || a.stop_step >= a.stop_maxlen[b]) && a.stop_step >= a.stop_minlen[b]) {
                a.stop_len[b] = a.stop_step;
                atomicAdd(a.stop_ndone, 1);
            }
        }
    }
}

template <bool LN>
__global__ __launch_bounds__(512) void k_rowgemm(pk_rowgemm_args a) {
    if (a.stop_w && (int)blockIdx.x >= (a.N + 15) / 16) {   // (block-uniform: the extra workgroups of the launch)
        rg_stop_block(a);
        return;
    }
    constexpr int NWV = 8, CH = 4, CW = 16;
    __shared__ float red[NWV * ROWS * CW];     // red[(wave * 32 + m) * 16 + col], 16 KB
    __shared__ float stat[2 * NWV * ROWS];     // LayerNorm: every wave's partial sums of every row (sums | squared deviations)
    __shared__ float gl[ROWS * CW];            // LSTM epilogue: the gate pre-activations
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, j = lane >> 4;
    const int m0 = blockIdx.y * ROWS;
    const int rows = min(ROWS, a.M - m0);
    const int T = (a.K + 15) >> 4;                         // steps of 16 k
    const int S = T > wave ? (T - wave + NWV - 1) / NWV : 0;   // this wave's steps t = wave + 8 s
    const f32x4* wt = reinterpret_cast<const f32x4*>(a.Wt) + (long)blockIdx.x * (4 * T) * CW;   // [4 T][16] float4
    const float* x0 = a.x + (long)(m0 + min(r, rows - 1)) * a.ldx;
    const float* x1 = a.x + (long)(m0 + min(16 + r, rows - 1)) * a.ldx;
    // the epilogue's operands of this thread's output element (e = tid: row e / 16, column e % 16), requested now
    const int em = min(tid / CW, rows - 1), ec = tid % CW, en = min((int)blockIdx.x * CW + ec, a.N - 1);
    float e_bias = 0.f, e_res = 0.f;

    f32x4 xa[CH], xb[CH], wv[CH], lg[CH], lb[CH];
    auto request = [&](int s0, f32x4 (&pa)[CH], f32x4 (&pb)[CH], f32x4 (&pw)[CH]) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const bool ok = s0 + i < S;
            const int t = ok ? wave + NWV * (s0 + i) : 0;
            const int k0 = 16 * t + 4 * j;
            const int kx = k0 < a.K ? k0 : 0;              // beyond K: any valid address (the packed weights are zero there)
            pa[i] = *reinterpret_cast<const f32x4*>(x0 + kx);
            pb[i] = *reinterpret_cast<const f32x4*>(x1 + kx);
            pw[i] = wt[(4 * t + j) * CW + r];
            if (LN && s0 == 0) {
                lg[i] = *reinterpret_cast<const f32x4*>(a.ln_g + kx);
                lb[i] = *reinterpret_cast<const f32x4*>(a.ln_b + kx);
            }
        }
    };
    request(0, xa, xb, wv);
    if (a.bias) e_bias = a.bias[en];
    if (a.res) e_res = a.res[(long)(m0 + em) * a.ldr + en];

    if (LN) {   // K <= 512: S <= 4, the whole of this wave's share is in xa / xb
        // two passes over the registers (mean, then the squared deviations), each reduced over the wave's four lane groups by
        // shuffles and over the eight waves through LDS in wave order.  (First version: one pass with Chan's merge of (count,
        // mean, M2) partials -- 29 divisions and 35 branches in a dependent chain, 3 us of a 9 us launch.)
        const float inv_k = 1.0f / (float)a.K;
        auto reduce_rows = [&](float v0, float v1, float* buf, float& t0, float& t1) {
            v0 += __shfl_xor(v0, 16);
            v1 += __shfl_xor(v1, 16);
            v0 += __shfl_xor(v0, 32);
            v1 += __shfl_xor(v1, 32);
            if (j == 0) {
                buf[wave * ROWS + r] = v0;
                buf[wave * ROWS + 16 + r] = v1;
            }
            __syncthreads();
            t0 = 0.f;
            t1 = 0.f;
#pragma unroll
            for (int w = 0; w < NWV; ++w) {
                t0 += buf[w * ROWS + r];
                t1 += buf[w * ROWS + 16 + r];
            }
        };
        bool okk[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) okk[i] = i < S && 16 * (wave + NWV * i) + 4 * j < a.K;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            s0 += okk[i] ? (xa[i][0] + xa[i][1]) + (xa[i][2] + xa[i][3]) : 0.f;
            s1 += okk[i] ? (xb[i][0] + xb[i][1]) + (xb[i][2] + xb[i][3]) : 0.f;
        }
        float tm0, tm1;
        reduce_rows(s0, s1, stat, tm0, tm1);
        tm0 *= inv_k;
        tm1 *= inv_k;
        float q0 = 0.f, q1 = 0.f;
#pragma unroll
        for (int i = 0; i < CH; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float d0 = xa[i][c] - tm0, d1 = xb[i][c] - tm1;
                q0 += okk[i] ? d0 * d0 : 0.f;
                q1 += okk[i] ? d1 * d1 : 0.f;
            }
        float tq0, tq1;
        reduce_rows(q0, q1, stat + NWV * ROWS, tq0, tq1);
        const float rs0 = 1.0f / sqrtf(tq0 * inv_k + a.ln_eps), rs1 = 1.0f / sqrtf(tq1 * inv_k + a.ln_eps);
#pragma unroll
        for (int i = 0; i < CH; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                xa[i][c] = (xa[i][c] - tm0) * rs0 * lg[i][c] + lb[i][c];
                xb[i][c] = (xb[i][c] - tm1) * rs1 * lg[i][c] + lb[i][c];
            }
    }

    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < S; s0 += CH) {
        f32x4 na[CH], nb[CH], nw[CH];
        const bool more = s0 + CH < S;   // (wave-uniform)
        if (more) request(s0 + CH, na, nb, nw);
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const bool ok = s0 + i < S;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float w = ok ? wv[i][c] : 0.f;
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[i][c], w, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[i][c], w, acc1, 0, 0, 0);
            }
        }
        if (more) {
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                xa[i] = na[i];
                xb[i] = nb[i];
                wv[i] = nw[i];
            }
        }
    }
    // lane (r, j) holds rows 4 j + q (acc0) and 16 + 4 j + q (acc1) of column r
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        red[(wave * ROWS + 4 * j + q) * CW + r] = acc0[q];
        red[(wave * ROWS + 16 + 4 * j + q) * CW + r] = acc1[q];
    }
    __syncthreads();
    if (a.lstm_c) {
        // gates of this workgroup's 4 units for every row -> LDS, then one thread per (row, unit) finishes the cell
        {
            const int m = tid / CW, c = tid - m * CW;
            const int nn = blockIdx.x * CW + c;
            float s = 0.f;
#pragma unroll
            for (int p = 0; p < NWV; ++p) s += red[(p * ROWS + m) * CW + c];
            if (a.bias && nn < a.N) s += a.bias[nn];
            gl[tid] = s;
        }
        __syncthreads();
        if (tid < ROWS * 4) {
            const int m = tid >> 2, jj = tid & 3;
            const int u = blockIdx.x * 4 + jj;
            if (m < rows && u < a.lstm_H) {
                const float* g = gl + m * CW;
                const float gi = rg_sigmoid(g[jj]), gf = rg_sigmoid(g[4 + jj]), gg = tanhf(g[8 + jj]), go = rg_sigmoid(g[12 + jj]);
                float* cp = a.lstm_c + (long)(m0 + m) * a.lstm_H + u;
                const float cn = gf * *cp + gi * gg;
                const float h = go * tanhf(cn);
                *cp = cn;
                a.lstm_h1[(long)(m0 + m) * a.lstm_ld1 + u] = h;
                a.lstm_h2[(long)(m0 + m) * a.lstm_ld2 + u] = h;
            }
        }
        return;
    }
    static_assert(ROWS * CW == 512, "one output element per thread");
    {
        const int m = tid / CW, c = tid - m * CW;
        const int nn = blockIdx.x * CW + c;
        if (m >= rows || nn >= a.N) return;
        float s = 0.f;
#pragma unroll
        for (int p = 0; p < NWV; ++p) s += red[(p * ROWS + m) * CW + c];
        if (a.bias) s += e_bias;
        if (a.act == PK_ACT_RELU) s = fmaxf(s, 0.f);
        if (a.dropout) {
            const unsigned long long eidx = (a.drop_base * (unsigned long long)a.drop_J + (unsigned long long)a.drop_j) *
                                                (unsigned long long)a.N + (unsigned long long)nn;
            unsigned w4[4];
            pk_dropout_words(eidx & ~3ull, a.drop_seeds ? a.drop_seeds[m0 + m] : 0ull, w4);
            s = w4[eidx & 3ull] >= a.drop_thr ? s * a.drop_scale : 0.f;
        }
        if (a.res) s += e_res;
        a.y[(long)(m0 + m) * a.ldy + nn] = s;
    }
}
}  // namespace

int pk_rowgemm_launch(pk_ctx* ctx, const char* prof_name, const pk_rowgemm_args& a) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) PK_FAIL(PK_EINVAL, "row GEMM: empty problem");
    if (a.K % 4 != 0 || a.ldx % 4 != 0)
        PK_FAIL(PK_EUNSUPPORTED, "row GEMM: K (%d) and ldx (%d) must be multiples of 4", a.K, a.ldx);
    if (a.ln_g && (a.K > KC || !a.ln_b)) PK_FAIL(PK_EUNSUPPORTED, "row GEMM: the LayerNorm prologue needs K <= %d", KC);
    if (a.act != PK_ACT_NONE && a.act != PK_ACT_RELU) PK_FAIL(PK_EUNSUPPORTED, "row GEMM: activation %d", a.act);
    if (a.lstm_c && (a.N != 4 * a.lstm_H || a.lstm_H % 4 != 0 || a.act != PK_ACT_NONE || a.dropout || a.res || !a.lstm_h1 ||
                     !a.lstm_h2))
        PK_FAIL(PK_EINVAL, "row GEMM: LSTM epilogue needs N == 4 * H, H %% 4 == 0, two h destinations and no act / dropout / res");
    const int cw = pk_rowgemm_cw(a.N);
    if (a.stop_w && (a.M > ROWS || (a.ln_g && a.K > 1024) || !a.stop_probs || !a.stop_len || !a.stop_ndone ||
                     (a.stop_kind == 0 && (!a.stop_minlen || !a.stop_maxlen)) || (a.stop_kind == 1 && a.ln_g) || a.stop_kind < 0 || a.stop_kind > 1))
        PK_FAIL(PK_EINVAL, "row GEMM: the stop-token head needs M <= %d, K <= 1024 under LayerNorm, and its arrays", ROWS);
    dim3 grid((a.N + cw - 1) / cw + (a.stop_w ? (a.M + 7) / 8 : 0), (a.M + ROWS - 1) / ROWS);
    if (a.ln_g) PK_LAUNCH(ctx, prof_name, k_rowgemm<true>, grid, dim3(512), 0, a);
    else PK_LAUNCH(ctx, prof_name, k_rowgemm<false>, grid, dim3(512), 0, a);
    return PK_OK;
}
