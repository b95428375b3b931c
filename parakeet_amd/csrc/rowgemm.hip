// rowgemm.hip -- see pk_rowgemm.h.
#include "pk_rowgemm.h"

#include "pk_gemm.h"
#include "pk_philox.h"

void pk_rowgemm_pack(const float* Wkn, int K, int N, std::vector<float>& out) {
    const int cw = pk_rowgemm_cw(N), nb = (N + cw - 1) / cw;
    out.assign((size_t)nb * K * cw, 0.f);
    for (int b = 0; b < nb; ++b)
        for (int k = 0; k < K; ++k)
            for (int c = 0; c < cw && b * cw + c < N; ++c)
                out[((size_t)b * K + k) * cw + c] = Wkn[(size_t)k * N + b * cw + c];
}

void pk_rowgemm_lstm_perm(int H, std::vector<int>& perm) {
    perm.resize((size_t)4 * H);
    for (int u = 0; u < H; ++u)
        for (int g = 0; g < 4; ++g) perm[(size_t)(u / 4) * 16 + g * 4 + (u % 4)] = g * H + u;
}

namespace {
constexpr int KC = PK_RG_KC, ROWS = PK_RG_ROWS;

__device__ __forceinline__ float rg_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

// CW columns per workgroup, KSUB = 64 / CW consecutive k per wave-wide load, 8 * KSUB K-parts per workgroup.
//
// Latency is everything here (32 rows: a launch is 10 - 40 MFLOP), so the kernel makes ONE trip to memory before it
// computes: the first chunk's weights (16 loads per lane), the activation tile (8 float4 per thread), the epilogue's bias and
// residual are all requested up front, unconditionally, with clamped indices (round 4; rounds 2 / 3 read the rows a first time
// for the LayerNorm statistics, four rows per wave one after the other, before anything else was requested: five dependent
// round trips, 12.5 us per launch in the TransformerTTS decoder against 2 us of work).  A thread's 8 float4 all belong to ONE
// row (m = tid % 32; 512 % 32 == 0), 32 of its K <= 512 values, so the LayerNorm statistics come from the registers that will
// be staged anyway: per thread mean and M2 of its 32 values, the row's 16 partials merged through LDS by Chan's formula
// (equal counts: mean = avg(mean_i), M2 = sum M2_i + n sum (mean_i - mean)^2 -- two-pass accuracy without a second pass).
template <int CW>
__global__ __launch_bounds__(512) void k_rowgemm(pk_rowgemm_args a) {
    constexpr int KSUB = 64 / CW, PARTS = 8 * KSUB;
    __shared__ __attribute__((aligned(16))) float xs[KC * ROWS];   // xs[k * 32 + m], 64 KB
    __shared__ float red[PARTS * ROWS * CW];                       // red[(part * 32 + m) * CW + col], 64 KB
    __shared__ float stat[2 * 16 * ROWS];                          // LayerNorm: (mean, M2) of the 16 partials of every row
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform by construction: keep it in an SGPR
    const int col = lane % CW, ks = lane / CW;
    const int part = wave * KSUB + ks;                            // this lane's K-part: k = part, part + PARTS, ...
    const int m0 = blockIdx.y * ROWS;
    const int rows = min(ROWS, a.M - m0);
    // A lane's k of a chunk: part, part + PARTS, ... -- at most KC / PARTS = 16 of them.
    constexpr int WG = KC / PARTS;
    constexpr int XR = (KC / 4) * ROWS / 512;   // float4 per thread per chunk
    static_assert(WG <= 16 && XR == 8, "register budget");
    const float* slab = a.Wt + (long)blockIdx.x * a.K * CW;   // this workgroup's [K][CW] slab
    const int xm = tid & (ROWS - 1), xp = tid >> 5;            // this thread's row of the tile and its K-part of it (16 parts)
    float wn[WG];
    float4 xn[XR];
    // the epilogue's operands of this thread's output element (e = tid: row e / CW, column e % CW), requested now
    const int em = min(tid / CW, rows - 1), ec = tid % CW, en = min((int)blockIdx.x * CW + ec, a.N - 1);
    float e_bias = 0.f, e_res = 0.f;
    {
        const int kc = min(KC, a.K);
#pragma unroll
        for (int g = 0; g < WG; ++g) wn[g] = slab[(long)min(part + PARTS * g, kc - 1) * CW + col];
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            const int k4 = min(xp + 16 * i, (kc >> 2) - 1);
            xn[i] = *reinterpret_cast<const float4*>(a.x + (long)(m0 + min(xm, rows - 1)) * a.ldx + 4 * k4);
        }
        if (a.bias) e_bias = a.bias[en];
        if (a.res) e_res = a.res[(long)(m0 + em) * a.ldr + en];
    }
    float ln_mean = 0.f, ln_rstd = 1.f;
    if (a.ln_g) {   // (uniform) K <= KC: the whole row is in the 16 threads' registers
        const int n4 = a.K >> 2;           // float4 per row
        float s = 0.f;
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < XR; ++i)
            if (xp + 16 * i < n4) {
                s += (xn[i].x + xn[i].y) + (xn[i].z + xn[i].w);
                cnt += 4;
            }
        const float mu = cnt ? s / (float)cnt : 0.f;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < XR; ++i)
            if (xp + 16 * i < n4) {
                const float d0 = xn[i].x - mu, d1 = xn[i].y - mu, d2 = xn[i].z - mu, d3 = xn[i].w - mu;
                q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
        stat[(xp * ROWS + xm) * 2] = mu;
        stat[(xp * ROWS + xm) * 2 + 1] = q;
        __syncthreads();
        // merge the row's 16 partials (counts differ only when K is not a multiple of 64: weight them)
        float tot = 0.f, msum = 0.f;
#pragma unroll
        for (int pp = 0; pp < 16; ++pp) {
            const int c4 = (n4 - pp + 15) / 16;   // float4 the partial pp holds
            const float w = (float)(4 * max(c4, 0));
            msum += w * stat[(pp * ROWS + xm) * 2];
            tot += w;
        }
        ln_mean = msum / tot;
        float m2 = 0.f;
#pragma unroll
        for (int pp = 0; pp < 16; ++pp) {
            const int c4 = (n4 - pp + 15) / 16;
            const float w = (float)(4 * max(c4, 0));
            const float d = stat[(pp * ROWS + xm) * 2] - ln_mean;
            m2 += stat[(pp * ROWS + xm) * 2 + 1] + w * d * d;
        }
        ln_rstd = 1.0f / sqrtf(m2 / tot + a.ln_eps);
    }
    float acc[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < a.K; k0 += KC) {
        const int kc = min(KC, a.K - k0);
        float w[WG];
#pragma unroll
        for (int g = 0; g < WG; ++g) w[g] = wn[g];
        if (k0 > 0) __syncthreads();   // the previous chunk is consumed
        // stage x[m0 + m][k0 + 4 * k4 ..] -> xs[(4 * k4 + i) * 32 + m]
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            const int k4 = xp + 16 * i;
            if (k4 < (kc >> 2)) {
                float4 v = xn[i];
                if (xm >= rows) v = make_float4(0.f, 0.f, 0.f, 0.f);
                else if (a.ln_g) {
                    const float4 g = *reinterpret_cast<const float4*>(a.ln_g + k0 + 4 * k4);
                    const float4 bb = *reinterpret_cast<const float4*>(a.ln_b + k0 + 4 * k4);
                    v.x = (v.x - ln_mean) * ln_rstd * g.x + bb.x;
                    v.y = (v.y - ln_mean) * ln_rstd * g.y + bb.y;
                    v.z = (v.z - ln_mean) * ln_rstd * g.z + bb.z;
                    v.w = (v.w - ln_mean) * ln_rstd * g.w + bb.w;
                }
                float* d = xs + (4 * k4) * ROWS + xm;
                d[0] = v.x;
                d[ROWS] = v.y;
                d[2 * ROWS] = v.z;
                d[3 * ROWS] = v.w;
            }
        }
        __syncthreads();
        {
            // the next chunk's operands; when there is none the clamps fold every load onto one cache line
            const bool has_next = k0 + KC < a.K;
            const int k0n = has_next ? k0 + KC : 0;
            const int kcn = has_next ? min(KC, a.K - k0n) : 4;
            const int mlim = has_next ? rows - 1 : 0;
#pragma unroll
            for (int g = 0; g < WG; ++g) wn[g] = slab[(long)(k0n + min(part + PARTS * g, kcn - 1)) * CW + col];
#pragma unroll
            for (int i = 0; i < XR; ++i) {
                const int k4 = min(xp + 16 * i, (kcn >> 2) - 1);
                xn[i] = *reinterpret_cast<const float4*>(a.x + (long)(m0 + min(xm, mlim)) * a.ldx + k0n + 4 * k4);
            }
        }
#pragma unroll
        for (int g = 0; g < WG; ++g) {
            const int kk = part + PARTS * g;
            const int k = min(kk, kc - 1);
            const float wv = kk < kc ? w[g] : 0.f;
            const float4* xr = reinterpret_cast<const float4*>(xs + k * ROWS);
#pragma unroll
            for (int q = 0; q < ROWS / 4; ++q) {
                const float4 xv = xr[q];
                acc[4 * q + 0] = fmaf(xv.x, wv, acc[4 * q + 0]);
                acc[4 * q + 1] = fmaf(xv.y, wv, acc[4 * q + 1]);
                acc[4 * q + 2] = fmaf(xv.z, wv, acc[4 * q + 2]);
                acc[4 * q + 3] = fmaf(xv.w, wv, acc[4 * q + 3]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) red[(part * ROWS + r) * CW + col] = acc[r];
    __syncthreads();
    if (a.lstm_c) {
        // gates of this workgroup's 4 units for every row -> LDS, then one thread per (row, unit) finishes the cell
        float* gl = xs;   // the activations are consumed: [32][16] gate pre-activations
        for (int e = tid; e < ROWS * CW; e += 512) {
            const int m = e / CW, c = e - m * CW;
            const int nn = blockIdx.x * CW + c;
            float s = 0.f;
#pragma unroll 8
            for (int p = 0; p < PARTS; ++p) s += red[(p * ROWS + m) * CW + c];
            if (a.bias && nn < a.N) s += a.bias[nn];
            gl[e] = s;
        }
        __syncthreads();
        if (tid < ROWS * 4) {
            const int m = tid >> 2, j = tid & 3;
            const int u = blockIdx.x * 4 + j;
            if (m < rows && u < a.lstm_H) {
                const float* g = gl + m * CW;
                const float gi = rg_sigmoid(g[j]), gf = rg_sigmoid(g[4 + j]), gg = tanhf(g[8 + j]), go = rg_sigmoid(g[12 + j]);
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
        const int e = tid;
        const int m = e / CW, c = e - m * CW;
        const int nn = blockIdx.x * CW + c;
        if (m >= rows || nn >= a.N) return;
        float s = 0.f;
#pragma unroll 8
        for (int p = 0; p < PARTS; ++p) s += red[(p * ROWS + m) * CW + c];
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
    if (a.K % 8 != 0 || a.ldx % 4 != 0)
        PK_FAIL(PK_EUNSUPPORTED, "row GEMM: K (%d) must be a multiple of 8 and ldx (%d) of 4", a.K, a.ldx);
    if (a.ln_g && (a.K > KC || !a.ln_b)) PK_FAIL(PK_EUNSUPPORTED, "row GEMM: the LayerNorm prologue needs K <= %d", KC);
    if (a.act != PK_ACT_NONE && a.act != PK_ACT_RELU) PK_FAIL(PK_EUNSUPPORTED, "row GEMM: activation %d", a.act);
    if (a.lstm_c && (a.N != 4 * a.lstm_H || a.lstm_H % 4 != 0 || a.act != PK_ACT_NONE || a.dropout || a.res || !a.lstm_h1 ||
                     !a.lstm_h2))
        PK_FAIL(PK_EINVAL, "row GEMM: LSTM epilogue needs N == 4 * H, H %% 4 == 0, two h destinations and no act / dropout / res");
    const int cw = pk_rowgemm_cw(a.N);
    dim3 grid((a.N + cw - 1) / cw, (a.M + ROWS - 1) / ROWS);
    if (cw != 16) PK_FAIL(PK_EUNSUPPORTED, "row GEMM: only the 16-column tiling is built");
    PK_LAUNCH(ctx, prof_name, k_rowgemm<16>, grid, dim3(512), 0, a);
    return PK_OK;
}
