// ops.hip -- generic primitives of parakeet/modules exposed on the engine's kernels:
//   sinusoid_position_encoding   parakeet/modules/positional_encoding.py:20-39
//   scaled_dot_product_attention parakeet/modules/attention.py:22-58 (float mask, returns the weights)
//   Conv1dBatchNorm.forward      parakeet/modules/conv.py:186-260 (eval mode, NLC layout)
// Not on the FastSpeech2/PWG/WaveFlow path; they serve the other models' inference code.
#include <cmath>
#include <vector>

#include "pk_gemm.h"
#include "pk_philox.h"

namespace {

// One thread per Philox block = 4 normals (see pk_synth.h for the exact recipe).
__global__ void k_randn(float* __restrict__ out, long n, unsigned long long seed, unsigned long long block0) {
    const long blk = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (blk * 4 >= n) return;
    const unsigned long long ctr = block0 + (unsigned long long)blk;
    unsigned c0 = (unsigned)ctr, c1 = (unsigned)(ctr >> 32), c2 = 0u, c3 = 0u;
    philox4x32_10(c0, c1, c2, c3, (unsigned)seed, (unsigned)(seed >> 32));
    const float two_m32 = 2.3283064365386963e-10f;
    float z[4];
    {
        const float u1 = ((float)c0 + 1.0f) * two_m32, u2 = (float)c1 * two_m32;
        const float r = sqrtf(-2.0f * logf(u1));
        float sn, cs;
        sincosf(6.283185307179586f * u2, &sn, &cs);
        z[0] = r * cs;
        z[1] = r * sn;
    }
    {
        const float u1 = ((float)c2 + 1.0f) * two_m32, u2 = (float)c3 * two_m32;
        const float r = sqrtf(-2.0f * logf(u1));
        float sn, cs;
        sincosf(6.283185307179586f * u2, &sn, &cs);
        z[2] = r * cs;
        z[3] = r * sn;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (blk * 4 + e < n) out[blk * 4 + e] = z[e];
}

// out[r] = src[r] >= 0 ? enc[src[r]] : 0   (row gather of the expansion / length regulator)
__global__ __launch_bounds__(128) void k_gather_rows(const float* __restrict__ enc, const int* __restrict__ src, int C,
                                                     float* __restrict__ out) {
    const long r = blockIdx.x;
    const int s = src[r];
    for (int c = threadIdx.x; c < C; c += blockDim.x) out[r * C + c] = s >= 0 ? enc[(long)s * C + c] : 0.f;
}

// enc[pos][2i] = sin(p), enc[pos][2i+1] = cos(p), p = (start + pos) * omega / 10000^(2i / size)
__global__ void k_sinusoid(float* __restrict__ out, int num_positions, int size, float omega, int start_pos) {
    const int pos = blockIdx.x;
    for (int c = threadIdx.x; c < size; c += blockDim.x) {
        const float channel = (float)(c & ~1);
        const float p = ((float)(start_pos + pos) * omega) / powf(10000.0f, channel / (float)size);
        out[(long)pos * size + c] = (c & 1) ? cosf(p) : sinf(p);
    }
}

// One wave per (batch, query): weights = softmax(q.k^T / sqrt(d) + (1 - mask) * -1e9), out = weights . v
// mask: float, strides given so that (B,1,Tk), (B,Tq,Tk) or (1,Tq,Tk) broadcast.
__global__ __launch_bounds__(256) void k_sdpa(const float* __restrict__ q, const float* __restrict__ k,
                                              const float* __restrict__ v, const float* __restrict__ mask,
                                              long mask_sb, long mask_sq, int B, int Tq, int Tk, int d, int dv,
                                              float* __restrict__ out, float* __restrict__ weights) {
    extern __shared__ float sh[];   // per wave: d floats of q + Tk floats of scores
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + wave;
    if (row >= (long)B * Tq) return;
    const int b = (int)(row / Tq), tq = (int)(row % Tq);
    float* qs = sh + (size_t)wave * (d + Tk);
    float* sc = qs + d;
    const float* qp = q + row * d;
    for (int c = lane; c < d; c += 64) qs[c] = qp[c];
    // [wave-lds-exchange] qs[] is read below by every lane of the wave (wave-private LDS: no workgroup barrier needed)
    const float scale = 1.0f / sqrtf((float)d);
    float mx = -INFINITY;
    for (int j = lane; j < Tk; j += 64) {
        const float* kp = k + ((long)b * Tk + j) * d;
        float s = 0.f;
        for (int c = 0; c < d; ++c) s = fmaf(qs[c], kp[c], s);
        s *= scale;
        if (mask) s += (1.0f - mask[b * mask_sb + tq * mask_sq + j]) * -1e9f;
        sc[j] = s;
        mx = fmaxf(mx, s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int j = lane; j < Tk; j += 64) {
        const float e = expf(sc[j] - mx);
        sc[j] = e;
        sum += e;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.0f / sum;
    for (int j = lane; j < Tk; j += 64) {
        sc[j] *= inv;
        if (weights) weights[row * Tk + j] = sc[j];
    }
    // [wave-lds-exchange] sc[] (one key per lane) is read below by every lane of the wave
    for (int c = lane; c < dv; c += 64) {
        float acc = 0.f;
        for (int j = 0; j < Tk; ++j) acc = fmaf(sc[j], v[((long)b * Tk + j) * dv + c], acc);
        out[row * dv + c] = acc;
    }
}

// Conv1dCell.add_input (modules/conv.py:166-183), two launches:
//   k_cell_shift: buffer[b][ci][:] <- concat(buffer[b][ci][1:], x_t[b][ci])   (update_buffer :141-151); one thread per (b, ci)
//   k_cell_out:   y[b][co] = bias[co] + sum_ci sum_j W[co][ci][j] * buffer[b][ci][j * dilation]   (:176-182); one wave per (b, co)
__global__ __launch_bounds__(256) void k_cell_shift(float* __restrict__ buf, const float* __restrict__ x, int rows, int r) {
    const int q = blockIdx.x * 256 + threadIdx.x;   // (b, ci)
    if (q >= rows) return;
    float* p = buf + (long)q * r;
    for (int j = 0; j + 1 < r; ++j) p[j] = p[j + 1];
    p[r - 1] = x[q];
}
__global__ __launch_bounds__(256) void k_cell_out(const float* __restrict__ buf, const float* __restrict__ W,
                                                  const float* __restrict__ bias, int B, int Cin, int Cout, int k, int dil,
                                                  int r, float* __restrict__ y) {
    const long o = (long)blockIdx.x * 4 + (threadIdx.x >> 6);   // (b, co)
    const int lane = threadIdx.x & 63;
    if (o >= (long)B * Cout) return;
    const int b = (int)(o / Cout), co = (int)(o - (long)b * Cout);
    const float* w = W + (long)co * Cin * k;
    const float* x = buf + (long)b * Cin * r;
    float s = 0.f;
    for (int e = lane; e < Cin * k; e += 64) {
        const int ci = e / k, j = e - ci * k;
        s = fmaf(w[e], x[(long)ci * r + j * dil], s);
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) s += __shfl_xor(s, m);
    if (lane == 0) y[o] = s + (bias ? bias[co] : 0.f);
}

// rows of a (B, T, C) NLC tensor -> row timeline with `gap` zero rows around every sequence
__global__ void k_nlc_to_timeline(const float* __restrict__ x, int T, int C, int gap, float* __restrict__ tl) {
    const int r = blockIdx.x;                       // timeline row
    const int per = T + gap;
    const int b = (r - gap) / per, t = (r - gap) - b * per;
    const bool valid = r >= gap && t < T;
    for (int c = threadIdx.x; c < C; c += blockDim.x)
        tl[(long)r * C + c] = valid ? x[((long)b * T + t) * C + c] : 0.f;
}

// copy an [M][K] row-major matrix into [rows_alloc][Kp] with zero padding (columns K..Kp, rows M..rows_alloc)
__global__ void k_pad_rows(const float* __restrict__ x, int M, int K, int Kp, float* __restrict__ y) {
    const int r = blockIdx.x;
    for (int c = threadIdx.x; c < Kp; c += blockDim.x)
        y[(long)r * Kp + c] = (r < M && c < K) ? x[(long)r * K + c] : 0.f;
}

}  // namespace

int pk_randn_device(pk_ctx* ctx, float* d_out, long n, unsigned long long seed, unsigned long long offset) {
    if (offset & 3) PK_FAIL(PK_EINVAL, "pk_randn: offset must be a multiple of 4");
    if (n <= 0) return PK_OK;
    const long blocks = (n + 3) / 4;
    PK_LAUNCH(ctx, "randn", k_randn, dim3((unsigned)pk_div_up(blocks, 256)), dim3(256), 0, d_out, n, seed, offset / 4);
    return PK_OK;
}

extern "C" int pk_randn(pk_ctx* ctx, float* out, int64_t n, uint64_t seed, uint64_t offset, int32_t flags) {
    if (!ctx || (!out && n > 0)) PK_FAIL(PK_EINVAL, "pk_randn: NULL argument");
    if (n < 0) PK_FAIL(PK_EINVAL, "pk_randn: n must be >= 0");
    PK_DEVICE(ctx->device);
    if (!(flags & PK_HOST_IO)) return pk_randn_device(ctx, out, n, seed, offset);
    float* d = nullptr;
    if (n == 0) return PK_OK;
    PK_HIP(hipMalloc(&d, (size_t)n * 4));
    int rc = pk_randn_device(ctx, d, n, seed, offset);
    if (rc == PK_OK) {
        hipError_t e = hipMemcpyAsync(out, d, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { (void)hipFree(d); PK_FAIL(PK_EHIP, "pk_randn: %s", hipGetErrorString(e)); }
    }
    (void)hipFree(d);
    return rc;
}

extern "C" int pk_op_expand(pk_ctx* ctx, const float* encodings, const int64_t* durations, int32_t B, int32_t T,
                            int32_t C, int32_t t_dec, float* out) {
    if (!ctx || !encodings || !durations || (!out && t_dec > 0)) PK_FAIL(PK_EINVAL, "pk_op_expand: NULL argument");
    if (B <= 0 || T <= 0 || C <= 0 || t_dec < 0) PK_FAIL(PK_EINVAL, "expand: bad shape");
    PK_DEVICE(ctx->device);
    std::vector<int> src((size_t)B * t_dec, -1);
    for (int b = 0; b < B; ++b) {
        long k = 0;
        for (int t = 0; t < T; ++t) {
            const int64_t d = durations[(size_t)b * T + t];
            if (d < 0) PK_FAIL(PK_EINVAL, "expand: negative duration at (%d, %d)", b, t);
            if (k + d > t_dec) PK_FAIL(PK_ESHAPE, "expand: utterance %d needs more than t_dec = %d frames", b, t_dec);
            for (int64_t i = 0; i < d; ++i) src[(size_t)b * t_dec + k + i] = b * T + t;
            k += d;
        }
    }
    if (t_dec == 0) return PK_OK;
    pk_dbuf d_src;
    int rc = pk_upload(ctx, d_src, src.data(), src.size() * sizeof(int));
    if (rc == PK_OK) {
        hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((size_t)B * t_dec)), dim3(128), 0, ctx->stream, encodings,
                           d_src.as<int>(), C, out);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);   // d_src is released below
        if (e != hipSuccess) { d_src.release(); PK_FAIL(PK_EHIP, "pk_op_expand: %s", hipGetErrorString(e)); }
    }
    d_src.release();
    return rc;
}

extern "C" int pk_op_sinusoid_position_encoding(pk_ctx* ctx, int32_t num_positions, int32_t feature_size,
                                                float omega, int32_t start_pos, float* out) {
    if (!ctx || !out) PK_FAIL(PK_EINVAL, "pk_op_sinusoid_position_encoding: NULL argument");
    if (num_positions <= 0 || feature_size <= 0) PK_FAIL(PK_EINVAL, "sinusoid_position_encoding: empty table");
    PK_DEVICE(ctx->device);
    PK_LAUNCH(ctx, "op_sinusoid", k_sinusoid, dim3(num_positions), dim3(128), 0, out, num_positions, feature_size,
              omega, start_pos);
    return PK_OK;
}

extern "C" int pk_op_scaled_dot_product_attention(pk_ctx* ctx, const float* q, const float* k, const float* v,
                                                  const float* mask, int32_t mask_mode, int32_t B, int32_t Tq,
                                                  int32_t Tk, int32_t d, int32_t dv, float* out, float* weights) {
    if (!ctx || !q || !k || !v || !out) PK_FAIL(PK_EINVAL, "pk_op_scaled_dot_product_attention: NULL argument");
    if (B <= 0 || Tq <= 0 || Tk <= 0 || d <= 0 || dv <= 0) PK_FAIL(PK_EINVAL, "attention: empty problem");
    PK_DEVICE(ctx->device);
    long sb = 0, sq = 0;
    switch (mask ? mask_mode : -1) {
        case -1: break;
        case 0: sb = Tk; sq = 0; break;                  // (B, 1, Tk)
        case 1: sb = (long)Tq * Tk; sq = Tk; break;      // (B, Tq, Tk)
        case 2: sb = 0; sq = Tk; break;                  // (1, Tq, Tk)
        default: PK_FAIL(PK_EINVAL, "attention: unknown mask mode %d", mask_mode);
    }
    const size_t shmem = (size_t)4 * (d + Tk) * sizeof(float);
    if (shmem > 64 * 1024) PK_FAIL(PK_EUNSUPPORTED, "attention: d + Tk = %d too large for this primitive", d + Tk);
    const long rows = (long)B * Tq;
    PK_LAUNCH(ctx, "op_sdpa", k_sdpa, dim3(pk_div_up(rows, 4)), dim3(256), shmem, q, k, v, mask, sb, sq, B, Tq, Tk,
              d, dv, out, weights);
    return PK_OK;
}

// y = BatchNorm1D_eval(Conv1D(x)) for NLC x (B, T, Cin), stride 1, symmetric padding `pad`
// (T_out = T + 2*pad - k + 1).  bn_* may be NULL (plain conv).  Weights are packed per call.
extern "C" int pk_op_conv1d_batchnorm_nlc(pk_ctx* ctx, const float* x, int32_t B, int32_t T, int32_t Cin,
                                          int32_t Cout, int32_t k, int32_t pad, const float* weight,
                                          const float* bias, const float* bn_weight, const float* bn_bias,
                                          const float* bn_mean, const float* bn_var, float eps, float* y) {
    if (!ctx || !x || !weight || !y) PK_FAIL(PK_EINVAL, "pk_op_conv1d_batchnorm_nlc: NULL argument");
    if (B <= 0 || T <= 0 || k <= 0 || pad < 0 || k > PK_GEMM_MAX_TAPS) PK_FAIL(PK_EINVAL, "conv1d: bad shape");
    if (Cin % PK_GEMM_BK != 0) PK_FAIL(PK_EUNSUPPORTED, "conv1d: in_channels must be a multiple of %d", PK_GEMM_BK);
    const int Tout = T + 2 * pad - k + 1;
    if (Tout <= 0) PK_FAIL(PK_ESHAPE, "conv1d: kernel larger than padded input");
    PK_DEVICE(ctx->device);
    // fold BN: w' = w * g / sqrt(var + eps), b' = (b - mean) * g / sqrt(var + eps) + beta
    std::vector<float> w((size_t)Cout * Cin * k), b(Cout, 0.f), kn, packed;
    for (int o = 0; o < Cout; ++o) {
        double s = 1.0, sh = bias ? bias[o] : 0.0;
        if (bn_weight) {
            s = (double)bn_weight[o] / std::sqrt((double)bn_var[o] + eps);
            sh = (sh - bn_mean[o]) * s + bn_bias[o];
        }
        for (size_t i = 0; i < (size_t)Cin * k; ++i) w[o * (size_t)Cin * k + i] = (float)(weight[o * (size_t)Cin * k + i] * s);
        b[o] = (float)sh;
    }
    pk_conv_to_kn(w.data(), Cout, Cin, k, kn);
    pk_gemm_pack(kn.data(), Cin * k, Cout, packed);
    const int gap = k;   // >= both paddings
    const int rows = gap + B * (T + gap);
    const int rows_alloc = ((rows + PK_GEMM_BM - 1) / PK_GEMM_BM) * PK_GEMM_BM + 2 * gap;
    pk_dbuf d_w, d_b, d_tl, d_map;
    int st = PK_OK;
    auto cleanup = [&]() { d_w.release(); d_b.release(); d_tl.release(); d_map.release(); };
    std::vector<int> rowmap(rows_alloc, -1);
    // output row (b, t) = timeline row gap + b*(T+gap) + t - pad + (k-1)/2 ... expressed through the tap offsets below
    for (int bb = 0; bb < B; ++bb)
        for (int t = 0; t < Tout; ++t) rowmap[gap + bb * (T + gap) + t] = bb * Tout + t;
    if ((st = pk_upload(ctx, d_w, packed.data(), packed.size() * 4)) != PK_OK ||
        (st = pk_upload(ctx, d_b, b.data(), b.size() * 4)) != PK_OK ||
        (st = pk_upload(ctx, d_map, rowmap.data(), rowmap.size() * 4)) != PK_OK ||
        (st = d_tl.reserve((size_t)(rows_alloc + gap) * Cin * 4)) != PK_OK) {
        cleanup();
        return st;
    }
    float* tl = d_tl.as<float>() + (size_t)gap * Cin;
    hipLaunchKernelGGL(k_nlc_to_timeline, dim3(rows), dim3(128), 0, ctx->stream, x, T, Cin, gap, tl);
    pk_gemm_args g;
    g.A = tl;
    g.lda = Cin;
    g.Cin = Cin;
    g.ntaps = k;
    for (int t = 0; t < k; ++t) {       // output t reads inputs t - pad + tap
        g.tap_off[t] = (long)(t - pad) * Cin;
        g.tap_w[t] = t;
    }
    g.wslabs_total = Cin * k / PK_GEMM_BK;
    g.Wp = d_w.as<float>();
    g.bias = d_b.as<float>();
    g.C = y;
    g.ldc = Cout;
    g.out_rowmap = d_map.as<int>();
    g.M = rows;
    g.N = Cout;
    st = pk_gemm_launch(ctx, "op_conv1d_bn", g);
    if (st == PK_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) {
        pk_set_error("conv1d: stream sync failed");
        st = PK_EHIP;
    }
    cleanup();
    return st;
}

// y[M][N] = x[M][K] . w[K][N] (+ bias[N]) on the exact-fp32 MFMA GEMM; K is zero-padded to the kernel's slab.
extern "C" int pk_op_matmul(pk_ctx* ctx, const float* x, int32_t M, int32_t K, int32_t N, const float* w,
                            const float* bias, float* y) {
    if (!ctx || !x || !w || !y) PK_FAIL(PK_EINVAL, "pk_op_matmul: NULL argument");
    if (M <= 0 || K <= 0 || N <= 0) PK_FAIL(PK_EINVAL, "pk_op_matmul: bad shape");
    PK_DEVICE(ctx->device);
    const int Kp = ((K + PK_GEMM_BK - 1) / PK_GEMM_BK) * PK_GEMM_BK;
    const int rows_alloc = ((M + PK_GEMM_BM - 1) / PK_GEMM_BM) * PK_GEMM_BM;
    std::vector<float> kn((size_t)Kp * N, 0.f), packed;
    memcpy(kn.data(), w, (size_t)K * N * sizeof(float));
    pk_gemm_pack(kn.data(), Kp, N, packed);
    pk_dbuf d_w, d_b, d_x;
    int st = PK_OK;
    auto cleanup = [&]() { d_w.release(); d_b.release(); d_x.release(); };
    if ((st = pk_upload(ctx, d_w, packed.data(), packed.size() * 4)) != PK_OK ||
        (bias && (st = pk_upload(ctx, d_b, bias, (size_t)N * 4)) != PK_OK) ||
        (st = d_x.reserve((size_t)rows_alloc * Kp * 4)) != PK_OK) {
        cleanup();
        return st;
    }
    hipLaunchKernelGGL(k_pad_rows, dim3(rows_alloc), dim3(128), 0, ctx->stream, x, M, K, Kp, d_x.as<float>());
    pk_gemm_args g;
    g.A = d_x.as<float>();
    g.lda = Kp;
    g.Wp = d_w.as<float>();
    g.bias = bias ? d_b.as<float>() : nullptr;
    g.C = y;
    g.ldc = N;
    g.M = M;
    g.N = N;
    g.Cin = Kp;
    g.taps = 1;
    g.pad = 0;
    st = pk_gemm_launch(ctx, "op_matmul", g);
    if (st == PK_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) {
        pk_set_error("pk_op_matmul: stream sync failed");
        st = PK_EHIP;
    }
    cleanup();
    return st;
}

extern "C" int pk_op_conv1d_cell_step(pk_ctx* ctx, float* buffer, const float* x_t, const float* weight, const float* bias,
                                      int32_t B, int32_t Cin, int32_t Cout, int32_t k, int32_t dilation, float* y) {
    if (!ctx || !x_t || !weight || !y) PK_FAIL(PK_EINVAL, "pk_op_conv1d_cell_step: NULL argument");
    if (B <= 0 || Cin <= 0 || Cout <= 0 || k <= 0 || dilation <= 0) PK_FAIL(PK_EINVAL, "pk_op_conv1d_cell_step: bad shape");
    const int r = 1 + (k - 1) * dilation;   // receptive field (:84)
    if (r > 1 && !buffer) PK_FAIL(PK_EINVAL, "pk_op_conv1d_cell_step: a receptive field of %d needs the buffer", r);
    PK_DEVICE(ctx->device);
    const float* in = x_t;   // receptive field 1: the step input itself (:178-179)
    if (r > 1) {
        PK_LAUNCH(ctx, "cell_shift", k_cell_shift, dim3(pk_div_up((long)B * Cin, 256)), dim3(256), 0, buffer, x_t, B * Cin, r);
        in = buffer;
    }
    PK_LAUNCH(ctx, "cell_out", k_cell_out, dim3(pk_div_up((long)B * Cout, 4)), dim3(256), 0, in, weight, bias, B, Cin, Cout, k,
              dilation, r, y);
    return PK_OK;
}
