// gst.hip -- global style tokens (pk_gst.h): StyleEncoder.forward of parakeet/modules/style_encoder.py on gfx950.
//
//   ReferenceEncoder.forward :187-215   Conv2D(k, stride, no bias) -> BatchNorm2D -> ReLU stack on (1, L, idim), then
//                                       transpose / reshape to (L', C * F') and a GRU whose last hidden state is the
//                                       reference embedding
//   StyleTokenLayer.forward  :266-288   multi-head attention of that embedding (the one query) over tanh(gst_embs)
//
// This runs once per utterance before the decoder loop and is tiny next to it (~50 MFLOP for 800 frames): the kernels
// are plain fp32 FMA loops, one thread per output element, sized for clarity, not for a roofline.  The key / value
// projections of the style tokens do not depend on the input and are folded at finalize.
// GRU semantics [paddle-semantics, from Paddle's API documentation of GRUCell]: gate order r, z, c along the 3H axis;
//   r = sigmoid(W_ir x + b_ir + W_hr h + b_hr), z = sigmoid(W_iz x + b_iz + W_hz h + b_hz),
//   c = tanh(W_ic x + b_ic + r * (W_hc h + b_hc)), h' = z * h + (1 - z) * c.
#include "pk_gst.h"

#include <algorithm>
#include <cmath>
#include <string>

namespace {

// One Conv2D + folded BatchNorm2D + ReLU layer, channels-first per utterance ([C][T][F], utterances one after another).
// tab: [in_off B][out_off B][Tin B][Tout B].  grid (ceil(max outputs / 256), B).
__global__ __launch_bounds__(256) void k_gst_conv(const float* __restrict__ in, float* __restrict__ out,
                                                  const long* __restrict__ tab, int B, const float* __restrict__ W,
                                                  const float* __restrict__ bias, int Cin, int Cout, int Fin, int Fout,
                                                  int k, int stride, int pad) {
    const int b = blockIdx.y;
    const long in_off = tab[b], out_off = tab[B + b];
    const int Tin = (int)tab[2 * B + b], Tout = (int)tab[3 * B + b];
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)Cout * Tout * Fout) return;
    const int f = (int)(e % Fout);
    const long q = e / Fout;
    const int t = (int)(q % Tout), co = (int)(q / Tout);
    float acc = bias[co];
    for (int ci = 0; ci < Cin; ++ci) {
        const float* w = W + ((long)co * Cin + ci) * k * k;
        const float* x = in + in_off + (long)ci * Tin * Fin;
        for (int kt = 0; kt < k; ++kt) {
            const int tt = t * stride + kt - pad;
            if (tt < 0 || tt >= Tin) continue;
            for (int kf = 0; kf < k; ++kf) {
                const int ff = f * stride + kf - pad;
                if (ff < 0 || ff >= Fin) continue;
                acc = fmaf(w[kt * k + kf], x[(long)tt * Fin + ff], acc);
            }
        }
    }
    out[out_off + e] = fmaxf(acc, 0.f);
}

__device__ __forceinline__ float gst_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

// One GRU layer over the whole sequence of one utterance per workgroup.
//   convC > 0: the input is the conv stack's output [convC][T][convF] and x_t[c * convF + f] = in[(c * T + t) * convF + f]
//              (hs.transpose([0, 2, 1, 3]).reshape(B, T, -1), :203-207); else rows [T][In].
//   in_offs / seq_offs / Ts: per utterance offsets (floats) of its input and output sequence, and its length.
//   WihT [In][3H], WhhT [H][3H].  LDS: x[In] | h[H] | gx[3H] | gh[3H].
__global__ __launch_bounds__(256) void k_gst_gru(const float* __restrict__ in, const long* __restrict__ in_offs,
                                                 const long* __restrict__ seq_offs, const long* __restrict__ Ts, int convC,
                                                 int convF, int In, const float* __restrict__ WihT,
                                                 const float* __restrict__ WhhT, const float* __restrict__ bih,
                                                 const float* __restrict__ bhh, int H, float* __restrict__ seq,
                                                 float* __restrict__ last) {
    extern __shared__ float sm[];
    float* xs = sm;
    float* hs = xs + In;
    float* gx = hs + H;
    float* gh = gx + 3 * H;
    const int b = blockIdx.x, tid = threadIdx.x, G = 3 * H;
    const long in_off = in_offs[b], seq_off = seq_offs[b];
    const int T = (int)Ts[b];
    for (int u = tid; u < H; u += 256) hs[u] = 0.f;
    for (int t = 0; t < T; ++t) {
        for (int i = tid; i < In; i += 256) {
            long src;
            if (convC > 0) {
                const int c = i / convF, f = i - c * convF;
                src = ((long)c * T + t) * convF + f;
            } else {
                src = (long)t * In + i;
            }
            xs[i] = in[in_off + src];
        }
        __syncthreads();
        for (int g = tid; g < G; g += 256) {
            float a = bih[g], c = bhh[g];
            for (int i = 0; i < In; ++i) a = fmaf(WihT[(long)i * G + g], xs[i], a);
            for (int j = 0; j < H; ++j) c = fmaf(WhhT[(long)j * G + g], hs[j], c);
            gx[g] = a;
            gh[g] = c;
        }
        __syncthreads();
        for (int u = tid; u < H; u += 256) {
            const float r = gst_sigmoid(gx[u] + gh[u]);
            const float z = gst_sigmoid(gx[H + u] + gh[H + u]);
            const float c = tanhf(gx[2 * H + u] + r * gh[2 * H + u]);
            const float hn = z * hs[u] + (1.f - z) * c;
            hs[u] = hn;
            seq[seq_off + (long)t * H + u] = hn;
        }
        __syncthreads();
    }
    if (last)
        for (int u = tid; u < H; u += 256) last[(long)b * H + u] = hs[u];
}

// StyleTokenLayer: style[b] = linear_out(softmax_h(q_h . K_h^T / sqrt(dk)) . V_h), q = linear_q(ref[b]);
// K, V [Tk][A] are the projected tanh(gst_embs).  grid (B).  LDS: q[A] | p[heads * Tk] | ctx[A].
__global__ __launch_bounds__(256) void k_gst_style(const float* __restrict__ ref, int Hg, const float* __restrict__ Wq,
                                                   const float* __restrict__ bq, const float* __restrict__ K,
                                                   const float* __restrict__ V, int Tk, int heads, int A,
                                                   const float* __restrict__ Wo, const float* __restrict__ bo,
                                                   float* __restrict__ style) {
    extern __shared__ float sm[];
    float* q = sm;
    float* p = q + A;
    float* ctx = p + heads * Tk;
    const int b = blockIdx.x, tid = threadIdx.x, dk = A / heads;
    const float* r = ref + (long)b * Hg;
    for (int c = tid; c < A; c += 256) {
        float a = bq[c];
        for (int j = 0; j < Hg; ++j) a = fmaf(r[j], Wq[(long)j * A + c], a);
        q[c] = a;
    }
    __syncthreads();
    const float scale = 1.f / sqrtf((float)dk);
    for (int e = tid; e < heads * Tk; e += 256) {
        const int hd = e / Tk, t = e - hd * Tk;
        float s = 0.f;
        for (int d = 0; d < dk; ++d) s = fmaf(q[hd * dk + d], K[(long)t * A + hd * dk + d], s);
        p[e] = s * scale;
    }
    __syncthreads();
    if (tid < heads) {
        float* ph = p + tid * Tk;
        float m = -INFINITY, sum = 0.f;
        for (int t = 0; t < Tk; ++t) m = fmaxf(m, ph[t]);
        for (int t = 0; t < Tk; ++t) {
            ph[t] = expf(ph[t] - m);
            sum += ph[t];
        }
        const float inv = 1.f / sum;
        for (int t = 0; t < Tk; ++t) ph[t] *= inv;
    }
    __syncthreads();
    for (int c = tid; c < A; c += 256) {
        const float* ph = p + (c / dk) * Tk;
        float a = 0.f;
        for (int t = 0; t < Tk; ++t) a = fmaf(ph[t], V[(long)t * A + c], a);
        ctx[c] = a;
    }
    __syncthreads();
    for (int c = tid; c < A; c += 256) {
        float a = bo[c];
        for (int j = 0; j < A; ++j) a = fmaf(ctx[j], Wo[(long)j * A + c], a);
        style[(long)b * A + c] = a;
    }
}

int find_first(const pk_param_map& P, const std::vector<std::string>& names, int64_t numel, std::vector<float>& out) {
    for (const std::string& n : names) {
        auto it = P.find(n);
        if (it == P.end()) continue;
        if (it->second.numel() != numel)
            PK_FAIL(PK_ESHAPE, "parameter %s has %lld elements, expected %lld", n.c_str(), (long long)it->second.numel(),
                    (long long)numel);
        out = it->second.data;
        return PK_OK;
    }
    PK_FAIL(PK_ESTATE, "parameter %s was never set", names.empty() ? "?" : names.back().c_str());
}

int conv_out(int n, int k, int stride, int pad) { return (n - k + 2 * pad) / stride + 1; }
}  // namespace

int pk_gst_check(const pk_gst_cfg& c) {
    if (c.tokens <= 0 || c.tokens > 256 || c.heads <= 0 || c.token_dim <= 0 || c.token_dim % c.heads != 0)
        PK_FAIL(PK_EINVAL, "GST: gst_tokens in [1, 256], gst_heads > 0 dividing adim");
    if (c.conv_layers < 1 || c.conv_layers > PK_GST_MAX_CONV)
        PK_FAIL(PK_EUNSUPPORTED, "GST: gst_conv_layers must be in [1, %d]", PK_GST_MAX_CONV);
    if (c.conv_kernel_size < 1 || c.conv_kernel_size % 2 == 0 || c.conv_kernel_size > 7)
        PK_FAIL(PK_EINVAL, "GST: kernel size must be odd. (style_encoder.py:152)");
    if (c.conv_stride < 1 || c.conv_stride > 4) PK_FAIL(PK_EUNSUPPORTED, "GST: gst_conv_stride must be in [1, 4]");
    for (int i = 0; i < c.conv_layers; ++i)
        if (c.conv_chans[i] <= 0) PK_FAIL(PK_EINVAL, "GST: gst_conv_chans_list needs gst_conv_layers positive entries (:153-155)");
    if (c.gru_layers < 1 || c.gru_layers > 4 || c.gru_units <= 0) PK_FAIL(PK_EUNSUPPORTED, "GST: gst_gru_layers in [1, 4], gst_gru_units > 0");
    int F = c.idim;
    for (int i = 0; i < c.conv_layers; ++i) F = conv_out(F, c.conv_kernel_size, c.conv_stride, (c.conv_kernel_size - 1) / 2);
    const long in0 = (long)F * c.conv_chans[c.conv_layers - 1];
    if (F <= 0) PK_FAIL(PK_EINVAL, "GST: the conv stack leaves no frequency bins");
    const size_t lds = ((size_t)std::max<long>(in0, c.gru_units) + 7 * (size_t)c.gru_units) * sizeof(float);
    if (lds > 60 * 1024) PK_FAIL(PK_EUNSUPPORTED, "GST: GRU of %ld inputs / %d units exceeds the kernel's LDS budget", in0, c.gru_units);
    if (((size_t)2 * c.token_dim + (size_t)c.heads * c.tokens) * sizeof(float) > 60 * 1024)
        PK_FAIL(PK_EUNSUPPORTED, "GST: style token layer too large for the kernel's LDS budget");
    return PK_OK;
}

int pk_gst_finalize(pk_fft_arena& ar, const pk_param_map& P, const std::string& prefix, pk_gst& g) {
    const pk_gst_cfg& c = g.cfg;
    PK_TRY(pk_gst_check(c));
    const int k = c.conv_kernel_size, pad = (k - 1) / 2;
    g.conv_f[0] = c.idim;
    for (int i = 0; i < c.conv_layers; ++i) {
        const int Cin = i == 0 ? 1 : c.conv_chans[i - 1], Cout = c.conv_chans[i];
        const std::string cv = prefix + ".ref_enc.convs." + std::to_string(3 * i), bn = prefix + ".ref_enc.convs." + std::to_string(3 * i + 1);
        std::vector<float> w, gam, bet, mean, var;
        PK_TRY(pk_get_weight(P, cv, {Cout, Cin, k, k}, w));
        PK_TRY(pk_get_vector(P, bn + ".weight", Cout, gam));
        PK_TRY(pk_get_vector(P, bn + ".bias", Cout, bet));
        PK_TRY(pk_get_vector(P, bn + "._mean", Cout, mean));
        PK_TRY(pk_get_vector(P, bn + "._variance", Cout, var));
        std::vector<float> b(Cout);
        for (int co = 0; co < Cout; ++co) {
            const double s = (double)gam[co] / std::sqrt((double)var[co] + 1e-5);   // BatchNorm2D eval, epsilon 1e-5
            for (long j = 0; j < (long)Cin * k * k; ++j) w[(size_t)co * Cin * k * k + j] = (float)((double)w[(size_t)co * Cin * k * k + j] * s);
            b[co] = (float)((double)bet[co] - (double)mean[co] * s);
        }
        g.conv_w[i] = ar.put(w);
        g.conv_b[i] = ar.put(b);
        g.conv_f[i + 1] = conv_out(g.conv_f[i], k, c.conv_stride, pad);
    }
    const int H = c.gru_units, G3 = 3 * H;
    g.gru_wih.assign(c.gru_layers, 0);
    g.gru_whh.assign(c.gru_layers, 0);
    g.gru_bih.assign(c.gru_layers, 0);
    g.gru_bhh.assign(c.gru_layers, 0);
    for (int l = 0; l < c.gru_layers; ++l) {
        const int In = l == 0 ? g.conv_f[c.conv_layers] * c.conv_chans[c.conv_layers - 1] : H;
        const std::string a = prefix + ".ref_enc.gru.", cl = a + std::to_string(l) + ".cell.", sfx = "_l" + std::to_string(l);
        std::vector<float> wih, whh, bih, bhh;
        PK_TRY(find_first(P, {a + "weight_ih" + sfx, cl + "weight_ih"}, (int64_t)G3 * In, wih));
        PK_TRY(find_first(P, {a + "weight_hh" + sfx, cl + "weight_hh"}, (int64_t)G3 * H, whh));
        PK_TRY(find_first(P, {a + "bias_ih" + sfx, cl + "bias_ih"}, G3, bih));
        PK_TRY(find_first(P, {a + "bias_hh" + sfx, cl + "bias_hh"}, G3, bhh));
        std::vector<float> wihT((size_t)In * G3), whhT((size_t)H * G3);
        for (int gi = 0; gi < G3; ++gi) {
            for (int i = 0; i < In; ++i) wihT[(size_t)i * G3 + gi] = wih[(size_t)gi * In + i];
            for (int j = 0; j < H; ++j) whhT[(size_t)j * G3 + gi] = whh[(size_t)gi * H + j];
        }
        g.gru_wih[l] = ar.put(wihT);
        g.gru_whh[l] = ar.put(whhT);
        g.gru_bih[l] = ar.put(bih);
        g.gru_bhh[l] = ar.put(bhh);
    }
    {
        // K = linear_k(tanh(gst_embs)), V = linear_v(tanh(gst_embs)) (:278-285): constants of the model
        const int A = c.token_dim, dk = A / c.heads, Tk = c.tokens;
        const std::string m = prefix + ".stl.mha.";
        std::vector<float> e, wk, wv, bk, bv, wq, bq, wo, bo;
        PK_TRY(find_first(P, {prefix + ".stl.gst_embs"}, (int64_t)Tk * dk, e));   // a bare parameter, no ".weight"
        PK_TRY(pk_get_weight(P, m + "linear_k", {dk, A}, wk));
        PK_TRY(pk_get_weight(P, m + "linear_v", {dk, A}, wv));
        PK_TRY(pk_get_vector(P, m + "linear_k.bias", A, bk));
        PK_TRY(pk_get_vector(P, m + "linear_v.bias", A, bv));
        PK_TRY(pk_get_weight(P, m + "linear_q", {H, A}, wq));
        PK_TRY(pk_get_vector(P, m + "linear_q.bias", A, bq));
        PK_TRY(pk_get_weight(P, m + "linear_out", {A, A}, wo));
        PK_TRY(pk_get_vector(P, m + "linear_out.bias", A, bo));
        std::vector<float> K((size_t)Tk * A), V((size_t)Tk * A);
        for (int t = 0; t < Tk; ++t)
            for (int o = 0; o < A; ++o) {
                double sk = bk[o], sv = bv[o];
                for (int d = 0; d < dk; ++d) {
                    const double te = std::tanh((double)e[(size_t)t * dk + d]);
                    sk += te * wk[(size_t)d * A + o];
                    sv += te * wv[(size_t)d * A + o];
                }
                K[(size_t)t * A + o] = (float)sk;
                V[(size_t)t * A + o] = (float)sv;
            }
        g.stl_k = ar.put(K);
        g.stl_v = ar.put(V);
        g.stl_wq = ar.put(wq);
        g.stl_bq = ar.put(bq);
        g.stl_wo = ar.put(wo);
        g.stl_bo = ar.put(bo);
    }
    return PK_OK;
}

int pk_gst_run(pk_fft_core* h, pk_gst& g, const float* speech, const int* lens, int B, float* d_style) {
    pk_ctx* ctx = h->ctx;
    const pk_gst_cfg& c = g.cfg;
    const int NL = c.conv_layers, k = c.conv_kernel_size, pad = (k - 1) / 2, H = c.gru_units;
    // per layer and utterance: offsets and lengths; activations ping-pong between d_a and d_b
    // table rows: layer i -> [in_off B][out_off B][Tin B][Tout B]; then the GRU's [in_off B][seq_off B][T B]
    std::vector<long> tab((size_t)(4 * NL + 3) * B);
    std::vector<int> T(lens, lens + B);
    size_t cap[2] = {0, 0};
    long max_out[PK_GST_MAX_CONV] = {0};
    long total_in = 0;
    for (int b = 0; b < B; ++b) {
        if (lens[b] <= 0) PK_FAIL(PK_EINVAL, "GST: reference spectrogram %d has %d frames", b, lens[b]);
        total_in += (long)lens[b] * c.idim;
    }
    cap[0] = (size_t)total_in;
    std::vector<long> off_in(B), off_out(B);
    {
        long o = 0;
        for (int b = 0; b < B; ++b) {
            off_in[b] = o;
            o += (long)lens[b] * c.idim;
        }
    }
    for (int i = 0; i < NL; ++i) {
        const int Cout = c.conv_chans[i];
        long o = 0;
        for (int b = 0; b < B; ++b) {
            const int Tout = conv_out(T[b], k, c.conv_stride, pad);
            long* row = tab.data() + (size_t)4 * i * B;
            row[b] = off_in[b];
            row[B + b] = o;
            row[2 * B + b] = T[b];
            row[3 * B + b] = Tout;
            const long n = (long)Cout * Tout * g.conv_f[i + 1];
            max_out[i] = std::max(max_out[i], n);
            off_out[b] = o;
            o += n;
            T[b] = Tout;
        }
        cap[(i + 1) & 1] = std::max(cap[(i + 1) & 1], (size_t)o);
        off_in = off_out;
    }
    long seq_rows = 0;
    {
        long* row = tab.data() + (size_t)4 * NL * B;
        for (int b = 0; b < B; ++b) {
            row[b] = off_in[b];
            row[B + b] = seq_rows * H;
            row[2 * B + b] = T[b];
            seq_rows += T[b];
        }
    }
    PK_TRY(g.d_a.reserve(std::max<size_t>(cap[0], 1) * sizeof(float)));
    PK_TRY(g.d_b.reserve(std::max<size_t>(cap[1], 1) * sizeof(float)));
    PK_TRY(g.d_seq0.reserve((size_t)seq_rows * H * sizeof(float)));
    PK_TRY(g.d_seq1.reserve((size_t)seq_rows * H * sizeof(float)));
    PK_TRY(g.d_ref.reserve((size_t)B * H * sizeof(float)));
    PK_TRY(pk_upload(ctx, g.d_tab, tab.data(), tab.size() * sizeof(long)));
    PK_HIP(hipMemcpyAsync(g.d_a.p, speech, (size_t)total_in * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    PK_HIP(hipStreamSynchronize(ctx->stream));   // the caller's host buffer may go away
    const long* dtab = g.d_tab.as<long>();
    for (int i = 0; i < NL; ++i) {
        const float* in = (i & 1) ? g.d_b.as<float>() : g.d_a.as<float>();
        float* out = (i & 1) ? g.d_a.as<float>() : g.d_b.as<float>();
        PK_LAUNCH(ctx, "gst_conv", k_gst_conv, dim3((unsigned)pk_div_up(max_out[i], 256), B), dim3(256), 0, in, out,
                  dtab + (size_t)4 * i * B, B, h->W(g.conv_w[i]), h->W(g.conv_b[i]), i == 0 ? 1 : c.conv_chans[i - 1],
                  c.conv_chans[i], g.conv_f[i], g.conv_f[i + 1], k, c.conv_stride, pad);
    }
    const float* conv = (NL & 1) ? g.d_b.as<float>() : g.d_a.as<float>();
    const int convC = c.conv_chans[NL - 1], convF = g.conv_f[NL];
    const long* gtab = dtab + (size_t)4 * NL * B;
    for (int l = 0; l < c.gru_layers; ++l) {
        const int In = l == 0 ? convC * convF : H;
        const float* in = l == 0 ? conv : ((l & 1) ? g.d_seq0.as<float>() : g.d_seq1.as<float>());
        float* seq = (l & 1) ? g.d_seq1.as<float>() : g.d_seq0.as<float>();
        const size_t lds = ((size_t)In + 7 * (size_t)H) * sizeof(float);
        // layers >= 1 read the previous layer's sequence: their input offsets are the sequence offsets
        PK_LAUNCH(ctx, "gst_gru", k_gst_gru, dim3(B), dim3(256), lds, in, l == 0 ? gtab : gtab + B, gtab + B, gtab + 2 * B,
                  l == 0 ? convC : 0, convF, In,
                  h->W(g.gru_wih[l]), h->W(g.gru_whh[l]), h->W(g.gru_bih[l]), h->W(g.gru_bhh[l]), H, seq,
                  l == c.gru_layers - 1 ? g.d_ref.as<float>() : (float*)nullptr);
    }
    const size_t lds = ((size_t)2 * c.token_dim + (size_t)c.heads * c.tokens) * sizeof(float);
    PK_LAUNCH(ctx, "gst_style", k_gst_style, dim3(B), dim3(256), lds, g.d_ref.as<float>(), H, h->W(g.stl_wq), h->W(g.stl_bq),
              h->W(g.stl_k), h->W(g.stl_v), c.tokens, c.heads, c.token_dim, h->W(g.stl_wo), h->W(g.stl_bo), d_style);
    return PK_OK;
}

void pk_gst_release(pk_gst& g) {
    pk_dbuf* bufs[] = {&g.d_a, &g.d_b, &g.d_tab, &g.d_seq0, &g.d_seq1, &g.d_ref};
    for (pk_dbuf* b : bufs) b->release();
}
