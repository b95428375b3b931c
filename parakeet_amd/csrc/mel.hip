// mel.hip -- STFT -> magnitude/power -> mel -> log on gfx950 (feature / metric path).
//
// Reference: parakeet/modules/audio.py STFT.forward :161-200 (reflect pad + F.conv1d with the
// windowed DFT basis [2*n_bin, 1, n_fft], stride hop), power :198-200, magnitude :202-215,
// MelScale.forward :226-229; host twin parakeet/data/get_feats.py LogMelFBank :56-88
// (log10(clip(mel_basis . |STFT|, 1e-10))).
//
// The strided conv IS a GEMM whose A operand is the padded signal itself with leading dimension
// = hop: row r (frame r) = xpad[r*hop .. r*hop + n_fft).  Utterances sit on one padded-sample axis
// at hop-aligned offsets, so every frame of every utterance is one row of a single GEMM
// (frames that straddle two utterances are computed and dropped by the row map).
#include <cmath>

#include "pk_gemm.h"

namespace {

// xpad[poff[b] + i] = x[reflect(i - pad)] for i in [0, len + 2*pad)   (F.pad mode='reflect', :175-179)
__global__ void k_reflect_pad(const float* __restrict__ wav, const int* __restrict__ woff,
                              const int* __restrict__ wlen, const long* __restrict__ poff, int pad,
                              float* __restrict__ xpad) {
    const int b = blockIdx.y;
    const int n = wlen[b];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n + 2 * pad) return;
    int s = i - pad;
    if (pad > 0) {
        if (s < 0) s = -s;
        if (s >= n) s = 2 * (n - 1) - s;
    }
    xpad[poff[b] + i] = wav[(long)woff[b] + s];
}

// spec[r][k] = re^2 + im^2 (power) or its sqrt (magnitude); columns [n_bin, ld) zeroed (GEMM K padding)
__global__ void k_magnitude(const float* __restrict__ reim, int ld_in, int n_bin, int rows, int power,
                            float* __restrict__ spec, int ld_out) {
    const int r = blockIdx.y;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows || k >= ld_out) return;
    float v = 0.f;
    if (k < n_bin) {
        const float re = reim[(long)r * ld_in + k], im = reim[(long)r * ld_in + n_bin + k];
        v = re * re + im * im;
        if (!power) v = sqrtf(v);
    }
    spec[(long)r * ld_out + k] = v;
}

// y = log_base(max(y, floor))   (np.clip(mel, a_min=1e-10) + log10 / log, get_feats.py:82-87)
__global__ void k_clip_log(float* __restrict__ y, long n, float floor_v, int base10) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = fmaxf(y[i], floor_v);
    y[i] = base10 ? log10f(v) : logf(v);
}

}  // namespace

struct pk_mel {
    pk_ctx* ctx = nullptr;
    pk_mel_cfg cfg;
    int n_bin = 0, ldspec = 0;
    pk_dbuf d_dft, d_melw;          // packed GEMM weights
    pk_dbuf ws_tab, ws_ltab, ws_wav, ws_xpad, ws_reim, ws_spec, ws_out;
};

extern "C" int pk_mel_create(pk_ctx* ctx, const pk_mel_cfg* cfg, const float* window, const float* mel_basis,
                             pk_mel** out) {
    if (!ctx || !cfg || !window || !out) PK_FAIL(PK_EINVAL, "pk_mel_create: NULL argument");
    *out = nullptr;
    const pk_mel_cfg& c = *cfg;
    if (c.n_fft <= 0 || c.n_fft % PK_GEMM_BK != 0) PK_FAIL(PK_EUNSUPPORTED, "STFT: n_fft must be a multiple of 16");
    if (c.hop_length <= 0 || c.hop_length % 4 != 0) PK_FAIL(PK_EUNSUPPORTED, "STFT: hop_length must be a multiple of 4");
    if (c.n_mels < 0 || (c.n_mels > 0 && !mel_basis)) PK_FAIL(PK_EINVAL, "pk_mel_create: mel basis missing");
    PK_DEVICE(ctx->device);
    pk_mel* h = new pk_mel();
    h->ctx = ctx;
    h->cfg = c;
    const int N = c.n_fft, nb = 1 + N / 2;
    h->n_bin = nb;
    h->ldspec = ((nb + PK_GEMM_BK - 1) / PK_GEMM_BK) * PK_GEMM_BK;
    // windowed DFT basis, np.fft.fft(np.eye(n_fft))[:n_bin] * window (:146-153): [K = n][N = re(k) | im(k)]
    {
        std::vector<float> kn((size_t)N * 2 * nb), packed;
        for (int n = 0; n < N; ++n)
            for (int k = 0; k < nb; ++k) {
                const double ang = -2.0 * M_PI * (double)(((long)n * k) % N) / N;
                kn[(size_t)n * 2 * nb + k] = (float)(std::cos(ang) * (double)window[n]);
                kn[(size_t)n * 2 * nb + nb + k] = (float)(std::sin(ang) * (double)window[n]);
            }
        pk_gemm_pack(kn.data(), N, 2 * nb, packed);
        int s = pk_upload(ctx, h->d_dft, packed.data(), packed.size() * sizeof(float));
        if (s != PK_OK) { delete h; return s; }
    }
    if (c.n_mels > 0) {
        std::vector<float> kn((size_t)h->ldspec * c.n_mels, 0.f), packed;   // [K = bin][N = mel]
        for (int m = 0; m < c.n_mels; ++m)
            for (int k = 0; k < nb; ++k) kn[(size_t)k * c.n_mels + m] = mel_basis[(size_t)m * nb + k];
        pk_gemm_pack(kn.data(), h->ldspec, c.n_mels, packed);
        int s = pk_upload(ctx, h->d_melw, packed.data(), packed.size() * sizeof(float));
        if (s != PK_OK) { delete h; return s; }
    }
    *out = h;
    return PK_OK;
}

extern "C" int pk_mel_num_frames(pk_mel* h, int32_t n_samples, int32_t* frames) {
    if (!h || !frames) PK_FAIL(PK_EINVAL, "pk_mel_num_frames: NULL argument");
    const int pad = h->cfg.center ? h->cfg.n_fft / 2 : 0;
    const long padded = (long)n_samples + 2 * pad;
    *frames = padded < h->cfg.n_fft ? 0 : (int32_t)(1 + (padded - h->cfg.n_fft) / h->cfg.hop_length);
    return PK_OK;
}

extern "C" int pk_mel_run(pk_mel* h, const float* wav, const int32_t* lens, int32_t B, float* out,
                          int32_t what, int32_t flags) {
    if (!h || !wav || !lens || !out) PK_FAIL(PK_EINVAL, "pk_mel_run: NULL argument");
    if (B <= 0) PK_FAIL(PK_EINVAL, "pk_mel_run: batch size must be positive");
    if (what < 0 || what > 2) PK_FAIL(PK_EINVAL, "pk_mel_run: what must be 0 (re|im), 1 (spectrum) or 2 (mel)");
    if (what == 2 && h->cfg.n_mels <= 0) PK_FAIL(PK_ESTATE, "pk_mel_run: no mel basis was given");
    pk_ctx* ctx = h->ctx;
    PK_DEVICE(ctx->device);
    const pk_mel_cfg& c = h->cfg;
    const int N = c.n_fft, hop = c.hop_length, nb = h->n_bin, pad = c.center ? N / 2 : 0;
    std::vector<int> woff(B), nfr(B), row0(B);
    std::vector<long> poff(B);
    long p = 0, sumS = 0;
    int sumF = 0, maxlen = 0;
    for (int b = 0; b < B; ++b) {
        if (lens[b] <= pad) PK_FAIL(PK_EINVAL, "pk_mel_run: utterance %d too short for reflect padding", b);
        woff[b] = (int)sumS;
        sumS += lens[b];
        maxlen = lens[b] > maxlen ? lens[b] : maxlen;
        const long padded = (long)lens[b] + 2 * pad;
        nfr[b] = padded < N ? 0 : (int)(1 + (padded - N) / hop);
        poff[b] = p;
        row0[b] = (int)(p / hop);
        p += ((padded + hop - 1) / hop) * hop;   // next utterance starts hop-aligned
        sumF += nfr[b];
    }
    if (sumF == 0) return PK_OK;
    const int rows = (int)(p / hop);             // every hop position is a candidate row
    const int rows_alloc = ((rows + PK_GEMM_BM - 1) / PK_GEMM_BM) * PK_GEMM_BM;
    std::vector<int> rowmap(rows_alloc, -1), tab;
    {
        int o = 0;
        for (int b = 0; b < B; ++b)
            for (int f = 0; f < nfr[b]; ++f) rowmap[row0[b] + f] = o++;
    }
    tab.insert(tab.end(), woff.begin(), woff.end());
    tab.insert(tab.end(), lens, lens + B);
    tab.insert(tab.end(), rowmap.begin(), rowmap.end());
    PK_TRY(h->ws_tab.reserve(tab.size() * sizeof(int)));
    PK_TRY(h->ws_ltab.reserve(poff.size() * sizeof(long)));
    PK_HIP(hipMemcpyAsync(h->ws_tab.p, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    PK_HIP(hipMemcpyAsync(h->ws_ltab.p, poff.data(), poff.size() * sizeof(long), hipMemcpyHostToDevice, ctx->stream));
    PK_HIP(hipStreamSynchronize(ctx->stream));
    const int* d_tab = h->ws_tab.as<int>();
    const float* d_wav = wav;
    const int out_cols = what == 0 ? 2 * nb : (what == 1 ? nb : c.n_mels);
    float* d_out = out;
    if (flags & PK_HOST_IO) {
        PK_TRY(h->ws_wav.reserve((size_t)sumS * 4));
        PK_TRY(h->ws_out.reserve((size_t)sumF * out_cols * 4));
        PK_HIP(hipMemcpyAsync(h->ws_wav.p, wav, (size_t)sumS * 4, hipMemcpyHostToDevice, ctx->stream));
        d_wav = h->ws_wav.as<float>();
        d_out = h->ws_out.as<float>();
    }
    // padded signal (+ one tile of slack so the last GEMM row tile can read n_fft samples)
    const size_t xpad_floats = (size_t)rows_alloc * hop + N + 64;
    PK_TRY(h->ws_xpad.reserve(xpad_floats * 4));
    PK_HIP(hipMemsetAsync(h->ws_xpad.p, 0, xpad_floats * 4, ctx->stream));
    PK_LAUNCH(ctx, "mel_reflect_pad", k_reflect_pad, dim3(pk_div_up(maxlen + 2 * pad, 256), B), dim3(256), 0, d_wav,
              d_tab, d_tab + B, h->ws_ltab.as<long>(), pad, h->ws_xpad.as<float>());
    // STFT GEMM: [rows x n_fft] (lda = hop) x [n_fft x 2*n_bin]
    pk_gemm_args g;
    g.A = h->ws_xpad.as<float>();
    g.lda = hop;
    g.Cin = N;
    g.taps = 1;
    g.pad = 0;
    g.Wp = h->d_dft.as<float>();
    g.M = rows;
    g.N = 2 * nb;
    if (what == 0) {
        g.C = d_out;
        g.ldc = 2 * nb;
        g.out_rowmap = d_tab + 2 * B;
        PK_TRY(pk_gemm_launch(ctx, "mel_stft_gemm", g));
    } else {
        PK_TRY(h->ws_reim.reserve((size_t)rows_alloc * 2 * nb * 4));
        g.C = h->ws_reim.as<float>();
        g.ldc = 2 * nb;
        PK_TRY(pk_gemm_launch(ctx, "mel_stft_gemm", g));
        if (what == 1) {
            // spectrum straight into the packed output (row map applied by a tiny second pass: reuse k_magnitude
            // on timeline rows, then gather) -- keep it simple: compute on the timeline, gather rows with a GEMM-free copy
            PK_TRY(h->ws_spec.reserve((size_t)rows_alloc * h->ldspec * 4));
            PK_LAUNCH(ctx, "mel_magnitude", k_magnitude, dim3(pk_div_up(h->ldspec, 256), rows), dim3(256), 0,
                      h->ws_reim.as<float>(), 2 * nb, nb, rows, c.power ? 1 : 0, h->ws_spec.as<float>(), h->ldspec);
            for (int b = 0, o = 0; b < B; ++b) {
                if (nfr[b] > 0)
                    PK_HIP(hipMemcpy2DAsync(d_out + (size_t)o * nb, (size_t)nb * 4,
                                            h->ws_spec.as<float>() + (size_t)row0[b] * h->ldspec, (size_t)h->ldspec * 4,
                                            (size_t)nb * 4, nfr[b], hipMemcpyDeviceToDevice, ctx->stream));
                o += nfr[b];
            }
        } else {
            PK_TRY(h->ws_spec.reserve((size_t)(rows_alloc + 8) * h->ldspec * 4));
            PK_LAUNCH(ctx, "mel_magnitude", k_magnitude, dim3(pk_div_up(h->ldspec, 256), rows), dim3(256), 0,
                      h->ws_reim.as<float>(), 2 * nb, nb, rows, c.power ? 1 : 0, h->ws_spec.as<float>(), h->ldspec);
            pk_gemm_args m;
            m.A = h->ws_spec.as<float>();
            m.lda = h->ldspec;
            m.Cin = h->ldspec;
            m.taps = 1;
            m.pad = 0;
            m.Wp = h->d_melw.as<float>();
            m.C = d_out;
            m.ldc = c.n_mels;
            m.out_rowmap = d_tab + 2 * B;
            m.M = rows;
            m.N = c.n_mels;
            PK_TRY(pk_gemm_launch(ctx, "mel_filterbank_gemm", m));
            if (c.log_base != 0) {
                const long n = (long)sumF * c.n_mels;
                PK_LAUNCH(ctx, "mel_clip_log", k_clip_log, dim3(pk_div_up(n, 256)), dim3(256), 0, d_out, n, c.log_floor,
                          c.log_base == 10 ? 1 : 0);
            }
        }
    }
    if (flags & PK_HOST_IO) {
        PK_HIP(hipMemcpyAsync(out, d_out, (size_t)sumF * out_cols * 4, hipMemcpyDeviceToHost, ctx->stream));
        PK_HIP(hipStreamSynchronize(ctx->stream));
    }
    return PK_OK;
}

extern "C" void pk_mel_destroy(pk_mel* h) {
    if (!h) return;
    pk_device_guard _dg(h->ctx->device);
    (void)hipStreamSynchronize(h->ctx->stream);
    pk_dbuf* bufs[] = {&h->d_dft, &h->d_melw, &h->ws_tab, &h->ws_ltab, &h->ws_wav, &h->ws_xpad, &h->ws_reim,
                       &h->ws_spec, &h->ws_out};
    for (auto* b : bufs) b->release();
    delete h;
}
