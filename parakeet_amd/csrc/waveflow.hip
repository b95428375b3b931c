// waveflow.hip -- ConditionalWaveFlow.infer on gfx950: kernels + pk_wf_* entry points.
//
// Reference: parakeet/models/waveflow.py  UpsampleNet.forward(trim) :103-132, fold :32-51,
// ResidualBlock.add_input :248-294, ResidualNet.add_input :368-392, Flow.inverse :515-556
// (_predict_row_parameters :496-501, _inverse_transform_row :503-505), WaveFlow.inverse :674-711
// (_create_perm :602-615, _trim :617-625), ConditionalWaveFlow.infer :785-805;
// parakeet/modules/geometry.py shuffle_dim :18-50.
//
// Layout.  The folded signal is (H = n_group rows) x (W = T / n_group positions).  All utterances of
// a batch share one position axis ("width timeline") framed by GAPW >= 2^(n_layers-1) zero positions,
// so the width-dilated "same" convolutions see the reference's zero padding and ragged batches are
// exact.  Per-position feature rows are channels-last [pos][C] fp32, i.e. GEMM A operands:
//   hist[l][slot][pos][C]   inputs of residual layer l for the last 3 rows (the reference's
//                           _conv_buffer, kept as a ring instead of concat-shifting every step)
//   cond[h][pos][n_mels]    upsampled mel, folded; row permutations of the flows are index maps
//   cur/nxt[h][pos]         the folded latent / signal
// One autoregressive step (flow f, row i) = for each of the 8 layers: GEMM1 (3x3 dilated causal conv
// as <=9 shifted taps + condition_proj as a second K block, gated-tanh epilogue) -> GEMM2 (out_proj,
// residual into the next layer's ring slot, skip accumulated), then k_wf_step (output_proj, affine
// inverse, input_proj of the new row).  Rows that do not exist yet (steps < 1) are skipped as taps
// instead of being stored as zeros.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "pk_gemm.h"
#include "pk_grid.h"
#include "pk_wf_layer.h"

namespace {

constexpr int WF_LEAD = 256;  // margin positions around every per-position buffer (>= largest tap shift)

// One Conv2DTranspose(1,1,(3,2f),stride (1,f),padding (1,f//2)) + trim + leaky_relu(0.4) layer.
// in[rows][M] (time-major, M mel bins), out likewise -- or, for the last layer, the folded
// cond[h][pos][M] layout (pos = woff[b] + t / G, h = t % G, only t < pruned[b]).
__global__ void k_wf_upsample(const float* __restrict__ in, const int* __restrict__ in_off,
                              const int* __restrict__ in_len, float* __restrict__ out,
                              const int* __restrict__ out_off, const float* __restrict__ w, float bias, int f,
                              int M, int fold_G, const int* __restrict__ woff, const int* __restrict__ pruned,
                              long cond_row_stride, int cond_ld, int blocked) {
    const int b = blockIdx.z;
    const int Tin = in_len[b];
    const int Tout = f * Tin - f;  // (Tin-1)*f - 2*(f/2) + 2f, minus the trimmed (2f - f) columns
    const int t = blockIdx.x * (blockDim.x / M) + threadIdx.x / M;
    const int c = threadIdx.x % M;
    if (t >= Tout || threadIdx.x >= (blockDim.x / M) * M) return;
    const float* src = in + (long)in_off[b] * M;
    float acc = bias;
    // out[c][t] += in[c + 1 - ky][ix] * w[ky][kx],  kx = t + f/2 - f*ix in [0, 2f)
    const int tt = t + f / 2;
    const int ix_hi = tt / f;                 // kx = tt - f*ix >= 0
    const int ix_lo = (tt - 2 * f) / f + 1;   // kx < 2f  (floor division handled below)
    for (int ix = ix_hi; ix >= 0 && ix >= ix_hi - 1; --ix) {
        const int kx = tt - f * ix;
        if (kx < 0 || kx >= 2 * f || ix >= Tin) continue;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int ci = c + 1 - ky;
            if (ci >= 0 && ci < M) acc = fmaf(src[(long)ix * M + ci], w[ky * 2 * f + kx], acc);
        }
    }
    (void)ix_lo;
    acc = acc > 0.f ? acc : 0.4f * acc;
    if (fold_G == 0) {
        out[((long)out_off[b] + t) * M + c] = acc;
    } else if (t < pruned[b]) {
        const long pos = (long)woff[b] + t / fold_G;
        // blocked: [pos / 32][cond_ld][32] (pk_wf_layer.h), else [pos][cond_ld]
        const long off = blocked ? (pos >> 5) * ((long)cond_ld * 32) + (long)c * 32 + (pos & 31) : pos * cond_ld + c;
        out[(long)(t % fold_G) * cond_row_stride + off] = acc;
    }
}

// cur[h][pos] = z[zoff[b] + G*w + h]  (fold :32-51 + transpose :697-698); gap positions -> 0
__global__ void k_wf_fold(const float* __restrict__ z, const int* __restrict__ pos_utt,
                          const int* __restrict__ pos_w, const int* __restrict__ zoff, int G, int npos,
                          long row_stride, float* __restrict__ cur) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npos) return;
    const int b = pos_utt[p];
    for (int h = 0; h < G; ++h)
        cur[(long)h * row_stride + p] = b >= 0 ? z[(long)zoff[b] + (long)G * pos_w[p] + h] : 0.f;
}

__global__ void k_wf_unfold(const float* __restrict__ cur, const int* __restrict__ pos_utt,
                            const int* __restrict__ pos_w, const int* __restrict__ ooff, int G, int npos,
                            long row_stride, float* __restrict__ wav) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npos) return;
    const int b = pos_utt[p];
    if (b < 0) return;
    for (int h = 0; h < G; ++h) wav[(long)ooff[b] + (long)G * pos_w[p] + h] = cur[(long)h * row_stride + p];
}

// Start of a flow: x[0] = z'[0] (z' = rows of cur permuted, :704,540-542) and the input projection
// of that first row into ring slot 1 of layer 0.  One wave per position.
// Every step: params = output_proj(sum of skips) (:499-500), x[i] = (z'[i] - b) * exp(-logs) (:503-505),
// then h0 = input_proj(x[i]) for the next step (:497).
__global__ __launch_bounds__(256) void k_wf_step(const float* __restrict__ skipsum, int C,
                                                 const float* __restrict__ w_out, float b_logs, float b_b,
                                                 const float* __restrict__ z_row, float* __restrict__ x_row,
                                                 const float* __restrict__ w_in, const float* __restrict__ b_in,
                                                 float* __restrict__ h0_next, const int* __restrict__ pos_utt,
                                                 int npos, int first) {
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= npos) return;
    const int lane = threadIdx.x & 63;
    const bool valid = pos_utt[p] >= 0;
    float xn = 0.f;
    if (valid) {
        if (first) {
            xn = z_row[p];
        } else {
            const float* s = skipsum + (long)p * C;
            float l = 0.f, bb = 0.f;
            for (int c = lane; c < C; c += 64) {
                const float v = s[c];
                l = fmaf(w_out[c], v, l);
                bb = fmaf(w_out[C + c], v, bb);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                l += __shfl_xor(l, o);
                bb += __shfl_xor(bb, o);
            }
            xn = (z_row[p] - (bb + b_b)) * expf(-(l + b_logs));
        }
    }
    if (lane == 0) x_row[p] = xn;
    if (h0_next)
        for (int c = lane; c < C; c += 64) h0_next[(long)p * C + c] = valid ? fmaf(w_in[c], xn, b_in[c]) : 0.f;
}

}  // namespace

// ================================================================== host side
struct WfLayerW {
    size_t w1, b1, w2, b2;   // packed GEMM weights (floats offsets into the arena)
    size_t w1h, w2h;         // split-fp16 fragments (halves offsets into arena16)
    WflPacked fl;            // the fused layer kernel's fragments (64-channel model), offsets into the same arenas
};

struct WfFlowW {
    size_t w_in, b_in, w_out;
    float b_logs, b_b;
    float b_logs_f, b_b_f;   // with the layers' folded skip biases (fused layer kernel, pk_wf_layer.h)
    std::vector<WfLayerW> layers;
};

struct pk_wf {
    pk_ctx* ctx = nullptr;
    pk_wf_cfg cfg;
    pk_param_map params;
    bool finalized = false;
    int gapw = 128;
    std::vector<float> arena_h;
    pk_dbuf arena;
    std::vector<uint16_t> arena16_h;
    pk_dbuf arena16;
    int math = PK_GEMM_MATH_F16X3;
    unsigned long long seed = 0, rng_offset = 0;   // internal latent stream (z == NULL)
    bool no_fuse = pk_prof_env("PK_WF_NO_FUSE") != nullptr;   // measurement switch: separate out_proj launches
    int mp = 96;              // mel channels padded to a multiple of 32 (GEMM K block of the condition)
    int layer_waves = 0;      // option "layer_waves": 0 = the launcher chooses, 8 / 12 = waves per workgroup of the fused layer kernel
    bool persistent = false;  // option "persistent": the layers of a row in ONE launch behind grid barriers (pk_grid.h).  Built, correct
                              // (bit-identical, tests/test_waveflow_gpu.py) and SLOWER on this part: a barrier across 236 workgroups
                              // on 8 XCDs costs 11 us of serialised atomics + 16 - 27 us of L2 write-back / invalidate
                              // (tools/micro/grid_barrier.hip, profiles/r04_grid_barrier_micro.txt) against the 5 - 6 us between two
                              // dependent launches -- 1 089 us per row instead of 8 x 55.  Off by default.
    bool fuse_step = true;    // option "fuse_step": the row's step in the launch of its last layer (else a kernel of its own)
    std::vector<WfFlowW> flows;
    std::vector<size_t> up_w;
    std::vector<float> up_b;
    // workspace
    pk_dbuf ws_tab, ws_mel, ws_z, ws_wav, ws_u[2], ws_cond, ws_cur, ws_nxt, ws_hist, ws_zbuf, ws_skip, ws_hamax, ws_camax, ws_trace, ws_bar, ws_desc, ws_replay;
    std::vector<WflLayer> desc_host;   // host image of ws_desc (kept until the next inference: the upload is asynchronous)
    const float* W(size_t off) const { return arena.as<float>() + off; }
};

extern "C" int pk_wf_create(pk_ctx* ctx, const pk_wf_cfg* cfg, pk_wf** out) {
    if (!ctx || !cfg || !out) PK_FAIL(PK_EINVAL, "pk_wf_create: NULL argument");
    *out = nullptr;
    const pk_wf_cfg& c = *cfg;
    if (c.n_group % 2 || c.n_flows % 2 || c.n_group <= 0 || c.n_flows <= 0)
        PK_FAIL(PK_EINVAL, "number of flows and number of group must be even since a permutation along "
                           "group among flows is used.");  // waveflow.py:586-589 (ValueError)
    if (c.n_group != 8 && c.n_group != 16)
        PK_FAIL(PK_EUNSUPPORTED, "WaveFlow: n_group %d needs height dilations > 1 (not implemented)", c.n_group);
    if (c.n_layers != 8) PK_FAIL(PK_EINVAL, "number of dilations_h should equals num of layers");  // :328-331
    if (c.kernel_h != 3 || c.kernel_w != 3) PK_FAIL(PK_EUNSUPPORTED, "WaveFlow: kernel_size must be (3, 3)");
    if (c.channels % 64 != 0 || c.channels > 256) PK_FAIL(PK_EUNSUPPORTED, "WaveFlow: channels must be 64/128/192/256");
    if (c.n_mels % PK_GEMM_BK != 0) PK_FAIL(PK_EUNSUPPORTED, "WaveFlow: n_mels must be a multiple of 16");
    if (c.n_upsample < 1 || c.n_upsample > 4) PK_FAIL(PK_EINVAL, "WaveFlow: 1..4 upsample layers");
    for (int i = 0; i < c.n_upsample; ++i)
        if (c.upsample_factors[i] < 2 || c.upsample_factors[i] % 2 || c.upsample_factors[i] > 64)
            PK_FAIL(PK_EUNSUPPORTED, "WaveFlow: upsample factor %d unsupported", c.upsample_factors[i]);
    pk_wf* h = new pk_wf();
    h->ctx = ctx;
    h->cfg = c;
    h->gapw = 1 << (c.n_layers - 1);
    h->mp = ((c.n_mels + PK_GEMM_HBK - 1) / PK_GEMM_HBK) * PK_GEMM_HBK;
    if (const char* e = pk_prof_env("PK_WF_MATH")) h->math = strcmp(e, "f32") == 0 ? PK_GEMM_MATH_F32 : PK_GEMM_MATH_F16X3;
    *out = h;
    return PK_OK;
}

extern "C" int pk_wf_set_param(pk_wf* h, const char* name, const float* data, const int64_t* shape, int32_t ndim) {
    if (!h) PK_FAIL(PK_EINVAL, "pk_wf_set_param: handle is NULL");
    h->finalized = false;
    return pk_store_param(h->params, name, data, shape, ndim);
}

namespace {
struct Arena {
    std::vector<float>& v;
    size_t put(const std::vector<float>& x) {
        size_t o = (v.size() + 3) & ~(size_t)3;
        v.resize(o);
        v.insert(v.end(), x.begin(), x.end());
        return o;
    }
};
size_t put16(std::vector<uint16_t>& v, const std::vector<uint16_t>& x) {
    size_t o = (v.size() + 7) & ~(size_t)7;
    v.resize(o);
    v.insert(v.end(), x.begin(), x.end());
    return o;
}
}  // namespace

extern "C" int pk_wf_set_seed(pk_wf* h, uint64_t seed) {
    if (!h) PK_FAIL(PK_EINVAL, "pk_wf_set_seed: handle is NULL");
    h->seed = seed;
    h->rng_offset = 0;
    return PK_OK;
}

// ONE predicate for "this model runs on the fused layer kernel (wf_layer.hip)": the kernel is built for 64 / 128 channels and a
// condition block of WFL_MP = 96 channels of which one must be free -- channel n_mels carries the folded biases
// (k_wf_cond_planes).  Used by finalize (what gets packed), by pk_wf_infer (which path runs) and by pk_wf_set_math.
static bool wfl_usable(const pk_wf* h) {
    return wfl_supports(h->cfg.channels) && h->mp == WFL_MP && h->cfg.n_mels < WFL_MP && !h->no_fuse;
}

extern "C" int pk_wf_set_option(pk_wf* h, const char* key, int64_t value) {
    if (!h || !key) PK_FAIL(PK_EINVAL, "pk_wf_set_option: NULL argument");
    if (strcmp(key, "layer_waves") == 0) {
        if (value != 0 && value != 6 && value != 8 && value != 12) PK_FAIL(PK_EINVAL, "pk_wf_set_option: layer_waves %lld (0, 6, 8, 12)", (long long)value);
        if ((value == 12 || value == 6) && h->cfg.channels != 64) PK_FAIL(PK_EUNSUPPORTED, "pk_wf_set_option: 12- / 6-wave workgroups are built for the 64-channel model");
        h->layer_waves = (int)value;
    } else if (strcmp(key, "persistent") == 0) {
        // Round 5: the residual stack of a row as one cooperative launch was never the default (slower than eight launches); at
        // sizes beyond the tests' it is not deterministic either (tools/wf_race_bisect.py: thousands of samples off by 1e-3 per
        // call).  The product refuses it; the profile build keeps it for the barrier measurements.
        if (value != 0 && !wfl_measurement_configs_allowed()) PK_FAIL(PK_EUNSUPPORTED, "pk_wf_set_option: 'persistent' is not in the product (not deterministic beyond small sizes)");
        h->persistent = value != 0;
    }
    else if (strcmp(key, "fuse_step") == 0) h->fuse_step = value != 0;
    else PK_FAIL(PK_EINVAL, "pk_wf_set_option: unknown option '%s'", key);
    return PK_OK;
}

extern "C" int pk_wf_set_math(pk_wf* h, int32_t mode) {
    if (!h) PK_FAIL(PK_EINVAL, "pk_wf_set_math: handle is NULL");
    if (mode != PK_GEMM_MATH_F32 && mode != PK_GEMM_MATH_F16X3 && mode != PK_GEMM_MATH_F16)
        PK_FAIL(PK_EINVAL, "pk_wf_set_math: unknown mode %d", mode);
    if (mode == PK_GEMM_MATH_F16 && !wfl_usable(h))
        PK_FAIL(PK_EUNSUPPORTED, "pk_wf_set_math: the fp16-operand mode runs on the fused layer kernel (64 or 128 channels, n_mels in (64, 96))");
    h->math = mode;
    return PK_OK;
}

extern "C" int pk_wf_finalize(pk_wf* h) {
    if (!h) PK_FAIL(PK_EINVAL, "pk_wf_finalize: handle is NULL");
    pk_ctx* ctx = h->ctx;
    PK_DEVICE(ctx->device);
    const pk_wf_cfg& c = h->cfg;
    const pk_param_map& P = h->params;
    const int C = c.channels, M = c.n_mels;
    h->arena_h.clear();
    h->arena16_h.clear();
    Arena ar{h->arena_h};
    const int MP = h->mp;
    h->up_w.resize(c.n_upsample);
    h->up_b.resize(c.n_upsample);
    for (int i = 0; i < c.n_upsample; ++i) {
        const int f = c.upsample_factors[i];
        std::vector<float> w, b;
        PK_TRY(pk_get_weight(P, "encoder." + std::to_string(i), {1, 1, 3, 2 * f}, w));
        PK_TRY(pk_get_vector(P, "encoder." + std::to_string(i) + ".bias", 1, b));
        h->up_w[i] = ar.put(w);
        h->up_b[i] = b[0];
    }
    h->flows.resize(c.n_flows);
    for (int fl = 0; fl < c.n_flows; ++fl) {
        const std::string p = "decoder." + std::to_string(fl);
        WfFlowW& F = h->flows[fl];
        std::vector<float> w, b;
        PK_TRY(pk_get_weight(P, p + ".input_proj", {C, 1, 1, 1}, w));
        PK_TRY(pk_get_vector(P, p + ".input_proj.bias", C, b));
        F.w_in = ar.put(w);
        F.b_in = ar.put(b);
        PK_TRY(pk_get_weight(P, p + ".output_proj", {2, C, 1, 1}, w));
        PK_TRY(pk_get_vector(P, p + ".output_proj.bias", 2, b));
        F.w_out = ar.put(w);   // [logs row (C)] [b row (C)]  (chunk(params, 2, axis=1) :500)
        F.b_logs = b[0];
        F.b_b = b[1];
        const std::vector<float> w_out_flow = w;
        double fold_l = b[0], fold_b = b[1];
        F.layers.resize(c.n_layers);
        for (int l = 0; l < c.n_layers; ++l) {
            const std::string q = p + ".resnet." + std::to_string(l);
            std::vector<float> wc, bc, wp, bp, wo, bo;
            PK_TRY(pk_get_weight(P, q + ".conv", {2 * C, C, 3, 3}, wc));
            PK_TRY(pk_get_vector(P, q + ".conv.bias", 2 * C, bc));
            PK_TRY(pk_get_weight(P, q + ".condition_proj", {2 * C, M, 1, 1}, wp));
            PK_TRY(pk_get_vector(P, q + ".condition_proj.bias", 2 * C, bp));
            PK_TRY(pk_get_weight(P, q + ".out_proj", {2 * C, C, 1, 1}, wo));
            PK_TRY(pk_get_vector(P, q + ".out_proj.bias", 2 * C, bo));
            // GEMM1 weight [K = (kr*3 + kc)*C + ci | 9C + m][N = 2C], gate-permuted columns
            const int K1 = 9 * C + MP;   // condition block zero-padded to MP channels
            std::vector<float> kn((size_t)K1 * 2 * C, 0.f), perm, bias(2 * C), pbias, packed;
            std::vector<uint16_t> ph;
            for (int co = 0; co < 2 * C; ++co) {
                for (int ci = 0; ci < C; ++ci)
                    for (int kr = 0; kr < 3; ++kr)
                        for (int kc = 0; kc < 3; ++kc)
                            kn[((size_t)(kr * 3 + kc) * C + ci) * 2 * C + co] = wc[(((size_t)co * C + ci) * 3 + kr) * 3 + kc];
                for (int m = 0; m < M; ++m) kn[((size_t)9 * C + m) * 2 * C + co] = wp[(size_t)co * M + m];
                bias[co] = bc[co] + bp[co];   // conv bias + condition_proj bias (:277-278)
            }
            pk_gemm_gate_permute(kn.data(), K1, C, perm);
            pk_gemm_gate_permute_bias(bias.data(), C, pbias);
            pk_gemm_pack(perm.data(), K1, 2 * C, packed);
            F.layers[l].w1 = ar.put(packed);
            pk_gemm_pack_h3(perm.data(), K1, 2 * C, ph);
            F.layers[l].w1h = put16(h->arena16_h, ph);
            F.layers[l].b1 = ar.put(pbias);
            // GEMM2: out_proj [K = C][N = 2C] (res | skip, chunk :282)
            std::vector<float> kn2((size_t)C * 2 * C), packed2;
            for (int co = 0; co < 2 * C; ++co)
                for (int ci = 0; ci < C; ++ci) kn2[(size_t)ci * 2 * C + co] = wo[(size_t)co * C + ci];
            pk_gemm_pack(kn2.data(), C, 2 * C, packed2);
            F.layers[l].w2 = ar.put(packed2);
            pk_gemm_pack_h3(kn2.data(), C, 2 * C, ph);
            F.layers[l].w2h = put16(h->arena16_h, ph);
            F.layers[l].b2 = ar.put(bo);
            if (wfl_usable(h))
            {
                F.layers[l].fl = wfl_pack(C, wc.data(), bc.data(), wp.data(), bp.data(), M, wo.data(), bo.data(),
                                          w_out_flow.data(), h->arena16_h, h->arena_h);
                fold_l += F.layers[l].fl.cso[0];
                fold_b += F.layers[l].fl.cso[1];
            }
        }
        F.b_logs_f = (float)fold_l;
        F.b_b_f = (float)fold_b;
    }
    PK_TRY(pk_upload(ctx, h->arena, h->arena_h.data(), h->arena_h.size() * sizeof(float)));
    h->arena_h.clear();
    h->arena_h.shrink_to_fit();
    PK_TRY(pk_upload(ctx, h->arena16, h->arena16_h.data(), h->arena16_h.size() * sizeof(uint16_t)));
    h->arena16_h.clear();
    h->arena16_h.shrink_to_fit();
    h->finalized = true;
    return PK_OK;
}

static int wf_len_after(const pk_wf_cfg& c, int t_mel) {
    long t = t_mel;
    for (int i = 0; i < c.n_upsample; ++i) t = t * c.upsample_factors[i] - c.upsample_factors[i];
    return (int)std::max<long>(t, 0);
}

extern "C" int pk_wf_cond_length(pk_wf* h, int32_t t_mel, int32_t* cond_len, int32_t* wav_len) {
    if (!h || !cond_len || !wav_len) PK_FAIL(PK_EINVAL, "pk_wf_cond_length: NULL argument");
    const int t = wf_len_after(h->cfg, t_mel);
    *cond_len = t;
    *wav_len = t / h->cfg.n_group * h->cfg.n_group;
    return PK_OK;
}


// ---- profile build only (round 6, HISTORY 10): PK_WF_REPLAY="<launch>:<repeats>" replays ONE layer launch of the inference
// -- a flow's first layer: it writes prm instead of accumulating, so the launch is a pure function of buffers it does not
// modify -- `repeats` times with the kernel the options select, and compares every run with the 8-wave kernel's result of
// the same launch (bit-identical by construction).  Per differing 32-position tile: the wave that owned it, the positions
// whose (logs, b) sums differ, the positions / channel octets whose stored planes differ.
__global__ void k_wf_replay_compare(const unsigned* __restrict__ out, const unsigned* __restrict__ ref, const float* __restrict__ prm,
                                    const float* __restrict__ prm_ref, int ntiles, int C, unsigned* __restrict__ rec) {
    // rec[tile * 4 + {0: differing plane words, 1: position mask (planes), 2: octet mask (planes), 3: position mask (prm)}]
    const int tile = blockIdx.x, t = threadIdx.x;
    const int words = C * 32;   // C * 128 bytes per 32-position block
    unsigned n = 0, pm = 0, om = 0;
    for (int w = t; w < words; w += blockDim.x) {
        if (out[(size_t)tile * words + w] != ref[(size_t)tile * words + w]) {
            ++n;
            pm |= 1u << ((w & 127) >> 2);
            om |= 1u << (w >> 8);
        }
    }
    unsigned qm = 0;
    if (t < 64 && prm[(size_t)tile * 64 + t] != prm_ref[(size_t)tile * 64 + t]) qm = 1u << (t >> 1);
    if (n) atomicAdd(&rec[tile * 4], n);
    if (pm) atomicOr(&rec[tile * 4 + 1], pm);
    if (om) atomicOr(&rec[tile * 4 + 2], om);
    if (qm) atomicOr(&rec[tile * 4 + 3], qm);
    (void)ntiles;
}

extern "C" int pk_wf_infer(pk_wf* h, const float* mel, const int32_t* frames, int32_t B, const float* z,
                           float* wav, int32_t flags) {
    if (!h || !mel || !frames || !wav) PK_FAIL(PK_EINVAL, "pk_wf_infer: NULL argument");
    if (!h->finalized) PK_FAIL(PK_ESTATE, "pk_wf_infer: call pk_wf_finalize first");
    if (B <= 0) PK_FAIL(PK_EINVAL, "pk_wf_infer: batch size must be positive");
    pk_ctx* ctx = h->ctx;
    PK_DEVICE(ctx->device);
    const pk_wf_cfg& c = h->cfg;
    const int C = c.channels, M = c.n_mels, G = c.n_group, NL = c.n_layers, MP = h->mp;
    // ---- per-utterance sizes
    std::vector<int> cuT(B + 1, 0), clen(B), pruned(B), Wb(B), woff(B), zoff(B), ooff(B);
    long sumZ = 0, sumO = 0;
    int pos = h->gapw;
    for (int b = 0; b < B; ++b) {
        if (frames[b] < 2) PK_FAIL(PK_EINVAL, "pk_wf_infer: utterance %d needs >= 2 mel frames", b);
        cuT[b + 1] = cuT[b] + frames[b];
        clen[b] = wf_len_after(c, frames[b]);
        pruned[b] = clen[b] / G * G;
        Wb[b] = pruned[b] / G;
        if (Wb[b] <= 0) PK_FAIL(PK_EINVAL, "pk_wf_infer: utterance %d too short", b);
        woff[b] = pos;
        pos += Wb[b] + h->gapw;
        zoff[b] = (int)sumZ;
        ooff[b] = (int)sumO;
        sumZ += clen[b];
        sumO += pruned[b];
    }
    const int npos = pos;
    const int npos_alloc = ((npos + PK_GEMM_BM - 1) / PK_GEMM_BM) * PK_GEMM_BM;
    const long pstride = (long)npos_alloc + 2 * WF_LEAD;   // positions per buffer row incl. margins
    // ---- tables
    std::vector<int> pos_utt(npos_alloc, -1), pos_w(npos_alloc, 0);
    for (int b = 0; b < B; ++b)
        for (int w = 0; w < Wb[b]; ++w) {
            pos_utt[woff[b] + w] = b;
            pos_w[woff[b] + w] = w;
        }
    std::vector<int> tab;
    auto push = [&](const std::vector<int>& v) {
        size_t o = tab.size();
        tab.insert(tab.end(), v.begin(), v.end());
        return o;
    };
    const size_t o_putt = push(pos_utt), o_pw = push(pos_w), o_woff = push(woff), o_zoff = push(zoff),
                 o_ooff = push(ooff), o_pruned = push(pruned);
    // per upsample layer: in_off, in_len, out_off
    std::vector<size_t> o_inoff(c.n_upsample), o_inlen(c.n_upsample), o_outoff(c.n_upsample);
    std::vector<long> layer_rows(c.n_upsample + 1);
    {
        std::vector<int> len(frames, frames + B), off(cuT.begin(), cuT.begin() + B);
        layer_rows[0] = cuT[B];
        for (int i = 0; i < c.n_upsample; ++i) {
            const int f = c.upsample_factors[i];
            std::vector<int> olen(B), ooff2(B);
            long acc = 0;
            for (int b = 0; b < B; ++b) {
                olen[b] = f * len[b] - f;
                ooff2[b] = (int)acc;
                acc += olen[b];
            }
            o_inoff[i] = push(off);
            o_inlen[i] = push(len);
            o_outoff[i] = push(ooff2);
            layer_rows[i + 1] = acc;
            len = olen;
            off = ooff2;
        }
    }
    PK_TRY(h->ws_tab.reserve(tab.size() * sizeof(int)));
    PK_HIP(hipMemcpyAsync(h->ws_tab.p, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    PK_HIP(hipStreamSynchronize(ctx->stream));
    const int* d_tab = h->ws_tab.as<int>();

    // ---- io staging
    const float* d_mel = mel;
    const float* d_z = z;
    float* d_wav = wav;
    if (flags & PK_HOST_IO) {
        PK_TRY(h->ws_mel.reserve((size_t)cuT[B] * M * 4));
        PK_TRY(h->ws_wav.reserve((size_t)sumO * 4));
        PK_HIP(hipMemcpyAsync(h->ws_mel.p, mel, (size_t)cuT[B] * M * 4, hipMemcpyHostToDevice, ctx->stream));
        d_mel = h->ws_mel.as<float>();
        d_wav = h->ws_wav.as<float>();
        if (z) {
            PK_TRY(h->ws_z.reserve((size_t)sumZ * 4));
            PK_HIP(hipMemcpyAsync(h->ws_z.p, z, (size_t)sumZ * 4, hipMemcpyHostToDevice, ctx->stream));
            d_z = h->ws_z.as<float>();
        }
    }
    if (!z) {   // z = randn(...) (:801) drawn by the engine: next range of the handle's stream
        PK_TRY(h->ws_z.reserve((size_t)sumZ * 4));
        PK_TRY(pk_randn_device(ctx, h->ws_z.as<float>(), sumZ, h->seed, h->rng_offset));
        h->rng_offset += ((unsigned long long)sumZ + 3) / 4 * 4;
        d_z = h->ws_z.as<float>();
    }
    // ---- workspaces
    const long cond_row = pstride * MP;         // floats per folded cond row (channels padded to MP, pad = 0)
    const long feat_row = pstride * C;          // floats per [pos][C] buffer
    PK_TRY(h->ws_cond.reserve((size_t)G * cond_row * 4));
    PK_TRY(h->ws_cur.reserve((size_t)G * pstride * 4));
    PK_TRY(h->ws_nxt.reserve((size_t)G * pstride * 4));
    PK_TRY(h->ws_hist.reserve((size_t)(NL + 1) * 3 * feat_row * 4));
    PK_TRY(h->ws_zbuf.reserve((size_t)feat_row * 4));
    PK_TRY(h->ws_skip.reserve((size_t)feat_row * 4));
    // block scaling of the split-fp16 GEMMs (pk_split.h): max|row| of every hist / cond row, kept next to the data
    // so that a launch only scans the one row set that is new (zero = margins, gaps and rows not yet written)
    // (the fused layer kernel keeps block maxima instead, one per 32 positions, in the same buffers)
    const bool use_wfl = wfl_usable(h) && (h->math == PK_GEMM_MATH_F16X3 || h->math == PK_GEMM_MATH_F16);
    const long bstride = pstride / WFL_BLK;   // blocks per buffer row incl. margins
    PK_TRY(h->ws_bar.reserve(2 * sizeof(unsigned)));   // grid barrier counter | its time-out flag (pk_grid.h)
    PK_HIP(hipMemsetAsync(h->ws_bar.p, 0, 2 * sizeof(unsigned), ctx->stream));
    PK_TRY(h->ws_hamax.reserve((size_t)(NL + 1) * 3 * pstride * 4));
    PK_TRY(h->ws_camax.reserve((size_t)G * pstride * 4));
    PK_HIP(hipMemsetAsync(h->ws_hamax.p, 0, (size_t)(NL + 1) * 3 * pstride * 4, ctx->stream));
    PK_HIP(hipMemsetAsync(h->ws_camax.p, 0, (size_t)G * pstride * 4, ctx->stream));
    for (int i = 0; i + 1 < c.n_upsample; ++i) PK_TRY(h->ws_u[i & 1].reserve((size_t)layer_rows[i + 1] * M * 4));
    // cond gaps / margins are read by masked rows only, but must be finite
    PK_HIP(hipMemsetAsync(h->ws_cond.p, 0, (size_t)G * cond_row * 4, ctx->stream));
    PK_HIP(hipMemsetAsync(h->ws_hist.p, 0, (size_t)(NL + 1) * 3 * feat_row * 4, ctx->stream));
    PK_HIP(hipMemsetAsync(h->ws_zbuf.p, 0, (size_t)feat_row * 4, ctx->stream));
    float* cond = h->ws_cond.as<float>() + (size_t)WF_LEAD * MP;
    float* cur = h->ws_cur.as<float>() + WF_LEAD;
    float* nxt = h->ws_nxt.as<float>() + WF_LEAD;
    float* hist = h->ws_hist.as<float>() + (size_t)WF_LEAD * C;
    float* zbuf = h->ws_zbuf.as<float>() + (size_t)WF_LEAD * C;
    float* skip = h->ws_skip.as<float>() + (size_t)WF_LEAD * C;
    float* prm = skip;   // the fused path keeps two floats per position here instead of C
    auto hist_ptr = [&](int layer, int slot) { return hist + ((size_t)layer * 3 + slot) * feat_row; };
    float* hamax = h->ws_hamax.as<float>() + WF_LEAD;
    float* camax = h->ws_camax.as<float>() + WF_LEAD;
    auto hamax_ptr = [&](int layer, int slot) { return hamax + ((size_t)layer * 3 + slot) * pstride; };
    const bool split_math = h->math == PK_GEMM_MATH_F16X3;

    // ---- upsample (encoder)
    {
        const float* in = d_mel;
        for (int i = 0; i < c.n_upsample; ++i) {
            const int f = c.upsample_factors[i];
            const bool last = i == c.n_upsample - 1;
            float* out = last ? cond : h->ws_u[i & 1].as<float>();
            int maxT = 0;
            {
                // largest output length of this layer
                std::vector<int> len(frames, frames + B);
                for (int k = 0; k <= i; ++k)
                    for (int b = 0; b < B; ++b) len[b] = c.upsample_factors[k] * len[b] - c.upsample_factors[k];
                for (int b = 0; b < B; ++b) maxT = std::max(maxT, len[b]);
            }
            const int tpb = 256 / M;  // time steps per block
            dim3 grid(pk_div_up(maxT, tpb), 1, B);
            PK_LAUNCH(ctx, "wf_upsample", k_wf_upsample, grid, dim3(256), 0, in, d_tab + o_inoff[i], d_tab + o_inlen[i],
                      out, d_tab + o_outoff[i], h->W(h->up_w[i]), h->up_b[i], f, M, last ? G : 0, d_tab + o_woff,
                      d_tab + o_pruned, cond_row, MP, use_wfl ? 1 : 0);
            in = out;
        }
    }
    unsigned* hbmax = h->ws_hamax.as<unsigned>() + WF_LEAD / WFL_BLK;   // block maxima (fused layer kernel)
    unsigned* cbmax = h->ws_camax.as<unsigned>() + WF_LEAD / WFL_BLK;
    auto hbmax_ptr = [&](int layer, int slot) { return hbmax + ((size_t)layer * 3 + slot) * bstride; };
    if (use_wfl) PK_TRY(wfl_cond_planes_launch(ctx, cond, cond_row, G, npos_alloc / WFL_BLK, bstride, cbmax, M));   // in place
    else if (split_math) PK_TRY(pk_row_amax_launch(ctx, cond, MP, MP, 0, (long)G * pstride - WF_LEAD, camax));
    // ---- fold z
    PK_LAUNCH(ctx, "wf_fold", k_wf_fold, dim3(pk_div_up(npos, 256)), dim3(256), 0, d_z, d_tab + o_putt, d_tab + o_pw,
              d_tab + o_zoff, G, npos, pstride, cur);

    // profiling only (PK_WF_ABLATE=16): the layer kernel's s_memtime stamps, printed after the last launch
    unsigned long long* d_trace = nullptr;
    static const bool want_trace = pk_prof_env("PK_WF_ABLATE") && (atoi(pk_prof_env("PK_WF_ABLATE")) & ~64) == 16;
    static const bool want_verify = pk_prof_env("PK_WF_ABLATE") && atoi(pk_prof_env("PK_WF_ABLATE")) == 128;   // the idle wave's slab verifier
    constexpr size_t VERIFY_WORDS = 1 + 6 * 200;
    static const bool verify_off = pk_prof_env("PK_WF_VERIFY_OFF") != nullptr;   // the verifier's instantiation without its traffic (A/B of the two)
    if (want_trace || (want_verify && !verify_off)) {
        const size_t bytes = std::max((size_t)12 * 2 * 24, VERIFY_WORDS) * sizeof(unsigned long long);
        PK_TRY(h->ws_trace.reserve(bytes));
        PK_HIP(hipMemsetAsync(h->ws_trace.p, 0, bytes, ctx->stream));
        d_trace = h->ws_trace.as<unsigned long long>();
    }
    int launch_seq = 0;
    static const char* replay_env = pk_prof_env("PK_WF_REPLAY");   // "<launch>:<repeats>" (see k_wf_replay_compare)
    const int replay_seq = replay_env ? atoi(replay_env) : -1;
    const int replay_reps = replay_env && strchr(replay_env, ':') ? atoi(strchr(replay_env, ':') + 1) : 0;
    // ---- the layer descriptors of the fused kernel, one per (flow, ring slot of the current row, layer): weights of the
    // (flow, layer), input ring of the layer, output = the next layer's ring at the current row's slot
    if (use_wfl) {
        h->desc_host.assign((size_t)c.n_flows * 3 * NL, WflLayer());
        for (int fl = 0; fl < c.n_flows; ++fl)
            for (int slot = 0; slot < 3; ++slot)
                for (int l = 0; l < NL; ++l) {
                    const WfLayerW& L = h->flows[fl].layers[l];
                    WflLayer& wl = h->desc_host[((size_t)fl * 3 + slot) * NL + l];
                    wl.w.w1 = h->arena16.as<uint16_t>() + L.fl.w1;
                    wl.w.w2 = h->arena16.as<uint16_t>() + L.fl.w2;
                    wl.w.b2r = h->W(L.fl.b2r);
                    wl.w.wso = h->W(L.fl.wso);
                    wl.w.k1 = L.fl.k1;
                    wl.w.k2res = L.fl.k2res;
                    wl.in0 = hist_ptr(l, 0);
                    wl.in_amax0 = hbmax_ptr(l, 0);
                    wl.out = l + 1 < NL ? hist_ptr(l + 1, slot) : nullptr;   // the last layer's residual output is unused (:390)
                    wl.out_amax = l + 1 < NL ? hbmax_ptr(l + 1, slot) : nullptr;
                    wl.first = l == 0;
                    wl.dil = 1 << l;
                }
        PK_TRY(h->ws_desc.reserve(h->desc_host.size() * sizeof(WflLayer)));
        PK_HIP(hipMemcpyAsync(h->ws_desc.p, h->desc_host.data(), h->desc_host.size() * sizeof(WflLayer), hipMemcpyHostToDevice, ctx->stream));
    }
    // ---- flows, reversed (:703-706)
    std::vector<int> cidx(G);
    for (int i = 0; i < G; ++i) cidx[i] = i;
    const int* rowvalid = d_tab + o_putt;
    for (int fl = c.n_flows - 1; fl >= 0; --fl) {
        std::vector<int> perm(G);   // _create_perm :602-615
        for (int i = 0; i < G; ++i)
            perm[i] = (fl < c.n_flows / 2) ? (G - 1 - i) : (i < G / 2 ? G / 2 - 1 - i : G + G / 2 - 1 - i);
        std::vector<int> cnew(G);
        for (int i = 0; i < G; ++i) cnew[i] = cidx[perm[i]];   // cumulative shuffle of the condition
        cidx = cnew;
        const WfFlowW& F = h->flows[fl];
        // row 0: copy + input_proj into slot 1 of layer 0
        if (use_wfl) {
            PK_TRY(wfl_step_launch(ctx, C, prm, F.b_logs_f, F.b_b_f, cur + (long)perm[0] * pstride, nxt,
                                   h->W(F.w_in), h->W(F.b_in), hist_ptr(0, 1), hbmax_ptr(0, 1), rowvalid, npos_alloc, 1));
        } else {
        PK_LAUNCH(ctx, "wf_step", k_wf_step, dim3(pk_div_up(npos, 4)), dim3(256), 0, skip, C, h->W(F.w_out), F.b_logs,
                  F.b_b, cur + (long)perm[0] * pstride, nxt, h->W(F.w_in), h->W(F.b_in), hist_ptr(0, 1), rowvalid,
                  npos, 1);
        if (split_math) PK_TRY(pk_row_amax_launch(ctx, hist_ptr(0, 1), C, C, 0, npos, hamax_ptr(0, 1)));
        }
        // the fused kernel needs the 64-channel shape (one 128-column block) and the split-fp16 path
        const bool fuse_proj = C == 64 && h->math == PK_GEMM_MATH_F16X3 && MP % PK_GEMM_HBK == 0 && !h->no_fuse;
        for (int i = 1; i < G; ++i) {
            const int slot = i % 3;
            if (use_wfl) {
                // the fused layer kernel (wf_layer.hip): conv taps + condition + gate + res projection + folded skip path.
                // One launch runs `per` consecutive layers of this row: all NL behind grid barriers (option "persistent",
                // OFF by default -- 632 us per row against 8 x 57 us, DESIGN 4.3 -- 120 launches per batch instead of 960 + 120), or one; the launch that
                // holds the last layer also finishes the row (the step: x[i], then the next row's layer-0 input).
                WflLaunch w;
                memset(&w, 0, sizeof(w));
                w.C = C;
                w.f16 = h->math == PK_GEMM_MATH_F16;
                w.slot_stride = feat_row;
                w.amax_stride = bstride;
                w.cur_slot = slot;
                w.prm = prm;
                w.cond = cond + (long)cidx[i] * cond_row;
                w.cond_amax = cbmax + (long)cidx[i] * bstride;
                w.ntap = 0;
                for (int kr = 0; kr < 3; ++kr) {
                    const int step = i - 2 + kr;   // kernel row kr reads the layer input of this step
                    if (step < 1) continue;        // rows before the sequence start are zeros (:287-290)
                    for (int kc = 0; kc < 3; ++kc) {
                        w.tap_slot[w.ntap] = step % 3;
                        w.tap_col[w.ntap] = kc - 1;
                        w.tap_w[w.ntap] = kr * 3 + kc;
                        ++w.ntap;
                    }
                }
                w.pos_utt = rowvalid;
                w.npos_alloc = npos_alloc;
                w.trace = d_trace;
                w.seq = launch_seq;      // (flow, row, layer) = (n_flows - 1 - seq / ((G - 1) NL), 1 + seq / NL % (G - 1), seq % NL)
                w.waves = h->layer_waves;
                w.bar = h->ws_bar.as<unsigned>();
                w.err = h->ws_bar.as<int>() + 1;
                const bool persistent = h->persistent && pk_grid_available() && NL <= WFL_MAX_LAYERS && !d_trace;
                const int per = persistent ? NL : 1;
                const bool fuse = h->fuse_step && C == 64;   // (the 128-channel kernel has no registers for the fused step)
                float* h0n = (i + 1 < G) ? hist_ptr(0, (i + 1) % 3) : nullptr;
                for (int l0 = 0; l0 < NL; l0 += per) {
                    w.nl = per;
                    w.layers = h->ws_desc.as<WflLayer>() + ((size_t)(fl * 3 + slot) * NL + l0);   // (flow, ring slot, layer)
                    w.l0 = h->desc_host[(size_t)(fl * 3 + slot) * NL + l0];                        // ... and its host image (per == 1)
                    if (l0 + per == NL && fuse) {
                        w.step_z = cur + (long)perm[i] * pstride;
                        w.step_x = nxt + (long)i * pstride;
                        w.step_w_in = h->W(F.w_in);
                        w.step_b_in = h->W(F.b_in);
                        w.step_h0 = h0n;
                        w.step_h0_amax = h0n ? hbmax_ptr(0, (i + 1) % 3) : nullptr;
                        w.step_b_logs = F.b_logs_f;
                        w.step_b_b = F.b_b_f;
                    }
                    if (PK_PROFILE_BUILD && replay_reps > 0 && launch_seq == replay_seq && per == 1 && w.l0.first && w.l0.out) {
                        const int ntl = npos_alloc / 32;
                        const size_t out_bytes = (size_t)npos_alloc * C * 4, prm_bytes = (size_t)npos_alloc * 8;
                        PK_TRY(h->ws_replay.reserve(out_bytes + prm_bytes + (size_t)ntl * 16));
                        char* rb = h->ws_replay.as<char>();
                        unsigned* rec = reinterpret_cast<unsigned*>(rb + out_bytes + prm_bytes);
                        WflLaunch r8 = w;
                        r8.waves = 8;
                        PK_TRY(wfl_layer_launch(ctx, r8));
                        PK_HIP(hipMemcpyAsync(rb, w.l0.out, out_bytes, hipMemcpyDeviceToDevice, ctx->stream));
                        PK_HIP(hipMemcpyAsync(rb + out_bytes, prm, prm_bytes, hipMemcpyDeviceToDevice, ctx->stream));
                        std::vector<unsigned> rec_h((size_t)ntl * 4);
                        const int tpw = std::max(1, (ntl + ctx->n_cu - 1) / ctx->n_cu);
                        long events = 0, runs_bad = 0;
                        std::vector<long> by_wave(16, 0), by_pos_prm(32, 0), by_oct(16, 0), by_npos(33, 0);
                        for (int rep = 0; rep < replay_reps; ++rep) {
                            PK_HIP(hipMemsetAsync(rec, 0, (size_t)ntl * 16, ctx->stream));
                            PK_TRY(wfl_layer_launch(ctx, w));
                            hipLaunchKernelGGL(k_wf_replay_compare, dim3(ntl), dim3(256), 0, ctx->stream, reinterpret_cast<const unsigned*>(w.l0.out),
                                               reinterpret_cast<const unsigned*>(rb), prm, reinterpret_cast<const float*>(rb + out_bytes), ntl, C, rec);
                            PK_HIP(hipMemcpyAsync(rec_h.data(), rec, (size_t)ntl * 16, hipMemcpyDeviceToHost, ctx->stream));
                            PK_HIP(hipStreamSynchronize(ctx->stream));
                            bool any = false;
                            for (int t = 0; t < ntl; ++t) {
                                const unsigned* e = &rec_h[(size_t)t * 4];
                                if (!e[0] && !e[3]) continue;
                                any = true;
                                ++events;
                                ++by_wave[(t % tpw) & 15];
                                ++by_npos[__builtin_popcount(e[3])];
                                for (int b = 0; b < 32; ++b) by_pos_prm[b] += (e[3] >> b) & 1;
                                for (int b = 0; b < 16; ++b) by_oct[b] += (e[2] >> b) & 1;
                                if (events <= 24)
                                    fprintf(stderr, "wf_replay: run %d tile %d = workgroup %d wave %d: %u plane words differ, positions %08x octets %04x; prm positions %08x\n",
                                            rep, t, t / tpw, t % tpw, e[0], e[1], e[2], e[3]);
                                if (events <= 6 && e[3]) {   // the (logs, b) sums of the tile: got | expected, per position
                                    float got[64], want[64];
                                    PK_HIP(hipMemcpy(got, prm + (size_t)t * 64, sizeof(got), hipMemcpyDeviceToHost));
                                    PK_HIP(hipMemcpy(want, rb + out_bytes + (size_t)t * 256, sizeof(want), hipMemcpyDeviceToHost));
                                    for (int q = 0; q < 32; ++q)
                                        fprintf(stderr, "wf_replay:   pos %2d logs %.9g (want %.9g, diff %.4g)  b %.9g (want %.9g, diff %.4g)\n", q, got[2 * q], want[2 * q],
                                                got[2 * q] - want[2 * q], got[2 * q + 1], want[2 * q + 1], got[2 * q + 1] - want[2 * q + 1]);
                                }
                            }
                            runs_bad += any;
                        }
                        fprintf(stderr, "wf_replay: launch %d (ntap %d, %d tiles, %d per workgroup): %ld of %d runs differ from the 8-wave kernel, %ld tiles in all\n",
                                launch_seq, w.ntap, ntl, tpw, runs_bad, replay_reps, events);
                        fprintf(stderr, "wf_replay: tiles by wave of the workgroup:");
                        for (int i = 0; i < 12; ++i) fprintf(stderr, " %ld", by_wave[i]);
                        fprintf(stderr, "\nwf_replay: tiles by number of positions whose (logs, b) differ (0 .. 32):");
                        for (int i = 0; i <= 32; ++i) fprintf(stderr, " %ld", by_npos[i]);
                        fprintf(stderr, "\nwf_replay: by position:");
                        for (int i = 0; i < 32; ++i) fprintf(stderr, " %ld", by_pos_prm[i]);
                        fprintf(stderr, "\nwf_replay: by channel octet of the stored planes:");
                        for (int i = 0; i < C / 8; ++i) fprintf(stderr, " %ld", by_oct[i]);
                        fprintf(stderr, "\n");
                    }
                    PK_TRY(wfl_layer_launch(ctx, w));
                    w.seq = ++launch_seq;
                }
                if (!fuse)
                    PK_TRY(wfl_step_launch(ctx, C, prm, F.b_logs_f, F.b_b_f, cur + (long)perm[i] * pstride,
                                           nxt + (long)i * pstride, h->W(F.w_in), h->W(F.b_in), h0n,
                                           h0n ? hbmax_ptr(0, (i + 1) % 3) : nullptr, rowvalid, npos_alloc, 0));
                continue;
            }
            for (int l = 0; l < NL && !use_wfl; ++l) {
                const WfLayerW& L = F.layers[l];
                pk_gemm_args g;
                g.A = hist_ptr(l, 0);   // taps carry the slot offsets
                g.lda = C;
                g.Cin = C;
                g.ntaps = 0;
                const long dil = 1L << l;
                for (int kr = 0; kr < 3; ++kr) {
                    const int step = i - 2 + kr;   // kernel row kr reads the layer input of this step
                    if (step < 1) continue;        // rows before the sequence start are zeros (:287-290)
                    for (int kc = 0; kc < 3; ++kc) {
                        g.tap_off[g.ntaps] = (long)(step % 3) * feat_row + (long)(kc - 1) * dil * C;
                        g.tap_w[g.ntaps] = kr * 3 + kc;
                        ++g.ntaps;
                    }
                }
                g.a_amax = hamax_ptr(l, 0);
                g.a2_amax = camax + (long)cidx[i] * pstride;
                g.A2 = cond + (long)cidx[i] * cond_row;
                g.lda2 = MP;
                g.Cin2 = MP;
                g.w2_slab0 = 9 * C / PK_GEMM_BK;
                g.wslabs_total = (9 * C + MP) / PK_GEMM_BK;
                g.Wp = h->W(L.w1);
                g.Wh = h->arena16.as<uint16_t>() + L.w1h;
                g.math = h->math;
                g.bias = h->W(L.b1);
                g.rowvalid = rowvalid;
                g.M = npos;
                g.N = 2 * C;
                if (fuse_proj) {
                    // conv + gate + res|skip projection in one launch: z never reaches HBM
                    g.epi = PK_EPI_GATE_PROJ;
                    g.Wh2 = h->arena16.as<uint16_t>() + L.w2h;
                    g.bias2 = h->W(L.b2);
                    g.res = hist_ptr(l, slot);
                    g.ldr = C;
                    g.C = hist_ptr(l + 1, slot);
                    g.ldc = C;
                    g.nsplit = C;
                    g.C2 = skip;
                    g.ldc2 = C;
                    g.acc2 = l > 0;
                    PK_TRY(pk_gemm_launch(ctx, "wf_gemm_conv_gate_proj", g));
                    if (l + 1 < NL) PK_TRY(pk_row_amax_launch(ctx, hist_ptr(l + 1, slot), C, C, 0, npos, hamax_ptr(l + 1, slot)));
                    continue;
                }
                g.epi = PK_EPI_GATE;
                g.C = zbuf;
                g.ldc = C;
                PK_TRY(pk_gemm_launch(ctx, "wf_gemm_conv_gate", g));
                pk_gemm_args o;
                o.A = zbuf;
                o.lda = C;
                o.Cin = C;
                o.taps = 1;
                o.pad = 0;
                o.Wp = h->W(L.w2);
                o.Wh = h->arena16.as<uint16_t>() + L.w2h;
                o.math = h->math;
                o.bias = h->W(L.b2);
                o.res = hist_ptr(l, slot);
                o.ldr = C;
                o.C = hist_ptr(l + 1, slot);
                o.ldc = C;
                o.nsplit = C;
                o.C2 = skip;
                o.ldc2 = C;
                o.acc2 = l > 0;
                o.rowvalid = rowvalid;
                o.M = npos;
                o.N = 2 * C;
                PK_TRY(pk_gemm_launch(ctx, "wf_gemm_out_proj", o));
                if (split_math && l + 1 < NL)
                    PK_TRY(pk_row_amax_launch(ctx, hist_ptr(l + 1, slot), C, C, 0, npos, hamax_ptr(l + 1, slot)));
            }
            float* h0n = (i + 1 < G) ? hist_ptr(0, (i + 1) % 3) : nullptr;
            PK_LAUNCH(ctx, "wf_step", k_wf_step, dim3(pk_div_up(npos, 4)), dim3(256), 0, skip, C, h->W(F.w_out),
                      F.b_logs, F.b_b, cur + (long)perm[i] * pstride, nxt + (long)i * pstride, h->W(F.w_in),
                      h->W(F.b_in), h0n, rowvalid, npos, 0);
            if (split_math && h0n) PK_TRY(pk_row_amax_launch(ctx, h0n, C, C, 0, npos, hamax_ptr(0, (i + 1) % 3)));
        }
        std::swap(cur, nxt);
    }
    if (d_trace && want_verify) {
        std::vector<unsigned long long> tr(VERIFY_WORDS);
        PK_HIP(hipMemcpyAsync(tr.data(), d_trace, VERIFY_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
        PK_HIP(hipStreamSynchronize(ctx->stream));
        fprintf(stderr, "wf_verify: %llu LDS weight chunks differed from memory in %d layer launches\n", tr[0], launch_seq);
        for (unsigned long long k = 0; k < std::min<unsigned long long>(tr[0], 200); ++k) {
            const unsigned long long* r = tr.data() + 1 + 6 * k;
            fprintf(stderr, "wf_verify: launch %llu workgroup %llu slab %llu chunk %llu (idle wave %llu, %llu working waves): got %016llx %016llx want %016llx %016llx\n",
                    r[0] >> 32, (r[0] >> 16) & 0xffff, (r[0] >> 12) & 15, r[0] & 0xfff, r[1] >> 32, r[1] & 0xffffffffu, r[2], r[3], r[4], r[5]);
        }
    } else if (d_trace) {
        unsigned long long tr[12 * 2 * 24];
        PK_HIP(hipMemcpyAsync(tr, d_trace, sizeof(tr), hipMemcpyDeviceToHost, ctx->stream));
        PK_HIP(hipStreamSynchronize(ctx->stream));
        unsigned long long t0 = ~0ull;
        for (unsigned long long v : tr) if (v && v < t0) t0 = v;
        for (int wv = 0; wv < 12; ++wv)
            for (int r = 0; r < 2; ++r) {
                if (!tr[(wv * 2 + r) * 24]) continue;
                fprintf(stderr, "wf_trace wave %d round %d:", wv, r);
                for (int i = 0; i < 24; ++i) fprintf(stderr, " %lld", tr[(wv * 2 + r) * 24 + i] ? (long long)(tr[(wv * 2 + r) * 24 + i] - t0) : -1LL);
                fprintf(stderr, "\n");
            }
    }
    PK_LAUNCH(ctx, "wf_unfold", k_wf_unfold, dim3(pk_div_up(npos, 256)), dim3(256), 0, cur, d_tab + o_putt,
              d_tab + o_pw, d_tab + o_ooff, G, npos, pstride, d_wav);
    // The row kernel's grid barrier gives up after a bounded spin and raises ws_bar[1]: a waveform computed past a timed-out
    // barrier is garbage, so the flag is checked on EVERY path that ran the row kernel -- with device-resident output too,
    // at the price of one 4-byte copy and a stream synchronisation (the option is off by default and slower than the
    // per-layer launches; a caller who pipelines does not want it anyway).
    const bool ran_row_kernel = h->persistent && use_wfl && pk_grid_available() && NL <= WFL_MAX_LAYERS && !d_trace;
    if (flags & PK_HOST_IO) PK_HIP(hipMemcpyAsync(wav, d_wav, (size_t)sumO * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (ran_row_kernel) {
        int berr = 0;
        PK_HIP(hipMemcpyAsync(&berr, h->ws_bar.as<int>() + 1, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        PK_HIP(hipStreamSynchronize(ctx->stream));
        if (berr) PK_FAIL(PK_EHIP, "pk_wf_infer: a grid barrier of the row kernel timed out (code %d): the result is invalid", berr);
    } else if (flags & PK_HOST_IO) {
        PK_HIP(hipStreamSynchronize(ctx->stream));
    }
    return PK_OK;
}

extern "C" void pk_wf_destroy(pk_wf* h) {
    if (!h) return;
    pk_device_guard _dg(h->ctx->device);
    (void)hipStreamSynchronize(h->ctx->stream);
    pk_dbuf* bufs[] = {&h->arena, &h->arena16, &h->ws_tab, &h->ws_mel, &h->ws_z, &h->ws_wav, &h->ws_u[0], &h->ws_u[1],
                       &h->ws_cond, &h->ws_cur, &h->ws_nxt, &h->ws_hist, &h->ws_zbuf, &h->ws_skip,
                       &h->ws_hamax, &h->ws_camax, &h->ws_trace, &h->ws_bar, &h->ws_desc, &h->ws_replay};
    for (auto* b : bufs) b->release();
    delete h;
}
