/* pk_synth.h -- C ABI of libpk_synth.so, the MI355X (gfx950) synthesis engine
 * behind Parakeet's inference API.
 *
 * The reference (PaddlePaddle/Parakeet) is pure Python over Paddle ops; it has
 * no FFI of its own.  The seam this ABI replaces is the Python class API of
 * the synthesis path; each entry point names the reference method whose device
 * work it performs (paths relative to the reference repository root):
 *
 *   pk_pwg_*    parakeet/models/parallel_wavegan/parallel_wavegan.py
 *               PWGGenerator.__init__ :369-443, set_state_dict,
 *               remove_weight_norm :485-496, inference :498-520 (forward :445-472),
 *               PWGInference.forward :772-775
 *   pk_fs2_*    parakeet/models/fastspeech2/fastspeech2.py
 *               FastSpeech2.__init__ :52-296, inference :468-558
 *               (_forward(is_inference=True) :377-466),
 *               FastSpeech2Inference.forward :668-671
 *   pk_wf_*     parakeet/models/waveflow.py ConditionalWaveFlow.infer :785-805
 *   pk_tts_*    parakeet/models/transformer_tts/transformer_tts.py TransformerTTS.inference :511-647,
 *               TransformerTTSInference.forward :757-767
 *   pk_taco_*   parakeet/models/tacotron2.py Tacotron2.infer :781-840 (Tacotron2Decoder.infer :474-541)
 *   pk_stft_mel parakeet/modules/audio.py STFT.magnitude :202-215 + MelScale :226-229,
 *               parakeet/data/get_feats.py LogMelFBank.get_log_mel_fbank :80-88
 *
 * Conventions
 *   - Every function returns PK_OK (0) or a negative pk_status; the message of
 *     the last failure on the calling thread is pk_last_error().  Nothing
 *     aborts.  The Python shim maps the codes back to the exception classes the
 *     reference raises (ValueError / AssertionError / NotImplementedError).
 *   - Plain pointers and sizes only.  Parameter arrays handed to *_set_param are
 *     HOST pointers and are copied (the caller may free them at once, like
 *     set_state_dict).  Data pointers of the compute calls are DEVICE pointers
 *     unless the call's `flags` has PK_HOST_IO, in which case they are host
 *     pointers and the engine stages them (synchronous on return).
 *   - A pk_ctx owns one HIP device + one stream.  Handles created on a context
 *     are not thread-safe; different contexts may be used concurrently.  Calls
 *     are asynchronous on the context's stream unless stated; use pk_sync().
 *   - Batches are "packed ragged": utterance b owns rows [cu[b], cu[b+1]) of a
 *     row-major array; lengths are given per utterance on the host.
 */
#ifndef PK_SYNTH_H
#define PK_SYNTH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    PK_OK = 0,
    PK_EINVAL = -1,        /* bad argument value              -> ValueError          */
    PK_ESHAPE = -2,        /* shape / size mismatch           -> AssertionError      */
    PK_EUNSUPPORTED = -3,  /* configuration not implemented   -> NotImplementedError */
    PK_EHIP = -4,          /* HIP runtime failure             -> RuntimeError        */
    PK_ENOMEM = -5,        /* allocation failure              -> MemoryError         */
    PK_ESTATE = -6         /* call order (e.g. not finalized) -> RuntimeError        */
} pk_status;

enum {
    PK_HOST_IO = 1,            /* flags bit: data pointers are host memory */
    PK_PWG_C_HAS_CONTEXT = 2,  /* pk_pwg_infer: mel rows already carry aux_context_window frames on both
                                  sides of every utterance (PWGGenerator.forward); default: the engine
                                  replicate-pads (PWGGenerator.inference) */
    PK_APPLY_NORMALIZER = 4,   /* pk_fs2_decode / pk_ss_decode / pk_pwg_infer: apply the ZScore registered with
                                  pk_*_set_normalizer in THIS call (the *Inference wrappers of the reference:
                                  FastSpeech2Inference.forward fastspeech2.py:668-671, PWGInference.forward
                                  parallel_wavegan.py:772-775, SpeedySpeechInference.forward :221-231).
                                  Without it the call stays in the model's own (normalised) domain, as
                                  model.inference() does in the reference -- the registered statistics are
                                  per-handle state, their use is per call */
    PK_TTS_KEEP_ATT = 8        /* pk_tts_infer: keep the encoder-decoder attention weights (TransformerTTS.inference's
                                  third return value) for pk_tts_read */
};

typedef struct pk_ctx pk_ctx;
typedef struct pk_pwg pk_pwg;
typedef struct pk_fs2 pk_fs2;
typedef struct pk_wf pk_wf;
typedef struct pk_mel pk_mel;

/* ---------------------------------------------------------------- context */
const char* pk_last_error(void);
const char* pk_version(void);
/* Create a context on HIP device `device_id` with its own stream. */
int pk_ctx_create(int device_id, pk_ctx** out);
/* Run on an externally owned hipStream_t (e.g. torch's current stream), so
 * that events recorded by the caller on that stream bracket the kernels. */
int pk_ctx_set_stream(pk_ctx* ctx, void* hip_stream);
int pk_sync(pk_ctx* ctx);
void pk_ctx_destroy(pk_ctx* ctx);

/* Per-kernel timing: when enabled every launch is bracketed by HIP events on
 * the context's stream.  pk_prof_read() synchronises and returns, for the
 * kernel class `name`, the number of launches and their summed duration (ms)
 * since the last pk_prof_reset().  Used by bench.py for roofline.achieved. */
int pk_prof_enable(pk_ctx* ctx, int on);
int pk_prof_reset(pk_ctx* ctx);
int pk_prof_read(pk_ctx* ctx, const char* name, int64_t* launches, double* total_ms);
/* Writes a '\n'-separated "name launches total_ms" listing into buf. */
int pk_prof_dump(pk_ctx* ctx, char* buf, int64_t buflen);

/* ------------------------------------------------------ Parallel WaveGAN */
/* Constructor arguments of PWGGenerator (parallel_wavegan.py:369-388) that
 * change device work.  Unsupported combinations -> PK_EUNSUPPORTED. */
typedef struct {
    int32_t in_channels;        /* 1 */
    int32_t out_channels;       /* 1 */
    int32_t kernel_size;        /* 3 */
    int32_t layers;             /* 30, must be a multiple of stacks (:398) */
    int32_t stacks;             /* 3  */
    int32_t residual_channels;  /* 64 */
    int32_t gate_channels;      /* 128 */
    int32_t skip_channels;      /* 64 */
    int32_t aux_channels;       /* 80 */
    int32_t aux_context_window; /* 2 */
    int32_t n_upsample;         /* 4 */
    int32_t upsample_scales[8]; /* 4,4,4,4 (LJSpeech, hop 256) or 4,5,3,5 (baker / vctk, hop 300); any product 32..1024 */
    int32_t use_causal_conv;    /* 0 only */
} pk_pwg_cfg;

int pk_pwg_create(pk_ctx* ctx, const pk_pwg_cfg* cfg, pk_pwg** out);
/* set_state_dict, one entry at a time.  `name` is the reference's state-dict
 * key ("conv_layers.3.conv.weight", "...weight_g", "...weight_v", ...);
 * float32 host data of the given shape. */
int pk_pwg_set_param(pk_pwg* h, const char* name, const float* data,
                     const int64_t* shape, int32_t ndim);
/* PWGInference's normalizer (ZScore, parakeet/modules/normalizer.py:18-33):
 * mel_in -> (mel_in - mu) / sigma.  NULL,NULL = identity. */
int pk_pwg_set_normalizer(pk_pwg* h, const float* mu, const float* sigma, int32_t n);
/* Arithmetic of the residual-block contractions (activations, weights, accumulators and every stored
 * tensor are fp32 in all modes).  Default PK_PWG_MATH_F16X3.
 *   PK_PWG_MATH_F32     exact fp32 products on v_mfma_f32_32x32x2_f32 (bitwise an fmaf chain);
 *   PK_PWG_MATH_F16X3   each fp32 product as a_hi*b_hi + a_lo*b_hi + a_hi*b_lo with fp16 parts
 *                       (11 + 11 significant bits per operand, dropped term 2^-22) on
 *                       v_mfma_f32_32x32x16_f16, fp32 accumulation.  Measured on the 30-layer generator:
 *                       relative max error 7e-7 (rms 2.3e-7) vs the fp64 oracle -- the class of the exact
 *                       path (5e-7, rms 1.9e-7) and of a CPU fp32 run (6e-7).  5.3x less matrix-pipe time.
 *                       Operands are block scaled by powers of two before the split (csrc/pk_split.h): the
 *                       error does not depend on the magnitude of weights or activations (floor 2^-39 of the
 *                       block maximum, fp32 range);
 *   PK_PWG_MATH_BF16X3  the same with bf16 parts (fp32 range, 8 + 8 bits): error 3.7e-6. */
enum { PK_PWG_MATH_F32 = 0, PK_PWG_MATH_BF16X3 = 1, PK_PWG_MATH_F16X3 = 2 };
int pk_pwg_set_math(pk_pwg* h, int32_t mode);
/* Scheduling of the residual stack (no effect on results): the batch is processed in chunks of whole
 * utterances of at most `samples` samples, all layers over one chunk before the next, so that the chunk's
 * activations (768 B per sample) stay in the 256 MB Infinity Cache between layers (327 680 = two 7.4 s
 * utterances = 252 MB).  Default: one chunk (measured faster, see pwg.hip). */
int pk_pwg_set_chunk_samples(pk_pwg* h, int64_t samples);
/* Named integer options of a handle.  The library reads NO environment variable (the measurement / ablation switches of
 * earlier rounds exist only in the profile build, parakeet_amd/build.py build(profile=True)); what a caller may choose
 * is listed here.  Unknown key or value -> PK_EINVAL.
 *   "planes"       1 (default) = under PK_PWG_MATH_F16X3 the residual stream x is stored as pre-split fp16 planes at one
 *                  a-priori scale per utterance and layer (pwg.hip, k_pwg_tile_scales); 0 = x stored as fp32 with one
 *                  measured scale per 32-sample block (round 2's path: ~4 % slower, no magnitude bound involved).
 *   "scale_guard"  0 = off; 1 (default) = the FIRST inference after pk_pwg_finalize also measures max|x| per utterance and
 *                  layer on the planes path, and when the a-priori bound overshoots the measured maximum by more than
 *                  2^10 anywhere, the call is repeated on the "planes" = 0 path, which the handle then keeps; 2 = every call.
 *                  A guarded call synchronises the stream once and does two small blocking copies INSIDE pk_pwg_infer: the
 *                  first call after finalize stalls a pipelined caller and must not be stream-captured.
 *                  Under 1 every "scale_guard_every"-th later inference is SAMPLED as well (the layer kernel folds each layer's maxima
 *                  in from its epilogue on such calls: one extra small launch, not a pass over x per layer), its
 *                  verdict deferred: the maxima are copied to pinned host memory behind an event and judged at the start of
 *                  a later call -- no stall, no allocation (the pinned buffer, 0.5 MB for up to 4096 utterances, and the
 *                  event are created by pk_pwg_finalize; a call with more utterances is not sampled; a sample whose event
 *                  reports an error is dropped, not judged); a verdict above 2^10 moves the handle to the "planes" = 0
 *                  path from then on (the sampled call keeps its result: at 2^10 the planes still carry 26 bits of the
 *                  actual maximum).  A sampled call adds two device-to-host copies and an event record to the stream: like
 *                  a guarded call it must not be stream-captured (capture with "scale_guard" 0).
 *   "noise_fed_first"  1 (default) = on the "planes" path at hop 256 the first residual block is fed from the noise itself: first_conv is
 *                  Conv1D(1 -> 64, k = 1), so the block's dilated conv over x = w n + b is a 3-tap conv on n with weights folded (in fp64) by
 *                  pk_pwg_finalize -- no first_conv launch, no x planes for layer 0 (round 6); 0 = first_conv, then the ordinary first block.
 *                  Both are within the engine's error bars of the reference; they differ from each other in the last bits.
 *   "scale_guard_every"  the sampling period under "scale_guard" 1 (default 16; 0 = never re-sample).
 *                  pk_pwg_scale_overshoot reports what was measured. */
int pk_pwg_set_option(pk_pwg* h, const char* key, int64_t value);
/* log2(a-priori bound / measured max|x|) per layer input, l = 0 .. layers (n = layers + 1 floats), the maximum over the
 * utterances of the last guarded or sampled inference ("scale_guard"; a sample still in flight is waited for); *fell_back
 * (nullable) = 1 when the handle has left the planes path inside a guarded call, 2 when a deferred sample moved it there.
 * PK_ESTATE if no guarded inference has run. */
int pk_pwg_scale_overshoot(pk_pwg* h, float* log2_overshoot, int32_t n, int32_t* fell_back);
/* remove_weight_norm + packing into the kernels' layouts + upload. */
int pk_pwg_finalize(pk_pwg* h);
/* PWGGenerator.inference for a packed batch.
 *   mel    (sum(frames), aux_channels) float32, row-major, packed by utterance
 *          (with PK_PWG_C_HAS_CONTEXT: frames[b] + 2*ctx rows per utterance)
 *   frames (B) host int32, frames per utterance (>= 1)
 *   noise  (sum(frames)*hop) float32 packed; the x = randn(...) of :515-516, passed in so that
 *          results are reproducible; NULL = drawn internally (pk_randn stream of pk_pwg_set_seed)
 *   wav    (sum(frames)*hop) float32 packed output
 * flags: PK_HOST_IO if mel/noise/wav are host pointers. */
int pk_pwg_infer(pk_pwg* h, const float* mel, const int32_t* frames, int32_t B,
                 const float* noise, float* wav, int32_t flags);
/* Debug / test taps: copy internal activations of the LAST pk_pwg_infer call
 * for utterance b to host.  what: 0 = layer 0's conv1x1_aux(upsampled c) (gate, S_b),
 * 1 = residual-stack output x (residual, S_b), 2 = skip sum before the sqrt(1/layers)
 * scale (skip, S_b), all channel-major. */
int pk_pwg_debug_read(pk_pwg* h, int32_t what, int32_t b, float* host_out, int64_t n_floats);
void pk_pwg_destroy(pk_pwg* h);

/* ------------------------------------------------------------ FastSpeech2 */
/* Constructor arguments of FastSpeech2 (fastspeech2.py:52-118) that change the
 * inference computation.  Dropout rates, init_type and the loss switches are
 * training-only and have no field.  Unsupported values -> PK_EUNSUPPORTED. */
typedef struct {
    int32_t idim, odim;
    int32_t adim, aheads;
    int32_t elayers, eunits, dlayers, dunits;
    int32_t positionwise_conv_kernel_size;
    int32_t positionwise_layer_type;            /* 0 "conv1d" (MultiLayeredConv1d), 1 "linear"
                                                 * (PositionwiseFeedForward), 2 "conv1d-linear" (Conv1dLinear);
                                                 * encoder.py:145-170 */
    int32_t duration_predictor_layers, duration_predictor_chans, duration_predictor_kernel_size;
    int32_t pitch_predictor_layers, pitch_predictor_chans, pitch_predictor_kernel_size;
    int32_t energy_predictor_layers, energy_predictor_chans, energy_predictor_kernel_size;
    int32_t pitch_embed_kernel_size, energy_embed_kernel_size;   /* 1 in every reference recipe */
    int32_t postnet_layers, postnet_chans, postnet_filts;
    int32_t use_batch_norm;
    int32_t use_scaled_pos_enc;
    int32_t encoder_normalize_before, decoder_normalize_before;  /* 0: post-norm blocks, no after_norm (encoder.py:142-143) */
    int32_t reduction_factor;                                    /* r mel frames per decoder row (feat_out: adim -> odim * r) */
    /* multi-speaker recipes (aishell3 / vctk: spk_embed_dim 256, "concat"): spk_embedding_table
     * [num_speakers, spk_embed_dim] (padding_idx 0) + spk_projection (fastspeech2.py:147-151,190-194). */
    int32_t num_speakers;                /* 0 with spk_embed_dim: only external embeddings (spembs) */
    int32_t spk_embed_dim;               /* 0 = single speaker */
    int32_t spk_embed_integration_type;  /* 0 = "add", 1 = "concat" */
    /* tone embedding (fastspeech2.py:153-157,197-202,404-408): tone_embedding_table [num_tones, tone_embed_dim]
     * (padding_idx 0) + tone_projection; only integration type "add" (0) -- with the 1-D tone ids that
     * FastSpeech2.inference forwards, the reference's "concat" branch (:606-610) cannot broadcast. */
    int32_t num_tones;
    int32_t tone_embed_dim;              /* 0 = no tone embedding */
    int32_t tone_embed_integration_type; /* 0 = "add" only */
    int32_t encoder_concat_after, decoder_concat_after;   /* concat_linear after the self-attention (encoder_layer.py:103-106) */
} pk_fs2_cfg;

int pk_fs2_create(pk_ctx* ctx, const pk_fs2_cfg* cfg, pk_fs2** out);
/* set_state_dict entry; names are the reference's keys ("encoder.encoders.0.self_attn.linear_q.weight", ...).
 * Linear weights are [in, out], Conv1D weights [Cout, Cin, k] (Paddle layouts). */
int pk_fs2_set_param(pk_fs2* h, const char* name, const float* data,
                     const int64_t* shape, int32_t ndim);
/* FastSpeech2Inference's normalizer: output mel -> mel * sigma + mu (ZScore.inverse,
 * fastspeech2.py:668-671).  NULL,NULL = return the normalised mel (FastSpeech2.inference). */
int pk_fs2_set_normalizer(pk_fs2* h, const float* mu, const float* sigma, int32_t n);
/* Arithmetic of the dense layers (Linear / Conv1D GEMMs) and of the two attention contractions: 0 = exact fp32
 * MFMA, 1 = block-scaled 3-term split-fp16 MFMA with fp32 accumulation (default; same construction and error
 * class as PK_PWG_MATH_F16X3; layers whose input channel count is not a multiple of 32 stay on the exact path).
 * LayerNorm, softmax, the duration arithmetic and every stored tensor are fp32 in both modes. */
int pk_fs2_set_math(pk_fs2* h, int32_t mode);
/* Named integer options (as pk_pwg_set_option; scheduling / tiling choices, results stay within the math mode's error class):
 *   "ffn_planes"             1 (default) = the feed-forward convs, q|k|v and attention-out projections of pre-norm FFT blocks
 *                            run on the planes kernels (csrc/ffn_planes.hip); 0 = on the tile GEMM
 *   "ffn_planes_min_blocks"  timelines shorter than this many 32-row blocks stay on the tile GEMM (default 0)
 *   "ffn_one_tile_max"       timelines with at most this many (32-row block, 32-column) tiles in a feed-forward conv run one tile
 *                            per wave (default 4096: the second conv up to sixteen utterances of 640 frames, the encoder's at thirty-two.  One
 *                            utterance 2.42 -> 2.0 ms, eight 3.1 -> 2.9, sixteen 3.73 -> 3.62, thirty-two 4.72 -> 4.64 on an MI355X;
 *                            8192 and more lose at thirty-two: four times the operand traffic per product); 0 = never
 *   "ffnp_variant"           0 (default) = tiling by shape; 88 / 84 / 48 / 44: first digit 8 / 4 = 256 / 128 output channels
 *                            per wave in the first conv, second digit = waves per workgroup of the second conv
 *   "attn_waves"             0 (default) = by shape; 4 / 8 query tiles per attention workgroup */
int pk_fs2_set_option(pk_fs2* h, const char* key, int64_t value);
int pk_fs2_finalize(pk_fs2* h);
/* Speaker conditioning of the NEXT pk_fs2_encode call (_forward :396-402, _integrate_with_spk_embed
 * :560-586): spk_id HOST int64 (B) looked up in spk_embedding_table, or spembs HOST float32
 * (B, spk_embed_dim) external embeddings (win over spk_id, as in the reference); both NULL (or never
 * called) = no integration, as when the reference gets neither.  B must equal the batch of that call;
 * the setting is consumed by it. */
int pk_fs2_set_speakers(pk_fs2* h, const int64_t* spk_id, const float* spembs, int32_t B);
/* Tone conditioning of the NEXT pk_fs2_encode call (:404-408, _integrate_with_tone_embed :588-604 with the
 * 1-D tone ids of FastSpeech2.inference: hs[t] += tone_projection(normalize(tone_embedding_table[tone[t]]))).
 * tone_id: HOST int64 packed by utterance like the token ids (n = sum of token counts); NULL clears. */
int pk_fs2_set_tones(pk_fs2* h, const int64_t* tone_id, int64_t n);
/* Phase 1 of inference (_forward :390-432): encoder, pitch/energy/duration predictors, prefix
 * sums.  ids: HOST int64, packed by utterance (sum(tok_lens)); tok_lens: HOST (B).
 * alpha = LengthRegulator speed control.  out_frames (HOST, B) receives the number of mel
 * frames of each utterance -- the output length is data dependent, so this call synchronises. */
int pk_fs2_encode(pk_fs2* h, const int64_t* ids, const int32_t* tok_lens, int32_t B, float alpha,
                  int32_t* out_frames);
/* Phase 2 (:432-466): length regulator, decoder, feat_out, postnet, (de)normalisation.
 * mel_out: (sum(out_frames), odim) float32 packed by utterance; device pointer, or host with
 * PK_HOST_IO. */
int pk_fs2_decode(pk_fs2* h, float* mel_out, int32_t flags);
/* Test taps of the last encode/decode for utterance b, copied to host:
 * 0 hs (T,adim) | 1 pitch (T) | 2 energy (T) | 3 durations (T) | 4 length-regulated hs (L,adim; needs
 * pk_fs2_set_debug(1)) | 5 decoder output (L,adim) | 6 before_outs (L,odim). */
int pk_fs2_set_debug(pk_fs2* h, int32_t on);
int pk_fs2_debug_read(pk_fs2* h, int32_t what, int32_t b, float* host_out, int64_t n_floats);
void pk_fs2_destroy(pk_fs2* h);

/* --------------------------------------------------------------- WaveFlow */
/* Constructor arguments of ConditionalWaveFlow (waveflow.py:741-757). */
typedef struct {
    int32_t n_upsample;
    int32_t upsample_factors[4];   /* 16, 16 */
    int32_t n_flows;               /* 8, even (:586-589) */
    int32_t n_layers;              /* 8 = len(dilations_dict[n_group]) (:328-331) */
    int32_t n_group;               /* 16, even */
    int32_t channels;              /* 64 (paper small model) / 128 (repo default) */
    int32_t n_mels;                /* 80 */
    int32_t kernel_h, kernel_w;    /* 3, 3 */
} pk_wf_cfg;

int pk_wf_create(pk_ctx* ctx, const pk_wf_cfg* cfg, pk_wf** out);
/* state-dict keys of ConditionalWaveFlow: encoder.{i}.*, decoder.{f}.input_proj.*,
 * decoder.{f}.resnet.{l}.{conv,condition_proj,out_proj}.*, decoder.{f}.output_proj.*;
 * weight_g / weight_v pairs are folded (recursively_remove_weight_norm). */
int pk_wf_set_param(pk_wf* h, const char* name, const float* data, const int64_t* shape, int32_t ndim);
/* 0 = exact fp32 MFMA, 1 = 3-term split-fp16 MFMA products with fp32 accumulation (default: fp32-equivalent error, as
 * pk_fs2_set_math), 2 = fp16 operands rounded to nearest, ONE MFMA per product, fp32 accumulation -- the precision the
 * reference itself synthesises at (examples/waveflow/synthesize.py:40 runs under paddle.amp.auto_cast), about 1e-4 of the
 * waveform's peak away from the fp64 oracle; 64- and 128-channel models only (PK_EUNSUPPORTED otherwise). */
int pk_wf_set_math(pk_wf* h, int32_t mode);
/* Named integer options (as pk_pwg_set_option; scheduling only, results do not change):
 *   "layer_waves"  0 (default) = the fused layer kernel of the 64-channel model runs in 12-wave workgroups (three waves per SIMD)
 *                  where that saves a round over 8-wave ones (BASELINE config 5's 8 x 640 frames: 11 tiles per workgroup), in both
 *                  maths; 6 / 8 / 12 = forced (6: two 6-wave workgroups per CU).  Same waveform bit for bit whatever the value.
 *                  (Round 5 refused three waves per SIMD in the default math: sporadic wrong tiles.  Round 6 found the cause -- a
 *                  packed fp32 FMA with op_sel, DESIGN.md 4.3 "the op_sel rule" -- and every configuration is back.)
 *   "persistent"   0 only.  (1 = the layers of a row in ONE cooperative launch with a barrier across the grid between two layers:
 *                  measured slower than eight launches in round 4; PK_EUNSUPPORTED in the product, a measurement configuration of
 *                  the profile build)
 *   "fuse_step"    1 (default) = a row's affine step and the next row's input projection happen in the launch of its last
 *                  layer; 0 = in a kernel of their own */
int pk_wf_set_option(pk_wf* h, const char* key, int64_t value);
int pk_wf_finalize(pk_wf* h);
/* For t_mel frames: length of the trimmed upsampled condition (= length of the z the reference
 * draws, waveflow.py:799-801) and of the returned waveform (pruned to a multiple of n_group, :695). */
int pk_wf_cond_length(pk_wf* h, int32_t t_mel, int32_t* cond_len, int32_t* wav_len);
/* ConditionalWaveFlow.infer (:785-805) for a packed batch.
 *   mel    (sum(frames), n_mels) float32 packed by utterance, time-major (the reference's
 *          (B, C_mel, T_mel) transposed)
 *   frames (B) host int32, >= 2
 *   z      packed latent, cond_len(frames[b]) floats per utterance (the randn of :801);
 *          NULL = drawn internally (pk_randn stream of pk_wf_set_seed)
 *   wav    packed output, wav_len(frames[b]) floats per utterance */
int pk_wf_infer(pk_wf* h, const float* mel, const int32_t* frames, int32_t B, const float* z,
                float* wav, int32_t flags);
void pk_wf_destroy(pk_wf* h);

/* ------------------------------------------------------------ SpeedySpeech */
/* SpeedySpeech(vocab_size, encoder_hidden_size, encoder_kernel_size, encoder_dilations,
 * duration_predictor_hidden_size, decoder_hidden_size, decoder_output_size, decoder_kernel_size,
 * decoder_dilations, tone_size) -- parakeet/models/speedyspeech/speedyspeech.py:142-166. */
typedef struct {
    int32_t vocab_size;
    int32_t tone_size;                    /* 0 = no tone embedding (tone_size=None) */
    int32_t encoder_hidden_size, encoder_kernel_size, n_encoder_dilations;
    int32_t encoder_dilations[32];
    int32_t duration_predictor_hidden_size;
    int32_t decoder_hidden_size, decoder_output_size, decoder_kernel_size, n_decoder_dilations;
    int32_t decoder_dilations[32];
    /* Every convolution of the model is nn.Conv1D(..., dilation=d, padding="same").  Paddle's conv kernels
     * (UpdatePaddingAndDilation, paddle/fluid/operators/conv_op.h, release 2.1) compute the "SAME" pads from
     * the undilated kernel -- before = (k-1)/2, after = (k-1) - before -- and RESET THE DILATION TO 1, so the
     * reference as it runs on Paddle does not dilate [paddle-semantics, unverified: Paddle is not installable
     * in the build image].  1 = that behaviour (what released checkpoints were trained with), 0 = the
     * convolution as written: dilation d, pads d*(k-1) split the same way. */
    int32_t same_padding_resets_dilation;
} pk_ss_cfg;
typedef struct pk_ss pk_ss;

int pk_ss_create(pk_ctx* ctx, const pk_ss_cfg* cfg, pk_ss** out);
/* set_state_dict entry ("encoder.res_blocks.3.blocks.1.0.weight", "...2._variance", ...). */
int pk_ss_set_param(pk_ss* h, const char* name, const float* data, const int64_t* shape, int32_t ndim);
/* SpeedySpeechInference's normalizer (:221-231): mel -> mel * sigma + mu.  NULL,NULL = SpeedySpeech.inference. */
int pk_ss_set_normalizer(pk_ss* h, const float* mu, const float* sigma, int32_t n);
/* 0 = exact fp32 MFMA, 1 = 3-term split-fp16 MFMA GEMMs (default, as pk_fs2_set_math). */
int pk_ss_set_math(pk_ss* h, int32_t mode);
int pk_ss_finalize(pk_ss* h);
/* Phase 1 of SpeedySpeech.inference (:178-196) for a packed batch: encoder, duration predictor,
 * durations = round(exp(.)).  text / tones: HOST int64, packed by utterance (tones may be NULL, :183);
 * tok_lens: HOST (B).  out_frames (B) host: decoder lengths (the sync the reference has at :194-196). */
int pk_ss_encode(pk_ss* h, const int64_t* text, const int64_t* tones, const int32_t* tok_lens, int32_t B,
                 int32_t* out_frames);
/* Phase 2 (:197-218): expand, + sinusoid_position_encoding, decoder -> packed (sum(frames), output_size). */
int pk_ss_decode(pk_ss* h, float* mel_out, int32_t flags);
/* Test taps of the last encode: 0 = encodings (T_b, H), 1 = log-durations (T_b), 2 = durations (T_b). */
int pk_ss_debug_read(pk_ss* h, int32_t what, int32_t b, float* host_out, int64_t n_floats);
void pk_ss_destroy(pk_ss* h);

/* ----------------------------------------------------------- dropout stream */
/* Tacotron2-style decoder prenets keep dropout ON at inference -- TransformerTTS through
 * modules/tacotron2/decoder.py:78-81 (F.dropout(x): p = 0.5, training = True whatever model.eval() says),
 * Tacotron2 through models/tacotron2.py:76-79 (training=True) -- so the reference's outputs depend on Paddle's
 * generator.  The engine draws the masks from a counter-based stream instead, a pure function of (seed, element
 * index): Philox4x32-10 with key = seed and counter = (lo(e >> 2), hi(e >> 2), 0, 0x44524F50); element e uses
 * word e & 3 of that block; keep <=> word >= floor(p * 2^32); kept values are multiplied by 1 / (1 - p)
 * (upscale_in_train, paddle.nn.functional.dropout's default mode).  oracle/philox_ref.py restates it; the golden
 * vectors of tests/golden/ come from the reference source run with this stream injected.
 * The element index of a model is stated with its entry point. */

/* ----------------------------------------------------------- TransformerTTS */
/* TransformerTTS(idim, odim, **model_cfg) -- parakeet/models/transformer_tts/transformer_tts.py:172-358.
 * Built: the embedding or conv-prenet encoder input layer, pre-norm blocks, the decoder prenet, the stop token,
 * the postnet, speaker embeddings ("add" / "concat"), both positional encodings, the "linear" decoder input layer
 * (dprenet_layers == 0), reduction_factor >= 1, global style tokens (use_gst: modules/style_encoder.py), post-norm and
 * concat_after blocks in the encoder and the decoder (encoder_layer.py:64-115, decoder_layer.py:104-151). */
typedef struct {
    int32_t idim, odim;
    int32_t embed_dim, eprenet_conv_layers, eprenet_conv_chans, eprenet_conv_filts;   /* layers 0: nn.Embedding(idim, adim) (:272-277) */
    int32_t dprenet_layers, dprenet_units;
    int32_t adim, aheads;
    int32_t elayers, eunits, dlayers, dunits;
    int32_t postnet_layers, postnet_chans, postnet_filts;
    int32_t positionwise_layer_type;       /* encoder only, 0 conv1d, 1 linear, 2 conv1d-linear (encoder.py:145-170) */
    int32_t positionwise_conv_kernel_size;
    int32_t use_scaled_pos_enc, use_batch_norm;
    int32_t encoder_normalize_before, decoder_normalize_before;
    int32_t encoder_concat_after, decoder_concat_after;
    int32_t reduction_factor;
    int32_t spk_embed_dim;                 /* 0 = None */
    int32_t use_gst;
    int32_t spk_embed_integration_type;    /* 0 = "add", 1 = "concat" (:313-317) */
    /* StyleEncoder(idim=odim, gst_token_dim=adim, ...) (:299-310), read when use_gst != 0 */
    int32_t gst_tokens, gst_heads;
    int32_t gst_conv_layers, gst_conv_kernel_size, gst_conv_stride;
    int32_t gst_gru_layers, gst_gru_units;
    int32_t gst_conv_chans[8];             /* gst_conv_chans_list, gst_conv_layers entries used */
} pk_tts_cfg;
typedef struct pk_tts pk_tts;

int pk_tts_create(pk_ctx* ctx, const pk_tts_cfg* cfg, pk_tts** out);
/* set_state_dict entry ("decoder.decoders.2.src_attn.linear_q.weight", "decoder.embed.0.0.prenet.1.0.bias", ...). */
int pk_tts_set_param(pk_tts* h, const char* name, const float* data, const int64_t* shape, int32_t ndim);
/* TransformerTTSInference's normalizer (:757-767): mel -> mel * sigma + mu, applied by pk_tts_read under
 * PK_APPLY_NORMALIZER.  NULL,NULL removes it. */
int pk_tts_set_normalizer(pk_tts* h, const float* mu, const float* sigma, int32_t n);
/* 0 = exact fp32 MFMA, 1 = 3-term split-fp16 MFMA GEMMs (default, as pk_fs2_set_math). */
int pk_tts_set_math(pk_tts* h, int32_t mode);
/* Named integer options: those of pk_fs2_set_option (the encoder's FFT stack), and
 *   "kv_prefix"  0 (default); 1 = decoder layer 0 projects k | v only for the prefix rows and q for the new rows with a
 *                row GEMM (measured neutral on an MI355X, kept as an option)
 *   "overlap_prefix"  1 (default) = the next decoding step's prefix work (prenet with its fresh dropout, input layer, layer 0's
 *                q | k | v of the row blocks that exist already) runs on a side stream under the current step's layer chain;
 *                0 = everything in order on one stream.  Same spectrogram bit for bit
 *   "fuse_prenet"  1 (default) = the decoder prenet (two layers) and the input layer of a step's new rows run as one launch;
 *                0 = three row GEMMs.  Same result up to summation order
 *   "fuse_src_q"  1 (default) = with 64-wide heads, at most 256 memory rows and adim <= 512 the encoder-decoder attention of a
 *                decoding step projects its own query (norm2 + linear_q inside the attention kernel: one launch less per layer);
 *                0 = a row GEMM of its own.  Same result up to summation order
 *   "overlap_cu_mask"  0 (default) = that side stream is an ordinary stream at the least urgent priority, the decoding loop's own
 *                stream at the most urgent; 1 = the side stream is confined to every other CU; 2 = and the loop's stream to the
 *                others.  Measured within 3 % of one another on an MI355X.  Read when the streams are created */
int pk_tts_set_option(pk_tts* h, const char* key, int64_t value);
/* Decoder-prenet dropout: 1 (default) = the dropout stream above with p = 0.5, element index
 * ((s*(s-1)/2 + pos) * dprenet_layers + layer) * dprenet_units + unit for decoding step s = 1, 2, ... and prefix
 * position pos < s (the reference re-applies the prenet to the whole prefix at every step, decoder.py:210);
 * 0 = no dropout (deterministic variant; not what the reference computes). */
int pk_tts_set_dropout(pk_tts* h, int32_t on);
int pk_tts_finalize(pk_tts* h);
/* Speaker embeddings of the NEXT pk_tts_infer call (_integrate_with_spk_embed :725-755, applied to the encoder output
 * :591-593): spembs HOST float32 (B, spk_embed_dim), one row per utterance.  Consumed by that call; NULL clears it.
 * A model with spk_embed_dim > 0 refuses to infer without them (the reference fails on spemb = None). */
int pk_tts_set_speakers(pk_tts* h, const float* spembs, int32_t B);
/* Reference spectrograms of the NEXT pk_tts_infer call for a use_gst model (`speech` of inference(), :586-588): speech
 * HOST float32 packed (sum(lens), odim), lens HOST (B) frames per utterance.  Consumed by that call; NULL clears it. */
int pk_tts_set_style_reference(pk_tts* h, const float* speech, const int32_t* lens, int32_t B);
/* TransformerTTS.inference (:511-647) for a packed batch, up to (not including) the postnet: <eos> = idim - 1 is
 * appended to every utterance (:563-565), the encoder runs once, then the decoder is stepped until every utterance
 * has stopped: utterance b ends at the first step s >= int(T_b * minlenratio) with sigmoid(prob_out) >= threshold
 * or s >= int(T_b * maxlenratio), T_b counting <eos> (:597-598, :638-642).
 *   ids      HOST int64, packed by utterance, WITHOUT <eos>;  tok_lens HOST (B)
 *   seeds    HOST (B) dropout-stream seed per utterance, or NULL = seed 0 for every utterance
 *   out_frames (B) host: L_b = decoder steps * reduction_factor (the per-step sync the reference has at :638)
 * flags: PK_TTS_KEEP_ATT keeps the encoder-decoder attention weights for pk_tts_read. */
int pk_tts_infer(pk_tts* h, const int64_t* ids, const int32_t* tok_lens, int32_t B, double threshold,
                 double minlenratio, double maxlenratio, const uint64_t* seeds, int32_t flags, int32_t* out_frames);
/* Postnet + outputs of the last pk_tts_infer (:644-651).
 *   mel_out   packed (sum(L_b), odim): outs + postnet(outs), de-normalised under PK_APPLY_NORMALIZER
 *   probs_out packed (sum(L_b)) stop probabilities, or NULL
 *   att_out   per utterance (dlayers, aheads, L_b / reduction_factor, T_b) encoder-decoder attention weights (one row
 *             per decoder step, :623-636), utterances one after another, or NULL; needs PK_TTS_KEEP_ATT at infer
 * flags: PK_HOST_IO if the three are host pointers. */
int pk_tts_read(pk_tts* h, float* mel_out, float* probs_out, float* att_out, int32_t flags);
/* Test taps of the last infer: 0 = encoder output hs (T_b, adim), 1 = outs before the postnet (L_b, odim),
 * 2 = last decoder layer's output rows (L_b / reduction_factor, adim). */
int pk_tts_debug_read(pk_tts* h, int32_t what, int32_t b, float* host_out, int64_t n_floats);
void pk_tts_destroy(pk_tts* h);

/* ---------------------------------------------------------------- Tacotron2 */
/* Tacotron2(vocab_size, n_tones, d_mels, d_encoder, ...) -- parakeet/models/tacotron2.py:626-689.
 * Refused with PK_EUNSUPPORTED: reduction_factor != 1 (Tacotron2.infer itself cannot run it: the postnet receives the
 * (B, T, d_mels * r) decoder output, :822-826). */
typedef struct {
    int32_t vocab_size;
    int32_t n_tones;                 /* 0 = None */
    int32_t d_mels, reduction_factor;
    int32_t d_encoder, encoder_conv_layers, encoder_kernel_size;
    int32_t d_prenet, d_attention_rnn, d_decoder_rnn;
    int32_t d_attention, attention_filters, attention_kernel_size;
    int32_t d_postnet, postnet_kernel_size, postnet_conv_layers;
    int32_t d_global_condition;      /* 0 = None; else a multiple of 16: the decoder's memory is d_encoder + this wide (:668-669) */
    int32_t use_stop_token;
    float p_prenet_dropout;          /* DecoderPreNet applies it with training=True (:76-79) */
} pk_taco_cfg;
typedef struct pk_taco pk_taco;

int pk_taco_create(pk_ctx* ctx, const pk_taco_cfg* cfg, pk_taco** out);
/* set_state_dict entry.  paddle.nn.LSTM registers each cell parameter twice ("encoder.lstm.0.cell_fw.weight_ih" and
 * "encoder.lstm.weight_ih_l0"); either name is accepted, the cuDNN-style one wins when both are given. */
int pk_taco_set_param(pk_taco* h, const char* name, const float* data, const int64_t* shape, int32_t ndim);
/* 0 = exact fp32 MFMA, 1 = 3-term split-fp16 MFMA GEMMs (default, as pk_fs2_set_math). */
int pk_taco_set_math(pk_taco* h, int32_t mode);
/* Decoder-prenet dropout: 1 (default) = the dropout stream with p = p_prenet_dropout, element index
 * (step * 2 + layer) * d_prenet + unit for decoding step 0, 1, ...; 0 = no dropout (not what the reference computes). */
int pk_taco_set_dropout(pk_taco* h, int32_t on);
int pk_taco_finalize(pk_taco* h);
/* Global condition of the NEXT pk_taco_infer call (:816-821): g HOST float32 (B, d_global_condition), one row per
 * utterance, concatenated to every encoder output row of that utterance.  Consumed by that call; NULL clears it.
 * A model with d_global_condition > 0 refuses to infer without it (the reference fails on the shapes). */
int pk_taco_set_global_condition(pk_taco* h, const float* g, int32_t B);
/* Tacotron2.infer (:781-840) for a packed batch, up to (not including) the postnet: embedding (+ tones), encoder
 * (conv stack, bidirectional LSTM), then the attention decoder is stepped in lockstep until every utterance has
 * ended: sigmoid(stop_logit) > 0.5 with a stop token (:515-518), else the "content exhausted" rule on the argmax of
 * the alignment (:520-525), or max_decoder_steps (:526-528).
 *   ids / tones  HOST int64, packed by utterance (tones NULL unless n_tones > 0);  tok_lens HOST (B)
 *   seeds        HOST (B) dropout-stream seed per utterance, or NULL = 0
 *   out_frames   (B) host: decoder steps L_b */
int pk_taco_infer(pk_taco* h, const int64_t* ids, const int64_t* tones, const int32_t* tok_lens, int32_t B,
                  int32_t max_decoder_steps, const uint64_t* seeds, int32_t flags, int32_t* out_frames);
/* Outputs of the last pk_taco_infer (:825-838), each packed by utterance, any of them may be NULL:
 *   mel_output (sum(L_b), d_mels); mel_outputs_postnet = mel_output + postnet(mel_output), same shape;
 *   alignments: per utterance (L_b, T_b); stop_logits (sum(L_b)), only with a stop token.
 * flags: PK_HOST_IO if they are host pointers. */
int pk_taco_read(pk_taco* h, float* mel_output, float* mel_outputs_postnet, float* alignments, float* stop_logits,
                 int32_t flags);
/* Test tap of the last infer: 0 = encoder outputs (T_b, d_encoder). */
int pk_taco_debug_read(pk_taco* h, int32_t what, int32_t b, float* host_out, int64_t n_floats);
void pk_taco_destroy(pk_taco* h);

/* ------------------------------------------------- STFT / mel / log features */
/* parakeet/modules/audio.py STFT (:74-215) + MelScale (:218-229); host twin
 * parakeet/data/get_feats.py LogMelFBank (:20-88). */
typedef struct {
    int32_t n_fft;        /* 1024 */
    int32_t hop_length;   /* 256 */
    int32_t center;       /* 1: reflect-pad n_fft/2 on both sides (:175-179) */
    int32_t power;        /* 0: magnitude sqrt(re^2+im^2) (:202-215); 1: power (:198-200) */
    int32_t n_mels;       /* 80; 0 = no mel stage */
    int32_t log_base;     /* 0 none, 10 (TTS features, get_feats.py:84-85), 2 = natural log (:86-87) */
    float log_floor;      /* 1e-10 (np.clip a_min, :83) */
} pk_mel_cfg;

/* window: n_fft floats (scipy.signal.get_window(..., fftbins=True), centre-padded to n_fft, :133-141);
 * mel_basis: (n_mels, 1 + n_fft/2) row-major (librosa.filters.mel), may be NULL if n_mels == 0. */
int pk_mel_create(pk_ctx* ctx, const pk_mel_cfg* cfg, const float* window, const float* mel_basis,
                  pk_mel** out);
/* frames = 1 + (n_samples + 2*pad - n_fft) / hop  (:103-105). */
int pk_mel_num_frames(pk_mel* h, int32_t n_samples, int32_t* frames);
/* wav: packed samples, lens (B) host.  out is packed by utterance, time-major:
 * what 0: (frames, 2*n_bin) real | imag (STFT.forward); 1: (frames, n_bin) magnitude / power;
 * 2: (frames, n_mels) mel (then log if configured). */
int pk_mel_run(pk_mel* h, const float* wav, const int32_t* lens, int32_t B, float* out,
               int32_t what, int32_t flags);
void pk_mel_destroy(pk_mel* h);

/* ------------------------------------------------------------ normal noise */
/* Standard-normal floats on the device: out[i] for i in [0, n), a pure function of (seed, offset + i).
 * Replaces the paddle.randn calls of PWGGenerator.inference (parallel_wavegan.py:515-516) and
 * ConditionalWaveFlow.infer (waveflow.py:801) for callers that hold no device RNG of their own.
 * Philox4x32-10 (Salmon et al., SC'11; key = seed, counter = (offset + i) / 4) -> 4 x uint32 ->
 * Box-Muller on pairs: u1 = (a + 1) * 2^-32 in (0, 1], u2 = b * 2^-32,
 * r = sqrt(-2 ln u1), (z0, z1) = r * (cos, sin)(2 pi u2).  offset must be a multiple of 4.
 * flags: PK_HOST_IO if out is a host pointer (then synchronous). */
int pk_randn(pk_ctx* ctx, float* out, int64_t n, uint64_t seed, uint64_t offset, int32_t flags);
/* Seed of the internal generator used when pk_pwg_infer / pk_wf_infer get noise == NULL (default 0);
 * every such call consumes a fresh, non-overlapping range of the stream. */
int pk_pwg_set_seed(pk_pwg* h, uint64_t seed);
int pk_wf_set_seed(pk_wf* h, uint64_t seed);

/* ------------------------------------------ generic primitives (parakeet/modules) */
/* sinusoid_position_encoding (positional_encoding.py:20-39) -> out (num_positions, feature_size), device. */
int pk_op_sinusoid_position_encoding(pk_ctx* ctx, int32_t num_positions, int32_t feature_size,
                                     float omega, int32_t start_pos, float* out);
/* scaled_dot_product_attention (attention.py:22-58), eval mode (no dropout).  q (B,Tq,d), k (B,Tk,d),
 * v (B,Tk,dv) device; mask float (zeros = padding, adds (1-mask)*-1e9) or NULL, mask_mode 0: (B,1,Tk),
 * 1: (B,Tq,Tk), 2: (1,Tq,Tk).  out (B,Tq,dv); weights (B,Tq,Tk) or NULL. */
int pk_op_scaled_dot_product_attention(pk_ctx* ctx, const float* q, const float* k, const float* v,
                                       const float* mask, int32_t mask_mode, int32_t B, int32_t Tq,
                                       int32_t Tk, int32_t d, int32_t dv, float* out, float* weights);
/* Conv1dBatchNorm.forward (conv.py:186-260) in eval mode, data_format "NLC", stride 1, symmetric
 * padding: x (B,T,Cin) device -> y (B, T+2*pad-k+1, Cout) device.  weight [Cout][Cin][k], bias [Cout] or
 * NULL, BatchNorm1D weight/bias/_mean/_variance [Cout] (all four or none) are HOST pointers.
 * Synchronous (weights are packed per call). */
int pk_op_conv1d_batchnorm_nlc(pk_ctx* ctx, const float* x, int32_t B, int32_t T, int32_t Cin,
                               int32_t Cout, int32_t k, int32_t pad, const float* weight,
                               const float* bias, const float* bn_weight, const float* bn_bias,
                               const float* bn_mean, const float* bn_var, float eps, float* y);

/* Conv1dCell.add_input (modules/conv.py:166-183): one step of a causal dilated Conv1D used as a cell.  buffer DEVICE
 * (B, Cin, r), r = 1 + (k - 1) * dilation, zeros before the first step (initialize_buffer :129-139), shifted by one
 * step and fed x_t DEVICE (B, Cin) here; weight DEVICE [Cout][Cin][k], bias DEVICE [Cout] or NULL; y DEVICE (B, Cout).
 * With r == 1 the buffer may be NULL.  Asynchronous on the context's stream. */
int pk_op_conv1d_cell_step(pk_ctx* ctx, float* buffer, const float* x_t, const float* weight, const float* bias,
                           int32_t B, int32_t Cin, int32_t Cout, int32_t k, int32_t dilation, float* y);

/* Plain row-major product y[M][N] = x[M][K] . w[K][N] (+ bias[N]) on the exact-fp32 MFMA GEMM: the
 * `paddle.matmul(self.weight, spectrogram)` of MelScale.forward (modules/audio.py:226-229) with rows =
 * (batch, frame).  x, y device; w, bias (or NULL) HOST.  Synchronous (the weight is packed per call). */
int pk_op_matmul(pk_ctx* ctx, const float* x, int32_t M, int32_t K, int32_t N, const float* w,
                 const float* bias, float* y);

/* expand (modules/expansion.py:19-37; LengthRegulator.expand length_regulator.py:46-66 without the alpha
 * scaling): token t of utterance b is repeated durations[b][t] times; sequences shorter than the longest are
 * zero-padded.  encodings (B, T, C) device; durations HOST int64 (B, T), >= 0; out (B, t_dec, C) device with
 * t_dec = max_b sum_t durations[b][t], which the caller computes to size `out` (0 is allowed: nothing is written). */
int pk_op_expand(pk_ctx* ctx, const float* encodings, const int64_t* durations, int32_t B, int32_t T, int32_t C,
                 int32_t t_dec, float* out);

#ifdef __cplusplus
}
#endif
#endif /* PK_SYNTH_H */
