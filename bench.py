#!/usr/bin/env python
"""Headline benchmark: audio samples/sec of FastSpeech2 + Parallel WaveGAN synthesis at 22.05 kHz.

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong]
    (N > 1: either launched by the driver as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
     --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`, or plain `python bench.py --gpus N`, which
     re-executes itself under torch.distributed.run with N ranks on 127.0.0.1)

One "step" = one pass of the synthesis hot path (FastSpeech2.inference ->
PWGGenerator.inference, mel stays in HBM) over one batch of synthetic
utterances of LJSpeech shape.  Per GPU the batch is BASELINE.json config 4's
per-GPU share: 256 utterances / 8 GPUs = 32 utterances of T = 128 phonemes,
every phoneme 5 frames -> L = 640 frames -> 163 840 samples (7.43 s) each.
Work per GPU is fixed as N grows (weak scaling, the default); `--scaling strong` instead splits the SAME 256
utterances (BASELINE config 4) over the N ranks with parakeet_amd.dist.shard_indices, each rank running its
share in mini-batches of 32.  Utterances are independent so there is no data-path collective
(parakeet_amd/dist.py); collecting the packed waveforms on every rank (`gather_ragged`, one RCCL all_gather)
is timed separately and reported as `gather_ms`, it is not part of `value`.  Inputs (token ids,
vocoder noise) are generated before the timed region and the noise is resident
in HBM; random-initialised weights of the reference architecture
(parakeet_amd/synthetic.py).  Everything is stored and accumulated in fp32.  The dense contractions use
the engine's default 3-term split-fp16 MFMA evaluation of each fp32 product (measured error = the
exact-fp32 path's); the all-exact-fp32 configuration is timed as well and reported
under `extras`.

Prints ONE JSON line on rank 0 with the driver's contract fields plus
`roofline` (dominant kernel: the PWG residual block) and, at N = 1,
`cpu_baseline` (the torch-CPU oracle on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SAMPLE_RATE = 22050
HOP = 256
UTT_PER_GPU = 32
TOKENS = 128
FRAMES_PER_TOKEN = 5
# algorithmic work of one ResidualBlock per output sample:
# dilated conv 2*192*128 + aux 1x1 2*80*128 + skip 2*64*64 + out 2*64*64
PWG_LAYER_FLOP_PER_SAMPLE = 86016
# SURVEY.md 8(d) layer-granular byte model per sample per layer (read x 256 + read c 320 + write x 256 +
# RMW skip 512); the engine never materialises c, its own minimum is 1024 B
PWG_LAYER_BYTES_PER_SAMPLE = 1344
PWG_LAYER_MIN_BYTES_PER_SAMPLE = 1024
FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense


def build_models(device):
    from parakeet_amd import synthetic as syn
    from parakeet_amd.fastspeech2 import FastSpeech2, FastSpeech2Inference
    from parakeet_amd.normalizer import ZScore
    from parakeet_amd.parallel_wavegan import PWGGenerator, PWGInference
    from parakeet_amd.synthesize import Synthesizer

    fs2_state = syn.fastspeech2_state(80, 80, fixed_duration=FRAMES_PER_TOKEN)
    pwg_state = syn.pwg_state()
    am = FastSpeech2(80, 80, **syn.FS2_LJSPEECH, device=device)
    am.set_state_dict(fs2_state)
    am.eval()
    voc = PWGGenerator(**syn.PWG_LJSPEECH, device=device)
    voc.set_state_dict(pwg_state)
    voc.remove_weight_norm()
    voc.eval()
    mu_f, sg_f = syn.mel_stats(seed=7)
    mu_p, sg_p = syn.mel_stats(seed=8)
    synth = Synthesizer(FastSpeech2Inference(ZScore(mu_f, sg_f), am), PWGInference(ZScore(mu_p, sg_p), voc))
    return synth, fs2_state, pwg_state, (mu_f, sg_f, mu_p, sg_p)


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(fs2_state, pwg_state, stats, ids, noise, warmup=2, timed=5, budget_s=30.0):
    """Time the torch-CPU oracle ("port") on a bounded sample of the same workload: utterance 0 of the
    benchmark batch (same ids, same noise), BASELINE.md section 2's protocol -- `warmup` untimed runs
    (on a quarter-length utterance: they only warm the thread pool / allocator / oneDNN primitives),
    up to `timed` full runs, median -- bounded to about `budget_s` seconds of CPU work: no further run is
    started once the budget is spent (a slow host gives fewer runs, never a bench that takes many minutes).
    Returns (record, logmel, wav) of the last run for the parity check."""
    from oracle import fastspeech2_ref, pwg_ref
    # threads: the cores this process may run on, at most 32 -- the GPU boxes expose a few hundred logical CPUs to a
    # container that owns a fraction of them, and an intra-op pool of that size does not finish (round 2: > 5 min
    # for what 32 threads do in 7 s).  `cores` reports the threads torch actually used.
    try:
        avail = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        avail = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(avail, 32)))
    cores = torch.get_num_threads()
    mu_f, sg_f, mu_p, sg_p = stats
    noise = torch.as_tensor(noise).float().cpu()

    def run(tok_ids, nz):
        with torch.no_grad():
            logmel = fastspeech2_ref.fastspeech2_inference(fs2_state, mu_f, sg_f, tok_ids)
            wav = pwg_ref.pwg_inference(pwg_state, mu_p, sg_p, logmel, nz)
        return logmel, wav

    short = max(len(ids) // 4, 1)
    for _ in range(warmup):
        run(ids[:short], noise[:short * FRAMES_PER_TOKEN * HOP])
    times = []
    for _ in range(timed):
        t0 = time.perf_counter()
        logmel, wav = run(ids, noise)
        times.append(time.perf_counter() - t0)
        if sum(times) + times[-1] > budget_s:
            break
    timed = len(times)
    dt = float(np.median(times))
    n = int(wav.shape[0])
    rec = {
        "value": n / dt, "unit": "samples/s", "cores": cores, "cores_available": avail, "kind": "port",
        "cpu_model": _cpu_model(),
        "cores_note": "torch intra-op threads = min(cores this process may run on, 32): the oracle's time is dominated by "
                      "oneDNN convolutions over ONE utterance (30 layers of 64-channel 1-D convs), which stop scaling near 32 "
                      "threads; with the pool at the box's few hundred logical CPUs the same run took > 5 min in round 2 "
                      "(thread-pool contention), so more threads would make the baseline slower, not faster",
        "sample": f"utterance 0 of the benchmark batch ({len(ids)} tokens -> {n // HOP} frames -> {n} samples), "
                  f"FastSpeech2+PWG torch-CPU fp32 oracle (Paddle-equivalent restatement); {warmup} warm-up + "
                  f"{timed} timed runs, median {dt:.2f} s (min {min(times):.2f}, max {max(times):.2f})",
        "x_realtime": n / dt / SAMPLE_RATE,
    }
    return rec, logmel.numpy(), wav[:, 0].numpy()


def waveflow_extra(channels, ctx, batch=8, frames=640, runs=5, math=None, oracle_check=False):
    """ConditionalWaveFlow.infer on BASELINE config 5's shape (batch 8 x 640 mel frames): median of `runs` batches, the
    layer kernel's average launch time from the engine's HIP-event profile and its roofline.
    Algorithmic work of one layer launch (per folded position): the (3,3) conv over the rows that exist +
    condition_proj + out_proj = 2 * (9 C + 80) * 2C + 2 * C * 2C FLOP at full taps; bytes: one fp32 read of each input row the
    kernel rows touch (1 - 3 x 4C), the condition row (4 * 80), the residual output (4C, not for the last layer) and the two
    folded output-projection sums per position (8 B read + 8 B write: the skip path is folded into them).  The
    reference's own data flow -- a C-wide skip sum read and written by every layer -- is reported next to it."""
    from parakeet_amd import synthetic as syn
    from parakeet_amd.waveflow import ConditionalWaveFlow
    wcfg = dict(syn.WAVEFLOW_LJSPEECH, channels=channels)
    wf = ConditionalWaveFlow(**wcfg)
    wf.set_state_dict(syn.waveflow_state(wcfg))
    wf.eval()
    if math:
        wf.set_math(math)
    g = torch.Generator(device="cuda").manual_seed(7)
    mels = [torch.clamp(torch.randn(80, frames, device="cuda", generator=g) * 2 - 4, min=float(np.log(1e-5)))
            for _ in range(batch)]
    zs = [torch.randn(wf.lengths(frames)[0], device="cuda", generator=g) for _ in range(batch)]
    out = wf.infer_batch(mels, zs)
    torch.cuda.synchronize()
    times = []
    for _ in range(runs):
        t1 = time.perf_counter()
        out = wf.infer_batch(mels, zs)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t1)
    dtw = float(np.median(times))
    nsw = sum(o.numel() for o in out)
    ctx.prof_enable(True)
    ctx.prof_reset()
    wf.infer_batch(mels, zs)
    prof = ctx.prof_dump()
    ctx.prof_enable(False)
    n_l, ms_l = prof.get("wf_layer", (0, 0.0))
    G, NL, NF, C = wcfg["n_group"], wcfg["n_layers"], wcfg["n_flows"], channels
    pos = nsw // G                                        # folded positions of the batch
    flop = byts = byts_ref = 0.0
    for i in range(1, G):                                 # row i of a flow: min(i, 3) input rows exist
        rows = min(i, 3)
        for l in range(NL):
            flop += pos * (2.0 * (3 * rows * C + 80) * 2 * C + 2.0 * C * 2 * C)
            # the data flow the engine implements: input rows + condition row + residual out + the two folded
            # output-projection sums per position (8 B read + 8 B write) -- the C-wide skip buffer does not exist
            byts += pos * (4.0 * (rows * C + 80 + (C if l + 1 < NL else 0)) + 16.0)
            # the reference's layer-granular data flow (C-wide skip read + write), the basis of the round-2/3 figures
            byts_ref += pos * 4.0 * (rows * C + 80 + (C if l + 1 < NL else 0) + (C if l == 0 else 2 * C))
    launches = NF * (G - 1) * NL
    ent = {
        "what": f"BASELINE config 5 shape (ConditionalWaveFlow, {channels} channels, batch {batch} x {frames} frames), " +
                ("fp16 operands / fp32 accumulation: the reference's own AMP inference precision "
                 "(examples/waveflow/synthesize.py:40), set_math('f16'); NOT fp32-equivalent (about 1e-4 of the peak)"
                 if math == "f16" else
                 "block-scaled split-fp16 products (fp32-equivalent error), layer inputs stored as pre-split fp16 planes") +
                ("; 12-wave workgroups (one round of 11 tiles per workgroup; the default math again since round 6: DESIGN.md 4.3, the op_sel "
                 "rule), the row's step fused into its last layer's launch" if channels == 64 else
                 "; 8-wave workgroups, two working waves per SIMD") +
                ("; only the hi parts of the weights travel to LDS (round 5)" if math == "f16" else ""),
        "samples_per_s": nsw / dtw, "x_realtime": nsw / dtw / SAMPLE_RATE, "ms_per_batch": dtw * 1e3,
        "ms_per_batch_runs": [t * 1e3 for t in times],
        "reference_published": "about 40x real time on V100 (docs/src/released_models.md:275-276)"}
    if n_l == launches and ms_l > 0:
        avg_s = ms_l / n_l * 1e-3
        b_l, b_ref, f_l = byts * NF / launches, byts_ref * NF / launches, flop * NF / launches
        traffic = None
        tpath = os.path.join(ROOT, "profiles", f"wf_layer_c{channels}_traffic.json")
        if math is None and os.path.exists(tpath):   # the counters were collected on the default-math kernel
            try:
                with open(tpath) as f:
                    traffic = json.load(f).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        ent["roofline"] = {
            "kernel": "k_wf_layer_p -- one WaveFlow residual layer of one row, %d launches per batch" % launches,
            "bound": "hbm", "achieved": b_l / avg_s / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": b_l / avg_s / 8e12,
            "traffic": traffic, "avg_launch_ms": avg_s * 1e3, "algorithmic_bytes_per_launch": b_l,
            "reference_dataflow_bytes_per_launch": b_ref, "frac_reference_dataflow": b_ref / avg_s / 8e12,
            "algorithmic_flop_per_launch": f_l, "algorithmic_tflops": f_l / avg_s / 1e12,
            "fp16_mfma_frac": (1.0 if math == "f16" else 3.0) * f_l / avg_s / 2.5e15,
            "note": "achieved / frac count the bytes of the IMPLEMENTED data flow (input rows, condition row, residual out, 16 B of "
                    "folded output-projection sums per position); frac_reference_dataflow counts the reference's C-wide skip "
                    "read-modify-write as well (the basis of the round-2/3 figures and targets); "
                    "averages over the launches of a batch (rows 1 and 2 of a flow read one and two input rows); "
                    "fp16_mfma_frac = issued fp16 MFMA FLOP (3 per product in the split mode, 1 in the fp16 mode) / 2.5 PFLOP/s"
                    + ("; traffic: profiles/wf_layer_c64_traffic.json (tools/pmc_traffic.py)"
                       if traffic is not None and channels == 64 else "")}
    if oracle_check:
        # BASELINE config 5 against the CPU oracle itself (the checker's role, like parity_check): utterance 0 of the timed call
        from oracle import waveflow_ref
        torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 32)))
        st = syn.waveflow_state(wcfg)
        t1 = time.perf_counter()
        with torch.no_grad():
            want = waveflow_ref.infer(st, mels[0].cpu()[None], zs[0].cpu()[None], wcfg)[0].numpy()
        got = out[0].cpu().numpy()
        ent["oracle_check"] = {"what": "utterance 0 of the 8 x 640 call vs oracle/waveflow_ref.infer (torch-CPU fp32)",
                               "relmax": float(np.abs(got - want).max() / np.abs(want).max()), "oracle_s": time.perf_counter() - t1,
                               "bar": 2e-3 if math == "f16" else 1e-5}
    del wf
    return ent


def _median_ms(fn, runs, sync):
    ts = []
    for _ in range(runs):
        t = time.perf_counter()
        fn()
        sync()
        ts.append((time.perf_counter() - t) * 1e3)
    return float(np.median(ts)), ts


def measurement_extras(synth, ctx, steps, part, ex):
    """the BASELINE configurations as stated (part "core": FastSpeech2 at 16 / 1, PWG alone at 32), the
    reference's own call shape, a batch sweep and the host-materialised step (part "rest").  None of this feeds `value`.
    Fills `ex`."""
    from parakeet_amd import synthetic as syn
    sync = torch.cuda.synchronize
    per_utt = TOKENS * FRAMES_PER_TOKEN * HOP
    frames_utt = TOKENS * FRAMES_PER_TOKEN
    texts64 = [syn.phoneme_ids(TOKENS, seed=10086 + i) for i in range(64)]
    g = torch.Generator(device="cuda").manual_seed(4242)
    noise64 = torch.randn(64 * per_utt, device="cuda", generator=g)

    # ---- BASELINE config 2 as stated: FastSpeech2, batch 16, T = 128 -> 640 frames each
    for b in ((32, 16, 1) if part == "core" else ()):
        synth.am.inference_batch(texts64[:b])
        sync()
        ms, runs = _median_ms(lambda: synth.am.inference_batch(texts64[:b]), max(steps, 10), sync)
        ex[f"fastspeech2_batch{b}"] = {
            "what": f"FastSpeech2 inference alone, {b} x {TOKENS} tokens -> {frames_utt} frames" +
                    (" (BASELINE config 2 as stated)" if b == 16 else
                     (" (the reference's call shape: one utterance per call)" if b == 1 else " (the headline step's acoustic half)")) +
                    ", default math, result left in HBM; median",
            "ms_per_batch": ms, "utterances_per_s": b / ms * 1e3, "algorithmic_tflops": 30.26e9 * b / ms / 1e9,
            "runs": len(runs)}

    if part == "core":
        _pwg_alone(synth, ctx, steps, ex, sync, per_utt, frames_utt)
        return ex
    return _measurement_rest(synth, steps, ex, sync, per_utt, texts64, noise64)


def _pwg_alone(synth, ctx, steps, ex, sync, per_utt, frames_utt):
    # ---- BASELINE config 3 as stated: PWG alone, batch 32, mel ~ N(0,1) from default_rng(42), noise from the
    # same generator passed explicitly, ZScore (0, 1)
    rng = np.random.default_rng(42)
    mel = torch.from_numpy(rng.standard_normal((UTT_PER_GPU * frames_utt, 80), dtype=np.float32)).cuda()
    nz = torch.from_numpy(rng.standard_normal(UTT_PER_GPU * per_utt, dtype=np.float32)).cuda()
    fr = np.full(UTT_PER_GPU, frames_utt, np.int32)
    run_pwg = lambda: synth.voc.infer_packed(mel, fr, noise=nz)
    run_pwg()
    sync()
    ms, runs = _median_ms(run_pwg, max(steps, 10), sync)
    ctx.prof_enable(True)
    ctx.prof_reset()
    wav = run_pwg()
    prof = ctx.prof_dump()
    ctx.prof_enable(False)
    key = next((k for k in ("pwg_layer_h3", "pwg_layer_b3", "pwg_layer") if k in prof), None)
    n_l, ms_l = prof.get(key, (0, 0.0)) if key else (0, 0.0)
    ns = UTT_PER_GPU * per_utt
    ent = {"what": "Parallel WaveGAN alone (BASELINE config 3 as stated): 32 x 640 frames of N(0,1) mel from default_rng(42), "
                   "explicit N(0,1) noise from the same generator, ZScore (0, 1); default math; waveform left in HBM; median",
           "ms_per_batch": ms, "samples_per_s": ns / ms * 1e3, "x_realtime": ns / ms * 1e3 / SAMPLE_RATE, "runs": len(runs),
           "finite": bool(torch.isfinite(wav).all()),
           "whole_call_hbm_frac": 40329.0 * ns / (ms * 1e-3) / 8e12,
           "whole_call_note": "SURVEY 8(d)'s 40 329 B per sample (30 layers x 1 344 B + first / last convs) / call time / 8 TB/s"}
    if n_l:
        avg = ms_l / n_l
        ent["roofline"] = {"kernel": "PWG ResidualBlock layer kernel, %d launches" % n_l, "bound": "hbm",
                           "achieved": PWG_LAYER_BYTES_PER_SAMPLE * ns / (avg * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                           "frac": PWG_LAYER_BYTES_PER_SAMPLE * ns / (avg * 1e-3) / 8e12, "avg_launch_ms": avg}
    try:
        over, fb = synth.voc.scale_overshoot()
        ent["scale_guard"] = {"what": "log2(a-priori bound of the planes path / measured max|x|) per layer input, maximum over "
                                      "the utterances of the handle's first (guarded) inference; limit 10",
                              "max": float(over.max()), "per_layer": [round(float(v), 2) for v in over], "fell_back": fb}
    except Exception as e:
        ent["scale_guard"] = {"error": repr(e)}
    ex["pwg_batch32"] = ent


def _measurement_rest(synth, steps, ex, sync, per_utt, texts64, noise64):
    # ---- the reference's actual call shape: one 128-token utterance per call, waveform materialised on the host
    # (examples/fastspeech2/ljspeech/synthesize_e2e.py:88-102 -> .numpy()), median of 20
    host1 = torch.empty(per_utt, dtype=torch.float32, pin_memory=True)

    def one():
        w, _ = synth.synthesize_packed(texts64[:1], noise=noise64[:per_utt])
        host1.copy_(w, non_blocking=True)
    one()
    sync()
    ms, runs = _median_ms(one, 20, sync)
    m_fs2 = (ex.get("fastspeech2_batch1") or {}).get("ms_per_batch", float("nan"))
    ex["latency_batch1"] = {
        "what": "one 128-token utterance end to end (FastSpeech2 -> PWG -> waveform in pinned host memory), the only call shape "
                "of the reference's recipe; median of 20 calls",
        "ms": ms, "min_ms": min(runs), "max_ms": max(runs), "x_realtime": per_utt / SAMPLE_RATE / (ms * 1e-3),
        "fastspeech2_ms": m_fs2, "vocoder_and_copy_ms": ms - m_fs2}

    # ---- batch sweep per GPU, end to end, device-resident result (the headline's step without the issue-ahead pipeline)
    sweep = {}
    for b in (1, 4, 16, 32, 64):
        f = lambda: synth.synthesize_packed(texts64[:b], noise=noise64[:b * per_utt])
        f()
        sync()
        ms, _ = _median_ms(f, 5, sync)
        sweep[str(b)] = {"ms_per_batch": ms, "samples_per_s": b * per_utt / ms * 1e3,
                         "x_realtime": b * per_utt / ms * 1e3 / SAMPLE_RATE}
    ex["batch_sweep"] = {"what": "FastSpeech2+PWG end to end per batch size on one GPU (128-token utterances, 640 frames each), "
                                 "unpipelined, waveform left in HBM; median of 5", "by_batch": sweep}

    # ---- BASELINE.md section 2's call boundary: model call + the waveform materialised on the host
    host32 = torch.empty(UTT_PER_GPU * per_utt, dtype=torch.float32, pin_memory=True)

    def step_host():
        w, _ = synth.synthesize_packed(texts64[:UTT_PER_GPU], noise=noise64[:UTT_PER_GPU * per_utt])
        host32.copy_(w, non_blocking=True)
    step_host()
    sync()
    ms_h, _ = _median_ms(step_host, max(steps, 10), sync)
    ms_d = sweep["32"]["ms_per_batch"]
    ex["host_io"] = {"what": "the 32-utterance step with the packed waveform (21 MB) copied to pinned host memory inside the timed "
                             "region (examples/GANVocoder/parallelwave_gan/synthesize.py:82-88 materialises it), unpipelined",
                     "host_io_ms_per_step": ms_h, "device_resident_ms_per_step": ms_d, "copy_ms": ms_h - ms_d,
                     "samples_per_s": UTT_PER_GPU * per_utt / ms_h * 1e3}
    return ex


def text_to_wav_extra(synth, n=UTT_PER_GPU):
    """VERDICT r4 "next" #5b / SURVEY 8f-3: the reference's whole request -- sentence -> English frontend -> ids -> FastSpeech2
    -> PWG -> waveform on the host -> WAV bytes (examples/fastspeech2/ljspeech/synthesize_e2e.py:88-107) -- with wall-clock
    per stage, as ONE batch of n sentences and as n single-sentence requests, frontend uncached and through the
    CachedTextToIds memo.  The synthetic acoustic model emits 5 frames per phone, so audio length follows the text."""
    from parakeet_amd.audio import wav_bytes
    from parakeet_amd.frontend import ARPABET_PHONEMES, CachedTextToIds, English, text_to_ids
    sync = torch.cuda.synchronize
    words = ("the speech was read with one hundred and twenty of the world in two thousand nineteen and she paid three "
             "dollars for this that we have not read to you or they all can be first from about five books").split()
    rng = np.random.default_rng(2021)
    sents = []
    for i in range(n):
        k = int(rng.integers(14, 26))
        w = [words[int(j)] for j in rng.integers(0, len(words), size=k)]
        w[int(rng.integers(0, k))] = str(int(rng.integers(3, 3000)))          # a number for the normaliser in every sentence
        sents.append((" ".join(w[:k // 2]) + ", " + " ".join(w[k // 2:]) + ".").capitalize())
    table = ["<pad>", "<unk>"] + sorted(p for p in ARPABET_PHONEMES if not p.startswith("<")) + ["sp", ",", ".", "?", "!", "<eos>"]
    assert len(table) <= 80                                                   # the bench model's vocabulary (idim 80)
    pmap = {p: i for i, p in enumerate(table)}
    en = English()
    clk = time.perf_counter

    def frontend_ms(fn, reps=5):
        ts = []
        for _ in range(reps):
            t = clk()
            ids = fn()
            ts.append((clk() - t) * 1e3)
        return float(np.median(ts)), ids

    fe_ms, ids = frontend_ms(lambda: [text_to_ids(en, s, pmap) for s in sents])
    cache = CachedTextToIds(en, pmap)
    cache.many(sents)
    fe_cached_ms, ids_c = frontend_ms(lambda: cache.many(sents))
    assert all(np.array_equal(a, b) for a, b in zip(ids, ids_c))
    tokens = [int(len(i)) for i in ids]
    hop_frames = FRAMES_PER_TOKEN * HOP

    def synth_batch(batch_ids, stage):
        t0 = clk()
        wav, frames = synth.synthesize_packed(batch_ids)                       # noise drawn by the engine, as in the recipe
        sync()
        t1 = clk()
        host = wav.cpu().numpy()
        t2 = clk()
        blobs, o = [], 0
        for f in frames:
            blobs.append(wav_bytes(host[o:o + int(f) * HOP], SAMPLE_RATE))
            o += int(f) * HOP
        t3 = clk()
        stage["gpu_ms"] += (t1 - t0) * 1e3
        stage["d2h_ms"] += (t2 - t1) * 1e3
        stage["wav_encode_ms"] += (t3 - t2) * 1e3
        return blobs

    def run(batched, cached):
        stage = {"frontend_ms": 0.0, "gpu_ms": 0.0, "d2h_ms": 0.0, "wav_encode_ms": 0.0}
        t0 = clk()
        if batched:
            t = clk()
            b = cache.many(sents) if cached else [text_to_ids(en, s, pmap) for s in sents]
            stage["frontend_ms"] += (clk() - t) * 1e3
            blobs = synth_batch(b, stage)
        else:
            blobs = []
            for s in sents:
                t = clk()
                b = [cache(s) if cached else text_to_ids(en, s, pmap)]
                stage["frontend_ms"] += (clk() - t) * 1e3
                blobs += synth_batch(b, stage)
        total = (clk() - t0) * 1e3
        return total, stage, blobs

    out = {"what": f"{n} English sentences ({min(tokens)}-{max(tokens)} phones, {sum(tokens)} in all -> {sum(tokens) * hop_frames} "
                   "samples; parakeet_amd.frontend.English over the demonstration lexicon, one number per sentence) -> ids -> FastSpeech2 "
                   "-> PWG (engine-drawn noise) -> waveform on the host -> 16-bit WAV bytes; wall-clock per stage, median of 5 passes "
                   "after one warm-up pass; 'requests' = one sentence per engine call, the reference's loop",
           "audio_seconds": sum(tokens) * hop_frames / SAMPLE_RATE, "frontend_ms_per_sentence_uncached": fe_ms / n,
           "frontend_ms_per_sentence_cached": fe_cached_ms / n}
    for name, batched, cached in (("one_batch", True, False), ("one_batch_cached_frontend", True, True),
                                  ("requests", False, False), ("requests_cached_frontend", False, True)):
        run(batched, cached)
        passes = [run(batched, cached) for _ in range(5)]
        passes.sort(key=lambda p: p[0])
        total, stage, blobs = passes[len(passes) // 2]
        ent = {"total_ms": total, "per_sentence_ms": total / n, "x_realtime": out["audio_seconds"] / (total * 1e-3)}
        ent.update({k: v for k, v in stage.items()})
        ent["frontend_share"] = stage["frontend_ms"] / total
        ent["wav_bytes"] = int(sum(len(b) for b in blobs))
        out[name] = ent
    out["conclusion"] = ("frontend share of a single-sentence request: %.1f %% uncached, %.1f %% through the memo; of the batch: %.1f %%"
                         % (100 * out["requests"]["frontend_share"], 100 * out["requests_cached_frontend"]["frontend_share"],
                            100 * out["one_batch"]["frontend_share"]))
    return out


def other_acoustic_models(synth, texts, steps):
    """SpeedySpeech and the autoregressive acoustic models (8f-4) at the headline's shape: 32 x 128 tokens ->
    640 frames each.  Sidecar only."""
    from parakeet_amd import synthetic as syn
    ex = {}
    try:
        from parakeet_amd.speedyspeech import SpeedySpeech
        ssm = SpeedySpeech(vocab_size=70, tone_size=7, **syn.SPEEDYSPEECH_BAKER)
        ssm.set_state_dict(syn.speedyspeech_state())
        ssm.eval()
        rng = np.random.default_rng(0)
        ph = [rng.integers(1, 70, size=TOKENS) for _ in range(UTT_PER_GPU)]
        tn = [rng.integers(1, 7, size=TOKENS) for _ in range(UTT_PER_GPU)]
        outs = ssm.inference_batch(ph, tn)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(steps):
            outs = ssm.inference_batch(ph, tn)
        torch.cuda.synchronize()
        dts = (time.perf_counter() - t1) / steps
        ex["speedyspeech_baker_batch32"] = {
            "what": "SpeedySpeech (baker configuration) inference alone, 32 x 128 phones with tones",
            "ms_per_batch": dts * 1e3, "utterances_per_s": UTT_PER_GPU / dts,
            "frames": int(sum(o.shape[0] for o in outs))}
        del ssm
    except Exception as e:
        ex["speedyspeech_baker_batch32"] = {"error": repr(e)}
    try:
        from parakeet_amd.tacotron2 import Tacotron2
        from parakeet_amd.transformer_tts import TransformerTTS
        rng = np.random.default_rng(0)
        frames_per_utt = TOKENS * 5
        tcfg = dict(syn.TRANSFORMER_TTS_LJSPEECH)
        ttm = TransformerTTS(idim=80, odim=80, **tcfg)
        ttm.set_state_dict(syn.transformer_tts_state(80, 80, tcfg, stop_bias=-8.0))
        ttm.eval()
        tx = [rng.integers(1, 79, size=TOKENS) for _ in range(UTT_PER_GPU)]
        ratio = (frames_per_utt + 0.5) / (TOKENS + 1)
        ttm.inference_batch(tx, maxlenratio=ratio, return_att=False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        outs = ttm.inference_batch(tx, maxlenratio=ratio, return_att=False)
        torch.cuda.synchronize()
        dtt = time.perf_counter() - t1
        nf = int(sum(o[0].shape[0] for o in outs))
        ex["transformer_tts_batch32"] = {
            "what": "TransformerTTS (LJSpeech recipe sizes) inference alone, 32 x 128 tokens decoded in lockstep for 640 "
                    "steps (the stop token is held off so that every utterance runs to int(129 * maxlenratio) = 640 "
                    "frames; the oracle comparison at these sizes, tests/test_ar_benchsize_gpu.py, covers the first 224 steps), "
                    "prenet dropout stream on, default math; the next step's prefix work runs on a side stream under the current step's "
                    "layer chain (option overlap_prefix)",
            "ms_per_batch": dtt * 1e3, "us_per_step": dtt / frames_per_utt * 1e6, "frames": nf,
            "utterances_per_s": UTT_PER_GPU / dtt, "x_realtime_mel_only": nf * 256 / SAMPLE_RATE / dtt}
        del ttm
        ccfg = dict(syn.TACOTRON2_LJSPEECH)
        tcm = Tacotron2(**ccfg)
        tcm.set_state_dict(syn.tacotron2_state(ccfg, stop_bias=-8.0))
        tcm.eval()
        cx = [rng.integers(1, 37, size=TOKENS) for _ in range(UTT_PER_GPU)]
        tcm.infer_batch(cx, max_decoder_steps=frames_per_utt)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        outs = tcm.infer_batch(cx, max_decoder_steps=frames_per_utt)
        torch.cuda.synchronize()
        dtc = time.perf_counter() - t1
        nf = int(sum(o["mel_output"].shape[0] for o in outs))
        ex["tacotron2_batch32"] = {
            "what": "Tacotron2 (examples/tacotron2/config.py sizes) inference alone, 32 x 128 tokens decoded in lockstep "
                    "for max_decoder_steps = 640 (stop token held off; tests/test_ar_benchsize_gpu.py compares 256 steps with the "
                    "oracle), prenet dropout stream on, default math",
            "ms_per_batch": dtc * 1e3, "us_per_step": dtc / frames_per_utt * 1e6, "frames": nf,
            "utterances_per_s": UTT_PER_GPU / dtc, "x_realtime_mel_only": nf * 256 / SAMPLE_RATE / dtc}
        del tcm
    except Exception as e:
        ex["autoregressive_models"] = {"error": repr(e)}
    return ex


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_spawn(args):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, PK_BENCH_SPAWNED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


class _DryStep:
    """--dry-run only (tests/test_bench_cpu.py): stands in for the engine so that the orchestration around it
    -- rank/shard bookkeeping, weight broadcast, barriers, max-over-ranks timing, ragged gather, the JSON
    line -- runs on CPU over gloo.  Produces zeros of the right length; never reports a throughput."""

    def __init__(self, n_tokens):
        self.n_tokens = n_tokens

    def __call__(self, texts, noise):
        frames = np.full(len(texts), self.n_tokens * FRAMES_PER_TOKEN, dtype=np.int32)
        return torch.zeros(int(frames.sum()) * HOP), frames


# ---- the ONE stdout line ------------------------------------------------------------------------------------------------------
# The driver keeps a bounded tail of stdout and parses the last line: round 5's 20.5 KB line (prose in `what` / `*_note` fields,
# every extra measurement inline) came back as BENCH_r05.json "parsed": null.  The line is now the contract fields + numbers;
# everything else -- notes, the per-kernel table, every extra measurement -- is the SIDECAR: profiles/bench_extras_last.json
# (and gpurun_out/ when that exists), with a pointer in the line.  tests/test_bench_cpu.py holds the line under LINE_LIMIT.
LINE_LIMIT = 8192
SIDECAR = os.path.join(ROOT, "profiles", "bench_extras_last.json")


def _num(v, nd=6):
    """Numbers of the line at a readable precision (floats to `nd` significant digits); everything else as it is."""
    if isinstance(v, float):
        return float(f"{v:.{nd}g}")
    return v


def _pick(d, keys, nd=6):
    return {k: _num(d[k], nd) for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def short_line(full):
    """The stdout line for a full record: the driver contract's fields, `roofline`, `cpu_baseline`, `parity_check` and a flat
    `others` of plain numbers (the extra measurements' headline figures); no prose beyond the short labels the contract asks
    for.  Pure function of `full` (tests call it with a stuffed record)."""
    out = _pick(full, ("metric", "value", "unit", "dry_run", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                       "scaling", "vs_baseline", "dtype", "data"), nd=9)
    for k in ("value", "vs_baseline"):            # null is meaningful for these two
        out.setdefault(k, full.get(k))
    cfg = full.get("config") or {}
    out["config"] = _pick(cfg, ("workload", "utterances_per_gpu", "utterances_this_rank", "global_batch", "minibatch",
                                "minibatches_per_step", "parallelism", "pipeline"))
    if "collectives" in cfg:                       # which collective carried what, in a few words (the long form: sidecar)
        out["config"]["collectives"] = {k: str(v)[:160] for k, v in cfg["collectives"].items()}
    roof = full.get("roofline")
    if roof:
        r = _pick(roof, ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_collected", "avg_launch_ms",
                         "algorithmic_bytes_per_launch", "algorithmic_flop_per_launch", "algorithmic_tflops",
                         "engine_min_bytes_per_launch"))
        r.setdefault("traffic", None)
        if roof.get("exact_f32"):
            r["exact_f32"] = _pick(roof["exact_f32"], ("samples_per_s", "ms_per_step", "layer_ms", "bound", "achieved_tflops",
                                                       "peak_tflops", "frac_of_fp32_mfma"))
        out["roofline"] = r
    cb = full.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "cores_available", "kind", "cpu_model", "sample", "x_realtime"))
        if "sample" in out["cpu_baseline"]:
            out["cpu_baseline"]["sample"] = out["cpu_baseline"]["sample"][:200]
    pc = full.get("parity_check")
    if pc:
        out["parity_check"] = _pick(pc, ("frames_equal", "mel_l1", "wav_relmax", "bars"))
    out.update(_pick(full, ("x_realtime", "value_unpipelined", "kernel_ms_sum", "rccl_world_size", "gather_ms", "gather_all_ms")))
    if full.get("rank_ms_per_step"):
        out["rank_ms_per_step"] = _pick(full["rank_ms_per_step"], ("min", "max"), nd=9)
    if full.get("pipeline_check"):
        out["pipeline_check"] = _pick(full["pipeline_check"], ("bit_identical_to_unpipelined", "unpipelined_ms_per_step"))
    ex = full.get("extras") or {}
    others = {}

    def put(name, ent, key, nd=5):
        if isinstance(ent, dict) and isinstance(ent.get(key), (int, float)):
            others[name] = _num(float(ent[key]), nd)
    for b in (32, 16, 1):
        put(f"fastspeech2_batch{b}_ms", ex.get(f"fastspeech2_batch{b}"), "ms_per_batch")
    put("pwg_batch32_ms", ex.get("pwg_batch32"), "ms_per_batch")
    put("latency_batch1_ms", ex.get("latency_batch1"), "ms")
    put("host_io_ms_per_step", ex.get("host_io"), "host_io_ms_per_step")
    for c in (64, 128):
        for suf in ("", "_fp16"):
            ent = ex.get(f"waveflow_c{c}_batch8{suf}")
            put(f"waveflow_c{c}_batch8{suf}_ms", ent, "ms_per_batch")
            if isinstance(ent, dict) and isinstance(ent.get("roofline"), dict):
                put(f"waveflow_c{c}_batch8{suf}_layer_us", {"v": ent["roofline"].get("avg_launch_ms", 0.0) * 1e3}, "v")
                put(f"waveflow_c{c}_batch8{suf}_hbm_frac", ent["roofline"], "frac", 3)
            if isinstance(ent, dict) and isinstance(ent.get("oracle_check"), dict):
                put(f"waveflow_c{c}_batch8{suf}_relmax_vs_oracle", ent["oracle_check"], "relmax", 3)
    put("speedyspeech_batch32_ms", ex.get("speedyspeech_baker_batch32"), "ms_per_batch")
    put("transformer_tts_us_per_step", ex.get("transformer_tts_batch32"), "us_per_step")
    put("tacotron2_us_per_step", ex.get("tacotron2_batch32"), "us_per_step")
    put("text_to_wav_one_batch_ms", (ex.get("text_to_wav") or {}).get("one_batch"), "total_ms")
    errs = sorted(k for k, v in ex.items() if isinstance(v, dict) and "error" in v)
    if errs:
        others["errors_in"] = errs
    if others:
        out["others"] = others
    if full.get("sidecar"):
        out["sidecar"] = full["sidecar"]
    line = json.dumps(out)
    if len(line) >= LINE_LIMIT:                    # never a line the driver cannot keep: shed the optional parts
        for k in ("others", "pipeline_check", "parity_check"):
            out.pop(k, None)
            line = json.dumps(out)
            if len(line) < LINE_LIMIT:
                break
    assert len(line) < LINE_LIMIT, f"bench line is {len(line)} bytes"
    return line


def write_sidecar(full):
    """Everything measured, prose included, next to the profiles (the driver's box: read it there; gpurun: comes back under
    gpurun_out/).  Returns the repository-relative path written, or None.  Never fails the bench."""
    rel = None
    for path in (SIDECAR, os.path.join(ROOT, "gpurun_out", "bench_extras_last.json")):
        try:
            if os.path.isdir(os.path.dirname(path)):
                with open(path, "wt") as f:
                    json.dump(full, f, indent=1)
                rel = rel or os.path.relpath(path, ROOT)
        except OSError:
            pass
    return rel


def emit(full):
    """Sidecar first, then the one stdout line (flushed at once: nothing that runs later can lose it)."""
    full = dict(full)
    full["sidecar"] = write_sidecar(full)
    if full["sidecar"]:
        write_sidecar(full)                        # (with its own name in it)
    print(short_line(full), flush=True)



_T0 = time.perf_counter()


def _log(msg):
    """Progress on stderr (stdout carries the one JSON line): shows where a slow run spends its time."""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench +{time.perf_counter() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: 32 utterances per GPU; strong: the same --global-batch utterances split over the ranks")
    ap.add_argument("--global-batch", type=int, default=256, help="utterances of a strong-scaling step")
    ap.add_argument("--minibatch", type=int, default=UTT_PER_GPU, help="utterances per engine call")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed-for-value extra measurements (= --extras none)")
    ap.add_argument("--extras", choices=("none", "core", "all"), default="all",
                    help="core: the strict-fp32 configuration, FastSpeech2 / PWG alone, WaveFlow at BASELINE config 5 -- measured "
                         "before the line is printed (their figures are in it); all (default): also the batch sweep, host io, "
                         "text -> wav, 128-channel WaveFlow, the other acoustic models -- measured AFTER the line is out, into "
                         "the sidecar file only")
    ap.add_argument("--pipeline", type=int, default=2,
                    help="2 (default): the next batch's acoustic model on a second engine handle / side stream, issued before this "
                         "batch's vocoder; 1: one handle, issued after the vocoder is queued (round 5); 0: in order")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU/gloo rehearsal of the multi-process orchestration with a stub in place of the engine "
                         "(test infrastructure; prints value null)")
    args = ap.parse_args()
    if args.no_extras:
        args.extras = "none"

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if not args.dry_run and (not torch.cuda.is_available() or torch.cuda.device_count() < args.gpus):
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {torch.cuda.device_count()} HIP device(s) visible")
        raise SystemExit(self_spawn(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    # PK_BENCH_FORCE_DIST=1 exercises the RCCL code path (init, weight broadcast, barriers, max-reduce) at N = 1; so does a
    # launcher (torch.distributed.run sets RANK and MASTER_ADDR even for one rank): the path the driver's N > 1 runs take is
    # the path a launched N = 1 run takes.  Plain `python bench.py` (the driver's N = 1 run) creates no process group.
    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ and not os.environ.get("PK_BENCH_SPAWNED_PLAIN")
    distributed = world > 1 or launched or bool(os.environ.get("PK_BENCH_FORCE_DIST"))
    dry = args.dry_run
    if not dry:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device (no CPU fallback for the product path)")
        if torch.cuda.device_count() <= local_rank:
            raise SystemExit(f"bench.py: rank {rank} needs HIP device {local_rank}, {torch.cuda.device_count()} visible")
        torch.cuda.set_device(local_rank)
    dev = "cpu" if dry else "cuda"

    def sync():
        if not dry:
            torch.cuda.synchronize()

    if distributed and dry:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
    elif distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(_free_port()))   # PK_BENCH_FORCE_DIST without a launcher
        # stdout is for the one JSON line: RCCL prints its NCCL_DEBUG=VERSION banner (set on the GPU boxes) with
        # printf when the communicator is created, so fd 1 points at stderr until the first collective has run
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            dist.barrier()
            torch.cuda.synchronize()
            assert dist.get_world_size() == args.gpus, "RCCL communicator does not span --gpus ranks"
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    from parakeet_amd import dist as pdist
    from parakeet_amd import synthetic as syn

    if dry:
        synth, fs2_state, pwg_state, stats = _DryStep(TOKENS), {"w": np.arange(8, dtype=np.float32)}, {}, None
        if distributed:
            got = pdist.broadcast_state_dict(fs2_state if rank == 0 else {"w": np.zeros(8, np.float32)}, src=0)
            assert np.array_equal(got["w"], fs2_state["w"])
    else:
        from parakeet_amd.runtime import Context
        synth, fs2_state, pwg_state, stats = build_models(local_rank)
        if distributed:
            # weights come from rank 0 over RCCL (one flat broadcast per model), as a deployment would do it
            fs2_b = pdist.broadcast_state_dict(fs2_state, src=0)
            pwg_b = pdist.broadcast_state_dict(pwg_state, src=0)
            synth.am.set_state_dict(fs2_b)
            synth.voc.set_state_dict(pwg_b)
            fs2_state = fs2_b   # (what the second acoustic lane below is built from)

    # this rank's utterances.  weak: UTT_PER_GPU per rank (global batch grows with N); strong: the same
    # --global-batch utterances for every N, dealt out by cost (all equal here) with shard_indices
    if args.scaling == "weak":
        own = list(range(rank * UTT_PER_GPU, (rank + 1) * UTT_PER_GPU))
        global_batch = UTT_PER_GPU * world
    else:
        global_batch = args.global_batch
        own = pdist.shard_indices([TOKENS] * global_batch, world, rank)
    texts_all = [syn.phoneme_ids(TOKENS, seed=10086 + i) for i in own]
    per_utt = TOKENS * FRAMES_PER_TOKEN * HOP
    n_samples = len(own) * per_utt                      # this rank's samples per step
    mb = max(1, args.minibatch)
    chunks = [(a, min(a + mb, len(own))) for a in range(0, len(own), mb)]
    gen = torch.Generator(device=dev)
    gen.manual_seed(42 + rank)
    noise = torch.randn(n_samples, device=dev, generator=gen)   # resident in HBM before the timed region

    def step():
        """One pass of the hot path over this rank's share; returns the last mini-batch's (wav, frames)."""
        out = None
        for a, b in chunks:
            nz = noise[a * per_utt:b * per_utt]
            out = synth(texts_all[a:b], nz) if dry else synth.synthesize_packed(texts_all[a:b], noise=nz)
        return out

    # --pipeline: the acoustic model of the NEXT step's batch is issued on a side stream while this step's vocoder
    # runs (Synthesizer.issue_acoustic / vocode_issued), so that its launches are queued before the GPU needs them.
    # Every step still does one acoustic pass and one vocoder pass over one batch; the acoustic pass belongs to the
    # following step's batch (all batches are the same synthetic utterances).  One mini-batch per step only.
    # --pipeline 2 (default): two acoustic lanes -- a second FastSpeech2 engine handle with the same weights on its own side
    # stream, and the next batch's acoustic model issued BEFORE this batch's vocoder (Synthesizer.issue_acoustic's comment;
    # profiles/r06_pipeline_lanes.txt: the holes that one handle leaves around the frame-count sync are covered by the
    # other lane's queued work).  --pipeline 1: round 5's one-lane order.
    pipelined = bool(args.pipeline) and not dry and len(chunks) == 1
    two_lanes = pipelined and args.pipeline >= 2
    pending = [None]
    step_no = [0]
    if two_lanes:
        from parakeet_amd.fastspeech2 import FastSpeech2, FastSpeech2Inference
        am2 = FastSpeech2(80, 80, **syn.FS2_LJSPEECH, device=local_rank)
        am2.set_state_dict(fs2_state)
        am2.eval()
        synth.add_acoustic_lane(FastSpeech2Inference(synth.am_inference.normalizer, am2))

    def step_pipelined():
        if not two_lanes:
            if pending[0] is None:
                pending[0] = synth.issue_acoustic(texts_all)
            out = synth.vocode_issued(pending[0], noise=noise)
            pending[0] = synth.issue_acoustic(texts_all)
            return out
        k = step_no[0]
        step_no[0] += 1
        if pending[0] is None:
            pending[0] = synth.issue_acoustic(texts_all, lane=k & 1)
        nxt = synth.issue_acoustic(texts_all, lane=(k + 1) & 1)
        out = synth.vocode_issued(pending[0], noise=noise)
        pending[0] = nxt
        return out

    plain_step = step
    if pipelined:
        step = step_pipelined

    def barrier():
        if distributed:
            import torch.distributed as dist
            dist.barrier()

    _log("models built; first-touch pass")
    wav, frames = step()  # build / first-touch pass (allocations, weight packing); never timed
    _log("warm-up")
    for _ in range(args.warmup):
        wav, frames = step()
    sync()
    a_last, b_last = chunks[-1]
    assert int(frames.sum()) * HOP == (b_last - a_last) * per_utt, "synthetic duration head must give 5 frames per token"
    assert bool(torch.isfinite(wav).all()), "non-finite waveform"

    barrier()
    sync()
    _log("timed region")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wav, frames = step()
    sync()
    barrier()
    elapsed = time.perf_counter() - t0
    _log(f"timed region done: {elapsed / args.steps * 1e3:.2f} ms/step")
    pipeline_check = None
    if pipelined:
        # back to the plain step for everything after the timed region; the pipelined waveform must be the plain one
        step = plain_step
        pending[0] = None
        w_plain, _ = step()
        sync()
        pipeline_check = {"bit_identical_to_unpipelined": bool(torch.equal(w_plain, wav))}
        # the same K steps without the pipeline, for the record (never `value`)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        dtp = (time.perf_counter() - t1) / args.steps
        pipeline_check["unpipelined_ms_per_step"] = dtp * 1e3
        pipeline_check["unpipelined_samples_per_s"] = n_samples / dtp
    gather_ms = gather_all_ms = None
    rank_ms = {"min": elapsed / args.steps * 1e3, "max": elapsed / args.steps * 1e3}
    comm_world = 1
    if distributed:
        import torch.distributed as dist
        comm_world = dist.get_world_size()          # the communicator's own count, for the SCALE record
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        tmin = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        elapsed = float(t.item())                   # `value` follows the slowest rank
        rank_ms = {"min": float(tmin.item()) / args.steps * 1e3, "max": elapsed / args.steps * 1e3}   # load imbalance
        # result collection: every rank's packed waveform of its last mini-batch gathered on rank 0 --
        # one all_gather of the lengths + direct sends of the exact sizes (gather_ragged_to); timed on its own, never part
        # of `value`.  The all-ranks variant (padded all_gather) is timed next to it for consumers that need that.
        lens = [int(f) * HOP for f in frames]

        def timed_gather(fn):
            fn()                            # communicator / buffer warm-up
            barrier()
            sync()
            tg = time.perf_counter()
            res = fn()
            sync()
            tgm = torch.tensor([time.perf_counter() - tg], device=dev, dtype=torch.float64)
            dist.all_reduce(tgm, op=dist.ReduceOp.MAX)
            return float(tgm.item()) * 1e3, res
        gather_ms, (bufs, meta) = timed_gather(lambda: pdist.gather_ragged_to(wav, lens, dst=0))
        if rank == 0:
            assert [int(b.numel()) for b in bufs] == [sum(m) for m in meta] and len(bufs) == world
        else:
            assert bufs is None
        gather_all_ms, (bufs, meta) = timed_gather(lambda: pdist.gather_ragged(wav, lens))
        assert [int(b.numel()) for b in bufs] == [sum(m) for m in meta] and len(bufs) == world

    # which collective carried what (the long form: DESIGN.md section 7)
    collectives = {
        "in_the_timed_step": "none (utterances are independent; each rank synthesises its own shard)",
        "weights": "one RCCL broadcast per model from rank 0 at start-up (FastSpeech2 148.5 MB, PWG 5.3 MB)"
                   if distributed else "single process: none",
        "results": "gather_ms: lengths by all_gather_object + one gather to rank 0 (ncclSend/Recv, one xGMI link per sender); "
                   "gather_all_ms: one padded all_gather" if distributed else "single process: none",
    }
    if dry:
        if rank == 0:
            print(short_line({
                "metric": "audio samples/sec, FastSpeech2+PWGAN 22.05kHz", "value": None, "unit": "samples/s",
                "dry_run": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": elapsed / args.steps * 1e3, "scaling": args.scaling, "gather_ms": gather_ms,
                "gather_all_ms": gather_all_ms, "rccl_world_size": comm_world, "rank_ms_per_step": rank_ms,
                "config": {"global_batch": global_batch, "utterances_this_rank": len(own),
                           "minibatches_per_step": len(chunks), "collectives": collectives}}), flush=True)
        if distributed:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- per-kernel durations (HIP events on the launch stream), outside the timed region
    ctx = Context.get(local_rank)
    ctx.prof_enable(True)
    ctx.prof_reset()
    prof_steps = 2
    for _ in range(prof_steps):
        step()
    prof = ctx.prof_dump()
    ctx.prof_enable(False)
    layer_samples = (chunks[0][1] - chunks[0][0]) * per_utt   # samples one PWG layer launch processes

    # ---- the engine's output for utterance 0 of this rank's first mini-batch (default math), kept for the
    # parity check against the CPU oracle below (same ids, same noise; outside the timed region)
    parity_src = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        a0, b0 = chunks[0]
        w0, _ = synth.synthesize_packed(texts_all[a0:b0], noise=noise[a0 * per_utt:b0 * per_utt])
        m0 = synth.am.decode_packed(denormalize=True)
        torch.cuda.synchronize()
        parity_src = (m0[:TOKENS * FRAMES_PER_TOKEN].cpu().numpy(), w0[:per_utt].cpu().numpy())

    # ---- extra measurements (none of them feeds `value`).  "core" runs before the line is printed -- its headline figures
    # are in the line (roofline.exact_f32, others) --, the rest afterwards, into the sidecar only.
    extras = {}
    do_extras = world == 1 and args.extras != "none" and args.scaling == "weak"
    texts = texts_all[:UTT_PER_GPU]

    def guarded(key, fn):
        try:
            fn()
        except Exception as e:  # never let an extra break the headline line
            extras[key] = {"error": repr(e)}

    def other_math(mode, key):
        """The same end-to-end step under another evaluation of the dense contractions (then back to the default)."""
        synth.voc.set_math(mode)
        synth.am.set_math("f32" if mode == "f32" else "f16x3")
        try:
            for _ in range(2):
                plain_step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                plain_step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / args.steps
            ctx.prof_enable(True)
            ctx.prof_reset()
            plain_step()
            p3 = ctx.prof_dump()
            ctx.prof_enable(False)
            pk = "pwg_layer" if mode == "f32" else "pwg_layer_b3"
            n3, ms3 = p3.get(pk, (0, 0.0))
            avg3 = ms3 / max(n3, 1)
            ent = {"samples_per_s": n_samples / dt, "x_realtime": n_samples / dt / SAMPLE_RATE,
                   "ms_per_step": dt * 1e3, "layer_kernel_avg_ms": avg3}
            if mode == "f32":
                tf = PWG_LAYER_FLOP_PER_SAMPLE * n_samples / (avg3 * 1e-3) / 1e12 if avg3 else 0.0
                ent["what"] = ("same end-to-end step with every contraction on the exact-fp32 matrix pipe "
                               "(v_mfma_f32_32x32x2_f32; pk_pwg_set_math / pk_fs2_set_math = F32)")
                ent["roofline"] = {"bound": "mfma", "achieved": tf, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                   "frac": tf / FP32_MFMA_PEAK_TFLOPS}
            else:
                ent["what"] = "same step with bf16 parts instead of fp16 parts (fp32 range, error 3.7e-6)"
            extras[key] = ent
        finally:
            synth.voc.set_math("f16x3")
            synth.am.set_math("f16x3")

    def waveflow(wf_c, wmath):
        key = f"waveflow_c{wf_c}_batch8" + ("_fp16" if wmath else "")
        check = wf_c == 64 and not args.no_cpu_baseline      # (config 5 itself: both maths against the oracle)
        guarded(key, lambda: extras.__setitem__(key, waveflow_extra(wf_c, ctx, math=wmath, oracle_check=check)))

    def extras_core():
        _log("extras (core): strict fp32, FastSpeech2 / PWG alone, WaveFlow config 5")
        guarded("all_exact_f32_mfma", lambda: other_math("f32", "all_exact_f32_mfma"))
        guarded("measurement_extras", lambda: measurement_extras(synth, ctx, args.steps, "core", extras))
        for wmath in (None, "f16"):      # BASELINE config 5 (64 channels, batch 8 x 640 frames)
            waveflow(64, wmath)

    def extras_rest():
        _log("extras (rest, sidecar only): bf16 parts, 128-channel WaveFlow, sweep / host io, text -> wav, other models")
        guarded("pwg_bf16x3_split", lambda: other_math("bf16x3", "pwg_bf16x3_split"))
        for wmath in (None, "f16"):      # the reference repository's default width
            waveflow(128, wmath)
        guarded("measurement_extras_rest", lambda: measurement_extras(synth, ctx, args.steps, "rest", extras))
        guarded("text_to_wav", lambda: extras.__setitem__("text_to_wav", text_to_wav_extra(synth)))
        guarded("acoustic_models", lambda: extras.update(other_acoustic_models(synth, texts, args.steps)))

    if do_extras:
        extras_core()

    full = None
    if rank == 0:
        total_samples = global_batch * per_utt             # whole job, all ranks, per step
        ms_per_step = elapsed / args.steps * 1e3
        value = total_samples * args.steps / elapsed
        layer_key = next((k for k in ("pwg_layer_h3", "pwg_layer_b3", "pwg_layer") if k in prof), "pwg_layer")
        n_layer, ms_layer = prof.get(layer_key, (0, 0.0))
        avg_ms = ms_layer / max(n_layer, 1)
        flop_per_launch = PWG_LAYER_FLOP_PER_SAMPLE * layer_samples
        bytes_per_launch = PWG_LAYER_BYTES_PER_SAMPLE * layer_samples
        # HBM bytes per launch from the PMC counters: collected by tools/pmc_traffic.py on THIS kernel in its own rocprofv3 passes
        # (counters cannot be read from inside the process); the file names the source hash of the library it was collected
        # on, and `traffic_collected` says whether that is the library timed here
        traffic = traffic_of = None
        tpath = os.path.join(ROOT, "profiles", "pwg_layer_traffic.json")
        if os.path.exists(tpath):
            try:
                with open(tpath) as f:
                    tj = json.load(f)
                if tj.get("prof_key", "pwg_layer") == layer_key and tj.get("samples_per_launch", 32 * per_utt) == layer_samples:
                    traffic = tj.get("hbm_bytes_per_launch")
                    from parakeet_amd import build as _pb
                    same = tj.get("kernel_source_sha256") == _pb.file_hash("pwg.hip")
                    traffic_of = (tj.get("collected", "earlier round") +
                                  (": this kernel source" if same else ": an earlier build of the kernel"))
            except Exception:
                traffic = traffic_of = None
        if layer_key == "pwg_layer":
            # exact-fp32 matrix pipe: compute bound (intensity 64 FLOP/B > ridge 19.7)
            achieved = flop_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
            roof = {"bound": "mfma", "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / FP32_MFMA_PEAK_TFLOPS}
        else:
            # split 16-bit matrix path: 5.3x less matrix-pipe time -> the layer-granular HBM stream binds
            achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            roof = {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0}
        total_prof_ms = sum(ms for _, ms in prof.values()) / prof_steps
        full = {
            "metric": "audio samples/sec, FastSpeech2+PWGAN 22.05kHz",
            "value": value,
            "unit": "samples/s",
            "x_realtime": value / SAMPLE_RATE,
            "rtf_reference_convention": SAMPLE_RATE / value,
            "n_gpus": world,
            "rccl_world_size": comm_world,
            "rank_ms_per_step": rank_ms,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32 (3-term split-fp16 MFMA, fp32-equivalent; exact-fp32 in roofline.exact_f32)",
            "dtype_note": "fp32 storage and accumulation everywhere; the dense contractions (PWG residual blocks and "
                          "last convs, FastSpeech2 Linear/Conv1D/attention) evaluate each fp32 product as a 3-term split-fp16 "
                          "MFMA sum (a_hi*b_hi + a_lo*b_hi + a_hi*b_lo) of BLOCK-SCALED operands (DESIGN.md 3 / HISTORY 4.7): the "
                          "error is the exact-fp32 MFMA path's independent of the magnitude of weights or activations; "
                          "parity_check compares this very batch with the fp32 CPU oracle; softmax / LayerNorm / durations are "
                          "plain fp32; the all-exact-fp32 configuration is timed as roofline.exact_f32",
            "data": "synthetic",
            "config": {
                "workload": "FastSpeech2+PWG end-to-end (BASELINE config 4 per-GPU share): "
                            f"{UTT_PER_GPU} utterances/GPU x {TOKENS} phonemes -> {TOKENS * FRAMES_PER_TOKEN} frames "
                            f"-> {TOKENS * FRAMES_PER_TOKEN * HOP} samples each, LJSpeech architecture, random-init weights",
                "utterances_per_gpu": len(own),
                "global_batch": global_batch,
                "minibatch": mb,
                "parallelism": f"dp{world} (utterance sharding, no data-path collective)",
                "collectives": collectives,
                "pipeline": ("next batch's acoustic model (second engine handle, own side stream) issued before this batch's vocoder" if two_lanes
                             else "next batch's acoustic model issued on a side stream under this batch's vocoder") if pipelined else "none",
            },
            "roofline": dict(roof, **{
                "kernel": {"pwg_layer": "k_pwg_layer (exact fp32 MFMA)", "pwg_layer_h3": "k_pwg_layer_b3<HALF> (3-term split-fp16 MFMA)",
                           "pwg_layer_b3": "k_pwg_layer_b3 (3-term split-bf16 MFMA)"}[layer_key] +
                          " -- PWG ResidualBlock, 30 launches/step",
                "traffic": traffic,
                "traffic_collected": traffic_of,
                "avg_launch_ms": avg_ms,
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "algorithmic_flop_per_launch": flop_per_launch,
                "algorithmic_tflops": flop_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0,
                "engine_min_bytes_per_launch": PWG_LAYER_MIN_BYTES_PER_SAMPLE * layer_samples,
                "note": "algorithmic bytes = SURVEY.md 8(d) layer-granular model, 1344 B/sample/layer x samples per "
                        "launch; the engine itself never materialises the upsampled conditioning (1024 B/sample); traffic = "
                        "profiles/pwg_layer_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, tools/pmc_traffic.py)",
            }),
            "kernel_ms_per_step": {k: ms / prof_steps for k, (_, ms) in sorted(prof.items())},
            "kernel_ms_sum": total_prof_ms,
        }
        ex32 = extras.get("all_exact_f32_mfma")
        if ex32 and ex32.get("layer_kernel_avg_ms"):
            # the strict-fp32 configuration of the SAME step (every contraction on v_mfma_f32_32x32x2_f32): its layer kernel is
            # bound by the fp32 matrix pipe, not by HBM
            full["roofline"]["exact_f32"] = {
                "samples_per_s": ex32["samples_per_s"], "ms_per_step": ex32["ms_per_step"],
                "layer_ms": ex32["layer_kernel_avg_ms"], "bound": "mfma",
                "achieved_tflops": ex32["roofline"]["achieved"], "peak_tflops": FP32_MFMA_PEAK_TFLOPS,
                "frac_of_fp32_mfma": ex32["roofline"]["frac"],
                "what": "pk_pwg_set_math / pk_fs2_set_math = F32: exact fp32 products, in order (no issue-ahead pipeline)"}
        if pipeline_check is not None:
            full["pipeline_check"] = pipeline_check
            full["value_unpipelined"] = pipeline_check["unpipelined_samples_per_s"]
            full["value_note"] = ("`value` = steady-state throughput with the next batch's acoustic model issued during this "
                                  "batch's vocoder; `value_unpipelined` = the same step issued strictly in order (the latency-true "
                                  "figure); extras.host_io = with the waveform copied to host memory inside the step")
        if extras:
            full["extras"] = extras
        if gather_ms is not None:
            full["gather_ms"] = gather_ms
            full["gather_all_ms"] = gather_all_ms
            full["gather_note"] = ("gather_ms: parakeet_amd.dist.gather_ragged_to -- every rank's packed waveform (last mini-batch) "
                                   "collected on rank 0 by direct sends; gather_all_ms: gather_ragged -- the same on every rank (one "
                                   "padded RCCL all_gather); neither is in `value` (config.collectives)")
        if world == 1 and not args.no_cpu_baseline:
            _log("cpu_baseline (torch-CPU oracle, bounded)")
            rec, ref_mel, ref_wav = cpu_baseline(fs2_state, pwg_state, stats, texts_all[0], noise[:per_utt].cpu())
            full["cpu_baseline"] = rec
            if parity_src is not None:
                got_mel, got_wav = parity_src
                full["parity_check"] = {
                    "what": "engine (default math, inside the 32-utterance batch) vs the fp32 CPU oracle run timed above: "
                            "same token ids, same vocoder noise, utterance 0",
                    "frames_equal": bool(got_mel.shape == ref_mel.shape),
                    "mel_l1": float(np.abs(got_mel - ref_mel).mean()) if got_mel.shape == ref_mel.shape else None,
                    "wav_relmax": float(np.abs(got_wav - ref_wav).max() / (np.abs(ref_wav).max() + 1e-30))
                    if got_wav.shape == ref_wav.shape else None,
                    "bars": {"mel_l1": 1e-4, "wav_relmax": 1e-4},
                }
        _log("the line")
        emit(full)
    if do_extras and args.extras == "all":
        extras_rest()
        if full is not None:
            full["extras"] = extras
            full["values_together"] = {
                "pipelined (value)": full["value"], "in_order (value_unpipelined)": full.get("value_unpipelined"),
                "in_order_host_materialised": (extras.get("host_io") or {}).get("samples_per_s")}
            full["sidecar"] = write_sidecar(full)
            _log(f"sidecar complete: {full['sidecar']}")
    _log("done")
    if distributed:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
