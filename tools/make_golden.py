#!/usr/bin/env python
"""Generate tests/golden/*.npz by executing the REFERENCE's own Python source
(/root/reference/parakeet/models/...) over the torch-backed paddle stand-in
(oracle/paddle_shim).  Runs only in the build container (needs /root/reference).

Weights are not stored: they are regenerated from seeds by parakeet_amd.synthetic, whose
state-dict keys the reference classes accept without remapping (set_state_dict asserts
an exact key match).  Stored: inputs (ids / mel / noise) and the reference outputs.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

ref_import.setup()
import paddle  # noqa: E402  (oracle/paddle_shim, or PaddlePaddle itself under PARAKEET_REAL_PADDLE=1)

from parakeet_amd import synthetic as syn  # noqa: E402

OUT = ref_import.golden_dir()   # tests/golden (stand-in) or tests/golden_paddle (PARAKEET_REAL_PADDLE=1)


def golden_fastspeech2():
    fsm = ref_import.load("parakeet.models.fastspeech2.fastspeech2")
    norm = ref_import.load("parakeet.modules.normalizer")
    cfg = dict(syn.FS2_LJSPEECH)
    state = syn.fastspeech2_state(80, 80, cfg, seed=2024)
    model = fsm.FastSpeech2(idim=80, odim=80, **cfg)
    model.set_state_dict(state)
    model.eval()
    mu, sigma = syn.mel_stats(seed=7)
    inf = fsm.FastSpeech2Inference(norm.ZScore(paddle.to_tensor(mu), paddle.to_tensor(sigma)), model)
    inf.eval()
    out = {"seed": np.array(2024), "mu": mu, "sigma": sigma}
    for i, (T, alpha) in enumerate([(9, 1.0), (14, 1.0), (11, 1.25)]):
        ids = syn.phoneme_ids(T, seed=500 + i)
        with paddle.no_grad():
            mel = model.inference(paddle.to_tensor(ids), alpha=alpha).numpy()
        out[f"ids{i}"] = ids
        out[f"alpha{i}"] = np.array(alpha, np.float32)
        out[f"mel{i}"] = mel.astype(np.float32)
    with paddle.no_grad():
        out["logmel0"] = inf(paddle.to_tensor(out["ids0"])).numpy().astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "fastspeech2_ljspeech.npz"), **out)
    print("fastspeech2:", {k: v.shape for k, v in out.items() if k.startswith("mel")})


def golden_fastspeech2_multispeaker():
    """aishell3 / vctk shape: spk_embed_dim 256; both integration types; spk_id (incl. the padding id 0)
    and an external speaker embedding."""
    fsm = ref_import.load("parakeet.models.fastspeech2.fastspeech2")
    out = {"seed": np.array(2025)}
    rng = np.random.default_rng(31)
    for kind in ("add", "concat"):
        cfg = dict(syn.FS2_LJSPEECH, spk_embed_dim=256, spk_embed_integration_type=kind)
        state = syn.fastspeech2_state(80, 80, cfg, seed=2025, num_speakers=6)
        model = fsm.FastSpeech2(idim=80, odim=80, num_speakers=6, **cfg)
        model.set_state_dict(state)
        model.eval()
        for i, spk in enumerate([3, 0]):
            ids = syn.phoneme_ids(8 + i, seed=600 + i)
            with paddle.no_grad():
                mel = model.inference(paddle.to_tensor(ids), spk_id=paddle.to_tensor(np.array([spk]))).numpy()
            out[f"{kind}_ids{i}"], out[f"{kind}_spk{i}"], out[f"{kind}_mel{i}"] = ids, np.array(spk), mel.astype(np.float32)
        ids = syn.phoneme_ids(7, seed=610)
        emb = rng.normal(size=(1, 256)).astype(np.float32)
        with paddle.no_grad():
            mel = model.inference(paddle.to_tensor(ids), spembs=paddle.to_tensor(emb[0])).numpy()
        out[f"{kind}_ids2"], out[f"{kind}_spemb2"], out[f"{kind}_mel2"] = ids, emb[0], mel.astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "fastspeech2_multispeaker.npz"), **out)
    print("fastspeech2 multi-speaker:", {k: v.shape for k, v in out.items() if "mel" in k})


def golden_fastspeech2_ffn_variants():
    """positionwise_layer_type "linear" and "conv1d-linear" (encoder.py:145-170); the LJSpeech recipe uses "conv1d"."""
    fsm = ref_import.load("parakeet.models.fastspeech2.fastspeech2")
    out = {"seed": np.array(2026)}
    for kind in ("linear", "conv1d-linear"):
        cfg = dict(syn.FS2_LJSPEECH, positionwise_layer_type=kind)
        state = syn.fastspeech2_state(80, 80, cfg, seed=2026, fixed_duration=2)   # 2 frames per token
        model = fsm.FastSpeech2(idim=80, odim=80, **cfg)
        model.set_state_dict(state)
        model.eval()
        tag = kind.replace("-", "_")
        for i in range(2):
            ids = syn.phoneme_ids(7 + 3 * i, seed=700 + i)
            with paddle.no_grad():
                mel = model.inference(paddle.to_tensor(ids)).numpy()
            out[f"{tag}_ids{i}"], out[f"{tag}_mel{i}"] = ids, mel.astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "fastspeech2_ffn_variants.npz"), **out)
    print("fastspeech2 ffn variants:", {k: v.shape for k, v in out.items() if "mel" in k})


FS2_BLOCK_VARIANTS = {
    # name -> overrides: post-norm blocks (no after_norm), concat_after, in the encoder / decoder stacks
    "postnorm": dict(encoder_normalize_before=False, decoder_normalize_before=False),
    "concat": dict(encoder_concat_after=True, decoder_concat_after=True),
    "mixed": dict(encoder_normalize_before=False, encoder_concat_after=True, positionwise_conv_kernel_size=3),
    # reduction_factor: r mel frames per decoder row (feat_out adim -> odim * r, reshaped; fastspeech2.py:271, :457)
    "r2": dict(reduction_factor=2),
    "r3_nopostnet": dict(reduction_factor=3, postnet_layers=0),
}


def golden_fastspeech2_block_variants():
    """normalize_before=False and concat_after=True (encoder_layer.py:64-115, encoder.py:142-143, 190-191); no recipe of the
    reference sets them, the constructor accepts them."""
    fsm = ref_import.load("parakeet.models.fastspeech2.fastspeech2")
    out = {"seed": np.array(2028)}
    for tag, over in FS2_BLOCK_VARIANTS.items():
        cfg = dict(syn.FS2_LJSPEECH, elayers=2, dlayers=2, **over)
        state = syn.fastspeech2_state(80, 80, cfg, seed=2028, fixed_duration=2)
        model = fsm.FastSpeech2(idim=80, odim=80, **cfg)
        model.set_state_dict(state)
        model.eval()
        for i in range(2):
            ids = syn.phoneme_ids(6 + 5 * i, seed=900 + i)
            with paddle.no_grad():
                mel = model.inference(paddle.to_tensor(ids)).numpy()
            out[f"{tag}_ids{i}"], out[f"{tag}_mel{i}"] = ids, mel.astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "fastspeech2_block_variants.npz"), **out)
    print("fastspeech2 block variants:", {k: v.shape for k, v in out.items() if "mel" in k})


def golden_fastspeech2_tones():
    """tone_embed_dim 64, "add"; tone ids forwarded as FastSpeech2.inference does (1-D)."""
    fsm = ref_import.load("parakeet.models.fastspeech2.fastspeech2")
    cfg = dict(syn.FS2_LJSPEECH, tone_embed_dim=64, tone_embed_integration_type="add")
    state = syn.fastspeech2_state(80, 80, cfg, seed=2027, num_tones=6, fixed_duration=2)
    model = fsm.FastSpeech2(idim=80, odim=80, num_tones=6, **cfg)
    model.set_state_dict(state)
    model.eval()
    out = {"seed": np.array(2027)}
    rng = np.random.default_rng(41)
    for i in range(2):
        ids = syn.phoneme_ids(9 + 4 * i, seed=800 + i)
        tones = rng.integers(0, 6, size=ids.shape[0]).astype(np.int64)   # includes the padding id 0
        with paddle.no_grad():
            mel = model.inference(paddle.to_tensor(ids), tone_id=paddle.to_tensor(tones)).numpy()
        out[f"ids{i}"], out[f"tones{i}"], out[f"mel{i}"] = ids, tones, mel.astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "fastspeech2_tones.npz"), **out)
    print("fastspeech2 tones:", {k: v.shape for k, v in out.items() if "mel" in k})


def golden_pwg():
    pw = ref_import.load("parakeet.models.parallel_wavegan.parallel_wavegan")
    norm = ref_import.load("parakeet.modules.normalizer")
    cfg = dict(syn.PWG_LJSPEECH)
    state = syn.pwg_state(cfg, seed=77, weight_norm=True)
    gen = pw.PWGGenerator(**{k: v for k, v in cfg.items()})
    gen.set_state_dict(state)
    gen.remove_weight_norm()
    gen.eval()
    rng = np.random.default_rng(5)
    out = {"seed": np.array(77)}
    # forward(x, c) on a batch, like tests/unit/test_pwg.py:135-136 (smaller)
    x = rng.normal(size=(2, 1, 4 * 256)).astype(np.float32)
    c = rng.normal(size=(2, 80, 4 + 4)).astype(np.float32)
    with paddle.no_grad():
        out["fwd_y"] = gen(paddle.to_tensor(x), paddle.to_tensor(c)).numpy().astype(np.float32)
    out["fwd_x"], out["fwd_c"] = x, c
    # inference(c) with the in-call randn replaced by a recorded draw
    mel = rng.normal(size=(3, 80)).astype(np.float32)
    noise = rng.normal(size=(1, 1, 3 * 256)).astype(np.float32)
    mu, sigma = syn.mel_stats(seed=8)
    with ref_import.fixed_randn(noise), paddle.no_grad():
        out["inf_wav"] = gen.inference(paddle.to_tensor(mel)).numpy().astype(np.float32)
        pinf = pw.PWGInference(norm.ZScore(paddle.to_tensor(mu), paddle.to_tensor(sigma)), gen)
        out["pinf_wav"] = pinf(paddle.to_tensor(mel * sigma + mu)).numpy().astype(np.float32)
    out["inf_mel"], out["inf_noise"], out["mu"], out["sigma"] = mel, noise.reshape(-1), mu, sigma
    np.savez_compressed(os.path.join(OUT, "pwg_ljspeech.npz"), **out)
    print("pwg:", out["fwd_y"].shape, out["inf_wav"].shape)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    golden_fastspeech2()
    golden_fastspeech2_multispeaker()
    golden_fastspeech2_ffn_variants()
    golden_fastspeech2_tones()
    golden_fastspeech2_block_variants()
    golden_pwg()
    from make_golden_waveflow import golden_waveflow
    golden_waveflow(OUT)
