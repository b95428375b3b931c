#!/usr/bin/env python
"""Golden vectors of the autoregressive acoustic models from the REFERENCE's own Python source
(/root/reference/parakeet/models/transformer_tts/transformer_tts.py, models/tacotron2.py) executed over the
torch-backed paddle stand-in.  Build container only.

Both models keep the decoder prenet's dropout ON at inference (modules/tacotron2/decoder.py:78-81,
models/tacotron2.py:61-80).  The stand-in's ``F.dropout`` hands every active call to DROPOUT_HOOK, which applies the
engine's counter-based dropout stream (oracle/philox_ref.py ``dropout_keep``), so the vectors are reproducible and the
HIP engine can be compared with them bit for bit in its decisions (keep / drop) and to fp32 accuracy in its values.
Weights are regenerated from seeds by parakeet_amd.synthetic."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

ref_import.setup()
import paddle  # noqa: E402  (stand-in or real, see ref_import)

from oracle import philox_ref  # noqa: E402
from oracle import tacotron2_ref as t2_ref  # noqa: E402
from oracle import transformer_tts_ref as tt_ref  # noqa: E402
from parakeet_amd import synthetic as syn  # noqa: E402

OUT = ref_import.golden_dir()


class TransformerTTSDropout:
    """F.dropout hook for TransformerTTS.inference: the prenet is applied to the whole prefix (1, step, units) once per
    decoding step, layer after layer, so the step is the row count and the layer is the call count modulo n_layers."""

    def __init__(self, seed, n_layers, units):
        self.drop = tt_ref.stream_dropout(seed, n_layers, units)
        self.n_layers, self.calls = n_layers, 0

    def __call__(self, shape, p):
        assert p == 0.5 and len(shape) == 3 and shape[0] == 1
        layer = self.calls % self.n_layers
        self.calls += 1
        return np.asarray(self.drop(shape[1], layer, shape[1], shape[2]))[None]


sys.path.insert(0, os.path.join(ref_import.ROOT, "tests"))
from ar_cases import T2_CASES, TTS_CASES  # noqa: E402


def golden_transformer_tts():
    ttm = ref_import.load("parakeet.models.transformer_tts.transformer_tts")
    out = {}
    for name, over, idim, T, seed, skw, kw in TTS_CASES:
        cfg = dict(syn.TRANSFORMER_TTS_LJSPEECH, **over)
        state = syn.transformer_tts_state(idim, 80, cfg, seed=seed, **skw)
        model = ttm.TransformerTTS(idim=idim, odim=80, **cfg)
        model.set_state_dict(state)
        model.eval()
        ids = syn.phoneme_ids(T, idim=idim, seed=700 + seed)
        spemb = None
        if cfg.get("spk_embed_dim"):
            spemb = np.random.default_rng(900 + seed).standard_normal(cfg["spk_embed_dim"]).astype(np.float32)
            out[f"{name}_spemb"] = spemb
        speech = None
        if cfg.get("use_gst"):
            speech = np.random.default_rng(950 + seed).standard_normal((70, 80)).astype(np.float32)
            out[f"{name}_speech"] = speech
        hook = TransformerTTSDropout(seed=seed, n_layers=max(cfg["dprenet_layers"], 1), units=cfg["dprenet_units"])
        with ref_import.dropout_hook(hook), paddle.no_grad():
            mel, probs, att = model.inference(paddle.to_tensor(ids), spembs=None if spemb is None else paddle.to_tensor(spemb),
                                              speech=None if speech is None else paddle.to_tensor(speech), **kw)
        out[f"{name}_ids"] = ids
        out[f"{name}_seed"] = np.array(seed)
        out[f"{name}_mel"] = mel.numpy().astype(np.float32)
        out[f"{name}_probs"] = probs.numpy().astype(np.float32)
        out[f"{name}_att"] = att.numpy().astype(np.float32)
        print("transformer_tts", name, out[f"{name}_mel"].shape, out[f"{name}_att"].shape,
              "probs", np.round(out[f"{name}_probs"], 3)[:12])
    np.savez_compressed(os.path.join(OUT, "transformer_tts.npz"), **out)


class Tacotron2Dropout:
    """F.dropout hook for Tacotron2.infer: only DecoderPreNet calls it with training=True (models/tacotron2.py:76-79),
    twice per decoding step on a (1, d_prenet) query."""

    def __init__(self, seed, units, p):
        self.drop = t2_ref.stream_dropout(seed, units, p)
        self.p, self.calls = p, 0

    def __call__(self, shape, p):
        assert p == self.p and len(shape) == 2 and shape[0] == 1
        step, layer = divmod(self.calls, 2)
        self.calls += 1
        return np.asarray(self.drop(step, layer, shape[1]))[None]


def golden_tacotron2():
    t2m = ref_import.load("parakeet.models.tacotron2")
    out = {}
    for name, over, T, seed, skw, max_steps in T2_CASES:
        cfg = dict(syn.TACOTRON2_LJSPEECH, **over)
        state = syn.tacotron2_state(cfg, seed=seed, **skw)
        model = t2m.Tacotron2(**cfg)
        model.set_state_dict(state)
        model.eval()
        rng = np.random.default_rng(800 + seed)
        ids = rng.integers(1, cfg["vocab_size"], size=(1, T)).astype(np.int64)
        tones = rng.integers(0, cfg["n_tones"], size=(1, T)).astype(np.int64) if cfg["n_tones"] else None
        gc = rng.standard_normal((1, cfg["d_global_condition"])).astype(np.float32) if cfg.get("d_global_condition") else None
        with ref_import.dropout_hook(Tacotron2Dropout(seed, cfg["d_prenet"], cfg["p_prenet_dropout"])), paddle.no_grad():
            o = model.infer(paddle.to_tensor(ids), max_decoder_steps=max_steps,
                            tones=None if tones is None else paddle.to_tensor(tones),
                            global_condition=None if gc is None else paddle.to_tensor(gc))
        out[f"{name}_ids"] = ids[0]
        if tones is not None:
            out[f"{name}_tones"] = tones[0]
        if gc is not None:
            out[f"{name}_global_condition"] = gc[0]
        for k in ("mel_output", "mel_outputs_postnet", "alignments", "stop_logits"):
            if k in o:
                out[f"{name}_{k}"] = o[k].numpy()[0].astype(np.float32)
        print("tacotron2", name, out[f"{name}_mel_output"].shape, out[f"{name}_alignments"].shape,
              np.round(1 / (1 + np.exp(-out[f"{name}_stop_logits"])), 3)[-4:] if f"{name}_stop_logits" in out else "")
    np.savez_compressed(os.path.join(OUT, "tacotron2.npz"), **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["transformer_tts", "tacotron2"]
    if "transformer_tts" in which:
        golden_transformer_tts()
    if "tacotron2" in which and "golden_tacotron2" in globals():
        golden_tacotron2()
