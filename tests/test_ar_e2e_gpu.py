"""Autoregressive acoustic models -> WaveFlow end to end on the engine (the pairing the reference uses:
examples/transformer_tts/synthesize.py, examples/tacotron2/synthesize.ipynb) vs the oracles chained, and the loaders /
recipe script on synthetic checkpoint files."""
import os
import pickle

import numpy as np
import pytest
import torch

from oracle import tacotron2_ref as t2
from oracle import transformer_tts_ref as tt
from oracle import waveflow_ref as wfr
from parakeet_amd import checkpoint as ck
from parakeet_amd import synthetic as syn

pytestmark = pytest.mark.gpu

TTS_CFG = dict(syn.TRANSFORMER_TTS_LJSPEECH, elayers=1, dlayers=2, postnet_layers=2)
T2_CFG = dict(syn.TACOTRON2_LJSPEECH, d_encoder=128, encoder_conv_layers=2, d_prenet=64, d_attention_rnn=128,
              d_decoder_rnn=128, d_attention=64, attention_filters=8, attention_kernel_size=7, d_postnet=64,
              postnet_conv_layers=3)
WF_CFG = dict(syn.WAVEFLOW_LJSPEECH, channels=64, n_flows=2)


def _wf(seed=4):
    from parakeet_amd.waveflow import ConditionalWaveFlow
    state = syn.waveflow_state(WF_CFG, seed=seed)
    m = ConditionalWaveFlow(**WF_CFG)
    m.set_state_dict(state)
    m.eval()
    return m, state


def _rel(a, b):
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


def test_transformer_tts_to_waveflow_ragged_batch():
    from parakeet_amd.normalizer import ZScore
    from parakeet_amd.synthesize import ARSynthesizer
    from parakeet_amd.transformer_tts import TransformerTTS, TransformerTTSInference
    state = syn.transformer_tts_state(40, 80, TTS_CFG, seed=31, stop_bias=-6.0)
    am = TransformerTTS(idim=40, odim=80, **TTS_CFG)
    am.set_state_dict(state)
    am.eval()
    mu = np.full(80, -4.0, np.float32)                    # WaveFlow's natural-log mel domain
    sd = np.full(80, 0.5, np.float32)
    voc, wstate = _wf()
    syn_ = ARSynthesizer(TransformerTTSInference(ZScore(mu, sd), am), voc)
    texts = [syn.phoneme_ids(T, idim=40, seed=900 + T) for T in (4, 2)]
    seeds = [3, 4]
    rng = np.random.default_rng(2)
    lens = [int((len(t) + 1) * 1.0) for t in texts]       # maxlenratio 1.0, the stop token held off
    zs = [rng.normal(size=(voc.lengths(L)[0],)).astype(np.float32) for L in lens]
    wavs = syn_.synthesize_batch(texts, seeds=seeds, zs=zs, maxlenratio=1.0)
    for t, sd_, z, w, L in zip(texts, seeds, zs, wavs, lens):
        mel = tt.inference(state, t, TTS_CFG, maxlenratio=1.0, seed=sd_, dtype=torch.float64)[0]
        assert mel.shape[0] == L
        logmel = (mel * torch.as_tensor(sd, dtype=torch.float64) + torch.as_tensor(mu, dtype=torch.float64))
        want = wfr.infer(wstate, logmel.T[None].to(torch.float64), torch.from_numpy(z)[None], WF_CFG, torch.float64)[0].numpy()
        assert w.shape == want.shape and _rel(w.numpy(), want) < 1e-3


def test_tacotron2_to_waveflow_and_from_pretrained(tmp_path):
    from parakeet_amd.synthesize import ARSynthesizer
    from parakeet_amd.tacotron2 import Tacotron2
    state = syn.tacotron2_state(T2_CFG, seed=41, stop_bias=-8.0)
    with open(tmp_path / "step-1000.pdparams", "wb") as f:
        pickle.dump({k: v for k, v in state.items()}, f, protocol=2)
    cfg = {"data": {"n_mels": 80, "sample_rate": 22050}, "model": dict(T2_CFG)}
    am = Tacotron2.from_pretrained(cfg, str(tmp_path / "step-1000"))
    voc, wstate = _wf(seed=6)
    ids = np.random.default_rng(7).integers(1, 37, size=6)
    z = np.random.default_rng(8).normal(size=(voc.lengths(5)[0],)).astype(np.float32)
    wav = ARSynthesizer(am, voc)(ids, seed=9, z=z, max_decoder_steps=5)
    mel = t2.infer(state, ids, T2_CFG, max_decoder_steps=5, seed=9, dtype=torch.float64)["mel_outputs_postnet"]
    want = wfr.infer(wstate, mel.T[None], torch.from_numpy(z)[None], WF_CFG, torch.float64)[0].numpy()
    assert wav.shape == want.shape and _rel(wav.numpy(), want) < 1e-3


def test_example_ar_script_writes_wavs(tmp_path):
    """examples/synthesize_ar.py (the arguments of examples/transformer_tts/synthesize.py) on synthetic checkpoints."""
    import subprocess
    import sys
    import wave
    import yaml
    state = syn.transformer_tts_state(40, 80, TTS_CFG, seed=31, stop_bias=-6.0)
    with open(tmp_path / "tts.pdz", "wb") as f:
        pickle.dump({"main_params": {k: ("t", v) for k, v in state.items()}}, f, protocol=2)
    with open(tmp_path / "waveflow.pdparams", "wb") as f:
        pickle.dump(dict(syn.waveflow_state(WF_CFG, seed=4)), f, protocol=2)
    np.save(tmp_path / "speech_stats.npy", np.stack([np.full(80, -4.0, np.float32), np.full(80, 0.5, np.float32)]))
    (tmp_path / "tts.yaml").write_text(yaml.safe_dump({"fs": 22050, "n_mels": 80, "model": dict(TTS_CFG)}))
    (tmp_path / "waveflow.yaml").write_text(yaml.safe_dump({"data": {"n_mels": 80}, "model": {
        k: (list(v) if isinstance(v, (list, tuple)) else v) for k, v in WF_CFG.items() if k != "n_mels"}}))
    phones = ["<pad>", "<unk>"] + ["P%d" % i for i in range(37)] + ["<eos>"]
    (tmp_path / "phone_id_map.txt").write_text("".join(f"{p} {i}\n" for i, p in enumerate(phones)))
    (tmp_path / "sentences.txt").write_text("001 P1 P2 P3\n002 P7 P9 P10 P11 P12\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "synthesize_ar.py"),
                        "--transformer-tts-config", str(tmp_path / "tts.yaml"),
                        "--transformer-tts-checkpoint", str(tmp_path / "tts.pdz"),
                        "--transformer-tts-stat", str(tmp_path / "speech_stats.npy"),
                        "--waveflow-config", str(tmp_path / "waveflow.yaml"),
                        "--waveflow-checkpoint", str(tmp_path / "waveflow"),
                        "--phones-dict", str(tmp_path / "phone_id_map.txt"), "--text", str(tmp_path / "sentences.txt"),
                        "--output-dir", str(tmp_path / "out")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    am = ck.load_transformer_tts(str(tmp_path / "tts.yaml"), tmp_path / "tts.pdz", tmp_path / "speech_stats.npy",
                                 tmp_path / "phone_id_map.txt")[0]
    voc = ck.load_waveflow(str(tmp_path / "waveflow.yaml"), str(tmp_path / "waveflow"))
    for i, (utt, ids) in enumerate((("001", [2, 3, 4]), ("002", [8, 10, 11, 12, 13]))):
        L = int(am.acoustic_model.inference(np.array(ids), seed=i)[0].shape[0])     # default maxlenratio 10
        with wave.open(str(tmp_path / "out" / f"{utt}.wav"), "rb") as w:
            assert w.getframerate() == 22050 and w.getnframes() == voc.lengths(L)[1]
