"""Waveform error of the WaveFlow engine against the fp64 oracle at bench.py's utterance shape (640 frames, all 8 flows), per
math mode.  usage: wf_error.py [channels]"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.set_num_threads(8)
from oracle import waveflow_ref as ref
from parakeet_amd import synthetic as syn
from parakeet_amd.waveflow import ConditionalWaveFlow
C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = dict(syn.WAVEFLOW_LJSPEECH, channels=C)
state = syn.waveflow_state(cfg, seed=31, weight_norm=True)
m = ConditionalWaveFlow(**cfg); m.set_state_dict(state); m.eval()
rng = np.random.default_rng(32)
mel = np.maximum(rng.normal(-4, 2, size=(80, 640)), np.log(1e-5)).astype(np.float32)
z = rng.normal(size=(m.lengths(640)[0],)).astype(np.float32)
with torch.no_grad():
    want = ref.infer(state, torch.from_numpy(mel)[None], torch.from_numpy(z)[None], cfg, torch.float64)[0].numpy()
    w32 = ref.infer(state, torch.from_numpy(mel)[None], torch.from_numpy(z)[None], cfg, torch.float32)[0].numpy()
pk = np.abs(want).max()
print(f"C={C}: oracle in fp32 (torch CPU)   rel max {np.abs(w32 - want).max() / pk:.3e}  rms {np.sqrt(np.mean((w32 - want) ** 2)) / pk:.3e}")
for math in ("f32", "f16x3", "f16"):
    m.set_math(math)
    got = m.infer_batch([mel], [z])[0].numpy()
    d = got - want
    print(f"C={C}: engine math={math:6s}          rel max {np.abs(d).max() / pk:.3e}  rms {np.sqrt(np.mean(d ** 2)) / pk:.3e}")
