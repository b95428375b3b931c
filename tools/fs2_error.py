"""Error of the two FastSpeech2 GEMM paths against the fp64 oracle."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from parakeet_amd import synthetic as syn
from parakeet_amd.fastspeech2 import FastSpeech2
from oracle import fastspeech2_ref as ref
state = syn.fastspeech2_state(80, 80, seed=77)
texts = [syn.phoneme_ids(T, seed=300 + i) for i, T in enumerate([60, 33, 90])]
cfg = {k: syn.FS2_LJSPEECH[k] for k in ref.DEFAULT_CFG if k in syn.FS2_LJSPEECH}
want = [ref.inference(state, t, cfg, dtype=torch.float64).numpy() for t in texts]
w32 = [ref.inference(state, t, cfg, dtype=torch.float32).numpy() for t in texts]
m = FastSpeech2(80, 80, **syn.FS2_LJSPEECH); m.set_state_dict(state); m.eval()
for mode in ("f32", "f16x3"):
    m.set_math(mode)
    outs = m.inference_batch(texts)
    l1 = max(np.abs(o.numpy() - w).mean() for o, w in zip(outs, want))
    mx = max(np.abs(o.numpy() - w).max() for o, w in zip(outs, want))
    print(f"engine {mode:6s}: mel L1 vs fp64 oracle {l1:.3e}, max abs {mx:.3e}")
print(f"torch-CPU fp32 oracle: mel L1 {max(np.abs(a - w).mean() for a, w in zip(w32, want)):.3e}, max abs {max(np.abs(a - w).max() for a, w in zip(w32, want)):.3e}")
