"""Error of the two PWG matrix paths against the fp64 oracle (full 30-layer generator)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from parakeet_amd import synthetic as syn
from parakeet_amd.parallel_wavegan import PWGGenerator
from oracle import pwg_ref
state = syn.pwg_state()
rng = np.random.default_rng(3)
frames = [24, 40]
mels = [rng.normal(size=(L, 80)).astype(np.float32) for L in frames]
noises = [rng.normal(size=(L * 256,)).astype(np.float32) for L in frames]
ref = [pwg_ref.generator_inference(state, torch.from_numpy(m), torch.from_numpy(n), dtype=torch.float64)[:, 0].numpy() for m, n in zip(mels, noises)]
ref32 = [pwg_ref.generator_inference(state, torch.from_numpy(m), torch.from_numpy(n), dtype=torch.float32)[:, 0].numpy() for m, n in zip(mels, noises)]
gen = PWGGenerator(**syn.PWG_LJSPEECH); gen.set_state_dict(state); gen.eval()
for mode in ("f32", "bf16x3", "f16x3"):
    gen.set_math(mode)
    outs = gen.inference_batch(mels, noises)
    e = max(np.abs(o.numpy()[:, 0] - r).max() / np.abs(r).max() for o, r in zip(outs, ref))
    rms = max(np.sqrt(((o.numpy()[:, 0] - r) ** 2).mean()) / np.sqrt((r ** 2).mean()) for o, r in zip(outs, ref))
    print(f"engine {mode:7s}: rel max err vs fp64 oracle {e:.3e}, rel rms {rms:.3e}")
e = max(np.abs(a - r).max() / np.abs(r).max() for a, r in zip(ref32, ref))
print(f"torch-CPU fp32 oracle: rel max err vs fp64 oracle {e:.3e}")
