import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np, torch
from parakeet_amd import synthetic as syn
from parakeet_amd.speedyspeech import SpeedySpeech
from parakeet_amd.runtime import Context
B, T = 32, 128
m = SpeedySpeech(vocab_size=70, tone_size=7, **syn.SPEEDYSPEECH_BAKER); m.set_state_dict(syn.speedyspeech_state()); m.eval()
rng = np.random.default_rng(0)
texts = [rng.integers(1, 70, size=T) for _ in range(B)]; tones = [rng.integers(1, 7, size=T) for _ in range(B)]
for _ in range(2): outs = m.inference_batch(texts, tones)
torch.cuda.synchronize(); t = time.time(); n = 5
for _ in range(n): outs = m.inference_batch(texts, tones)
torch.cuda.synchronize(); dt = (time.time() - t) / n
frames = sum(o.shape[0] for o in outs)
print(f"SpeedySpeech baker B={B} T={T}: {frames} frames, {dt*1e3:.2f} ms/batch, {B/dt:.0f} utt/s, {frames*256/22050/dt:.0f}x RT (mel only)")
ctx = Context.get(); ctx.prof_enable(True); ctx.prof_reset(); m.inference_batch(texts, tones)
for k, (n_, ms) in ctx.prof_dump().items(): print(f"  {k:24s} n={n_:3d} total={ms:8.3f} ms")
