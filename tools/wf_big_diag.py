"""Diagnostic (round 5): a 24-utterance WaveFlow call against two-utterance calls -- determinism, where the differences are, and
both against the fp64 oracle for the utterance that differs most."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import numpy as np, torch
from parakeet_amd import synthetic as syn
from parakeet_amd.waveflow import ConditionalWaveFlow
C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = dict(syn.WAVEFLOW_LJSPEECH, channels=C)
model = ConditionalWaveFlow(**cfg)
state = syn.waveflow_state(cfg, seed=77, weight_norm=True)
model.set_state_dict(state); model.eval()
rng = np.random.default_rng(78)
frames = [int(t) for t in rng.integers(500, 2001, size=24)]
mels = [np.maximum(rng.normal(-4, 2, size=(cfg["n_mels"], T)), np.log(1e-5)).astype(np.float32) for T in frames]
zs = [rng.normal(size=(model.lengths(T)[0],)).astype(np.float32) for T in frames]
print("frames", frames)
big1 = [o.numpy().copy() for o in model.infer_batch(mels, zs)]
big2 = [o.numpy().copy() for o in model.infer_batch(mels, zs)]
print("big call deterministic:", all(np.array_equal(a, b) for a, b in zip(big1, big2)))
worst = (0, -1)
for pair in ((0, 1), (5, 6), (11, 12), (22, 23)):
    s1 = [o.numpy().copy() for o in model.infer_batch([mels[b] for b in pair], [zs[b] for b in pair])]
    s2 = [o.numpy().copy() for o in model.infer_batch([mels[b] for b in pair], [zs[b] for b in pair])]
    print("pair", pair, "deterministic:", all(np.array_equal(a, b) for a, b in zip(s1, s2)))
    for o, b in zip(s1, pair):
        d = np.abs(o - big1[b]); m = np.abs(big1[b]).max()
        i = int(d.argmax())
        print(f"  utt {b} frames {frames[b]} n {o.size}: max diff {d.max()/m:.3e} at sample {i} ({i/o.size:.3f} of the utterance), "
              f"mean diff {d.mean()/m:.3e}, samples with diff > 1e-5: {(d/m > 1e-5).sum()}, max|wav| {m:.3f}")
        if d.max() / m > worst[0]: worst = (d.max() / m, b, o)
if len(sys.argv) > 2:
    from oracle import waveflow_ref as ref
    _, b, small = worst
    t = time.time()
    want = ref.infer(state, torch.from_numpy(mels[b])[None], torch.from_numpy(zs[b])[None], cfg, torch.float64)[0].numpy()
    m = np.abs(want).max()
    print(f"oracle for utt {b}: {time.time()-t:.0f} s; big call err {np.abs(big1[b]-want).max()/m:.3e}, small call err {np.abs(small-want).max()/m:.3e}")
    want32 = ref.infer(state, torch.from_numpy(mels[b])[None], torch.from_numpy(zs[b])[None], cfg, torch.float32)[0].numpy()
    print(f"   fp32 oracle err vs fp64: {np.abs(want32-want).max()/m:.3e}")
