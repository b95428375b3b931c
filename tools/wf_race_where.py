"""Round 5: where are the samples a non-deterministic 64-channel WaveFlow call gets wrong?  One flow only (no propagation between
flows); sample index -> (position in the packed row, row of the group) -> (workgroup, wave) of the layer kernel's tile."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import collections
import numpy as np
from parakeet_amd import synthetic as syn
from parakeet_amd.waveflow import ConditionalWaveFlow
shape = [int(v) for v in os.environ.get("WF_FRAMES", "1200,1200").split(",")]
NF = int(os.environ.get("WF_FLOWS", 8))
cfg = dict(syn.WAVEFLOW_LJSPEECH, channels=64, n_flows=NF)
state = syn.waveflow_state(cfg, seed=77, weight_norm=True)
rng = np.random.default_rng(78)
def make(math=None, **opts):
    m = ConditionalWaveFlow(**cfg); m.set_state_dict(state); m.eval()
    if math: m.set_math(math)
    for k, v in opts.items(): m.set_option(k, v)
    return m
ref_m = make("f32"); m = make(**({"layer_waves": int(os.environ["WF_WAVES"])} if os.environ.get("WF_WAVES") else {}))
mels = [np.maximum(rng.normal(-4, 2, size=(80, T)), np.log(1e-5)).astype(np.float32) for T in shape]
zs = [rng.normal(size=(m.lengths(T)[0],)).astype(np.float32) for T in shape]
ref = [o.numpy() for o in ref_m.infer_batch(mels, zs)]
G, GAP = 16, 128
Wb = [r.size // G for r in ref]
npos = GAP + sum(w + GAP for w in Wb)
ntiles = (npos + 127) // 128 * 128 // 32
tpw = max(1, (ntiles + 255) // 256)
print(f"frames {shape} flows {NF}: positions per row {npos}, tiles {ntiles}, tiles per workgroup {tpw}")
for rep in range(int(os.environ.get("WF_REP", 4))):
    outs = [o.numpy() for o in m.infer_batch(mels, zs)]
    off = GAP
    waves, rows, tiles = collections.Counter(), collections.Counter(), collections.Counter()
    nbad = 0
    for b, (o, r) in enumerate(zip(outs, ref)):
        peak = np.abs(r).max()
        bad = np.nonzero(np.abs(o - r) / peak > 1e-5)[0]
        nbad += bad.size
        for i in bad:
            w, h = int(i) // G, int(i) % G
            t = (off + w) // 32
            waves[t % tpw] += 1; rows[h] += 1; tiles[(t // tpw, t % tpw)] += 1
        off += Wb[b] + GAP
    if nbad:
        print(f"run {rep}: {nbad} bad samples; by wave of the workgroup {dict(sorted(waves.items()))}; by row {dict(sorted(rows.items()))}")
        print(f"        tiles (workgroup, wave) hit: {sorted(tiles.items())[:24]}")
