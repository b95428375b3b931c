"""Round 5: which WaveFlow path is not deterministic?  Runs one call shape several times per engine variant and counts the
samples that differ between repeats and from the exact-fp32 unfused path."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from parakeet_amd import synthetic as syn
from parakeet_amd.waveflow import ConditionalWaveFlow
C = int(os.environ.get("WF_C", 64))
shape = [int(v) for v in os.environ.get("WF_FRAMES", "1599,1128").split(",")]
REP = int(os.environ.get("WF_REP", 4))
cfg = dict(syn.WAVEFLOW_LJSPEECH, channels=C)
state = syn.waveflow_state(cfg, seed=77, weight_norm=True)
rng = np.random.default_rng(78)
def make(opts, math=None):
    m = ConditionalWaveFlow(**cfg); m.set_state_dict(state); m.eval()
    if math: m.set_math(math)
    for k, v in opts.items(): m.set_option(k, v)
    return m
base = make({})
mels = [np.maximum(rng.normal(-4, 2, size=(cfg["n_mels"], T)), np.log(1e-5)).astype(np.float32) for T in shape]
zs = [rng.normal(size=(base.lengths(T)[0],)).astype(np.float32) for T in shape]
def run(m): return np.concatenate([o.numpy() for o in m.infer_batch(mels, zs)])
ref = run(make({}, "f32")); ref2 = run(make({}, "f32"))
peak = np.abs(ref).max()
print(f"C={C} frames={shape} samples={ref.size}; f32 unfused deterministic: {np.array_equal(ref, ref2)}")
variants = [("default", {}, None), ("waves8", {"layer_waves": 8}, None), ("nofuse_step", {"fuse_step": 0}, None),
            ("waves8_nofuse_step", {"layer_waves": 8, "fuse_step": 0}, None), ("f16", {}, "f16"), ("f16_waves8", {"layer_waves": 8}, "f16")]
if C == 64: variants += [("waves6", {"layer_waves": 6}, None)]
variants += [("persistent", {"persistent": 1}, None)]
if os.environ.get("WF_VARIANTS"): variants = [v for v in variants if v[0] in os.environ["WF_VARIANTS"].split(",")]
for name, opts, math in variants:
    try:
        m = make(opts, math)
        outs = [run(m) for _ in range(REP)]
    except Exception as e:
        print(f"{name:22s} failed: {e}"); continue
    tol = 2e-2 if math == "f16" else 1e-5
    ndiff = [int((o != outs[0]).sum()) for o in outs[1:]]
    bad = [int((np.abs(o - ref) / peak > tol).sum()) for o in outs]
    worst = [float(np.abs(o - ref).max() / peak) for o in outs]
    if REP > 8:
        print(f"{name:22s} {REP} runs: {sum(1 for b in bad if b)} with samples off the f32 path by > {tol:g} (counts {sorted(b for b in bad if b)}), {sum(1 for n in ndiff if n)} differ from the first; worst {max(worst):.2e}")
    else:
        print(f"{name:22s} samples differing from the first repeat: {ndiff}; samples off the f32 path by > {tol:g}: {bad}; max {['%.2e' % w for w in worst]}")
