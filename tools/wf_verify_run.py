"""Round 6: the 12-wave default-math WaveFlow layer kernel (profile build, PK_WF_ALLOW_3WAVE=1) at BASELINE config 5's shape,
N calls; every call is compared with the exact-fp32 unfused path (another kernel family, deterministic), and -- under
PK_WF_ABLATE=128 -- the idle wave of every workgroup verifies the LDS weight slabs against memory while the working waves
read them (wf_layer.hip `verify_slab`; the C side prints `wf_verify:` lines per call on stderr).  Where the wrong samples of a
bad call lie (flow order, row, workgroup) is printed next to the verifier's records of the same call.
  PK_PROFILE_LIB=1 PK_WF_ALLOW_3WAVE=1 [PK_WF_ABLATE=128] python tools/wf_verify_run.py [calls] [waves]"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import collections
import numpy as np
from parakeet_amd import synthetic as syn
from parakeet_amd.waveflow import ConditionalWaveFlow
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
WAVES = int(sys.argv[2]) if len(sys.argv) > 2 else 12
shape = [int(v) for v in os.environ.get("WF_FRAMES", ",".join(["640"] * 8)).split(",")]
cfg = dict(syn.WAVEFLOW_LJSPEECH, channels=int(os.environ.get("WF_C", 64)))
state = syn.waveflow_state(cfg, seed=77, weight_norm=True)
rng = np.random.default_rng(78)
def make(math=None, **opts):
    m = ConditionalWaveFlow(**cfg); m.set_state_dict(state); m.eval()
    if math: m.set_math(math)
    for k, v in opts.items(): m.set_option(k, v)
    return m
ref_m = make("f32")
m = make(**({"layer_waves": WAVES} if WAVES else {}))
mels = [np.maximum(rng.normal(-4, 2, size=(80, T)), np.log(1e-5)).astype(np.float32) for T in shape]
zs = [rng.normal(size=(m.lengths(T)[0],)).astype(np.float32) for T in shape]
ref = [o.numpy() for o in ref_m.infer_batch(mels, zs)]
G, GAP = 16, 128
Wb = [r.size // G for r in ref]
npos = GAP + sum(w + GAP for w in Wb)
ntiles = (npos + 127) // 128 * 128 // 32
tpw = max(1, (ntiles + 255) // 256)
print(f"frames {shape}: positions per row {npos}, tiles {ntiles}, tiles per workgroup {tpw}, waves option {WAVES}", flush=True)
nbad_calls = 0
for rep in range(N):
    sys.stderr.write(f"--- call {rep}\n"); sys.stderr.flush()
    outs = [o.numpy() for o in m.infer_batch(mels, zs)]
    off = GAP
    wgs, rows = collections.Counter(), collections.Counter()
    nbad = 0
    for b, (o, r) in enumerate(zip(outs, ref)):
        peak = np.abs(r).max()
        bad = np.nonzero(np.abs(o - r) / peak > 1e-5)[0]
        nbad += bad.size
        for i in bad:
            w, h = int(i) // G, int(i) % G
            t = (off + w) // 32
            wgs[t // tpw] += 1; rows[h] += 1
        off += Wb[b] + GAP
    if nbad:
        nbad_calls += 1
        print(f"call {rep}: {nbad} bad samples; workgroups {dict(sorted(wgs.items())[:12])}{' ...' if len(wgs) > 12 else ''}; output rows {dict(sorted(rows.items()))}", flush=True)
print(f"{nbad_calls} of {N} calls wrong")
