import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np, torch
from parakeet_amd import synthetic as syn
from parakeet_amd.waveflow import ConditionalWaveFlow
from parakeet_amd.runtime import Context
B, L = 8, 640
C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
MATH = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "-" else None
WAVES = int(sys.argv[3]) if len(sys.argv) > 3 else 0     # option "layer_waves": 0 = the launcher's choice, 8, 12
cfg = dict(syn.WAVEFLOW_LJSPEECH, channels=C)
m = ConditionalWaveFlow(**cfg); m.set_state_dict(syn.waveflow_state(cfg)); m.eval()
if MATH: m.set_math(MATH)
if WAVES: m.set_option("layer_waves", WAVES)
if os.environ.get("PK_QWF_PERSISTENT"): m.set_option("persistent", int(os.environ["PK_QWF_PERSISTENT"]))   # 0: one launch per layer
rng = np.random.default_rng(0)
mels = [torch.tensor(np.maximum(rng.normal(-4, 2, size=(80, L)), np.log(1e-5)).astype(np.float32)).cuda() for _ in range(B)]
zs = [torch.randn(m.lengths(L)[0], device='cuda') for _ in range(B)]
ctx = Context.get()
out = m.infer_batch(mels, zs); torch.cuda.synchronize()
assert all(bool(torch.isfinite(o).all()) for o in out)
t=time.time(); n=5
for i in range(n): m.infer_batch(mels, zs)
t_enq=(time.time()-t)/n     # host time to ENQUEUE a batch (no synchronisation inside infer_batch with device-resident I/O)
torch.cuda.synchronize(); dt=(time.time()-t)/n
ns = sum(o.numel() for o in out)
print(f"WaveFlow C={C} math={MATH} waves={WAVES} persistent={os.environ.get('PK_QWF_PERSISTENT', 'default')} B={B} L={L}: {dt*1e3:.1f} ms/batch, {ns/dt/1e6:.2f} Msamples/s, {ns/dt/22050:.0f}x RT; host enqueue {t_enq*1e3:.1f} ms/batch")
ctx.prof_enable(True); ctx.prof_reset()
m.infer_batch(mels, zs)
for k,(n_,ms) in ctx.prof_dump().items(): print(f"  {k:20s} n={n_:5d} total={ms:9.3f} ms avg={ms/n_*1e3:8.1f} us")
