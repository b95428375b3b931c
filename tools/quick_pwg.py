import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np, torch
from parakeet_amd import synthetic as syn
from parakeet_amd.parallel_wavegan import PWGGenerator
from parakeet_amd.runtime import Context
B, L = int(os.environ.get("PK_QPWG_B", 32)), int(os.environ.get("PK_QPWG_L", 640))
cfg = dict(syn.PWG_LJSPEECH)
if os.environ.get("PK_QPWG_SCALES"): cfg["upsample_scales"] = [int(v) for v in os.environ["PK_QPWG_SCALES"].split(",")]   # e.g. 4,5,3,5 = hop 300 (baker / vctk)
HOP = int(np.prod(cfg["upsample_scales"]))
gen = PWGGenerator(**cfg); gen.set_state_dict(syn.pwg_state(cfg)); gen.eval()
rng = np.random.default_rng(42)
mels = [torch.tensor(rng.normal(size=(L,80)).astype(np.float32)).cuda() for _ in range(B)]
noises = [torch.randn(L*HOP, device='cuda') for _ in range(B)]
ctx = Context.get()
for i in range(2): gen.inference_batch(mels, noises)
torch.cuda.synchronize()
t=time.time(); n=3
for i in range(n): gen.inference_batch(mels, noises)
torch.cuda.synchronize(); dt=(time.time()-t)/n
print(f"PWG B={B} L={L} hop={HOP}: {dt*1e3:.1f} ms/batch, {B*L*HOP/dt/1e6:.2f} Msamples/s, {B*L*HOP/dt/22050:.0f}x RT")
ctx.prof_enable(True); ctx.prof_reset()
gen.inference_batch(mels, noises)
for k,(n_,ms) in ctx.prof_dump().items(): print(f"  {k:16s} n={n_:3d} total={ms:9.3f} ms avg={ms/n_:8.3f} ms")
ctx.prof_enable(False)
