import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np, torch
from parakeet_amd import synthetic as syn
from parakeet_amd.parallel_wavegan import PWGGenerator
from parakeet_amd.runtime import Context
B, L = 32, 640
gen = PWGGenerator(**syn.PWG_LJSPEECH); gen.set_state_dict(syn.pwg_state()); gen.eval()
rng = np.random.default_rng(42)
mels = [torch.tensor(rng.normal(size=(L,80)).astype(np.float32)).cuda() for _ in range(B)]
noises = [torch.randn(L*256, device='cuda') for _ in range(B)]
ctx = Context.get()
for i in range(2): gen.inference_batch(mels, noises)
torch.cuda.synchronize()
t=time.time(); n=3
for i in range(n): gen.inference_batch(mels, noises)
torch.cuda.synchronize(); dt=(time.time()-t)/n
print(f"PWG B={B} L={L}: {dt*1e3:.1f} ms/batch, {B*L*256/dt/1e6:.2f} Msamples/s, {B*L*256/dt/22050:.0f}x RT")
ctx.prof_enable(True); ctx.prof_reset()
gen.inference_batch(mels, noises)
for k,(n_,ms) in ctx.prof_dump().items(): print(f"  {k:16s} n={n_:3d} total={ms:9.3f} ms avg={ms/n_:8.3f} ms")
ctx.prof_enable(False)
