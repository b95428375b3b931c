import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np, torch
from parakeet_amd import synthetic as syn
from parakeet_amd.fastspeech2 import FastSpeech2
from parakeet_amd.runtime import Context
B, T = (int(sys.argv[1]) if len(sys.argv) > 1 else 32), 128
m = FastSpeech2(80, 80, **syn.FS2_LJSPEECH); m.set_state_dict(syn.fastspeech2_state(fixed_duration=5)); m.eval()
for kv in os.environ.get("PK_QFS2_OPTS", "").split(","):     # e.g. PK_QFS2_OPTS=ffn_planes=0
    if "=" in kv: m.set_option(kv.split("=")[0], int(kv.split("=")[1]))
texts = [syn.phoneme_ids(T, seed=i) for i in range(B)]
ctx = Context.get()
for i in range(2): m.inference_batch(texts)
torch.cuda.synchronize()
t=time.time(); n=5
for i in range(n): m.inference_batch(texts)
torch.cuda.synchronize(); dt=(time.time()-t)/n
print(f"FS2 B={B} T={T} L={T*5} opts={os.environ.get('PK_QFS2_OPTS', '')}: {dt*1e3:.2f} ms/batch, {B/dt:.0f} utt/s, {30.26*B/dt/1e3:.1f} TFLOP/s")
ctx.prof_enable(True); ctx.prof_reset()
m.inference_batch(texts)
tot=0
for k,(n_,ms) in ctx.prof_dump().items(): print(f"  {k:20s} n={n_:3d} total={ms:9.3f} ms avg={ms/n_:8.4f} ms"); tot+=ms
print("  sum", tot)
ctx.prof_enable(False)
