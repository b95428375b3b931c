"""One line per run for one-box A/Bs of the PWG layer kernel: batch time, the layer kernel's average launch, and a hash of the waveform
(fixed inputs: variants that only re-schedule must give the same bits).  PK_PROFILE_LIB=1 selects the profile / variant library."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hashlib, time, numpy as np, torch
from parakeet_amd import synthetic as syn
from parakeet_amd.parallel_wavegan import PWGGenerator
from parakeet_amd.runtime import Context
B, L, N = int(os.environ.get("PK_QPWG_B", 32)), 640, int(os.environ.get("PK_QPWG_N", 10))
gen = PWGGenerator(**syn.PWG_LJSPEECH); gen.set_state_dict(syn.pwg_state()); gen.eval()
if os.environ.get("PK_QPWG_NZ"): gen.set_option("noise_fed_first", int(os.environ["PK_QPWG_NZ"]))   # round 6: 0 = first_conv + the ordinary first block
rng = np.random.default_rng(42)
mels = [torch.tensor(rng.normal(size=(L, 80)).astype(np.float32)).cuda() for _ in range(B)]
noises = [torch.tensor(rng.normal(size=L * 256).astype(np.float32)).cuda() for _ in range(B)]
ctx = Context.get()
for i in range(3): out = gen.inference_batch(mels, noises)
torch.cuda.synchronize()
t = time.time()
for i in range(N): out = gen.inference_batch(mels, noises)
torch.cuda.synchronize(); dt = (time.time() - t) / N
ctx.prof_enable(True); ctx.prof_reset()
for i in range(3): gen.inference_batch(mels, noises)
p = ctx.prof_dump(); ctx.prof_enable(False)
k = next(k for k in ("pwg_layer_h3", "pwg_layer_b3", "pwg_layer") if k in p)
h = hashlib.sha256(torch.cat([o.reshape(-1) for o in out]).cpu().numpy().tobytes()).hexdigest()[:12]
edge = "  ".join(f"{n} {p[n][1]/p[n][0]*1e3:.1f}" for n in ("pwg_first", "pwg_last", "pwg_last_h3") if n in p)
print(f"{sys.argv[1] if len(sys.argv) > 1 else 'run':28s} {dt*1e3:7.2f} ms/batch  layer kernel {p[k][1]/p[k][0]*1e3:8.1f} us x {p[k][0]//3}  {edge}  wav {h}")
