"""What a guarded / sampled PWG call costs next to a plain one (32 x 640 frames): scale_guard 1 with no sample inside the timed window against
scale_guard 2 (every call measures).  PK_PROFILE_LIB=1 selects the profile / variant library."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hashlib, time, numpy as np, torch
from parakeet_amd import synthetic as syn
from parakeet_amd.parallel_wavegan import PWGGenerator
B, L, N = 32, 640, 10
gen = PWGGenerator(**syn.PWG_LJSPEECH); gen.set_state_dict(syn.pwg_state()); gen.eval()
rng = np.random.default_rng(42)
mels = [torch.tensor(rng.normal(size=(L, 80)).astype(np.float32)).cuda() for _ in range(B)]
noises = [torch.tensor(rng.normal(size=L * 256).astype(np.float32)).cuda() for _ in range(B)]
res = {}
for mode in (1, 2, 3):   # 3 = scale_guard 1 with EVERY later call sampled (verdict deferred: no in-call synchronisation)
    gen.set_option("scale_guard", 1 if mode == 3 else mode)
    gen.set_option("scale_guard_every", 0 if mode == 1 else (1 if mode == 3 else 16))
    for i in range(3): out = gen.inference_batch(mels, noises)
    torch.cuda.synchronize(); t = time.time()
    for i in range(N): out = gen.inference_batch(mels, noises)
    torch.cuda.synchronize(); res[mode] = (time.time() - t) / N * 1e3
    h = hashlib.sha256(torch.cat([o.reshape(-1) for o in out]).cpu().numpy().tobytes()).hexdigest()[:12]
    over, fb = gen.scale_overshoot()
    print(f"{sys.argv[1] if len(sys.argv) > 1 else 'run':16s} scale_guard {mode}: {res[mode]:7.2f} ms/batch  wav {h}  overshoot max {over.max():.4f} last {over[-1]:.4f} fell_back {fb}", flush=True)
print(f"{sys.argv[1] if len(sys.argv) > 1 else 'run':16s} a guarded call costs {res[2] - res[1]:+.2f} ms, a sampled call {res[3] - res[1]:+.2f} ms")
