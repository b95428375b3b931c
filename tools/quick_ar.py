"""Timing of the autoregressive acoustic models at LJSpeech shape (32 utterances x 128 tokens -> 640 frames each),
with the engine's per-kernel profile.  usage: quick_ar.py [tts|taco] [B] [frames] [math]"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np, torch
from parakeet_amd import synthetic as syn
from parakeet_amd.runtime import Context

which = sys.argv[1] if len(sys.argv) > 1 else "tts"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
L = int(sys.argv[3]) if len(sys.argv) > 3 else 640
math = sys.argv[4] if len(sys.argv) > 4 else "f16x3"
T = 128
rng = np.random.default_rng(0)
if which == "tts":
    from parakeet_amd.transformer_tts import TransformerTTS
    cfg = dict(syn.TRANSFORMER_TTS_LJSPEECH)
    m = TransformerTTS(idim=80, odim=80, **cfg); m.set_state_dict(syn.transformer_tts_state(80, 80, cfg, stop_bias=-8.0)); m.eval()
    m.set_math(math)
    for kv in os.environ.get("PK_QAR_OPTS", "").split(","):     # e.g. PK_QAR_OPTS=overlap_prefix=0,overlap_cu_mask=0
        if "=" in kv: m.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    texts = [rng.integers(1, 79, size=T) for _ in range(B)]
    ratio = (L + 0.5) / (T + 1)
    run = lambda: m.inference_batch(texts, maxlenratio=ratio, return_att=False)
    frames = lambda outs: sum(o[0].shape[0] for o in outs)
else:
    from parakeet_amd.tacotron2 import Tacotron2
    cfg = dict(syn.TACOTRON2_LJSPEECH)
    m = Tacotron2(**cfg); m.set_state_dict(syn.tacotron2_state(cfg, stop_bias=-8.0)); m.eval()
    m.set_math(math)
    texts = [rng.integers(1, 37, size=T) for _ in range(B)]
    run = lambda: m.infer_batch(texts, max_decoder_steps=L)
    frames = lambda outs: sum(o["mel_output"].shape[0] for o in outs)
outs = run()
torch.cuda.synchronize(); t = time.time(); n = 2
for _ in range(n): outs = run()
torch.cuda.synchronize(); dt = (time.time() - t) / n
f = frames(outs)
print(f"{which} B={B} T={T} math={math} opts={os.environ.get('PK_QAR_OPTS', '')}: {f} frames, {dt*1e3:.1f} ms/batch, {dt/L*1e6:.0f} us/step, {B/dt:.1f} utt/s, {f*256/22050/dt:.0f}x RT (mel only)")
ctx = Context.get(); ctx.prof_enable(True); ctx.prof_reset(); run()
tot = 0.0
for k, (n_, ms) in sorted(ctx.prof_dump().items(), key=lambda kv: -kv[1][1]):
    tot += ms
    print(f"  {k:28s} n={n_:6d} total={ms:9.3f} ms  avg={ms/n_*1e3:8.2f} us")
print(f"  kernel sum {tot:.1f} ms")
