"""Round 5 (HISTORY 9.9): the same call N times, bit for bit -- Parallel WaveGAN (32 x 640 frames), FastSpeech2 (32 x 128 tokens) and
the end-to-end step, the autoregressive models.  A hazard that needs two waves of a SIMD at the wrong cycle shows as a run that
differs from the first; none of these kernels has shown one (their determinism tests compare two runs; this compares many)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from parakeet_amd import synthetic as syn
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40


def report(name, outs):
    d = [int((o != outs[0]).sum().item()) for o in outs[1:]]
    print(f"{name:28s} {len(outs)} runs: {sum(1 for x in d if x)} differ from the first (elements differing: {sorted(x for x in d if x)[:8]})", flush=True)


from parakeet_amd.parallel_wavegan import PWGGenerator
gen = PWGGenerator(**syn.PWG_LJSPEECH); gen.set_state_dict(syn.pwg_state()); gen.eval()
g = torch.Generator(device="cuda").manual_seed(42)
mel = torch.randn(32 * 640, 80, device="cuda", generator=g); noise = torch.randn(32 * 640 * 256, device="cuda", generator=g)
report("PWG 32 x 640 frames", [gen.infer_packed(mel, [640] * 32, noise=noise).as_subclass(torch.Tensor).clone() for _ in range(N)])
for B in (3, 7):   # partial rounds of the persistent tile loop
    report(f"PWG {B} x 640 frames", [gen.infer_packed(mel[:B * 640], [640] * B, noise=noise[:B * 640 * 256]).as_subclass(torch.Tensor).clone() for _ in range(N)])
from parakeet_amd.fastspeech2 import FastSpeech2
am = FastSpeech2(80, 80, **syn.FS2_LJSPEECH); am.set_state_dict(syn.fastspeech2_state(fixed_duration=5)); am.eval()
for B in (32, 16, 1):
    texts = [syn.phoneme_ids(128, seed=i) for i in range(B)]
    outs = []
    for _ in range(N):
        outs.append(torch.cat([o.as_subclass(torch.Tensor).reshape(-1) for o in am.inference_batch(texts)]).clone())
    report(f"FastSpeech2 {B} x 128 tokens", outs)
texts = [syn.phoneme_ids(int(t), seed=100 + i) for i, t in enumerate(np.random.default_rng(1).integers(37, 129, size=16))]
outs = []
for _ in range(N):
    outs.append(torch.cat([o.as_subclass(torch.Tensor).reshape(-1) for o in am.inference_batch(texts)]).clone())
report("FastSpeech2 16 ragged", outs)
if "--ar" in sys.argv:
    rng = np.random.default_rng(0)
    from parakeet_amd.speedyspeech import SpeedySpeech
    ss = SpeedySpeech(vocab_size=70, tone_size=7, **syn.SPEEDYSPEECH_BAKER); ss.set_state_dict(syn.speedyspeech_state()); ss.eval()
    texts = [rng.integers(1, 70, size=128) for _ in range(32)]; tones = [rng.integers(1, 7, size=128) for _ in range(32)]
    report("SpeedySpeech 32 x 128", [torch.cat([o.as_subclass(torch.Tensor).reshape(-1) for o in ss.inference_batch(texts, tones)]).clone() for _ in range(N)])
    from parakeet_amd.tacotron2 import Tacotron2
    cfg = dict(syn.TACOTRON2_LJSPEECH)
    t2 = Tacotron2(**cfg); t2.set_state_dict(syn.tacotron2_state(cfg, stop_bias=-8.0)); t2.eval()
    texts = [rng.integers(1, 37, size=128) for _ in range(32)]
    def taco():
        return torch.cat([torch.as_tensor(np.asarray(o["mel_output"])).reshape(-1) for o in t2.infer_batch(texts, max_decoder_steps=320, seeds=list(range(32)))])
    report("Tacotron2 32 x 320 steps", [taco() for _ in range(max(4, N // 4))])
    from parakeet_amd.transformer_tts import TransformerTTS
    cfg = dict(syn.TRANSFORMER_TTS_LJSPEECH)
    tt = TransformerTTS(idim=80, odim=80, **cfg); tt.set_state_dict(syn.transformer_tts_state(80, 80, cfg, stop_bias=-8.0)); tt.eval()
    texts = [rng.integers(1, 79, size=128) for _ in range(32)]
    def tts():
        return torch.cat([torch.as_tensor(np.asarray(o[0])).reshape(-1) for o in tt.inference_batch(texts, maxlenratio=(160 + 0.5) / 129, return_att=False, seeds=list(range(32)))])
    report("TransformerTTS 32 x 160 steps", [tts() for _ in range(max(4, N // 6))])
