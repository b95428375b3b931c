"""One PWG (or e2e) pass for rocprofv3 --pmc collection.  usage: pmc_run.py [pwg|e2e] [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from parakeet_amd import synthetic as syn
mode = sys.argv[1] if len(sys.argv) > 1 else "pwg"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
L = 640
if mode == "pwg":
    from parakeet_amd.parallel_wavegan import PWGGenerator
    gen = PWGGenerator(**syn.PWG_LJSPEECH); gen.set_state_dict(syn.pwg_state()); gen.eval()
    gen.set_option("scale_guard", 0)   # the ONE call of this script would be the guarded first call: count the instantiation every other call runs
    rng = np.random.default_rng(42)
    mels = [torch.tensor(rng.normal(size=(L, 80)).astype(np.float32)).cuda() for _ in range(B)]
    noises = [torch.randn(L * 256, device="cuda") for _ in range(B)]
    gen.inference_batch(mels, noises)
    torch.cuda.synchronize()
elif mode in ("wf", "wf128", "wf16", "wf128_16"):                  # 64 / 128 channels, default math / fp16 operands
    from parakeet_amd.waveflow import ConditionalWaveFlow
    cfg = dict(syn.WAVEFLOW_LJSPEECH, channels=128 if "128" in mode else 64, n_flows=2)   # two flows = 240 layer launches are plenty for counters
    m = ConditionalWaveFlow(**cfg); m.set_state_dict(syn.waveflow_state(cfg)); m.eval()
    if mode.endswith("16"): m.set_math("f16")
    if os.environ.get("PK_QWF_PERSISTENT"): m.set_option("persistent", int(os.environ["PK_QWF_PERSISTENT"]))   # counters per LAYER launch
    rng = np.random.default_rng(0)
    Bw = min(B, 8)
    mels = [torch.tensor(np.maximum(rng.normal(-4, 2, size=(80, L)), np.log(1e-5)).astype(np.float32)).cuda() for _ in range(Bw)]
    zs = [torch.randn(m.lengths(L)[0], device="cuda") for _ in range(Bw)]
    m.infer_batch(mels, zs)
    torch.cuda.synchronize()
elif mode == "fs2":
    from parakeet_amd.fastspeech2 import FastSpeech2
    m = FastSpeech2(80, 80, **syn.FS2_LJSPEECH); m.set_state_dict(syn.fastspeech2_state(fixed_duration=5)); m.eval()
    texts = [syn.phoneme_ids(128, seed=i) for i in range(B)]
    m.inference_batch(texts)
    torch.cuda.synchronize()
else:
    sys.argv = [sys.argv[0]]
    import bench
    synth, *_ = bench.build_models(0)
    texts = [syn.phoneme_ids(128, seed=i) for i in range(B)]
    noise = torch.randn(B * L * 256, device="cuda")
    synth.synthesize_packed(texts, noise=noise)
    torch.cuda.synchronize()
