"""Round 6 experiment: how the acoustic model of the following batches is issued around the vocoder of the current one.
  v0  the product's pipeline: vocode(k), then issue_acoustic(k+1) -- one FastSpeech2 handle, one side stream
  v1  two FastSpeech2 handles on two side streams, issue_acoustic(k+1) BEFORE vocode(k)
  v2  two handles / streams, vocode(k) before issue_acoustic(k+1)
Every variant: 20 steps timed, the last waveform compared with the plain in-order step."""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from parakeet_amd import synthetic as syn
from parakeet_amd.fastspeech2 import FastSpeech2, FastSpeech2Inference
from parakeet_amd.normalizer import ZScore
from parakeet_amd.synthesize import Synthesizer

synth, fs2_state, pwg_state, (mu_f, sg_f, mu_p, sg_p) = bench.build_models(0)
am2 = FastSpeech2(80, 80, **syn.FS2_LJSPEECH, device=0); am2.set_state_dict(fs2_state); am2.eval()
synth2 = Synthesizer(FastSpeech2Inference(ZScore(mu_f, sg_f), am2), synth.voc_inference)   # second acoustic handle, the same vocoder
lanes = [synth, synth2]
B = 32
texts = [syn.phoneme_ids(bench.TOKENS, seed=10086 + i) for i in range(B)]
g = torch.Generator(device="cuda"); g.manual_seed(42)
noise = torch.randn(B * bench.TOKENS * bench.FRAMES_PER_TOKEN * bench.HOP, device="cuda", generator=g)
ref, _ = synth.synthesize_packed(texts, noise=noise); synth2.synthesize_packed(texts, noise=noise); torch.cuda.synchronize()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def run(name, stepper):
    st = {"pending": None, "k": 0}
    for _ in range(4): out = stepper(st)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(N): out = stepper(st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / N
    if st["pending"] is not None: st["pending"][2].synchronize()
    print(f"{name:4s} {dt*1e3:7.3f} ms/step  {B*163840/dt/1e6:7.2f} M samples/s  identical to in-order: {bool(torch.equal(out[0], ref))}", flush=True)


def v_inorder(st):
    return synth.synthesize_packed(texts, noise=noise)


def v0(st):
    if st["pending"] is None: st["pending"] = synth.issue_acoustic(texts)
    out = synth.vocode_issued(st["pending"], noise=noise)
    st["pending"] = synth.issue_acoustic(texts)
    return out


def v1(st):
    k = st["k"]; st["k"] += 1
    if st["pending"] is None: st["pending"] = lanes[k & 1].issue_acoustic(texts)
    nxt = lanes[(k + 1) & 1].issue_acoustic(texts)
    out = lanes[k & 1].vocode_issued(st["pending"], noise=noise)
    st["pending"] = nxt
    return out


def v2(st):
    k = st["k"]; st["k"] += 1
    if st["pending"] is None: st["pending"] = lanes[k & 1].issue_acoustic(texts)
    out = lanes[k & 1].vocode_issued(st["pending"], noise=noise)
    st["pending"] = lanes[(k + 1) & 1].issue_acoustic(texts)
    return out


for rep in range(3):
    run("ord", v_inorder); run("v0", v0); run("v1", v1); run("v2", v2)
