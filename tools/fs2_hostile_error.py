"""FastSpeech2 mel error against the fp64 oracle under hostile weights for the attention-context bound (values that cancel under
uniform attention, tests/test_fs2_gpu.py::_cancelling_values_state), per math mode.  usage: python tools/fs2_hostile_error.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from oracle import fastspeech2_ref as ref
from parakeet_amd.fastspeech2 import FastSpeech2
from test_fs2_gpu import _cfg, _oracle_cfg, _cancelling_values_state, _context_overshoot

cfg = _cfg()
ids = np.array([1, 2] * 32, dtype=np.int64)
for gain in (1.0, 16.0, 1024.0, 65536.0):
    state = _cancelling_values_state(cfg, 170, gain)
    want = ref.inference(state, ids, _oracle_cfg(cfg), dtype=torch.float64).numpy()
    cpu32 = ref.inference(state, ids, _oracle_cfg(cfg), dtype=torch.float32).numpy()
    m = FastSpeech2(80, 80, **cfg); m.set_state_dict(state); m.eval()
    out = {}
    for mode in ("f32", "f16x3"):
        m.set_math(mode)
        got = m.inference(ids).numpy()
        out[mode] = (float(np.abs(got - want).mean()), float(np.abs(got - want).max()))
    print(f"gain {gain:8.0f} (context bound loose by {_context_overshoot(state, cfg, ids):9.1f}): mel L1 / max vs fp64 oracle: exact fp32 MFMA {out['f32'][0]:.2e} / {out['f32'][1]:.2e}, split fp16 {out['f16x3'][0]:.2e} / {out['f16x3'][1]:.2e}, "
          f"torch CPU fp32 {np.abs(cpu32 - want).mean():.2e} / {np.abs(cpu32 - want).max():.2e}")
