"""WaveFlow golden vectors from the reference's own source (see tools/make_golden.py)."""
import os

import numpy as np

import ref_import

ref_import.setup()
import paddle  # noqa: E402  (stand-in or real, see ref_import)

from parakeet_amd import synthetic as syn  # noqa: E402


def golden_waveflow(out_dir):
    wfm = ref_import.load("parakeet.models.waveflow")
    cfg = dict(syn.WAVEFLOW_LJSPEECH, channels=64)
    state = syn.waveflow_state(cfg, seed=314, weight_norm=True)
    model = wfm.ConditionalWaveFlow(**cfg)
    model.set_state_dict(state)
    model.eval()
    for layer in model.sublayers():   # utils/layer_tools.recursively_remove_weight_norm (layer_tools.py:40-46)
        try:
            paddle.nn.utils.remove_weight_norm(layer)
        except ValueError:
            pass
    rng = np.random.default_rng(11)
    mel = np.maximum(rng.normal(-4, 2, size=(2, 80, 4)), np.log(1e-5)).astype(np.float32)
    t = 4
    for f in cfg["upsample_factors"]:
        t = f * t - f
    z = rng.normal(size=(2, t)).astype(np.float32)
    with ref_import.fixed_randn(z), paddle.no_grad():
        wav = model.infer(paddle.to_tensor(mel)).numpy().astype(np.float32)
    np.savez_compressed(os.path.join(out_dir, "waveflow_c64.npz"), seed=np.array(314), mel=mel, z=z, wav=wav)
    print("waveflow:", wav.shape)
