import sys, os; sys.path.insert(0, '/root/repo')
if os.environ.get('PK_EMU')=='1':
    sys.path.insert(0,'/root/repo/tools/hipemu'); import harness; harness.install()
import numpy as np, torch
torch.set_num_threads(8)
from parakeet_amd import synthetic as syn
from parakeet_amd.fastspeech2 import FastSpeech2
from oracle import fastspeech2_ref as ref
cfg = dict(syn.FS2_LJSPEECH)
keys = ("adim aheads elayers eunits dlayers dunits positionwise_conv_kernel_size "
        "duration_predictor_layers duration_predictor_chans duration_predictor_kernel_size "
        "pitch_predictor_layers pitch_predictor_chans pitch_predictor_kernel_size "
        "energy_predictor_layers energy_predictor_chans energy_predictor_kernel_size "
        "pitch_embed_kernel_size energy_embed_kernel_size postnet_layers postnet_chans postnet_filts").split()
ocfg = {k: cfg[k] for k in keys}
state = syn.fastspeech2_state(seed=5)
lens = [int(x) for x in (sys.argv[1:] or [9, 33, 2])]
texts = [syn.phoneme_ids(T, seed=i) for i,T in enumerate(lens)]
want = [ref.inference(state, ids, ocfg, dtype=torch.float64).numpy() for ids in texts]
for env in ('1', '0'):
    os.environ['PK_FS2_FFN_PLANES'] = env   # profile build (PK_PROFILE_LIB=1)
    m = FastSpeech2(80, 80, **cfg); m.set_state_dict(state); m.eval()
    o = [x.cpu().numpy() for x in m.inference_batch(texts)]
    print('planes' if env=='1' else 'gemm  ', ' '.join(f"L1 {np.abs(a-b).mean():.2e} max {np.abs(a-b).max():.2e}" for a,b in zip(o,want)))
