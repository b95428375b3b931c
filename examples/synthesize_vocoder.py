#!/usr/bin/env python
"""Vocoder-only synthesis from mel spectrogram files on the MI355X engine -- the counterparts of the reference's
examples/GANVocoder/parallelwave_gan/synthesize.py (``--config --checkpoint --test-metadata --output-dir``: a jsonlines
file of ``{"utt_id", "feats": path.npy}`` records, features already normalised with the vocoder's statistics, (T, n_mels))
and examples/waveflow/synthesize.py (``--config --checkpoint_path --input --output``: a directory of ``*.npy`` mel files,
(n_mels, T) log-magnitudes).

All utterances of the list are vocoded as ONE ragged batch (the reference loops one by one); the time and the
real-time factor are printed for the batch, as the reference prints them per utterance.
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from parakeet_amd import checkpoint  # noqa: E402
from parakeet_amd.audio import write_wav  # noqa: E402


def parallel_wavegan(args):
    import torch
    from parakeet_amd.parallel_wavegan import PWGGenerator
    cfg = checkpoint._config(args.config)
    gen = PWGGenerator(**cfg["generator_params"])
    gen.set_state_dict(checkpoint.load_params(args.checkpoint, "generator_params"))
    gen.remove_weight_norm()
    gen.eval()
    with open(args.test_metadata, "rt") as f:
        meta = [json.loads(line) for line in f if line.strip()]
    base = os.path.dirname(os.path.abspath(args.test_metadata))
    mels = [np.load(m["feats"] if os.path.isabs(m["feats"]) else os.path.join(base, m["feats"])).astype(np.float32) for m in meta]
    frames = np.array([m.shape[0] for m in mels], dtype=np.int32)
    packed = torch.from_numpy(np.concatenate(mels, axis=0))
    t0 = time.time()
    wav = gen.infer_packed(gen._ctx.to_device(packed), frames)          # features are already normalised (:86-88)
    gen._ctx.sync()
    dt = time.time() - t0
    wav = wav.cpu().numpy()
    os.makedirs(args.output_dir, exist_ok=True)
    o = 0
    for m, L in zip(meta, frames):
        n = int(L) * gen.upsample_factor
        write_wav(os.path.join(args.output_dir, m["utt_id"] + ".wav"), wav[o:o + n], cfg["fs"])
        o += n
    print(f"{len(meta)} utterances, {wav.size} samples, time: {dt:.3f}s, Hz: {wav.size / dt:.0f}, RTF: {cfg['fs'] / (wav.size / dt):.5f}.")


def waveflow(args):
    model = checkpoint.load_waveflow(args.config, args.checkpoint_path)
    cfg = checkpoint._config(args.config)
    files = sorted(glob.glob(os.path.join(os.path.expanduser(args.input), "*.npy")))
    mels = [np.load(f).astype(np.float32) for f in files]               # (n_mels, T) each (:36-38)
    t0 = time.time()
    wavs = model.infer_batch(mels)
    model._ctx.sync()
    dt = time.time() - t0
    out = os.path.expanduser(args.output)
    os.makedirs(out, exist_ok=True)
    total = 0
    for f, w in zip(files, wavs):
        path = os.path.join(out, os.path.splitext(os.path.basename(f))[0] + ".wav")
        w = w.cpu().numpy()
        total += w.size
        write_wav(path, w, cfg["data"]["sample_rate"])
        print("[synthesize] {} -> {}".format(f, path))
    print(f"{len(files)} utterances, {total} samples, time: {dt:.3f}s")


def main():
    ap = argparse.ArgumentParser(description="Synthesize with parallel wavegan / waveflow from mel spectrogram files.")
    sub = ap.add_subparsers(dest="vocoder", required=True)
    p = sub.add_parser("parallel_wavegan", help="examples/GANVocoder/parallelwave_gan/synthesize.py")
    p.add_argument("--config", required=True)
    p.add_argument("--checkpoint", required=True)
    p.add_argument("--test-metadata", required=True)
    p.add_argument("--output-dir", required=True)
    w = sub.add_parser("waveflow", help="examples/waveflow/synthesize.py")
    w.add_argument("--config", required=True)
    w.add_argument("--checkpoint_path", required=True)
    w.add_argument("--input", required=True)
    w.add_argument("--output", required=True)
    args = ap.parse_args()
    (parallel_wavegan if args.vocoder == "parallel_wavegan" else waveflow)(args)


if __name__ == "__main__":
    main()
