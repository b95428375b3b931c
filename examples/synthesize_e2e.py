#!/usr/bin/env python
"""FastSpeech2 + Parallel WaveGAN synthesis from released checkpoints on the MI355X engine -- the counterpart
of the reference's examples/fastspeech2/ljspeech/synthesize_e2e.py with the same arguments, minus Paddle.

The reference turns sentences into phones with ``parakeet.frontend.English`` (g2p_en + nltk), which this
image does not have; ``--text`` therefore holds *phone* sequences, one ``utt_id PH1 PH2 ...`` per line
(what ``frontend.phoneticize`` returns, :90-99; unknown phones and punctuation map to "sp" exactly as there).
All utterances are synthesised as ONE ragged batch (the reference loops one by one, :88-107).
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from parakeet_amd import checkpoint  # noqa: E402
from parakeet_amd.audio import write_wav  # noqa: E402
from parakeet_amd.synthesize import Synthesizer  # noqa: E402

PUNC = "：，；。？！“”‘’':,;.?!"   # synthesize_e2e.py:66


def main():
    ap = argparse.ArgumentParser(description="Synthesize with fastspeech2 & parallel wavegan.")
    ap.add_argument("--fastspeech2-config", required=True)
    ap.add_argument("--fastspeech2-checkpoint", required=True)
    ap.add_argument("--fastspeech2-stat", required=True)
    ap.add_argument("--pwg-config", required=True)
    ap.add_argument("--pwg-checkpoint", required=True)
    ap.add_argument("--pwg-stat", required=True)
    ap.add_argument("--phones-dict", default="phone_id_map.txt")
    ap.add_argument("--text", required=True, help="'utt_id PH1 PH2 ...' per line")
    ap.add_argument("--output-dir", required=True)
    ap.add_argument("--seed", type=int, default=0, help="seed of the engine's noise stream")
    args = ap.parse_args()

    am, phone_id_map = checkpoint.load_fastspeech2(args.fastspeech2_config, args.fastspeech2_checkpoint,
                                                   args.fastspeech2_stat, args.phones_dict)
    voc = checkpoint.load_pwg(args.pwg_config, args.pwg_checkpoint, args.pwg_stat)
    voc.pwg_generator.set_seed(args.seed)
    fs = checkpoint._config(args.fastspeech2_config)["fs"]

    utt_ids, batch = [], []
    with open(args.text, "rt") as f:
        for line in f:
            parts = line.strip().split()
            if not parts:
                continue
            phones = [p if (p in phone_id_map and p not in PUNC) else "sp" for p in parts[1:]]   # :94-98
            utt_ids.append(parts[0])
            batch.append([phone_id_map[p] for p in phones])
    os.makedirs(args.output_dir, exist_ok=True)
    t0 = time.perf_counter()
    wavs = Synthesizer(am, voc).synthesize_batch(batch)
    n = 0
    for utt_id, wav in zip(utt_ids, wavs):
        w = wav.numpy()
        n += w.shape[0]
        write_wav(os.path.join(args.output_dir, utt_id + ".wav"), w, fs)
        print(f"{utt_id} done!")
    dt = time.perf_counter() - t0
    print(f"{len(wavs)} utterances, {n / fs:.2f} s of audio in {dt:.3f} s ({n / fs / dt:.1f}x real time, incl. first-call setup)")


if __name__ == "__main__":
    main()
