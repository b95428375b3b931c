#!/usr/bin/env python
"""FastSpeech2 + Parallel WaveGAN synthesis of Mandarin text from released checkpoints on the MI355X engine -- the
counterpart of the reference's examples/fastspeech2/baker/synthesize_e2e.py and, with ``--speaker-dict``, of
examples/fastspeech2/aishell3/synthesize_e2e.py (same arguments, minus Paddle).

``--text`` holds one ``utt_id sentence`` per line; sentences go through ``parakeet_amd.frontend.Frontend`` over
``--lexicon`` (a pinyin lexicon file; the package's demonstration lexicon by default).  The multi-speaker recipe
synthesises speaker 0 only (aishell3/synthesize_e2e.py:89-90) and names its files ``{spk_id}_{utt_id}.wav``;
``--spk-id`` picks another one.  All utterances are synthesised as ONE ragged batch."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from parakeet_amd import checkpoint  # noqa: E402
from parakeet_amd.audio import write_wav  # noqa: E402
from parakeet_amd.frontend import Frontend  # noqa: E402
from parakeet_amd.synthesize import Synthesizer  # noqa: E402


def run(args):
    am, _ = checkpoint.load_fastspeech2(args.fastspeech2_config, args.fastspeech2_checkpoint, args.fastspeech2_stat,
                                        args.phones_dict, speaker_dict=args.speaker_dict)
    voc = checkpoint.load_pwg(args.pwg_config, args.pwg_checkpoint, args.pwg_stat)
    voc.pwg_generator.set_seed(args.seed)
    fs = checkpoint._config(args.fastspeech2_config)["fs"]
    frontend = Frontend(phone_vocab_path=args.phones_dict, lexicon=args.lexicon)
    utt_ids, phones = [], []
    with open(args.text, "rt", encoding="utf-8") as f:
        for line in f:
            parts = line.strip().split(maxsplit=1)
            if len(parts) == 2:
                utt_ids.append(parts[0])
                phones.append(frontend.get_input_ids(parts[1], merge_sentences=True)["phone_ids"][0])
    if frontend.missing:
        print("not in the lexicon (read as 'sp'):", " ".join(sorted(set(frontend.missing))), file=sys.stderr)
    multi = args.speaker_dict is not None
    wavs = Synthesizer(am, voc).synthesize_batch(phones, spk_ids=[args.spk_id] * len(phones) if multi else None)
    os.makedirs(args.output_dir, exist_ok=True)
    for utt_id, wav in zip(utt_ids, wavs):
        name = f"{args.spk_id}_{utt_id}" if multi else utt_id
        write_wav(os.path.join(args.output_dir, name + ".wav"), wav.numpy(), fs)
        print(f"{name} done!")


def main():
    ap = argparse.ArgumentParser(description="Synthesize with fastspeech2 & parallel wavegan.")
    ap.add_argument("--fastspeech2-config", required=True)
    ap.add_argument("--fastspeech2-checkpoint", required=True)
    ap.add_argument("--fastspeech2-stat", required=True)
    ap.add_argument("--pwg-config", required=True)
    ap.add_argument("--pwg-checkpoint", required=True)
    ap.add_argument("--pwg-stat", required=True)
    ap.add_argument("--phones-dict", default="phone_id_map.txt")
    ap.add_argument("--speaker-dict", default=None, help="speaker id map file of the multi-speaker recipes")
    ap.add_argument("--spk-id", type=int, default=0)
    ap.add_argument("--lexicon", default=None, help="pinyin lexicon for the Mandarin frontend")
    ap.add_argument("--text", required=True, help="'utt_id sentence' per line")
    ap.add_argument("--output-dir", required=True)
    ap.add_argument("--seed", type=int, default=0, help="seed of the engine's noise stream")
    run(ap.parse_args())


if __name__ == "__main__":
    main()
