#!/usr/bin/env python
"""SpeedySpeech + Parallel WaveGAN synthesis of Mandarin text from released checkpoints on the MI355X engine -- the
counterpart of the reference's examples/speedyspeech/baker/synthesize_e2e.py with the same arguments, minus Paddle and
minus its static-graph export.

``--text`` holds one ``utt_id sentence`` per line.  Sentences go through ``parakeet_amd.frontend.Frontend`` (the
reference's ``parakeet.frontend.zh_frontend.Frontend`` with its jieba / pypinyin dictionaries replaced by ``--lexicon``,
a pinyin lexicon file: ``word syl [syl ...] [#pos]``; the package's demonstration lexicon by default -- characters it
lacks are listed on stderr and read as "sp").  All utterances are synthesised as ONE ragged batch with the mel staying
in HBM (the reference loops sentence by sentence, :113-131)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from parakeet_amd import checkpoint  # noqa: E402
from parakeet_amd.audio import write_wav  # noqa: E402
from parakeet_amd.frontend import Frontend  # noqa: E402
from parakeet_amd.synthesize import Synthesizer  # noqa: E402


def main():
    ap = argparse.ArgumentParser(description="Synthesize with speedyspeech & parallel wavegan.")
    ap.add_argument("--speedyspeech-config", required=True)
    ap.add_argument("--speedyspeech-checkpoint", required=True)
    ap.add_argument("--speedyspeech-stat", required=True)
    ap.add_argument("--pwg-config", required=True)
    ap.add_argument("--pwg-checkpoint", required=True)
    ap.add_argument("--pwg-stat", required=True)
    ap.add_argument("--phones-dict", default="phone_id_map.txt")
    ap.add_argument("--tones-dict", default="tone_id_map.txt")
    ap.add_argument("--lexicon", default=None, help="pinyin lexicon for the Mandarin frontend")
    ap.add_argument("--text", required=True, help="'utt_id sentence' per line")
    ap.add_argument("--output-dir", required=True)
    ap.add_argument("--seed", type=int, default=0, help="seed of the engine's noise stream")
    args = ap.parse_args()

    am, _, _ = checkpoint.load_speedyspeech(args.speedyspeech_config, args.speedyspeech_checkpoint, args.speedyspeech_stat,
                                            args.phones_dict, args.tones_dict)
    voc = checkpoint.load_pwg(args.pwg_config, args.pwg_checkpoint, args.pwg_stat)
    voc.pwg_generator.set_seed(args.seed)
    fs = checkpoint._config(args.speedyspeech_config)["fs"]
    frontend = Frontend(phone_vocab_path=args.phones_dict, tone_vocab_path=args.tones_dict, lexicon=args.lexicon)

    utt_ids, phones, tones = [], [], []
    with open(args.text, "rt", encoding="utf-8") as f:
        for line in f:
            parts = line.strip().split(maxsplit=1)
            if len(parts) < 2:
                continue
            ids = frontend.get_input_ids(parts[1], merge_sentences=True, get_tone_ids=True)   # :114-117
            utt_ids.append(parts[0])
            phones.append(ids["phone_ids"][0])
            tones.append(ids["tone_ids"][0])
    if frontend.missing:
        print("not in the lexicon (read as 'sp'):", " ".join(sorted(set(frontend.missing))), file=sys.stderr)
    wavs = Synthesizer(am, voc).synthesize_batch(phones, tones=tones)
    os.makedirs(args.output_dir, exist_ok=True)
    for utt_id, wav in zip(utt_ids, wavs):
        write_wav(os.path.join(args.output_dir, utt_id + ".wav"), wav.numpy(), fs)
        print(f"{utt_id} done!")


if __name__ == "__main__":
    main()
