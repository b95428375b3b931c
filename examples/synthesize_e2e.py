#!/usr/bin/env python
"""FastSpeech2 + Parallel WaveGAN synthesis from released checkpoints on the MI355X engine -- the counterpart
of the reference's examples/fastspeech2/ljspeech/synthesize_e2e.py with the same arguments, minus Paddle; with
``--speaker-dict`` of examples/fastspeech2/vctk/synthesize_e2e.py (multi-speaker, speaker ``--spk-id``); with
``--test-metadata`` of examples/fastspeech2/synthesize.py (preprocessed phone ids, a speaker per utterance).

``--text`` holds one ``utt_id sentence`` per line as in the reference (:45-50).  Sentences go through
``parakeet_amd.frontend.English`` -- the reference's ``parakeet.frontend.English`` with its g2p_en backend replaced
by a lexicon-driven one (``--lexicon``: a CMUdict-format file; words it lacks fall back to letter-to-sound rules
and are listed on stderr) -- and the recipe's id mapping (start / end symbols dropped, unknown phones and
punctuation -> "sp", :88-97).  With ``--phones-input`` the lines hold phone sequences instead
(``utt_id PH1 PH2 ...``, what ``frontend.phoneticize`` returns).
All utterances are synthesised as ONE ragged batch (the reference loops one by one, :88-107).
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from parakeet_amd import checkpoint  # noqa: E402
from parakeet_amd.audio import write_wav  # noqa: E402
from parakeet_amd.frontend import English, phones_to_ids  # noqa: E402
from parakeet_amd.frontend.phone_map import RECIPE_PUNC as PUNC  # noqa: E402  (synthesize_e2e.py:67)
from parakeet_amd.synthesize import Synthesizer  # noqa: E402


def main():
    ap = argparse.ArgumentParser(description="Synthesize with fastspeech2 & parallel wavegan.")
    ap.add_argument("--fastspeech2-config", required=True)
    ap.add_argument("--fastspeech2-checkpoint", required=True)
    ap.add_argument("--fastspeech2-stat", required=True)
    ap.add_argument("--pwg-config", required=True)
    ap.add_argument("--pwg-checkpoint", required=True)
    ap.add_argument("--pwg-stat", required=True)
    ap.add_argument("--phones-dict", default="phone_id_map.txt")
    ap.add_argument("--text", help="'utt_id sentence' per line")
    ap.add_argument("--test-metadata", help="instead of --text: the jsonlines file of examples/fastspeech2/synthesize.py "
                                            "(records with utt_id, text = phone ids and, with --speaker-dict, spk_id)")
    ap.add_argument("--lexicon", default=None, help="CMUdict-format pronunciation lexicon for the English frontend "
                                                    "(default: the small demonstration lexicon of the package)")
    ap.add_argument("--phones-input", action="store_true", help="--text holds 'utt_id PH1 PH2 ...' lines")
    ap.add_argument("--speaker-dict", default=None, help="speaker id map of the multi-speaker recipe "
                                                         "(examples/fastspeech2/vctk/synthesize_e2e.py); files become {spk_id}_{utt_id}.wav")
    ap.add_argument("--spk-id", type=int, default=0, help="the vctk recipe synthesises speaker 0 only (:93-94)")
    ap.add_argument("--output-dir", required=True)
    ap.add_argument("--seed", type=int, default=0, help="seed of the engine's noise stream")
    args = ap.parse_args()

    am, phone_id_map = checkpoint.load_fastspeech2(args.fastspeech2_config, args.fastspeech2_checkpoint,
                                                   args.fastspeech2_stat, args.phones_dict, speaker_dict=args.speaker_dict)
    voc = checkpoint.load_pwg(args.pwg_config, args.pwg_checkpoint, args.pwg_stat)
    voc.pwg_generator.set_seed(args.seed)
    fs = checkpoint._config(args.fastspeech2_config)["fs"]

    if (args.text is None) == (args.test_metadata is None):
        ap.error("give --text or --test-metadata")
    frontend = None if (args.phones_input or args.test_metadata) else English(lexicon=args.lexicon)
    utt_ids, batch, spk = [], [], []
    if args.test_metadata:            # examples/fastspeech2/synthesize.py:37-53, :98-106: ids (and speakers) as preprocessed
        import json
        with open(args.test_metadata, "rt") as f:
            for line in f:
                if line.strip():
                    rec = json.loads(line)
                    utt_ids.append(rec["utt_id"])
                    batch.append([int(i) for i in rec["text"]])
                    spk.append(int(rec.get("spk_id", args.spk_id)))
    with open(args.text if args.text else os.devnull, "rt") as f:
        for line in f:
            parts = line.strip().split()
            if not parts:
                continue
            if frontend is None:
                ids = phones_to_ids(parts[1:], phone_id_map, PUNC, strip_start_end=False)
            else:
                ids = phones_to_ids(frontend.phoneticize(" ".join(parts[1:])), phone_id_map, PUNC)   # :89-97
                if frontend.backend.oov:
                    print(f"{parts[0]}: not in the lexicon, pronounced by rule: {frontend.backend.oov}", file=sys.stderr)
            utt_ids.append(parts[0])
            batch.append([int(i) for i in ids])
    os.makedirs(args.output_dir, exist_ok=True)
    t0 = time.perf_counter()
    multi = args.speaker_dict is not None
    if not spk:
        spk = [args.spk_id] * len(batch)
    wavs = Synthesizer(am, voc).synthesize_batch(batch, spk_ids=spk if multi else None)
    if multi and not args.test_metadata:          # the e2e recipe's file names (vctk/synthesize_e2e.py:106-110)
        utt_ids = [f"{s}_{u}" for s, u in zip(spk, utt_ids)]
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
