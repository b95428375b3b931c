#!/usr/bin/env python
"""TransformerTTS (or Tacotron2) + WaveFlow synthesis from released checkpoints on the MI355X engine -- the counterpart
of the reference's examples/transformer_tts/synthesize.py (same arguments, minus Paddle) and of
examples/tacotron2/synthesize.py + WaveFlow.

``--text`` holds one ``utt_id PH1 PH2 ...`` line per utterance (phones of ``--phones-dict``, as the recipe's
``test_metadata`` carries them); with ``--raw-text`` the lines are ``utt_id sentence`` and go through
``parakeet_amd.frontend.English`` (``--lexicon``) and the id mapping of
examples/transformer_tts/ljspeech/synthesize_e2e.py:84-90; with ``--tacotron2-config`` raw sentences go through
``parakeet_amd.frontend.EnglishCharacter`` like examples/tacotron2/synthesize.py:49.
All utterances are decoded in lockstep as ONE ragged batch and vocoded as one batch (the reference loops one by one).
The decoder prenets keep dropout on at inference (as in the reference); ``--seed`` selects the engine's dropout stream.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from parakeet_amd import checkpoint  # noqa: E402
from parakeet_amd.audio import write_wav  # noqa: E402
from parakeet_amd.synthesize import ARSynthesizer  # noqa: E402


def main():
    ap = argparse.ArgumentParser(description="Synthesize with transformer tts / tacotron2 & waveflow.")
    ap.add_argument("--transformer-tts-config")
    ap.add_argument("--transformer-tts-checkpoint")
    ap.add_argument("--transformer-tts-stat")
    ap.add_argument("--tacotron2-config", help="yaml with the model / data sections of examples/tacotron2/config.py")
    ap.add_argument("--tacotron2-checkpoint", help="path without the .pdparams suffix")
    ap.add_argument("--waveflow-config", required=True)
    ap.add_argument("--waveflow-checkpoint", required=True)
    ap.add_argument("--phones-dict", default="phone_id_map.txt")
    ap.add_argument("--text", required=True)
    ap.add_argument("--output-dir", required=True)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--max-decoder-steps", type=int, default=1000)
    ap.add_argument("--raw-text", action="store_true", help="--text holds 'utt_id sentence' lines (TransformerTTS)")
    ap.add_argument("--lexicon", default=None, help="CMUdict-format lexicon for the English frontend")
    args = ap.parse_args()

    voc = checkpoint.load_waveflow(args.waveflow_config, args.waveflow_checkpoint)
    utt_ids, batch, kw = [], [], {}
    if args.tacotron2_config:
        from parakeet_amd.frontend import EnglishCharacter
        am = checkpoint.load_tacotron2(args.tacotron2_config, args.tacotron2_checkpoint)
        frontend = EnglishCharacter()
        kw["max_decoder_steps"] = args.max_decoder_steps
        fs = checkpoint._config(args.tacotron2_config)["data"]["sample_rate"]
        with open(args.text, "rt") as f:
            for i, line in enumerate(f):
                if line.strip():
                    utt_ids.append(f"sentence_{i}")
                    batch.append(frontend(line.strip()))
    else:
        am, phone_id_map = checkpoint.load_transformer_tts(args.transformer_tts_config, args.transformer_tts_checkpoint,
                                                           args.transformer_tts_stat, args.phones_dict)
        fs = checkpoint._config(args.transformer_tts_config)["fs"]
        frontend = None
        if args.raw_text:
            from parakeet_amd.frontend import English, phones_to_ids_transformer_tts
            frontend = English(lexicon=args.lexicon)
        with open(args.text, "rt") as f:
            for line in f:
                parts = line.strip().split()
                if parts:
                    utt_ids.append(parts[0])
                    if frontend is None:
                        batch.append([phone_id_map[p] for p in parts[1:]])
                    else:
                        batch.append(phones_to_ids_transformer_tts(frontend.phoneticize(" ".join(parts[1:])), phone_id_map))
    wavs = ARSynthesizer(am, voc).synthesize_batch(batch, seeds=[args.seed + i for i in range(len(batch))], **kw)
    os.makedirs(args.output_dir, exist_ok=True)
    for utt_id, wav in zip(utt_ids, wavs):
        write_wav(os.path.join(args.output_dir, utt_id + ".wav"), wav.numpy(), fs)
        print(f"{utt_id} done!")


if __name__ == "__main__":
    main()
