"""From frontend phones to the ids a trained FastSpeech2 expects
(examples/fastspeech2/ljspeech/synthesize_e2e.py:53-56,66-67,88-97).

The acoustic model's vocabulary is the ``phone_id_map.txt`` written at preprocessing time (one ``phone id`` pair per
line, parakeet/datasets/preprocess_utils.py:92-103); the recipe drops the start / end symbols, drops whitespace
tokens, and maps everything the map does not know -- and every punctuation mark -- to the pause phone ``sp``.
"""
import numpy as np

__all__ = ["RECIPE_PUNC", "read_phone_id_map", "phones_to_ids", "text_to_ids", "phones_to_ids_transformer_tts", "CachedTextToIds"]

RECIPE_PUNC = "：，；。？！“”‘’':,;.?!"     # synthesize_e2e.py:67


def read_phone_id_map(path):
    """``phone id`` per line -> dict, ids as int (synthesize_e2e.py:53-56)."""
    table = {}
    with open(path, encoding="utf-8") as f:
        for line in f:
            fields = line.split()
            if not fields:
                continue
            if len(fields) != 2:
                raise ValueError(f"{path}: expected 'phone id', got {line!r}")
            table[fields[0]] = int(fields[1])
    return table


def phones_to_ids(phones, phone_id_map, punc=RECIPE_PUNC, strip_start_end=True):
    """The loop body of synthesize_e2e.py:88-97 on an already phoneticized sentence."""
    if strip_start_end:
        phones = phones[1:-1]                   # remove start_symbol and end_symbol (:90-91)
    phones = [p for p in phones if not p.isspace()]
    if "sp" not in phone_id_map and any((p not in phone_id_map or p in punc) for p in phones):
        raise KeyError("phone_id_map has no 'sp' entry to map unknown phones and punctuation to")
    phones = [p if (p in phone_id_map and p not in punc) else "sp" for p in phones]
    return np.asarray([phone_id_map[p] for p in phones], dtype=np.int64)


def text_to_ids(frontend, sentence, phone_id_map, punc=RECIPE_PUNC):
    """``frontend.phoneticize`` + the recipe's mapping: raw text -> int64 ids for ``FastSpeech2.inference``."""
    return phones_to_ids(frontend.phoneticize(sentence), phone_id_map, punc)


def phones_to_ids_transformer_tts(phones, phone_id_map, strip_start_end=True):
    """The loop body of examples/transformer_tts/ljspeech/synthesize_e2e.py:84-90: start / end symbols and whitespace
    tokens dropped, punctuation KEPT (that recipe's vocabulary has it), anything the map lacks -> ","."""
    if strip_start_end:
        phones = phones[1:-1]
    phones = [p for p in phones if not p.isspace()]
    if "," not in phone_id_map and any(p not in phone_id_map for p in phones):
        raise KeyError("phone_id_map has no ',' entry to map unknown phones to")
    return np.asarray([phone_id_map[p if p in phone_id_map else ","] for p in phones], dtype=np.int64)


class CachedTextToIds:
    """``text_to_ids`` behind a bounded least-recently-used memo keyed by the sentence: a serving process sees the same
    prompts, greetings and sentence fragments again and again, and the frontend (normalisation, tokenisation, lexicon
    look-ups -- host Python) is the only stage of a request that does not run on the GPU (bench.py ``extras.text_to_wav``
    gives its share).  ``many(sentences)`` maps a batch, computing each distinct new sentence once.  The arrays are shared
    between callers: treat them as read-only."""

    def __init__(self, frontend, phone_id_map, punc=RECIPE_PUNC, capacity=4096):
        from collections import OrderedDict
        self.frontend, self.phone_id_map, self.punc, self.capacity = frontend, phone_id_map, punc, int(capacity)
        self._memo = OrderedDict()
        self.hits = self.misses = 0

    def __call__(self, sentence):
        memo = self._memo
        ids = memo.get(sentence)
        if ids is not None:
            memo.move_to_end(sentence)
            self.hits += 1
            return ids
        self.misses += 1
        ids = text_to_ids(self.frontend, sentence, self.phone_id_map, self.punc)
        memo[sentence] = ids
        if len(memo) > self.capacity:
            memo.popitem(last=False)
        return ids

    def many(self, sentences):
        return [self(s) for s in sentences]
