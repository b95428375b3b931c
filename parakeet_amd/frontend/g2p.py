"""Lexicon-driven grapheme-to-phoneme conversion with the call contract of ``g2p_en.G2p`` (the backend of
parakeet/frontend/phonectic.py:48-52 and arpabet.py:127-129).

``G2p.__call__(text)`` normalises the text (the same steps as normalizer.normalize), tokenises it, looks every
word up in CMUdict (first pronunciation), predicts unknown words, and returns ONE flat list: the ARPAbet phones of
each word, a " " token between consecutive tokens, punctuation marks as tokens of their own.  This class does the
same from a pronunciation lexicon in CMUdict text format (``WORD  PH1 PH2 ...``; alternative pronunciations
``WORD(2)`` are ignored, ``;;;`` lines are comments) supplied by the caller.  Differences from g2p_en, all forced by
what is available offline: heteronyms are not disambiguated by part of speech (first pronunciation wins);
contractions stay one token (CMUdict lists ``don't``); words missing from the lexicon go through the
letter-to-sound rules below instead of g2p_en's neural predictor.
"""
import os
import re

from .normalizer import normalize

__all__ = ["LexiconG2p", "ARPABET_PHONEMES", "GRAPHEMES"]

# g2p_en's symbol tables (the reference builds its vocabularies from them, phonectic.py:49-52,136-139)
_VOWELS = ["AA", "AE", "AH", "AO", "AW", "AY", "EH", "ER", "EY", "IH", "IY", "OW", "OY", "UH", "UW"]
_CONSONANTS = ["B", "CH", "D", "DH", "F", "G", "HH", "JH", "K", "L", "M", "N", "NG", "P", "R", "S", "SH", "T", "TH",
               "V", "W", "Y", "Z", "ZH"]
_STRESSED = sorted([v + s for v in _VOWELS for s in "012"] + _CONSONANTS + ["UW"])
ARPABET_PHONEMES = ["<pad>", "<unk>", "<s>", "</s>"] + _STRESSED
GRAPHEMES = ["<pad>", "<unk>", "</s>"] + list("abcdefghijklmnopqrstuvwxyz")

_DEMO_LEXICON = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "demo_lexicon.txt")

_TOKEN_RE = re.compile(r"[a-z]+(?:'[a-z]+)*'?|\.\.\.|[.,?!\-]")

# ---- letter-to-sound rules for out-of-lexicon words: longest match first, left to right.  (pattern, phones)
_LTS = [
    ("tion", ["SH", "AH", "N"]), ("sion", ["ZH", "AH", "N"]), ("ture", ["CH", "ER"]), ("ough", ["AO"]),
    ("augh", ["AO"]), ("eigh", ["EY"]), ("igh", ["AY"]), ("tch", ["CH"]), ("dge", ["JH"]), ("sch", ["S", "K"]),
    ("ch", ["CH"]), ("sh", ["SH"]), ("th", ["TH"]), ("ph", ["F"]), ("wh", ["W"]), ("ck", ["K"]), ("ng", ["NG"]),
    ("qu", ["K", "W"]), ("kn", ["N"]), ("wr", ["R"]), ("gh", ["G"]), ("ee", ["IY"]), ("ea", ["IY"]),
    ("oo", ["UW"]), ("ou", ["AW"]), ("ow", ["OW"]), ("oi", ["OY"]), ("oy", ["OY"]), ("ai", ["EY"]), ("ay", ["EY"]),
    ("oa", ["OW"]), ("au", ["AO"]), ("aw", ["AO"]), ("ei", ["EY"]), ("ey", ["IY"]), ("ie", ["IY"]), ("ue", ["UW"]),
    ("ew", ["UW"]), ("ar", ["AA", "R"]), ("er", ["ER"]), ("ir", ["ER"]), ("ur", ["ER"]), ("or", ["AO", "R"]),
    ("a", ["AE"]), ("b", ["B"]), ("c", ["K"]), ("d", ["D"]), ("e", ["EH"]), ("f", ["F"]), ("g", ["G"]),
    ("h", ["HH"]), ("i", ["IH"]), ("j", ["JH"]), ("k", ["K"]), ("l", ["L"]), ("m", ["M"]), ("n", ["N"]),
    ("o", ["AA"]), ("p", ["P"]), ("q", ["K"]), ("r", ["R"]), ("s", ["S"]), ("t", ["T"]), ("u", ["AH"]),
    ("v", ["V"]), ("w", ["W"]), ("x", ["K", "S"]), ("y", ["Y"]), ("z", ["Z"]),
]
_LONG = {"a": "EY", "e": "IY", "i": "AY", "o": "OW", "u": "UW", "y": "AY"}
_VOWEL_SET = set(_VOWELS)


def letter_to_sound(word):
    """Rule-based pronunciation of an out-of-lexicon word: digraph table, soft c / g before e i y, vowel +
    consonant + final e -> long vowel with the e silent, final y -> IY; primary stress on the first vowel, the
    others unstressed."""
    w = re.sub(r"[^a-z]", "", word.lower())
    phones = []
    i = 0
    n = len(w)
    while i < n:
        rest = w[i:]
        # silent final e after a consonant ("make"); the vowel before it was made long below
        if rest == "e" and phones and phones[-1] not in _VOWEL_SET and any(p in _VOWEL_SET for p in phones):
            break
        if rest[0] in "aeiouy" and len(rest) == 3 and rest[1] not in "aeiouyrw" and rest[2] == "e" and \
                not (rest[0] == "y" and not phones):
            phones.append(_LONG[rest[0]])      # magic e
            i += 1
            continue
        if rest[0] == "c" and len(rest) > 1 and rest[1] in "eiy":
            phones.append("S")
            i += 1
            continue
        if rest[0] == "g" and len(rest) > 1 and rest[1] in "eiy" and i > 0:
            phones.append("JH")
            i += 1
            continue
        if rest == "y" and phones:
            phones.append("IY")
            break
        if rest[0] == "y" and phones and (len(rest) == 1 or rest[1] not in "aeiou"):
            phones.append("IH")
            i += 1
            continue
        for pat, ph in _LTS:
            if rest.startswith(pat):
                if len(pat) == 1 and i + 1 < n and w[i + 1] == pat and pat not in "aeiou":
                    i += 1                     # doubled consonant: once
                phones.extend(ph)
                i += len(pat)
                break
        else:
            i += 1
    seen = False
    out = []
    for p in phones:
        if p in _VOWEL_SET:
            out.append(p + ("0" if seen else "1"))
            seen = True
        else:
            out.append(p)
    return out


class LexiconG2p(object):
    """``LexiconG2p(lexicon=path)(text) -> List[str]`` (see the module docstring)."""

    phonemes = ARPABET_PHONEMES
    graphemes = GRAPHEMES

    def __init__(self, lexicon=None, extra=None):
        self.lexicon_path = lexicon or _DEMO_LEXICON
        self.cmu = self.read_lexicon(self.lexicon_path)
        if extra:
            for w, pron in extra.items():
                self.cmu[w.lower()] = [list(pron)]
        self.oov = []           # words of the last call that went through the letter-to-sound rules

    @staticmethod
    def read_lexicon(path):
        valid = set(_STRESSED)
        table = {}
        with open(path, encoding="utf-8", errors="replace") as f:
            for ln, line in enumerate(f, 1):
                line = line.strip()
                if not line or line.startswith(";;;") or line.startswith("#"):
                    continue
                fields = line.split()
                word, phones = fields[0].lower(), fields[1:]
                if not phones:
                    raise ValueError(f"{path}:{ln}: no pronunciation for {fields[0]!r}")
                bad = [p for p in phones if p not in valid]
                if bad:
                    raise ValueError(f"{path}:{ln}: unknown ARPAbet symbol(s) {bad} for {fields[0]!r}")
                word = re.sub(r"\(\d+\)$", "", word)       # WORD(2): alternative pronunciation
                table.setdefault(word, []).append(phones)
        return table

    def predict(self, word):
        return letter_to_sound(word)

    def __call__(self, text):
        text = normalize(text)
        self.oov = []
        prons = []
        for word in _TOKEN_RE.findall(text):
            if re.search("[a-z]", word) is None:
                pron = [word]
            elif word in self.cmu:
                pron = self.cmu[word][0]
            elif word.strip("'") in self.cmu:
                pron = self.cmu[word.strip("'")][0]
            else:
                pron = self.predict(word)
                self.oov.append(word)
                if not pron:
                    continue
            prons.extend(pron)
            prons.append(" ")
        return prons[:-1]
