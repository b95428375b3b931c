#!/usr/bin/env python
"""tests/golden/en_frontend.json: the REFERENCE's own English frontend -- parakeet/frontend/normalizer/{numbers,normalizer}.py,
phonectic.py (English, EnglishCharacter), arpabet.py (ARPABET, ARPABETWithStress), vocab.py, punctuation.py -- executed on the
sentences of tests/en_cases.py with stand-ins for the two third-party packages it imports (neither is installable here):

* ``inflect``: ``engine().number_to_words`` for the four argument combinations numbers.py:56-74 uses, written here from
  inflect's documented conventions, independently of parakeet_amd/frontend/normalizer.py (the thing under test);
* ``g2p_en``: ``G2p`` with g2p_en's symbol tables and call contract (normalise, tokenise, CMUdict lookup, one flat phone list
  with " " between tokens), answered from parakeet_amd's demonstration lexicon; out-of-lexicon words go to the engine-side
  letter-to-sound rules (g2p_en's neural predictor cannot be reproduced -- a dictionary resource, not logic under test).

What this pins is the reference's own logic: regular expressions and branch structure of the number normaliser, accent /
case / character filtering, vocabulary assembly and id order, start / end symbols, the in-vocabulary filter, stress removal,
``numericalize`` / ``reverse`` and the recipe's id mapping (synthesize_e2e.py:88-98).  Build container only."""
import importlib
import json
import os
import re
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_import  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tests"))
from en_cases import RECIPE_SENTENCES, SENTENCES  # noqa: E402


# ---- inflect stand-in -----------------------------------------------------------------------------------------------------
_ONES = "zero one two three four five six seven eight nine ten eleven twelve thirteen fourteen fifteen sixteen seventeen eighteen nineteen".split()
_TY = {2: "twenty", 3: "thirty", 4: "forty", 5: "fifty", 6: "sixty", 7: "seventy", 8: "eighty", 9: "ninety"}
_MILL = ["", " thousand", " million", " billion", " trillion", " quadrillion", " quintillion"]
_NTH = {"one": "first", "two": "second", "three": "third", "five": "fifth", "eight": "eighth", "nine": "ninth", "twelve": "twelfth"}


def _two(digits):
    v = int(digits)
    return _ONES[v] if v < 20 else _TY[v // 10] + ("-" + _ONES[v % 10] if v % 10 else "")


class _Engine:
    """inflect.engine(): number_to_words(num, andword="and", zero="zero", group=0).  group 0: the digit string is cut into
    threes from the right; a three "hTU" reads "<h> hundred[ <andword>] <TU>" + its scale word; non-empty threes are joined
    by ", "; when the LAST three has no hundreds digit and something precedes it, it attaches with " <andword> " instead of
    ", " (no comma), if andword is not empty.  group 2: the string is cut into twos from the LEFT ("1905" -> 19 | 05), a
    leading 0 reads <zero>, "00" reads "<zero> <zero>"; joined by ", ".  A trailing st / nd / rd / th makes the last word
    ordinal."""

    def number_to_words(self, num, andword="and", zero="zero", group=0):
        s = str(num).strip().lower()
        m = re.fullmatch(r"(\d+)(st|nd|rd|th)", s)
        if m:
            words = self.number_to_words(m.group(1), andword=andword, zero=zero, group=group)
            head, last = re.match(r"(.*?)([a-z]+)$", words).groups()
            last = _NTH.get(last) or (last[:-1] + "ieth" if last.endswith("y") else last + "th")
            return head + last
        s = s.lstrip("+")
        if not s.isdigit():
            raise ValueError(s)
        if group == 2:
            out = []
            for i in range(0, len(s), 2):
                pair = s[i:i + 2]
                if len(pair) == 1:
                    out.append(zero if pair == "0" else _ONES[int(pair)])
                elif pair[0] == "0":
                    out.append(zero + " " + (zero if pair[1] == "0" else _ONES[int(pair[1])]))
                else:
                    out.append(_two(pair))
            return ", ".join(out)
        if group != 0:
            raise NotImplementedError(group)
        if int(s) == 0:
            return zero
        s = s.lstrip("0")
        threes = []
        while s:
            threes.append(s[-3:].rjust(3, "0"))
            s = s[:-3]
        spoken = []          # (index of the three, text), most significant first
        for idx in range(len(threes) - 1, -1, -1):
            h, tu = int(threes[idx][0]), threes[idx][1:]
            if h == 0 and int(tu) == 0:
                continue
            if h:
                text = _ONES[h] + " hundred"
                if int(tu):
                    text += (" " + andword if andword else "") + " " + _two(tu)
            else:
                text = _two(tu)
            spoken.append((idx, h, text + _MILL[idx]))
        result = ""
        for n, (idx, h, text) in enumerate(spoken):
            if n == 0:
                result = text
            elif idx == 0 and h == 0 and andword:
                result += " " + andword + " " + text
            else:
                result += ", " + text
        return result


def install_stubs():
    from parakeet_amd.frontend.g2p import LexiconG2p, letter_to_sound
    inflect = types.ModuleType("inflect")
    inflect.engine = _Engine
    sys.modules["inflect"] = inflect

    cmu = LexiconG2p.read_lexicon(LexiconG2p().lexicon_path)
    vowels = "AA AE AH AO AW AY EH ER EY IH IY OW OY UH UW".split()
    cons = "B CH D DH F G HH JH K L M N NG P R S SH T TH V W Y Z ZH".split()

    class G2p:
        # g2p_en/g2p.py: the published symbol tables (the bare "UW" between UH2 and UW0 is g2p_en's own)
        graphemes = ["<pad>", "<unk>", "</s>"] + list("abcdefghijklmnopqrstuvwxyz")
        phonemes = ["<pad>", "<unk>", "<s>", "</s>"] + sorted([v + s for v in vowels for s in "012"] + cons + ["UW"])

        def __call__(self, text):
            norm = importlib.import_module("parakeet.frontend.normalizer.normalizer").normalize
            text = norm(text)       # g2p_en's own preprocessing is the code the reference's normalizer was taken from
            prons = []
            for word in re.findall(r"[a-z]+(?:'[a-z]+)*'?|\.\.\.|[.,?!\-]", text):
                if re.search("[a-z]", word) is None:
                    pron = [word]
                elif word in cmu:
                    pron = cmu[word][0]
                elif word.strip("'") in cmu:
                    pron = cmu[word.strip("'")][0]
                else:
                    pron = letter_to_sound(word)
                    if not pron:
                        continue
                prons.extend(pron)
                prons.append(" ")
            return prons[:-1]

    g2p_en = types.ModuleType("g2p_en")
    g2p_en.G2p = G2p
    sys.modules["g2p_en"] = g2p_en
    g2pm = types.ModuleType("g2pM")         # phonectic.py imports it at module level for its Chinese class

    class G2pM:
        def __init__(self, *a, **k):
            raise RuntimeError("g2pM is not part of this golden")
    g2pm.G2pM = G2pM
    sys.modules["g2pM"] = g2pm


def main():
    install_stubs()
    pkg = types.ModuleType("parakeet")
    pkg.__path__ = [os.path.join(ref_import.REF, "parakeet")]
    sys.modules["parakeet"] = pkg
    for sub in ("frontend", "frontend.normalizer"):
        m = types.ModuleType("parakeet." + sub)
        m.__path__ = [os.path.join(ref_import.REF, "parakeet", *sub.split("."))]
        sys.modules["parakeet." + sub] = m
    numbers = importlib.import_module("parakeet.frontend.normalizer.numbers")
    normalizer = importlib.import_module("parakeet.frontend.normalizer.normalizer")
    width = importlib.import_module("parakeet.frontend.normalizer.width")
    ph = importlib.import_module("parakeet.frontend.phonectic")
    arp = importlib.import_module("parakeet.frontend.arpabet")
    out = {"normalize_numbers": {}, "normalize": {}, "english": {}, "character": {}, "arpabet": {}, "arpabet_stress": {}}
    en, ch, a0, a1 = ph.English(), ph.EnglishCharacter(), arp.ARPABET(), arp.ARPABETWithStress()
    out["vocab"] = {"english": list(en.vocab.stoi), "character": list(ch.vocab.stoi), "arpabet": list(a0.vocab.stoi),
                    "arpabet_stress": list(a1.vocab.stoi),
                    "sizes": [en.vocab_size, ch.vocab_size, a0.vocab_size, a1.vocab_size],
                    "special_indices": [en.vocab.padding_index, en.vocab.unk_index, en.vocab.start_index, en.vocab.end_index]}
    for s in SENTENCES:
        out["normalize_numbers"][s] = numbers.normalize_numbers(s)
        out["normalize"][s] = normalizer.normalize(s)
        phones = en.phoneticize(s)
        ids = en(s)
        assert ids == en.numericalize(phones) and en.reverse(ids) == phones
        out["english"][s] = {"phones": phones, "ids": ids}
        out["character"][s] = {"text": ch.phoneticize(s), "ids": ch(s)}
        out["arpabet"][s] = {"phones": a0.phoneticize(s), "ids": a0(s), "with_start_end": a0(s, add_start_end=True)}
        out["arpabet_stress"][s] = {"phones": a1.phoneticize(s), "ids": a1(s), "with_start_end": a1(s, add_start_end=True)}
    out["width"] = {s: [width.full2half_width(s), width.half2full_width(s)] for s in ("Hello, World 123!", "Ａ　ｂ！", "")}
    # the recipe's phone -> id mapping over a phone_id_map of the released layout (synthesize_e2e.py:45-50, 88-98)
    table = ["<pad>", "<unk>"] + sorted(p for p in en.phonemes if not p.startswith("<")) + ["sp", ",", ".", "?", "!", "<eos>"]
    phone_id_map = {p: i for i, p in enumerate(table)}
    punc = "：，；。？！“”‘’':,;.?!"
    out["recipe"] = {"phone_id_map": table, "sentences": {}}
    for s in RECIPE_SENTENCES:
        phones = en.phoneticize(s)[1:-1]
        phones = [p for p in phones if not p.isspace()]
        phones = [p if (p in phone_id_map and p not in punc) else "sp" for p in phones]
        out["recipe"]["sentences"][s] = {"phones": phones, "ids": [phone_id_map[p] for p in phones]}
    path = os.path.join(ROOT, "tests", "golden", "en_frontend.json")
    json.dump(out, open(path, "wt", encoding="utf-8"), ensure_ascii=False, indent=0, sort_keys=True)
    print("wrote", path, len(SENTENCES), "sentences")


if __name__ == "__main__":
    main()
