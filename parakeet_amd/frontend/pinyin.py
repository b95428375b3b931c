"""The "simple Chinese phonology using pinyin symbols" of parakeet/frontend/pinyin.py: ``ParakeetPinyin`` (phones and
tones as two sequences, :55-140) and ``ParakeetPinyinWithTone`` (toned finals as single symbols, :143-215), the
phonologies of the Mandarin Tacotron2 recipes.

The reference asks pypinyin for the syllables of a sentence (``lazy_pinyin(style=TONE3, strict=True)`` with the neutral
tone written 5); here they come from the caller's pinyin lexicon (``PinyinLexicon``, as in zh_frontend.py of this
package).  Everything after that is the reference's own logic and is restated: the rewriting of standard pinyin into the
"parakeet convention" (:218-259 -- bo -> buo, ong -> ueng, iong -> veng, in / ing -> ien / ieng, un / ui / iu spelled out,
zi / zhi -> zii / zhiii, y / w glides, ju -> jv), the initial / final split, the symbol inventory and the vocabularies.

Two slips of the reference are kept as they are, because they decide what a model was trained on or what a caller sees:
``ParakeetPinyin`` with ``add_start_end=True`` overwrites the phone sequence with the tone sequence before adding the
markers (:63-71), so only ``<s> </s>`` survive the vocabulary filter; ``ParakeetPinyinWithTone`` with
``add_start_end=True`` reads an attribute it does not have (:151-153) and raises AttributeError.
"""
import re
from itertools import product

from .phonectic import Phonetics
from .vocab import Vocab
from .zh_frontend import PinyinLexicon

PUNCTUATIONS = ["，", "。", "？", "！"]
INITIALS = ["b", "p", "m", "f", "d", "t", "n", "l", "g", "k", "h", "j", "q", "x", "zh", "ch", "sh", "r", "z", "c", "s"]
FINALS = ["ii", "iii", "a", "o", "e", "ea", "ai", "ei", "ao", "ou", "an", "en", "ang", "eng", "er", "i", "ia", "io", "ie", "iai",
          "iao", "iou", "ian", "ien", "iang", "ieng", "u", "ua", "uo", "uai", "uei", "uan", "uen", "uang", "ueng", "v", "ve",
          "van", "ven", "veng"]
ERNIZED = ["&r"]
TONES = ["0", "1", "2", "3", "4", "5"]
PHONES = INITIALS + FINALS + ERNIZED + PUNCTUATIONS
TONED_PHONES = INITIALS + [f + t for f, t in product(FINALS, TONES[1:])] + ERNIZED + PUNCTUATIONS

# standard pinyin -> the inventory above, applied in this order to the toneless syllable (pinyin.py:224-257)
_REWRITES = (
    (re.compile(r"([bpmf])o$"), r"\1uo"),
    ("iong", "veng"), ("ong", "ueng"),
    ("ing", "ieng"), ("in", "ien"),
    ("un", "uen"), ("ui", "uei"), ("iu", "iou"),
    ("zi", "zii"), ("ci", "cii"), ("si", "sii"), ("zhi", "zhiii"), ("chi", "chiii"), ("shi", "shiii"), ("ri", "riii"),
    ("yi", "i"), ("yu", "v"), ("y", "i"),
    ("wu", "u"), ("w", "u"),
    ("ju", "jv"), ("qu", "qv"), ("xu", "xv"),
)


def to_parakeet_convention(syllable):
    """'zhong1' -> 'zhueng1'; the last character is the tone digit."""
    body, tone = syllable[:-1], syllable[-1]
    for pat, rep in _REWRITES:
        body = pat.sub(rep, body) if hasattr(pat, "sub") else body.replace(pat, rep)
    return body + tone


def split_syllable(syllable):
    """(phones, tones) of one syllable: the initial carries tone '0', the final the syllable's tone; a punctuation mark is a
    phone of tone '0' (:262-292)."""
    if syllable in PUNCTUATIONS:
        return [syllable], ["0"]
    s = to_parakeet_convention(syllable)
    body, tone = s[:-1], s[-1]
    if not body:          # a lone character that is neither pinyin nor one of the four marks: the reference raises
        return [], []     # IndexError here (:281); nothing of it would pass the vocabulary filter anyway
    for n in (2, 1):
        if body[:n] in INITIALS:
            return [body[:n], body[n:]], ["0", tone]
    return [body], [tone]


class _LexiconPhonology(Phonetics):
    def __init__(self, lexicon=None):
        self.lexicon = lexicon if isinstance(lexicon, PinyinLexicon) else PinyinLexicon(lexicon)

    def _syllables(self, sentence):
        """What ``lazy_pinyin(sentence, style=TONE3)`` gives: one tone-number syllable per character the lexicon can read;
        a run of other characters (letters, digits, punctuation) stays together as one item, as pypinyin leaves it."""
        out, run = [], ""
        for piece, _ in self.lexicon.segment(sentence):
            if piece in self.lexicon.words:
                if run:
                    out.append(run)
                    run = ""
                out += list(self.lexicon.words[piece][0])
            else:
                run += piece
        if run:
            out.append(run)
        return out


class ParakeetPinyin(_LexiconPhonology):
    def __init__(self, lexicon=None):
        super().__init__(lexicon)
        self.vocab_phonemes = Vocab(PHONES)
        self.vocab_tones = Vocab(TONES)

    def convert_pypinyin_tone3(self, syllables, add_start_end=False):
        phonemes, tones = [], []
        for s in syllables:
            p, t = split_syllable(s)
            phonemes += p
            tones += t
        if add_start_end:   # the reference's slip (:63-71): the phones are replaced by the tones here
            phonemes = [self.vocab_tones.start_symbol] + tones + [self.vocab_tones.end_symbol]
        phonemes = [p for p in phonemes if p in self.vocab_phonemes.stoi]
        tones = [t for t in tones if t in self.vocab_tones.stoi]
        return phonemes, tones

    def phoneticize(self, sentence, add_start_end=False):
        return self.convert_pypinyin_tone3(self._syllables(sentence), add_start_end=add_start_end)

    def numericalize(self, phonemes, tones):
        return [self.vocab_phonemes.lookup(p) for p in phonemes], [self.vocab_tones.lookup(t) for t in tones]

    def __call__(self, sentence, add_start_end=False):
        return self.numericalize(*self.phoneticize(sentence, add_start_end=add_start_end))

    @property
    def vocab_size(self):
        return len(self.vocab_phonemes)     # 70 = 62 phones + 4 punctuation marks + 4 special symbols

    @property
    def tone_vocab_size(self):
        return len(self.vocab_tones)        # 10


class ParakeetPinyinWithTone(_LexiconPhonology):
    def __init__(self, lexicon=None):
        super().__init__(lexicon)
        self.vocab = Vocab(TONED_PHONES)

    def convert_pypinyin_tone3(self, syllables, add_start_end=False):
        phonemes = []
        for s in syllables:
            if s in PUNCTUATIONS:
                phonemes.append(s)
                continue
            p, t = split_syllable(s)
            phonemes += p[:-1] + [p[-1] + t[-1]]      # the final carries the tone digit (:295-317)
        if add_start_end:
            raise AttributeError("'ParakeetPinyinWithTone' object has no attribute 'vocab_phonemes'")   # as :151-153
        return [p for p in phonemes if p in self.vocab.stoi]

    def phoneticize(self, sentence, add_start_end=False):
        return self.convert_pypinyin_tone3(self._syllables(sentence), add_start_end=add_start_end)

    def numericalize(self, phonemes):
        return [self.vocab.lookup(p) for p in phonemes]

    def __call__(self, sentence, add_start_end=False):
        return self.numericalize(self.phoneticize(sentence, add_start_end=add_start_end))

    @property
    def vocab_size(self):
        return len(self.vocab)              # 230
