"""Phonetic front ends (parakeet/frontend/phonectic.py:30-210; the module keeps the reference's spelling)."""
from abc import ABC, abstractmethod

from .g2p import LexiconG2p
from .normalizer import normalize
from .punctuation import get_punctuations
from .vocab import Vocab

__all__ = ["Phonetics", "English", "EnglishCharacter", "ARPABET", "ARPABETWithStress", "Chinese"]


class Phonetics(ABC):
    @abstractmethod
    def __call__(self, sentence):
        pass

    @abstractmethod
    def phoneticize(self, sentence):
        pass

    @abstractmethod
    def numericalize(self, phonemes):
        pass


class English(Phonetics):
    """Text -> ARPAbet phones (with stress marks) -> ids (phonectic.py:44-128).  ``lexicon`` is the CMUdict-format
    file of the G2P stand-in (frontend/g2p.py); ``backend`` may be any callable with g2p_en.G2p's contract."""

    def __init__(self, lexicon=None, backend=None):
        self.backend = backend if backend is not None else LexiconG2p(lexicon)
        self.phonemes = list(self.backend.phonemes)
        self.punctuations = get_punctuations("en")
        self.vocab = Vocab(self.phonemes + self.punctuations)

    def phoneticize(self, sentence):
        start, end = self.vocab.start_symbol, self.vocab.end_symbol
        phonemes = ([] if start is None else [start]) + self.backend(sentence) + ([] if end is None else [end])
        return [item for item in phonemes if item in self.vocab.stoi]

    def numericalize(self, phonemes):
        return [self.vocab.lookup(item) for item in phonemes if item in self.vocab.stoi]

    def reverse(self, ids):
        return [self.vocab.reverse(i) for i in ids]

    def __call__(self, sentence):
        return self.numericalize(self.phoneticize(sentence))

    @property
    def vocab_size(self):
        return len(self.vocab)


class EnglishCharacter(Phonetics):
    """Text -> normalised characters -> ids (phonectic.py:131-210)."""

    def __init__(self, backend=None):
        self.graphemes = list((backend if backend is not None else LexiconG2p).graphemes)
        self.punctuations = get_punctuations("en")
        self.vocab = Vocab(self.graphemes + self.punctuations)

    def phoneticize(self, sentence):
        return normalize(sentence)

    def numericalize(self, sentence):
        return [self.vocab.lookup(item) for item in sentence if item in self.vocab.stoi]

    def reverse(self, ids):
        return [self.vocab.reverse(i) for i in ids]

    def __call__(self, sentence):
        return self.numericalize(self.phoneticize(sentence))

    @property
    def vocab_size(self):
        return len(self.vocab)


_ARPABET_VOWELS = ("AA", "AE", "AH", "AO", "AW", "AY", "EH", "ER", "EY", "IH", "IY", "OW", "OY", "UW", "UH")


class ARPABET(Phonetics):
    """Text -> ARPAbet phones WITHOUT stress marks -> ids over a fixed 39-phone inventory + , . ? ! (arpabet.py:26-209:
    the phonology of the reference's Tacotron2 / TransformerTTS "phone" recipes).  ``lexicon`` / ``backend`` as English."""
    phonemes = ["AA", "AE", "AH", "AO", "AW", "AY", "B", "CH", "D", "DH", "EH", "ER", "EY", "F", "G", "HH", "IH", "IY", "JH",
                "K", "L", "M", "N", "NG", "OW", "OY", "P", "R", "S", "SH", "T", "TH", "UW", "UH", "V", "W", "Y", "Z", "ZH"]
    punctuations = [",", ".", "?", "!"]
    symbols = phonemes + punctuations
    keep_stress = False

    def __init__(self, lexicon=None, backend=None):
        self.backend = backend if backend is not None else LexiconG2p(lexicon)
        self.vocab = Vocab(self.phonemes + self.punctuations)

    def _remove_vowels(self, phone):
        """'AH0' -> 'AH' (the reference's name for dropping the stress digit of a vowel, arpabet.py:131-132)."""
        return phone[:-1] if phone[-1:] in "012" and phone[:-1] in _ARPABET_VOWELS else phone

    def phoneticize(self, sentence, add_start_end=False):
        phonemes = list(self.backend(sentence))
        if not self.keep_stress:
            phonemes = [self._remove_vowels(p) for p in phonemes]
        if add_start_end:
            phonemes = [self.vocab.start_symbol] + phonemes + [self.vocab.end_symbol]
        return [p for p in phonemes if p in self.vocab.stoi]

    def numericalize(self, phonemes):
        return [self.vocab.lookup(p) for p in phonemes]

    def reverse(self, ids):
        return [self.vocab.reverse(i) for i in ids]

    def __call__(self, sentence, add_start_end=False):
        return self.numericalize(self.phoneticize(sentence, add_start_end=add_start_end))

    @property
    def vocab_size(self):
        return len(self.vocab)    # 47 = 39 phones + 4 punctuation marks + 4 special tokens


class ARPABETWithStress(ARPABET):
    """The same with stress marks kept: 69 phones (arpabet.py:212-302)."""
    phonemes = [p for base in ("AA", "AE", "AH", "AO", "AW", "AY") for p in (base + "0", base + "1", base + "2")] + \
        ["B", "CH", "D", "DH"] + [p for base in ("EH", "ER", "EY") for p in (base + "0", base + "1", base + "2")] + \
        ["F", "G", "HH"] + [p for base in ("IH", "IY") for p in (base + "0", base + "1", base + "2")] + \
        ["JH", "K", "L", "M", "N", "NG"] + [p for base in ("OW", "OY") for p in (base + "0", base + "1", base + "2")] + \
        ["P", "R", "S", "SH", "T", "TH"] + [p for base in ("UH", "UW") for p in (base + "0", base + "1", base + "2")] + \
        ["V", "W", "Y", "Z", "ZH"]
    symbols = phonemes + ARPABET.punctuations
    keep_stress = True


class Chinese(Phonetics):
    """Whole-syllable Mandarin phonology (phonectic.py:213-300): a sentence -> tone-number syllables and punctuation marks,
    ``<s>`` / ``</s>`` around them, ids from a vocabulary of every syllable the backend knows plus the Chinese punctuation.

    The reference's backend is the g2pM network and its vocabulary is ``list(set(...))`` over g2pM's dictionary -- an order
    that changes with the interpreter's string hashing.  Here the backend is the caller's pinyin lexicon and the syllables
    are sorted, so ids are reproducible; a model trained with the reference needs the id table it was trained with."""

    def __init__(self, lexicon=None):
        from .zh_frontend import PinyinLexicon
        self.lexicon = lexicon if isinstance(lexicon, PinyinLexicon) else PinyinLexicon(lexicon)
        self.phonemes = sorted({syl for syls, _ in self.lexicon.words.values() for syl in syls})
        self.punctuations = get_punctuations("cn")
        self.vocab = Vocab(self.phonemes + self.punctuations)

    def backend(self, sentence):
        """g2pM's call contract (tone=True, char_split=False): one syllable per character it reads; a run of other
        characters stays together as one item."""
        out, run = [], ""
        for piece, _ in self.lexicon.segment(sentence):
            if piece in self.lexicon.words:
                if run:
                    out.append(run)
                    run = ""
                out += list(self.lexicon.words[piece][0])
            else:
                run += piece
        return out + ([run] if run else [])

    def _filter_symbols(self, phonemes):
        """Items of the vocabulary pass; anything else is looked at character by character (:254-263)."""
        out = []
        for item in phonemes:
            if item in self.vocab.stoi:
                out.append(item)
            else:
                out += [ch for ch in item if ch in self.vocab.stoi]
        return out

    def phoneticize(self, sentence):
        start, end = self.vocab.start_symbol, self.vocab.end_symbol
        return self._filter_symbols(([] if start is None else [start]) + self.backend(sentence) + ([] if end is None else [end]))

    def numericalize(self, phonemes):
        return [self.vocab.lookup(item) for item in phonemes]

    def __call__(self, sentence):
        return self.numericalize(self.phoneticize(sentence))

    @property
    def vocab_size(self):
        return len(self.vocab)
