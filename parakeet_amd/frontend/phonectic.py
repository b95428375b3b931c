"""Phonetic front ends (parakeet/frontend/phonectic.py:30-210; the module keeps the reference's spelling)."""
from abc import ABC, abstractmethod

from .g2p import LexiconG2p
from .normalizer import normalize
from .punctuation import get_punctuations
from .vocab import Vocab

__all__ = ["Phonetics", "English", "EnglishCharacter"]


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
