"""Symbol table with the four special symbols in front (parakeet/frontend/vocab.py:20-130)."""
from collections import OrderedDict

__all__ = ["Vocab"]


class Vocab(object):
    """Vocabulary: specials (<pad>, <unk>, <s>, </s>; any of them may be None / empty) take the first ids in that
    order, then ``symbols`` in their given order, duplicates ignored (vocab.py:46-66)."""

    def __init__(self, symbols, padding_symbol="<pad>", unk_symbol="<unk>", start_symbol="<s>", end_symbol="</s>"):
        self.special_symbols = OrderedDict()
        for item in (padding_symbol, unk_symbol, start_symbol, end_symbol):
            if item:
                self.special_symbols[item] = len(self.special_symbols)
        self.padding_symbol = padding_symbol
        self.unk_symbol = unk_symbol
        self.start_symbol = start_symbol
        self.end_symbol = end_symbol
        self.stoi = OrderedDict(self.special_symbols)
        for s in symbols:
            if s not in self.stoi:
                self.stoi[s] = len(self.stoi)
        self.itos = {v: k for k, v in self.stoi.items()}

    def __len__(self):
        return len(self.stoi)

    @property
    def num_specials(self):
        return len(self.special_symbols)

    @property
    def padding_index(self):
        return self.stoi.get(self.padding_symbol, -1)

    @property
    def unk_index(self):
        return self.stoi.get(self.unk_symbol, -1)

    @property
    def start_index(self):
        return self.stoi.get(self.start_symbol, -1)

    @property
    def end_index(self):
        return self.stoi.get(self.end_symbol, -1)

    def __repr__(self):
        return "Vocab(size: {},\nstoi:\n{})".format(len(self), self.stoi)

    __str__ = __repr__

    def lookup(self, symbol):
        return self.stoi[symbol]          # KeyError for unknown symbols, like the reference (:108-111)

    def reverse(self, index):
        return self.itos[index]

    def add_symbol(self, symbol):
        if symbol in self.stoi:
            return
        n = len(self.stoi)
        self.stoi[symbol] = n
        self.itos[n] = symbol

    def add_symbols(self, symbols):
        for s in symbols:
            self.add_symbol(s)
