"""Symbol table of the text frontends: the interface of the reference's ``Vocab`` (parakeet/frontend/vocab.py:20-130 --
``lookup`` / ``reverse`` / ``add_symbol(s)``, ``stoi`` / ``itos``, the four ``*_index`` properties) over one ordered
symbol list and its inverse index.  Ids are positions in that list: the special symbols that are set come first in the
order pad, unk, start, end, then the caller's symbols in first-seen order."""

from types import MappingProxyType

__all__ = ["Vocab"]

_ROLES = ("padding", "unk", "start", "end")


class Vocab:
    def __init__(self, symbols, padding_symbol="<pad>", unk_symbol="<unk>", start_symbol="<s>", end_symbol="</s>"):
        self._role = dict(zip(_ROLES, (padding_symbol, unk_symbol, start_symbol, end_symbol)))
        self._symbols = []            # id -> symbol
        self._index = {}              # symbol -> id
        self.add_symbols(s for s in self._role.values() if s)      # None / "" means "this table has no such symbol"
        self.num_specials = len(self._symbols)
        self.add_symbols(symbols)

    # -- growing the table -------------------------------------------------------------------------------------------
    def add_symbol(self, symbol):
        """Append ``symbol`` with the next free id; a symbol already present keeps its id."""
        if self._index.setdefault(symbol, len(self._symbols)) == len(self._symbols):
            self._symbols.append(symbol)

    def add_symbols(self, symbols):
        for symbol in symbols:
            self.add_symbol(symbol)

    # -- lookups -----------------------------------------------------------------------------------------------------
    def lookup(self, symbol):
        """Id of ``symbol``; unknown symbols raise KeyError (the callers map them to a fallback themselves)."""
        return self._index[symbol]

    def reverse(self, index):
        if not 0 <= index < len(self._symbols):
            raise KeyError(index)
        return self._symbols[index]

    def __len__(self):
        return len(self._symbols)

    def _special(self, role):
        return self._index.get(self._role[role], -1)

    padding_symbol = property(lambda self: self._role["padding"])
    unk_symbol = property(lambda self: self._role["unk"])
    start_symbol = property(lambda self: self._role["start"])
    end_symbol = property(lambda self: self._role["end"])
    padding_index = property(lambda self: self._special("padding"))
    unk_index = property(lambda self: self._special("unk"))
    start_index = property(lambda self: self._special("start"))
    end_index = property(lambda self: self._special("end"))

    # -- the reference's public dict views (read-only, in id order) -------------------------------------------
    @property
    def stoi(self):
        return MappingProxyType(self._index)

    def __contains__(self, symbol):
        return symbol in self._index

    @property
    def itos(self):
        return dict(enumerate(self._symbols))

    @property
    def special_symbols(self):
        return {s: i for i, s in enumerate(self._symbols[:self.num_specials])}

    def __repr__(self):
        return "Vocab(size: {},\nstoi:\n{})".format(len(self), dict(self._index))
