"""Symbol table of the text frontends: the interface of the reference's ``Vocab`` (parakeet/frontend/vocab.py:20-130 --
``lookup`` / ``reverse`` / ``add_symbol(s)``, ``stoi`` / ``itos``, the four ``*_index`` properties) over one ordered
symbol table.  Ids are insertion positions: the special symbols that are set come first in the order pad, unk, start, end,
then the caller's symbols in first-seen order.  ``stoi`` / ``itos`` / ``special_symbols`` are plain dict attributes kept in step
by ``add_symbol`` -- callers of the reference read AND write them (``vocab.stoi[s] = i``), so they are neither copies nor
read-only views (writing ``stoi`` directly leaves ``itos`` and ``len()`` behind, exactly as it does in the reference: use
``add_symbol``).  The four ``*_symbol`` names are plain instance attributes as in the reference (:57-60), assignable; the
``*_index`` properties look the CURRENT value up, -1 when the table does not hold it."""

__all__ = ["Vocab"]


class Vocab:
    def __init__(self, symbols, padding_symbol="<pad>", unk_symbol="<unk>", start_symbol="<s>", end_symbol="</s>"):
        self.padding_symbol, self.unk_symbol = padding_symbol, unk_symbol
        self.start_symbol, self.end_symbol = start_symbol, end_symbol
        self.stoi = {}                # symbol -> id (insertion ordered)
        self.itos = {}                # id -> symbol
        self.add_symbols(s for s in (padding_symbol, unk_symbol, start_symbol, end_symbol) if s)   # None / "": no such symbol
        self.special_symbols = dict(self.stoi)
        self.add_symbols(symbols)

    @property
    def num_specials(self):
        return len(self.special_symbols)

    # -- growing the table -------------------------------------------------------------------------------------------
    def add_symbol(self, symbol):
        """Append ``symbol`` with the next free id; a symbol already present keeps its id."""
        n = len(self.stoi)
        if self.stoi.setdefault(symbol, n) == n:
            self.itos[n] = symbol

    def add_symbols(self, symbols):
        for symbol in symbols:
            self.add_symbol(symbol)

    # -- lookups -----------------------------------------------------------------------------------------------------
    def lookup(self, symbol):
        """Id of ``symbol``; unknown symbols raise KeyError (the callers map them to a fallback themselves)."""
        return self.stoi[symbol]

    def reverse(self, index):
        return self.itos[index]

    def __len__(self):
        return len(self.stoi)

    padding_index = property(lambda self: self.stoi.get(self.padding_symbol, -1))
    unk_index = property(lambda self: self.stoi.get(self.unk_symbol, -1))
    start_index = property(lambda self: self.stoi.get(self.start_symbol, -1))
    end_index = property(lambda self: self.stoi.get(self.end_symbol, -1))

    def __contains__(self, symbol):
        return symbol in self.stoi

    def __repr__(self):
        return "Vocab(size: {},\nstoi:\n{})".format(len(self), dict(self.stoi))
