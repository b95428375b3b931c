"""Every pinyin syllable Mandarin phonotactics allows, with its initial / final spelling: ``generate_lexicon(with_tone,
with_erhua) -> OrderedDict {syllable: "initial final"}`` of parakeet/frontend/generate_lexicon.py:27-160 (the lexicon the
reference hands to the Montreal Forced Aligner, and the pinyin -> phones table of its g2pM path, zh_frontend.py:42-44).

Restated as two steps: ``combines(C, V)`` -- may this initial carry this final? (the rules of :52-118 as lookups) -- and
``spell(C, V)`` -- the orthography of the pair (y / w for the empty initial, ü written u after j q x, iou / uei / uen
contracted, ii / iii written i; :120-150).  Pure host-side logic.
"""
from collections import OrderedDict

INITIALS = ["b", "p", "m", "f", "d", "t", "n", "l", "g", "k", "h", "zh", "ch", "sh", "r", "z", "c", "s", "j", "q", "x"]
FINALS = ["a", "ai", "ao", "an", "ang", "e", "er", "ei", "en", "eng", "o", "ou", "ong", "ii", "iii", "i", "ia", "iao", "ian", "iang",
          "ie", "io", "iou", "iong", "in", "ing", "u", "ua", "uai", "uan", "uang", "uei", "uo", "uen", "ueng", "v", "ve", "van", "vn"]
SPECIALS = ["sil", "sp"]

_PALATALS = {"j", "q", "x"}
_NO_FRONT = {"f", "g", "k", "h", "zh", "ch", "sh", "r", "z", "c", "s"}          # never before an i- or ü- final
_LABIALS = {"b", "p", "m", "f"}


def _front(V):
    """齐齿呼 / 撮口呼: the final starts with i or ü (the apical vowels ii / iii do not count)."""
    return V not in ("ii", "iii") and V[0] in "iv"


def combines(C, V):
    if V == "ii":
        return C in ("z", "c", "s")
    if V == "iii":
        return C in ("zh", "ch", "sh", "r")
    if _front(V) and C in _NO_FRONT:
        return False
    if V[0] == "v" and C not in (_PALATALS | {""} | ({"n", "l"} if V in ("v", "ve") else set())):
        return False
    if C in _PALATALS and not _front(V):
        return False
    if C in _LABIALS and ((V[0] in "uv" and V != "u") or V == "ong"):
        return False
    if V in ("ua", "uai", "uang") and C in ("d", "t", "n", "l", "r", "z", "c", "s"):
        return False
    if V == "ong" and C == "sh":
        return False
    if V == "o" and C in ("d", "t", "n", "g", "k", "h", "zh", "ch", "sh", "r", "z", "c", "s"):
        return False
    if V in ("ueng", "er") and C != "":          # weng and er stand alone
        return False
    return True


_CONTRACT = {"iou": "iu", "uei": "ui", "uen": "un"}


def spell(C, V):
    if C == "":
        if V in ("i", "in", "ing"):
            C = "y"
        elif V == "u":
            C = "w"
        elif V[0] == "i" and V not in ("ii", "iii"):
            C, V = "y", V[1:]
        elif V[0] == "u":
            C, V = "w", V[1:]
        elif V[0] == "v":
            C, V = "yu", V[1:]
    else:
        if C in _PALATALS and V[0] == "v":
            V = V.replace("v", "u")
        V = _CONTRACT.get(V, V)
    s = C + V
    while "ii" in s:                              # ii / iii are written i
        s = s.replace("ii", "i")
    return s


def rule(C, V, R, T):
    """The syllable of (initial, final, erhua mark, tone), or None when the combination does not exist."""
    if not combines(C, V):
        return None
    s = spell(C, V)
    if R == "r" and s.endswith("r"):             # er takes no further erhua
        return None
    return s + R + T


def generate_lexicon(with_tone=False, with_erhua=False):
    out = OrderedDict()
    for C in [""] + INITIALS:
        for V in FINALS:
            for R in (["", "r"] if with_erhua else [""]):
                for T in (["1", "2", "3", "4", "5"] if with_tone else [""]):
                    s = rule(C, V, R, T)
                    if s:
                        out[s] = f"{C} {V}{R}{T}"
    return out
