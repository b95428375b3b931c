"""English text normalisation (parakeet/frontend/normalizer/normalizer.py:22-34, numbers.py:17-86, width.py:17-40).

``normalize_numbers`` follows the reference's regular expressions and branch structure; the spelling of a number,
which the reference obtains from the third-party ``inflect`` package, is produced by ``number_to_words`` below with
inflect's conventions for the cases the reference uses: groups of three joined by ", ", hyphenated tens-units,
``andword`` between hundreds and the rest (``''`` for cardinals, ``'and'`` for ordinals, as numbers.py:56-74 calls
it), and the two-digit grouping with ``zero='oh'`` for years.
"""
import re
import unicodedata

__all__ = ["normalize", "normalize_numbers", "number_to_words", "full2half_width", "half2full_width"]

_UNITS = ["zero", "one", "two", "three", "four", "five", "six", "seven", "eight", "nine", "ten", "eleven", "twelve",
          "thirteen", "fourteen", "fifteen", "sixteen", "seventeen", "eighteen", "nineteen"]
_TENS = ["", "", "twenty", "thirty", "forty", "fifty", "sixty", "seventy", "eighty", "ninety"]
_SCALES = ["", "thousand", "million", "billion", "trillion", "quadrillion", "quintillion", "sextillion"]
_ORDINAL_WORD = {"one": "first", "two": "second", "three": "third", "five": "fifth", "eight": "eighth",
                 "nine": "ninth", "twelve": "twelfth"}


def _below_100(n):
    if n < 20:
        return _UNITS[n]
    t, u = divmod(n, 10)
    return _TENS[t] + ("-" + _UNITS[u] if u else "")


def _below_1000(n, andword):
    h, r = divmod(n, 100)
    if h == 0:
        return _below_100(r)
    out = _UNITS[h] + " hundred"
    if r:
        out += (" " + andword if andword else "") + " " + _below_100(r)
    return out


def _cardinal(n, andword):
    if n == 0:
        return "zero"
    groups = []
    while n:
        n, g = divmod(n, 1000)
        groups.append(g)
    if len(groups) > len(_SCALES):
        raise ValueError("number too large to spell")
    parts = []
    for i in range(len(groups) - 1, -1, -1):
        if groups[i] == 0:
            continue
        # inflect puts the andword only inside a group ("one thousand, two hundred and three"), except that a last
        # group below 100 after larger groups also takes it ("one thousand and three")
        if i == 0 and groups[i] < 100 and parts and andword:
            words = andword + " " + _below_100(groups[i])
            parts[-1] = parts[-1] + " " + words
            continue
        words = _below_1000(groups[i], andword)
        parts.append(words + (" " + _SCALES[i] if i else ""))
    return ", ".join(parts)


def _ordinal_from_cardinal(words):
    head, sep, last = words.rpartition("-") if "-" in words.rsplit(" ", 1)[-1] else words.rpartition(" ")
    if last in _ORDINAL_WORD:
        last = _ORDINAL_WORD[last]
    elif last.endswith("y"):
        last = last[:-1] + "ieth"
    else:
        last = last + "th"
    return head + sep + last


def _group2(n, zero):
    digits = str(n)
    if len(digits) % 2:
        chunks = [digits[0]] + [digits[i:i + 2] for i in range(1, len(digits), 2)]
    else:
        chunks = [digits[i:i + 2] for i in range(0, len(digits), 2)]
    out = []
    for c in chunks:
        if len(c) == 2 and c[0] == "0":
            out.append(zero + " " + (zero if c[1] == "0" else _UNITS[int(c[1])]))
        else:
            out.append(_below_100(int(c)))
    return ", ".join(out)


def number_to_words(num, andword="and", zero="zero", group=0):
    """Spell an integer (or an ordinal such as ``"21st"``) the way ``inflect.engine().number_to_words`` does for
    the argument combinations numbers.py uses."""
    s = str(num).strip()
    m = re.fullmatch(r"([0-9]+)(st|nd|rd|th)", s)
    if m:
        return _ordinal_from_cardinal(_cardinal(int(m.group(1)), andword))
    n = int(s)
    if group == 2:
        return _group2(n, zero)
    return _cardinal(n, andword) if n else zero


_comma_number_re = re.compile(r"([0-9][0-9\,]+[0-9])")
_decimal_number_re = re.compile(r"([0-9]+\.[0-9]+)")
_pounds_re = re.compile(r"£([0-9\,]*[0-9]+)")
_dollars_re = re.compile(r"\$([0-9\.\,]*[0-9]+)")
_ordinal_re = re.compile(r"[0-9]+(st|nd|rd|th)")
_number_re = re.compile(r"[0-9]+")


def _expand_dollars(m):
    match = m.group(1)
    parts = match.split(".")
    if len(parts) > 2:
        return match + " dollars"           # unexpected format (numbers.py:38-39)
    dollars = int(parts[0]) if parts[0] else 0
    cents = int(parts[1]) if len(parts) > 1 and parts[1] else 0
    if dollars and cents:
        return "%s %s, %s %s" % (dollars, "dollar" if dollars == 1 else "dollars", cents,
                                 "cent" if cents == 1 else "cents")
    if dollars:
        return "%s %s" % (dollars, "dollar" if dollars == 1 else "dollars")
    if cents:
        return "%s %s" % (cents, "cent" if cents == 1 else "cents")
    return "zero dollars"


def _expand_number(m):
    num = int(m.group(0))
    if 1000 < num < 3000:                     # years and the like (numbers.py:63-72)
        if num == 2000:
            return "two thousand"
        if 2000 < num < 2010:
            return "two thousand " + number_to_words(num % 100)
        if num % 100 == 0:
            return number_to_words(num // 100) + " hundred"
        return number_to_words(num, andword="", zero="oh", group=2).replace(", ", " ")
    return number_to_words(num, andword="")


def normalize_numbers(text):
    """numbers.py:77-86: thousands separators, pounds, dollars, decimals, ordinals, then plain numbers."""
    text = re.sub(_comma_number_re, lambda m: m.group(1).replace(",", ""), text)
    text = re.sub(_pounds_re, r"\1 pounds", text)
    text = re.sub(_dollars_re, _expand_dollars, text)
    text = re.sub(_decimal_number_re, lambda m: m.group(1).replace(".", " point "), text)
    text = re.sub(_ordinal_re, lambda m: number_to_words(m.group(0)), text)
    text = re.sub(_number_re, _expand_number, text)
    return text


def normalize(sentence):
    """normalizer.py:22-34: numbers, accents stripped, lower case, everything but ``a-z ' . , ? ! -`` and space
    dropped, two abbreviations expanded."""
    sentence = str(sentence)
    sentence = normalize_numbers(sentence)
    sentence = "".join(ch for ch in unicodedata.normalize("NFD", sentence) if unicodedata.category(ch) != "Mn")
    sentence = sentence.lower()
    sentence = re.sub(r"[^ a-z'.,?!\-]", "", sentence)
    sentence = sentence.replace("i.e.", "that is")
    sentence = sentence.replace("e.g.", "for example")
    return sentence


def full2half_width(ustr):
    """width.py:17-27."""
    out = []
    for u in ustr:
        num = ord(u)
        if num == 0x3000:
            num = 32
        elif 0xFF01 <= num <= 0xFF5E:
            num -= 0xFEE0
        out.append(chr(num))
    return "".join(out)


def half2full_width(ustr):
    """width.py:30-40."""
    out = []
    for u in ustr:
        num = ord(u)
        if num == 32:
            num = 0x3000
        elif 0x21 <= num <= 0x7E:
            num += 0xFEE0
        out.append(chr(num))
    return "".join(out)
