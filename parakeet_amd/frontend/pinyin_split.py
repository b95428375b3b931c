"""Pinyin syllable -> (initial, final + tone) in the conventions the reference's Mandarin frontend gets from pypinyin
(``Style.INITIALS`` / ``Style.FINALS_TONE3`` with ``strict=True`` and ``neutral_tone_with_five=True``,
parakeet/frontend/zh_frontend.py:68-71): y- and w- are spellings, not initials, and finals are written in full --
the orthographic rules of Hanyu Pinyin undone:

    yi -> i     ya -> ia    ye -> ie    yao -> iao   you -> iou   yan -> ian   yin -> in   yang -> iang
    ying -> ing yong -> iong   yu -> v   yue -> ve   yuan -> van   yun -> vn
    wu -> u     wa -> ua    wo -> uo    wai -> uai   wei -> uei   wan -> uan   wen -> uen  wang -> uang  weng -> ueng
    j / q / x + u, ue, uan, un  ->  v, ve, van, vn        (ü is written v; nü / lü arrive as nv / lv)
    initial + iu, ui, un        ->  iou, uei, uen

[pypinyin-semantics: pypinyin is not installable here; the table is the standard scheme it documents for strict mode.]
The i -> ii / iii distinction (after z c s / zh ch sh r) is made by the frontend, as in the reference (:72-77).
"""
import re

_INITIALS = ("zh", "ch", "sh", "b", "p", "m", "f", "d", "t", "n", "l", "g", "k", "h", "j", "q", "x", "r", "z", "c", "s")
_Y = {"i": "i", "a": "ia", "e": "ie", "ao": "iao", "ou": "iou", "an": "ian", "in": "in", "ang": "iang", "ing": "ing",
      "ong": "iong", "u": "v", "ue": "ve", "uan": "van", "un": "vn", "o": "io"}
_W = {"u": "u", "a": "ua", "o": "uo", "ai": "uai", "ei": "uei", "an": "uan", "en": "uen", "ang": "uang", "eng": "ueng"}
_SYLLABLE = re.compile(r"^([a-zü:]+?)([1-5]?)$")


def split_syllable(syllable):
    """'zhong1' -> ('zh', 'ong1'); 'yuan2' -> ('', 'van2'); 'jiu3' -> ('j', 'iou3'); 'er5' -> ('', 'er5').
    Anything that is not a pinyin syllable (punctuation, an unknown character) -> (s, s), like pypinyin's pass-through."""
    m = _SYLLABLE.match(syllable.lower())
    if not m:
        return syllable, syllable
    body, tone = m.group(1).replace("u:", "v").replace("ü", "v"), m.group(2) or "5"
    if body[0] == "y":
        rest = body[1:]
        return "", _Y.get(rest, "i" + rest) + tone
    if body[0] == "w":
        rest = body[1:]
        return "", _W.get(rest, "u" + rest) + tone
    ini = next((i for i in _INITIALS if body.startswith(i)), "")
    fin = body[len(ini):]
    if not any(v in fin for v in "aeiouv"):   # syllabic consonants (m, n, ng, hm): no initial
        return "", body + tone
    if ini in ("j", "q", "x") and fin[0] == "u":
        fin = "v" + fin[1:]
    if ini:
        fin = {"iu": "iou", "ui": "uei", "un": "uen"}.get(fin, fin)
    return ini, fin + tone
