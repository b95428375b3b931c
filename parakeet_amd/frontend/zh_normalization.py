"""Mandarin text normalisation: sentence splitting and verbalisation of non-standard words (dates, times,
temperatures, fractions, percentages, phone numbers, ranges, signed / decimal numbers, numbers before measure words,
digit strings) -- the behaviour of parakeet/frontend/zh_normalization/ (text_normlization.py:48-98, num.py,
chronology.py, phonecode.py, quantifier.py), restated as one ordered rule table.

Host-side text processing; no device work.  Behaviours of the reference that look like slips are kept, because a
frontend that verbalises differently feeds the acoustic model different phones than it was trained on:
  * phone numbers read digit by digit with "一", not "幺" (phonecode.py passes alt_one=True, but num.py:196-200 drops the
    result of str.replace);
  * the temperature unit is always read "度" (quantifier.py:34-36 tests the decimal group instead of the unit group).
Not restated: the traditional -> simplified character table (char_convert.py; a data file of ~2 500 pairs): pass
``t2s=`` a mapping to enable it.
"""
import re

_DIGIT = "零一二三四五六七八九"
_UNIT = {1: "十", 2: "百", 3: "千", 4: "万", 8: "亿"}
_FULLWIDTH = {i + 0xFEE0: i for i in list(range(0x30, 0x3A)) + list(range(0x41, 0x5B)) + list(range(0x61, 0x7B))}
_FULLWIDTH[0x3000] = 0x20

# measure words a bare number is read as a cardinal in front of (num.py:32)
_MEASURE = ("(朵|匹|张|座|回|场|尾|条|个|首|阙|阵|网|炮|顶|丘|棵|只|支|袭|辆|挑|担|颗|壳|窠|曲|墙|群|腔|砣|座|客|贯|扎|捆|刀|令|打|手|罗|坡|山|岭|江|"
            "溪|钟|队|单|双|对|出|口|头|脚|板|跳|枝|件|贴|针|线|管|名|位|身|堂|课|本|页|家|户|层|丝|毫|厘|分|钱|两|斤|担|铢|石|钧|锱|忽|(千|毫|微)克|"
            "毫|厘|(公)分|分|寸|尺|丈|里|寻|常|铺|程|(千|分|厘|毫|微)米|米|撮|勺|合|升|斗|石|盘|碗|碟|叠|桶|笼|盆|盒|杯|钟|斛|锅|簋|篮|盘|桶|罐|瓶|壶|"
            "卮|盏|箩|箱|煲|啖|袋|钵|年|月|日|季|刻|时|周|天|秒|分|旬|纪|岁|世|更|夜|春|夏|秋|冬|代|伏|辈|丸|泡|粒|颗|幢|堆|条|根|支|道|面|片|张|颗|"
            "块|元|(亿|千万|百万|万|千|百)|(亿|千万|百万|万|千|百|美|)元|(亿|千万|百万|万|千|百|)块|角|毛|分)")


def verbalize_digit(digits):
    """'2021' -> '二零二一' (num.py:196-200)."""
    return "".join(_DIGIT[int(d)] for d in digits)


def _groups(s, zero=True):
    """Symbols of the cardinal reading of a digit string (num.py:160-176): split at the largest unit below the length."""
    t = s.lstrip("0")
    if not t:
        return []
    if len(t) == 1:
        return [_DIGIT[0], _DIGIT[int(t)]] if (zero and len(t) < len(s)) else [_DIGIT[int(t)]]
    power = max(p for p in _UNIT if p < len(t))
    return _groups(s[:-power]) + [_UNIT[power]] + _groups(s[-power:])


def verbalize_cardinal(digits):
    """'10086' -> '一万零八十六', '12' -> '十二', '000' -> '零' (num.py:179-193)."""
    if not digits:
        return ""
    digits = digits.lstrip("0")
    if not digits:
        return _DIGIT[0]
    sym = _groups(digits)
    if len(sym) >= 2 and sym[0] == _DIGIT[1] and sym[1] == _UNIT[1]:
        sym = sym[1:]            # 一十二 -> 十二
    return "".join(sym)


def num2str(value):
    """'3.20' -> '三点二', '.22' -> '零点二二' (num.py:203-224)."""
    parts = value.split(".")
    if len(parts) > 2:
        raise ValueError(f"The value string: '${value}' has more than one point in it.")
    out = verbalize_cardinal(parts[0])
    dec = parts[1].rstrip("0") if len(parts) == 2 else ""
    if dec:
        out = (out or _DIGIT[0]) + "点" + verbalize_digit(dec)
    return out


def _signed(sign, body):
    return ("负" if sign else "") + num2str(body)


def _number(m):
    """RE_NUMBER / RE_DECIMAL_NUM handler (num.py:118-137): group 5 is a bare '.5'."""
    return num2str(m.group(5)) if m.group(5) else _signed(m.group(1), m.group(2))


_RE_NUMBER = re.compile(r"(-?)((\d+)(\.\d+)?)|(\.(\d+))")


def _time_part(s):
    out = num2str(s.lstrip("0"))
    return _DIGIT[0] + out if s.startswith("0") else out


def _time(m):
    out = num2str(m.group(1)) + "点"
    if m.group(2).lstrip("0"):
        out += _time_part(m.group(2)) + "分"
    if m.group(4) and m.group(4).lstrip("0"):
        out += _time_part(m.group(4)) + "秒"
    return out


def _date(m):
    out = verbalize_digit(m.group(1)) + "年"
    if m.group(3):
        out += verbalize_cardinal(m.group(3)) + "月"
    if m.group(5):
        out += verbalize_cardinal(m.group(5)) + m.group(9)
    return out


def _date2(m):
    return verbalize_digit(m.group(1)) + "年" + verbalize_cardinal(m.group(3)) + "月" + verbalize_cardinal(m.group(4)) + "日"


def _mobile(m):
    return "".join(verbalize_digit(p) for p in m.group(0).strip("+").split())


def _telephone(m):
    return "".join(verbalize_digit(p) for p in m.group(0).split("-"))


def _range(m):
    return _RE_NUMBER.sub(_number, m.group(1)) + "到" + _RE_NUMBER.sub(_number, m.group(8))


# (pattern, handler) in the order text_normlization.py:77-93 applies them
_RULES = [
    (re.compile(r"(\d{4}|\d{2})年((0?[1-9]|1[0-2])月)?(((0?[1-9])|((1|2)[0-9])|30|31)([日号]))?"), _date),
    (re.compile(r"(\d{4})([- /.])(0[1-9]|1[012])\2(0[1-9]|[12][0-9]|3[01])"), _date2),
    (re.compile(r"([0-1]?[0-9]|2[0-3]):([0-5][0-9])(:([0-5][0-9]))?"), _time),
    (re.compile(r"(-?)(\d+(\.\d+)?)(°C|℃|度|摄氏度)"), lambda m: ("零下" if m.group(1) else "") + num2str(m.group(2)) + "度"),
    (re.compile(r"(-?)(\d+)/(\d+)"), lambda m: ("负" if m.group(1) else "") + num2str(m.group(3)) + "分之" + num2str(m.group(2))),
    (re.compile(r"(-?)(\d+(\.\d+)?)%"), lambda m: ("负" if m.group(1) else "") + "百分之" + num2str(m.group(2))),
    (re.compile(r"(?<!\d)((\+?86 ?)?1([38]\d|5[0-35-9]|7[678]|9[89])\d{8})(?!\d)"), _mobile),
    (re.compile(r"(?<!\d)((0(10|2[1-3]|[3-9]\d{2})-?)?[1-9]\d{7,8})(?!\d)"), _telephone),
    (re.compile(r"((-?)((\d+)(\.\d+)?)|(\.(\d+)))[-~]((-?)((\d+)(\.\d+)?)|(\.(\d+)))"), _range),
    (re.compile(r"(-)(\d+)"), lambda m: "负" + num2str(m.group(2))),
    (re.compile(r"(-?)((\d+)(\.\d+))|(\.(\d+))"), _number),
    (re.compile(r"(\d+)([多余几])?" + _MEASURE), lambda m: num2str(m.group(1)) + (m.group(2) or "") + m.group(3)),
    (re.compile(r"\d{3}\d*"), lambda m: verbalize_digit(m.group(0))),
    (_RE_NUMBER, _number),
]


class TextNormalizer:
    """``normalize(text) -> [sentence, ...]`` (text_normlization.py:48-98)."""

    def __init__(self, t2s=None):
        self._split_re = re.compile(r"([：，；。？！,;?!][”’]?)")
        self._t2s = {ord(k): v for k, v in (t2s or {}).items()}

    def _split(self, text):
        text = self._split_re.sub(r"\1\n", text).strip()
        return [s.strip() for s in re.split(r"\n+", text)]

    def normalize_sentence(self, sentence):
        if self._t2s:
            sentence = sentence.translate(self._t2s)
        sentence = sentence.translate(_FULLWIDTH)          # full-width letters, digits and the ideographic space
        for pattern, handler in _RULES:
            sentence = pattern.sub(handler, sentence)
        return sentence

    def normalize(self, text):
        return [self.normalize_sentence(s) for s in self._split(text)]
