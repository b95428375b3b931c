"""Mandarin frontend (parakeet_amd/frontend/zh_*.py, tone_sandhi.py, pinyin_split.py) against golden vectors produced by
the reference's own source run over dictionary stand-ins (tools/make_golden_zh.py), and against facts stated in the
reference's comments.  CPU only."""
import json
import os
import sys

import numpy as np
import pytest

from parakeet_amd.frontend.pinyin_split import split_syllable
from parakeet_amd.frontend.zh_frontend import Frontend, PinyinLexicon
from parakeet_amd.frontend.zh_normalization import TextNormalizer, num2str, verbalize_cardinal

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from zh_cases import PHONES, TONES  # noqa: E402

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "zh_frontend.json"), encoding="utf-8"))


def _frontend(tmp_path):
    pv, tv = tmp_path / "phones.txt", tmp_path / "tones.txt"
    pv.write_text("".join(f"{p} {i}\n" for i, p in enumerate(PHONES)))
    tv.write_text("".join(f"{t} {i}\n" for i, t in enumerate(TONES)))
    # the reference ran with its own neutral-tone word list; the ones the demonstration lexicon knows:
    return Frontend(phone_vocab_path=str(pv), tone_vocab_path=str(tv), neutral_words=set(GOLD["neutral_words_used"]))


def test_normalizer_matches_reference_source():
    n = TextNormalizer()
    for text, want in GOLD["normalize"].items():
        assert n.normalize(text) == want, text
    # facts from the reference's comments (num.py:108, 179-193, 203-224; chronology.py; quantifier.py:19)
    assert verbalize_cardinal("000") == "零" and verbalize_cardinal("12") == "十二" and verbalize_cardinal("10086") == "一万零八十六"
    assert num2str(".22".lstrip()) == "零点二二" and num2str("3.20") == "三点二"
    assert n.normalize("-3°C") == ["零下三度"] and n.normalize("00078") == ["零零零七八"]
    assert n.normalize("ＡＢ１２") == ["AB十二"]          # full-width forms are folded (the reference's tables are no-ops)


def test_tone_sandhi_matches_reference_source(tmp_path):
    fe = _frontend(tmp_path)
    for word, pos, finals, want in GOLD["sandhi"]:
        assert fe.tone_modifier.modified_tone(word, pos, list(finals)) == want, (word, pos)
    got = {w: f for w, _, _, f in GOLD["sandhi"]}
    # the worked examples in tone_sandhi.py's comments
    assert got["家里"] == ["ia1", "i5"] and got["看不懂"][1] == "u5" and got["不怕"][0] == "u2"
    assert got["一段"][0] == "i2" and got["一天"][0] == "i4" and got["第一"][1] == "i1" and got["看一看"][1] == "i5"
    assert got["展览馆"] == ["an2", "an2", "uan3"] and got["纸老虎"] == ["iii3", "ao2", "u3"]    # 2 + 1 and 1 + 2 third tones
    assert got["蒙古包"] == ["eng2", "u3", "ao1"] and got["所有人"][0] == "uo2" and got["你好"] == ["i2", "ao3"]
    assert got["男子"][1] == "i3" and got["桌子"][1] == "i5"                  # must_not_neural_tone_words vs 子
    for text, want in GOLD["merge"]:
        seg = fe.lexicon.segment(text)
        assert [[w, p] for w, p in fe.tone_modifier.pre_merge_for_modify(seg)] == want, text
    # _merge_yi's docstring example (:228-231)
    assert fe.tone_modifier._merge_yi([("听", "v"), ("一", "m"), ("听", "v")]) == [["听一听", "v"]]


def test_frontend_matches_reference_source(tmp_path):
    fe = _frontend(tmp_path)
    for text, want in GOLD["phonemes"].items():
        assert fe.get_phonemes(text) == want["merged"], text
        assert fe.get_phonemes(text, merge_sentences=False) == want["split"], text
        assert fe.get_phonemes(text, with_erhua=False) == want["no_erhua"], text
    for text, want in GOLD["ids"].items():
        ids = fe.get_input_ids(text, merge_sentences=True, get_tone_ids=True)
        assert [a.tolist() for a in ids["phone_ids"]] == want["phone_ids"], text
        assert [a.tolist() for a in ids["tone_ids"]] == want["tone_ids"], text
        ids2 = fe.get_input_ids(text, merge_sentences=False)
        assert [a.tolist() for a in ids2["phone_ids"]] == want["phone_ids_split_no_tones"], text
        assert all(a.dtype == np.int64 for a in ids["phone_ids"] + ids["tone_ids"])
    # erhua: 小孩儿 -> the r joins the previous final; 女儿 / 花儿 keep their own syllable (not_erhua)
    ph = fe.get_phonemes("小孩儿在胡同儿里看花儿，女儿也去")[0]
    assert "air2" in ph and "ongr5" in ph and ph.count("er2") == 2      # 胡同 is on the neutral-tone list: tong5 + r
    assert "龘" in fe.missing                                   # characters the lexicon lacks are reported


def test_pinyin_split_and_lexicon():
    for syl, want in (("zhong1", ("zh", "ong1")), ("yuan2", ("", "van2")), ("jiu3", ("j", "iou3")), ("hui4", ("h", "uei4")),
                      ("dun1", ("d", "uen1")), ("xue2", ("x", "ve2")), ("yi1", ("", "i1")), ("wu3", ("", "u3")),
                      ("lv4", ("l", "v4")), ("er5", ("", "er5")), ("ma", ("m", "a5")), ("ng2", ("", "ng2")), ("，", ("，", "，"))):
        assert split_syllable(syl) == want, syl
    lex = PinyinLexicon(entries={"测试": (("ce4", "shi4"), "vn")})
    assert lex.segment("测试hello 测") == [("测试", "vn"), ("hello", "eng"), ("测", "x")]
    assert lex.cut_for_search("测试") == ["测试"] and lex.pinyin("测试") == ["ce4", "shi4"]
    demo = PinyinLexicon()
    assert demo.cut_for_search("蒙古包") == ["蒙古", "蒙古包"] and demo.cut_for_search("纸老虎") == ["老虎", "纸老虎"]
    assert [w for w, _ in demo.segment("我们今天去北京")] == ["我们", "今天", "去", "北京"]


def test_arpabet_phonologies_match_reference_source():
    """ARPABET / ARPABETWithStress (arpabet.py:26-302) over the same G2P backend as the reference classes were run with."""
    from parakeet_amd.frontend import ARPABET, ARPABETWithStress
    for text, per_cls in GOLD["arpabet"].items():
        for cls in (ARPABET, ARPABETWithStress):
            want, fe = per_cls[cls.__name__], cls()
            assert fe.phoneticize(text) == want["phones"], (text, cls.__name__)
            assert fe.phoneticize(text, add_start_end=True) == want["phones_se"]
            assert fe(text, add_start_end=True) == want["ids_se"] and fe.vocab_size == want["vocab_size"]
            assert fe.reverse(fe(text)) == fe.phoneticize(text)
    assert ARPABET().vocab_size == 47 and ARPABETWithStress().vocab_size == 77      # arpabet.py:208, :301
    assert all(p[-1] not in "012" for p in ARPABET().phoneticize("Hello, world!"))   # stress marks dropped


def test_pinyin_phonologies_match_reference_source():
    """ParakeetPinyin / ParakeetPinyinWithTone (frontend/pinyin.py) over the lexicon vs the reference's own source run with a
    pypinyin stand-in that answers from the same lexicon (tools/make_golden_zh.py)."""
    from parakeet_amd.frontend import ParakeetPinyin, ParakeetPinyinWithTone
    from parakeet_amd.frontend.pinyin import split_syllable, to_parakeet_convention
    g = GOLD
    plain, toned = ParakeetPinyin(), ParakeetPinyinWithTone()
    assert (plain.vocab_size, plain.tone_vocab_size, toned.vocab_size) == (70, 10, 230)          # pinyin.py:131-140, :213
    assert g["pinyin_vocab"] == {"phones": 70, "tones": 10, "toned": 230}
    for text, ref in g["pinyin"].items():
        ph, tn = plain.phoneticize(text)
        assert (ph, tn) == (ref["phonemes"], ref["tones"]), text
        ids = plain(text)
        assert (ids[0], ids[1]) == (ref["phone_ids"], ref["tone_ids"]), text
        assert [list(v) for v in plain.phoneticize(text, add_start_end=True)] == ref["start_end"], text   # the kept slip
        assert toned.phoneticize(text) == ref["toned"] and toned(text) == ref["toned_ids"], text
    # the rewriting rules one by one (pinyin.py:224-257)
    for src, want in (("bo1", "buo1"), ("zhong1", "zhueng1"), ("xiong2", "xveng2"), ("jin1", "jien1"), ("ying2", "ieng2"),
                      ("lun2", "luen2"), ("gui4", "guei4"), ("liu2", "liou2"), ("zi3", "zii3"), ("shi4", "shiii4"), ("ri4", "riii4"),
                      ("yu3", "v3"), ("ye3", "ie3"), ("wu3", "u3"), ("wo3", "uo3"), ("ju2", "jv2"), ("xue2", "xve2")):
        assert to_parakeet_convention(src) == want, src
    assert split_syllable("zhong1") == (["zh", "ueng"], ["0", "1"]) and split_syllable("，") == (["，"], ["0"])
    assert split_syllable("a1") == (["a"], ["1"])
    with pytest.raises(AttributeError):
        toned.phoneticize("你好", add_start_end=True)


def test_generate_lexicon_matches_reference_source():
    """generate_lexicon (frontend/generate_lexicon.py:146-160): every syllable, in order, with its initial / final spelling,
    in all four modes -- compared through a digest of the whole table plus samples (tools/make_golden_zh.py)."""
    import hashlib
    from parakeet_amd.frontend.generate_lexicon import generate_lexicon, rule
    for key, ref in GOLD["generate_lexicon"].items():
        wt, we = key[4] == "1", key[-1] == "1"
        lex = generate_lexicon(with_tone=wt, with_erhua=we)
        items = list(lex.items())
        assert len(lex) == ref["n"], key
        assert [list(i) for i in items[:12]] == ref["head"] and [list(i) for i in items[::97]] == ref["every_97th"], key
        blob = "\n".join(f"{k}\t{v}" for k, v in items).encode("utf-8")
        assert hashlib.sha256(blob).hexdigest() == ref["sha256"], key
    assert GOLD["generate_lexicon"]["tone0_erhua0"]["n"] == 460
    assert rule("", "ueng", "", "1") == "weng1" and rule("j", "van", "", "") == "juan" and rule("b", "ong", "", "") is None
    assert rule("", "er", "r", "2") is None and rule("zh", "iii", "r", "4") == "zhir4"


def test_g2pm_path_matches_reference_source(tmp_path):
    """Frontend(g2p_model="g2pM") (zh_frontend.py:40-44, :78-92): syllables split with the generate_lexicon table; the lexicon
    answers in the place of the g2pM network, as the stand-in did when the golden was made."""
    pv, tv = tmp_path / "phones.txt", tmp_path / "tones.txt"
    pv.write_text("".join(f"{p} {i}\n" for i, p in enumerate(PHONES)), encoding="utf-8")
    tv.write_text("".join(f"{t} {i}\n" for i, t in enumerate(TONES)), encoding="utf-8")
    fe = Frontend(g2p_model="g2pM", phone_vocab_path=str(pv), tone_vocab_path=str(tv), neutral_words=set(GOLD["neutral_words_used"]))
    for text, ref in GOLD["g2pM"].items():
        assert fe.get_phonemes(text) == ref["merged"], text
        ids = fe.get_input_ids(text, merge_sentences=True, get_tone_ids=True)
        assert [t.tolist() for t in ids["phone_ids"]] == ref["phone_ids"], text
        assert [t.tolist() for t in ids["tone_ids"]] == ref["tone_ids"], text
    with pytest.raises(ValueError):
        Frontend(g2p_model="espeak")


def test_chinese_syllable_phonology_matches_reference_source():
    """phonectic.Chinese (:213-300): whole syllables + punctuation, <s> / </s>, the character-wise filter of what the vocabulary
    lacks -- the reference's source over a g2pM stand-in that answers from the lexicon.  Ids are compared through the symbols:
    the reference's vocabulary order is a set's."""
    from parakeet_amd.frontend import Chinese
    zh = Chinese()
    assert zh.vocab_size == GOLD["chinese"]["vocab_size"]
    for text, ref in GOLD["chinese"]["phoneticize"].items():
        ph = zh.phoneticize(text)
        assert ph == ref, text
        assert [zh.vocab.reverse(i) for i in zh(text)] == ref
    assert zh.phonemes == sorted(zh.phonemes) and zh.vocab.lookup("<s>") == 2
