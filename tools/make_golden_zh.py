#!/usr/bin/env python
"""Golden vectors of the Mandarin frontend from the REFERENCE's own source (parakeet/frontend/zh_frontend.py,
tone_sandhi.py, zh_normalization/) executed with stand-ins for the three dictionary packages it imports:
``pypinyin`` / ``jieba`` are answered from parakeet_amd's demonstration lexicon (the same resource the engine-side
frontend uses), ``g2pM`` is a dummy.  This pins the reference's own logic -- normalisation, merge rules, tone sandhi,
erhua, "sp", id mapping -- not the dictionaries.  Build container only.  Output: tests/golden/zh_frontend.json."""
import hashlib
import importlib
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_import  # noqa: E402

ref_import.setup()   # the paddle stand-in (zh_frontend.py wraps its ids in paddle tensors)
from parakeet_amd.frontend.pinyin_split import split_syllable  # noqa: E402
from parakeet_amd.frontend.zh_frontend import PinyinLexicon  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tests"))
from zh_cases import ARPABET_TEXTS, NORMALIZE, PINYIN_SENTENCES, SANDHI, SENTENCES, PHONES, TONES  # noqa: E402

LEX = PinyinLexicon()


def install_stubs():
    pp = types.ModuleType("pypinyin")
    ppc = types.ModuleType("pypinyin.constants")
    ppc.SUPPORT_UCS4 = True

    class Style:
        INITIALS, FINALS_TONE3 = "initials", "finals_tone3"

    def lazy_pinyin(word, neutral_tone_with_five=True, style=None):
        which = 0 if style == Style.INITIALS else 1
        return [split_syllable(s)[which] for s in LEX.pinyin(word)]

    pp.Style, pp.lazy_pinyin, pp.constants = Style, lazy_pinyin, ppc
    # pypinyin.core.Pinyin(converter).lazy_pinyin(sentence, style=Style.TONE3, strict=True) of frontend/pinyin.py: one
    # tone-number syllable per character the lexicon reads, runs of other characters kept together (as pypinyin does)
    core = types.ModuleType("pypinyin.core")

    class CoreStyle:
        TONE3 = "tone3"

    class Pinyin:
        def __init__(self, converter=None):
            pass

        def lazy_pinyin(self, sentence, style=None, strict=True):
            assert style == CoreStyle.TONE3
            out, run = [], ""
            for piece, _ in LEX.segment(sentence):
                if piece in LEX.words:
                    if run:
                        out.append(run)
                        run = ""
                    out += list(LEX.words[piece][0])
                else:
                    run += piece
            return out + ([run] if run else [])

    core.DefaultConverter, core.Pinyin, core.Style = type("DefaultConverter", (), {}), Pinyin, CoreStyle
    contrib = types.ModuleType("pypinyin.contrib")
    neutral = types.ModuleType("pypinyin.contrib.neutral_tone")
    neutral.NeutralToneWith5Mixin = type("NeutralToneWith5Mixin", (), {})
    contrib.neutral_tone = neutral
    pp.core, pp.contrib = core, contrib
    for name, mod in (("pypinyin.core", core), ("pypinyin.contrib", contrib), ("pypinyin.contrib.neutral_tone", neutral)):
        sys.modules[name] = mod
    jb = types.ModuleType("jieba")
    jb.cut_for_search = LEX.cut_for_search
    psg = types.ModuleType("jieba.posseg")
    psg.lcut = LEX.segment
    jb.posseg = psg
    g2pm = types.ModuleType("g2pM")

    class G2pM:   # the network's call contract (zh_frontend.py:79): tone-number syllables, ü written "u:"
        cedict = {w: list(syl) for w, (syl, _) in LEX.words.items()}   # phonectic.Chinese reads its syllable inventory here

        def __call__(self, word, tone=True, char_split=False):
            assert tone and not char_split
            if getattr(self, "plain_v", False):          # phonectic.Chinese: syllables as the dictionary spells them,
                out, run = [], ""                        # runs of other characters kept together
                for piece, _ in LEX.segment(word):
                    if piece in LEX.words:
                        out += ([run] if run else []) + list(LEX.words[piece][0])
                        run = ""
                    else:
                        run += piece
                return out + ([run] if run else [])
            return [p.replace("v", "u:") for p in LEX.pinyin(word)]

    g2pm.G2pM = G2pM
    from parakeet_amd.frontend.g2p import LexiconG2p
    g2pen = types.ModuleType("g2p_en")          # the English phonologies ask g2p_en for phones: the lexicon stand-in answers
    g2pen.G2p = LexiconG2p
    infl = types.ModuleType("inflect")          # imported by the reference's number normaliser, unused on these inputs
    infl.engine = lambda: None
    for name, mod in (("pypinyin", pp), ("pypinyin.constants", ppc), ("jieba", jb), ("jieba.posseg", psg), ("g2pM", g2pm),
                      ("g2p_en", g2pen), ("inflect", infl)):
        sys.modules[name] = mod
    for sub in ("frontend", "frontend.zh_normalization", "frontend.normalizer"):
        m = types.ModuleType("parakeet." + sub)
        m.__path__ = [os.path.join(ref_import.REF, "parakeet", *sub.split("."))]
        sys.modules["parakeet." + sub] = m


def main():
    install_stubs()
    zf = importlib.import_module("parakeet.frontend.zh_frontend")
    ts = importlib.import_module("parakeet.frontend.tone_sandhi")
    tn = importlib.import_module("parakeet.frontend.zh_normalization.text_normlization")
    tmp = os.path.join(ROOT, "gpurun_out")
    os.makedirs(tmp, exist_ok=True)
    pv, tv = os.path.join(tmp, "zh_phones.txt"), os.path.join(tmp, "zh_tones.txt")
    open(pv, "wt").write("".join(f"{p} {i}\n" for i, p in enumerate(PHONES)))
    open(tv, "wt").write("".join(f"{t} {i}\n" for i, t in enumerate(TONES)))
    fe = zf.Frontend(phone_vocab_path=pv, tone_vocab_path=tv)
    sandhi = ts.ToneSandhi()
    sandhi.must_neural_tone_words = set(sandhi.must_neural_tone_words)   # the reference's own list
    out = {"normalize": {}, "sandhi": [], "merge": [], "phonemes": {}, "ids": {}}
    norm = tn.TextNormalizer()
    for text in NORMALIZE:
        out["normalize"][text] = norm.normalize(text)
    for word, pos, finals in SANDHI:
        out["sandhi"].append([word, pos, finals, sandhi.modified_tone(word, pos, list(finals))])
    for text in SENTENCES:
        seg = LEX.segment(text)
        out["merge"].append([text, [[w, p] for w, p in sandhi.pre_merge_for_modify(seg)]])
        out["phonemes"][text] = {"merged": fe.get_phonemes(text), "split": fe.get_phonemes(text, merge_sentences=False),
                                 "no_erhua": fe.get_phonemes(text, with_erhua=False)}
        ids = fe.get_input_ids(text, merge_sentences=True, get_tone_ids=True)
        ids2 = fe.get_input_ids(text, merge_sentences=False)
        out["ids"][text] = {"phone_ids": [t.numpy().tolist() for t in ids["phone_ids"]],
                            "tone_ids": [t.numpy().tolist() for t in ids["tone_ids"]],
                            "phone_ids_split_no_tones": [t.numpy().tolist() for t in ids2["phone_ids"]]}
    ra = importlib.import_module("parakeet.frontend.arpabet")
    out["arpabet"] = {}
    for text in ARPABET_TEXTS:
        out["arpabet"][text] = {}
        for cls in (ra.ARPABET, ra.ARPABETWithStress):
            fe_en = cls()
            out["arpabet"][text][cls.__name__] = {
                "phones": fe_en.phoneticize(text), "phones_se": fe_en.phoneticize(text, add_start_end=True),
                "ids_se": fe_en(text, add_start_end=True), "vocab_size": fe_en.vocab_size}
    fe_m = zf.Frontend(g2p_model="g2pM", phone_vocab_path=pv, tone_vocab_path=tv)
    out["g2pM"] = {}
    for text in SENTENCES:
        ids = fe_m.get_input_ids(text, merge_sentences=True, get_tone_ids=True)
        out["g2pM"][text] = {"merged": fe_m.get_phonemes(text), "phone_ids": [t.numpy().tolist() for t in ids["phone_ids"]],
                             "tone_ids": [t.numpy().tolist() for t in ids["tone_ids"]]}
    rph = importlib.import_module("parakeet.frontend.phonectic")
    zh = rph.Chinese()
    zh.backend.plain_v = True
    out["chinese"] = {"vocab_size": zh.vocab_size, "phoneticize": {t: zh.phoneticize(t) for t in PINYIN_SENTENCES + ["未登录的字：龘abc，。"]}}
    rp = importlib.import_module("parakeet.frontend.pinyin")
    out["pinyin"] = {}
    plain, toned = rp.ParakeetPinyin(), rp.ParakeetPinyinWithTone()
    for text in PINYIN_SENTENCES:
        ph, tn_ = plain.phoneticize(text)
        ids = plain(text)
        out["pinyin"][text] = {"phonemes": ph, "tones": tn_, "phone_ids": ids[0], "tone_ids": ids[1],
                               "start_end": [list(v) for v in plain.phoneticize(text, add_start_end=True)],
                               "toned": toned.phoneticize(text), "toned_ids": toned(text)}
    out["pinyin_vocab"] = {"phones": plain.vocab_size, "tones": plain.tone_vocab_size, "toned": toned.vocab_size}
    gl = importlib.import_module("parakeet.frontend.generate_lexicon")
    out["generate_lexicon"] = {}
    for wt in (False, True):
        for we in (False, True):
            lex = gl.generate_lexicon(with_tone=wt, with_erhua=we)
            blob = "\n".join(f"{k}\t{v}" for k, v in lex.items()).encode("utf-8")
            items = list(lex.items())
            out["generate_lexicon"][f"tone{int(wt)}_erhua{int(we)}"] = {
                "n": len(lex), "sha256": hashlib.sha256(blob).hexdigest(), "head": items[:12], "every_97th": items[::97]}
    out["neutral_words_used"] = sorted(w for w in sandhi.must_neural_tone_words if w in LEX.words)
    path = os.path.join(ROOT, "tests", "golden", "zh_frontend.json")
    json.dump(out, open(path, "wt", encoding="utf-8"), ensure_ascii=False, indent=0)
    print("wrote", path, {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
